/* modup_reuse.c -- our own program against the rt_ant drop-in API: the runtime keeps the raised digits of a polynomial
 * across rotations of the same ciphertext (csrc/rt/rt_poly.cpp ModupCache); this is every way the digits could be stale.
 * The rotation is spelled at the polynomial level, call for call what the ACE POLY pass emits for Rotate()
 * (Decomp_modup per digit, key inner product with Hw_modmul / Hw_modadd per limb through one scratch limb, Mod_down,
 * + c0, Hw_rotate), because that call sequence is what the cache keys on.
 *  1. three rotations of one ciphertext in a row (the taps of a convolution): digits raised once;
 *  2. the ciphertext is changed in place by queued per-limb ops between two rotations;
 *  3. the ciphertext is rewritten at the same address and level by direct launches (Rescale into it) between two rotations;
 *  4. the ciphertext is freed and another one gets its memory.
 * Checked against the clear computation; every slot is printed so that the test can compare ACEHIP_MODUP_REUSE=0 / 1. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#define LEN 64

static void limbs_mac(POLY acc, POLY key, POLY ext, POLY tmp, uint32_t n, uint32_t acc_at, uint32_t key_at, MODULUS* m, uint32_t degree) {
  for (uint32_t i = 0; i < n; ++i, ++m) {
    Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key, key_at + i, degree), Coeffs(ext, acc_at + i, degree), m, degree);
    Hw_modadd(Coeffs(acc, acc_at + i, degree), Coeffs(acc, acc_at + i, degree), Coeffs(tmp, 0, degree), m, degree);
  }
}

/* takes the ciphertext BY VALUE like the generated function: the polynomial pointers are what identifies the source */
static CIPHERTEXT rotate_by_hand(CIPHERTEXT ct, int32_t idx) {
  uint32_t degree = Degree();
  CIPHERTEXT res;
  memset(&res, 0, sizeof(res));
  Init_ciph_same_scale(&res, &ct, 0);
  size_t lv = Poly_level(&ct._c1_poly);
  POLY acc0 = Alloc_poly(degree, lv, 1), acc1 = Alloc_poly(degree, lv, 1), ext = Alloc_poly(degree, lv, 1);
  POLY tmp = Alloc_poly(degree, 1, 0), d0 = Alloc_poly(degree, lv, 0), d1 = Alloc_poly(degree, lv, 0);
  SW_KEY swk = Swk(1, idx);
  for (uint32_t part = 0; part < Num_decomp(&ct._c1_poly); ++part) {
    Decomp_modup(ext, &ct._c1_poly, part);
    POLY k0 = Pk0_at(swk, part), k1 = Pk1_at(swk, part);
    uint32_t nq = Poly_level(ext), np = Num_p(ext), p_at = Num_alloc(ext) - np, kp_at = Poly_level(k0);
    limbs_mac(acc0, k0, ext, tmp, nq, 0, 0, Q_modulus(), degree);
    limbs_mac(acc1, k1, ext, tmp, nq, 0, 0, Q_modulus(), degree);
    /* p-limbs: the key keeps them behind ITS q-limbs */
    for (uint32_t i = 0; i < np; ++i) {
      MODULUS* m = P_modulus() + i;
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(k0, kp_at + i, degree), Coeffs(ext, p_at + i, degree), m, degree);
      Hw_modadd(Coeffs(acc0, p_at + i, degree), Coeffs(acc0, p_at + i, degree), Coeffs(tmp, 0, degree), m, degree);
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(k1, kp_at + i, degree), Coeffs(ext, p_at + i, degree), m, degree);
      Hw_modadd(Coeffs(acc1, p_at + i, degree), Coeffs(acc1, p_at + i, degree), Coeffs(tmp, 0, degree), m, degree);
    }
  }
  Mod_down(d0, acc0);
  Mod_down(d1, acc1);
  MODULUS* m = Q_modulus();
  int64_t* order = Auto_order(idx);
  for (uint32_t i = 0; i < Poly_level(d0); ++i, ++m) {
    Hw_modadd(Coeffs(d0, i, degree), Coeffs(d0, i, degree), Coeffs(&ct._c0_poly, i, degree), m, degree);
    Hw_rotate(Coeffs(&res._c0_poly, i, degree), Coeffs(d0, i, degree), order, m, degree);
    Hw_rotate(Coeffs(&res._c1_poly, i, degree), Coeffs(d1, i, degree), order, m, degree);
  }
  Free_poly(acc0);
  Free_poly(acc1);
  Free_poly(ext);
  Free_poly(tmp);
  Free_poly(d0);
  Free_poly(d1);
  return res;
}

static void add_into(CIPHER acc, CIPHER x) { Add_ciph(acc, acc, x); }

bool Main_graph() {
  CIPHERTEXT in = Get_input_data("input", 0);
  uint32_t degree = Degree();
  CIPHERTEXT x, acc, r, a2, b3, w, z;
  memset(&x, 0, sizeof(x));
  memset(&acc, 0, sizeof(acc));
  memset(&a2, 0, sizeof(a2));
  memset(&b3, 0, sizeof(b3));
  memset(&w, 0, sizeof(w));
  memset(&z, 0, sizeof(z));
  float twos[LEN], threes[LEN];
  for (int i = 0; i < LEN; ++i) twos[i] = 2.0f, threes[i] = 3.0f;
  PLAINTEXT p2, p3;
  memset(&p2, 0, sizeof(p2));
  memset(&p3, 0, sizeof(p3));
  Encode_plain_from_float(&p2, twos, LEN, 1, Level(&in));
  Encode_plain_from_float(&p3, threes, LEN, 1, Level(&in));
  Mul_plain(&a2, &in, &p2); /* 2x at Delta^2 */
  Mul_plain(&b3, &in, &p3); /* 3x at Delta^2 */
  /* x = rescale(2x), spelled per polynomial into a ciphertext allocated at the input's level: the polynomials of x keep
   * their addresses whatever is rescaled into them later */
  Init_ciph_same_scale(&x, &in, 0);
  Rescale(&x._c0_poly, &a2._c0_poly);
  Rescale(&x._c1_poly, &a2._c1_poly);
  x._scaling_factor = in._scaling_factor;
  x._sf_degree = 1;
  /* 1: three taps of the same ciphertext: 2x<<1 + 2x<<2 + 2x<<3 */
  acc = rotate_by_hand(x, 1);
  r = rotate_by_hand(x, 2);
  add_into(&acc, &r);
  Free_ciph_poly(&r, 1);
  r = rotate_by_hand(x, 3);
  add_into(&acc, &r);
  Free_ciph_poly(&r, 1);
  /* 2: x += x through per-limb ops (x = 4x), then the same rotation again: + 4x<<1 */
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < Poly_level(&x._c0_poly); ++i, ++m) {
    Hw_modadd(Coeffs(&x._c0_poly, i, degree), Coeffs(&x._c0_poly, i, degree), Coeffs(&x._c0_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&x._c1_poly, i, degree), Coeffs(&x._c1_poly, i, degree), Coeffs(&x._c1_poly, i, degree), m, degree);
  }
  r = rotate_by_hand(x, 1);
  add_into(&acc, &r);
  Free_ciph_poly(&r, 1);
  /* 3: x = rescale(3x) by direct launches into the SAME polynomials (same address, same level afterwards): + 3x<<1 */
  x._c0_poly._num_primes = a2._c0_poly._num_primes;
  x._c1_poly._num_primes = a2._c1_poly._num_primes;
  Rescale(&x._c0_poly, &b3._c0_poly);
  Rescale(&x._c1_poly, &b3._c1_poly);
  r = rotate_by_hand(x, 1);
  add_into(&acc, &r);
  Free_ciph_poly(&r, 1);
  /* 4: x is freed, a copy of another ciphertext (x<<2 of the 2x we started from: acc's first addend recomputed) takes
   * its memory: + (that)<<1 */
  Init_ciph_same_scale(&w, &in, 0);
  Rescale(&w._c0_poly, &a2._c0_poly);
  Rescale(&w._c1_poly, &a2._c1_poly); /* w = 2x */
  w._scaling_factor = in._scaling_factor;
  w._sf_degree = 1;
  Free_ciph_poly(&x, 1);
  r = rotate_by_hand(w, 2); /* hands the queue over: x's blocks are back in the pool */
  Free_ciph_poly(&r, 1);
  Copy_ciph(&z, &w); /* z may sit where x was */
  r = rotate_by_hand(z, 1); /* + 2x<<1 */
  add_into(&acc, &r);
  Free_ciph_poly(&r, 1);
  Set_output_data("output", 0, &acc);
  Free_plain(&p2);
  Free_plain(&p3);
  Free_ciph_poly(&in, 1);
  Free_ciph_poly(&a2, 1);
  Free_ciph_poly(&b3, 1);
  Free_ciph_poly(&w, 1);
  Free_ciph_poly(&z, 1);
  return true;
}

CKKS_PARAMS* Get_context_params() {
  static CKKS_PARAMS parm = {LIB_ANT, 16384, 0, 3, 60, 50, 2, 192, 3, {1, 2, 3}};
  return &parm;
}
DATA_SCHEME* Get_encode_scheme(int idx) {
  static DATA_SCHEME scheme = {"input", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
DATA_SCHEME* Get_decode_scheme(int idx) {
  static DATA_SCHEME scheme = {"output", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
RT_DATA_INFO* Get_rt_data_info() { return NULL; }
int Get_output_count() { return 1; }
int Get_input_count() { return 1; }

int main() {
  Prepare_context();
  double x[LEN];
  for (int i = 0; i < LEN; ++i) x[i] = sin(0.37 * i) * 0.9;
  TENSOR* t = Alloc_tensor(1, 1, 1, LEN, x);
  Prepare_input(t, "input");
  Free_tensor(t);
  Run_main_graph();
  double* r = Handle_output("output");
  Finalize_context();
  int bad = 0;
  double max_err = 0;
  for (int i = 0; i < LEN - 3; ++i) {
    /* 2x<<1 + 2x<<2 + 2x<<3 + 4x<<1 + 3x<<1 + 2x<<1 */
    double expect = 11.0 * x[i + 1] + 2.0 * x[i + 2] + 2.0 * x[i + 3];
    double err = fabs(r[i] - expect);
    if (err > max_err) max_err = err;
    if (err > 1e-3) {
      if (bad < 5) printf("index %d: %f != %f\n", i, r[i], expect);
      ++bad;
    }
    printf("slot %d = %.17g\n", i, r[i]);
  }
  free(r);
  printf("max_err = %.3e\n", max_err);
  printf(bad ? "FAILED!\n" : "SUCESS!\n");
  return bad ? 1 : 0;
}
