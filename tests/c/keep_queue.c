/* keep_queue.c -- our own program against the rt_ant drop-in API (no reference sources involved): the hazards of per-limb ops
 * that STAY QUEUED across a direct launch whose operand list says it does not need them (csrc/rt/rt_poly.cpp "keeping ops queued").
 * The program is written the way ACE-generated code is (per-limb Hw_* loops, Decomp_modup / Mod_down pairs, Hw_rotate, temporaries
 * freed right behind the loops that use them), and every part is built so that an op left behind wrongly -- or taken along in the
 * wrong order -- changes bits of the output:
 *  1. convolution taps with a rotation between two accumulations: acc += pt_k * rot_k(x); the products and the accumulations of
 *     tap k wait while the key-switch of tap k + 1 runs, the temporaries of tap k are freed (pinned in the pool until their readers
 *     have run) and new blocks are allocated in between;
 *  2. write-after-read: queued ops READ a polynomial that the next direct launch (Rescale) REWRITES -- they must run first;
 *  3. read-after-write: queued ops WRITE a polynomial that the next direct launch (Rescale) READS;
 *  4. write-after-write: a queued op writes a limb, a direct launch rewrites the whole polynomial, a queued op reads it -- the
 *     reader must see the launch's value, so the first writer may not be left behind the launch;
 *  5. a kept reader of a block that is freed, while later allocations and direct launches go on (the block may not be recycled).
 * Output slots are printed with %.17g: the test compares runs with ACEHIP_HW_KEEP=0 / 1, ACEHIP_HW_STAGES=0 / 1 and ACEHIP_POISON=1
 * under one ACEHIP_SEED bit for bit, and checks the clear computation to 1e-3.  With ACEHIP_POISON_SELFTEST=1 one input of the paired
 * Mod_down is left out of its declared list on purpose: the ops that produce it stay queued, and the poison check must abort. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#define LEN 64
#define TAPS 4

/* the generated form of a rotation: raise the digits of c1, inner product with the rotation key limb by limb through ONE scratch
 * limb, Mod_down of both accumulators, + c0, automorphism */
static CIPHERTEXT rotate_like_generated_code(CIPHERTEXT ciph, int32_t rot_idx) {
  CIPHERTEXT res;
  uint32_t degree = Degree();
  memset(&res, 0, sizeof(res));
  Init_ciph_same_scale(&res, &ciph, 0);
  size_t lv = Poly_level(&ciph._c1_poly);
  POLY k0 = Alloc_poly(degree, lv, 1), k1 = Alloc_poly(degree, lv, 1), ext = Alloc_poly(degree, lv, 1);
  POLY tmp = Alloc_poly(degree, 1, 0), d0 = Alloc_poly(degree, lv, 0), d1 = Alloc_poly(degree, lv, 0);
  SW_KEY swk = Swk(1, rot_idx);
  for (uint32_t part = 0; part < Num_decomp(&ciph._c1_poly); ++part) {
    Decomp_modup(ext, &ciph._c1_poly, part);
    POLY key0 = Pk0_at(swk, part), key1 = Pk1_at(swk, part);
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(ext); ++i, ++m) {
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key0, i, degree), Coeffs(ext, i, degree), m, degree);
      Hw_modadd(Coeffs(k0, i, degree), Coeffs(k0, i, degree), Coeffs(tmp, 0, degree), m, degree);
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key1, i, degree), Coeffs(ext, i, degree), m, degree);
      Hw_modadd(Coeffs(k1, i, degree), Coeffs(k1, i, degree), Coeffs(tmp, 0, degree), m, degree);
    }
    m = P_modulus();
    uint32_t p_ofst = Num_alloc(ext) - Num_p(ext), key_p_ofst = Poly_level(key0);
    for (uint32_t i = 0; i < Num_p(ext); ++i, ++m) {
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key0, i + key_p_ofst, degree), Coeffs(ext, i + p_ofst, degree), m, degree);
      Hw_modadd(Coeffs(k0, i + p_ofst, degree), Coeffs(k0, i + p_ofst, degree), Coeffs(tmp, 0, degree), m, degree);
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key1, i + key_p_ofst, degree), Coeffs(ext, i + p_ofst, degree), m, degree);
      Hw_modadd(Coeffs(k1, i + p_ofst, degree), Coeffs(k1, i + p_ofst, degree), Coeffs(tmp, 0, degree), m, degree);
    }
  }
  Mod_down(d0, k0);
  Mod_down(d1, k1);
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < lv; ++i, ++m) Hw_modadd(Coeffs(d0, i, degree), Coeffs(d0, i, degree), Coeffs(&ciph._c0_poly, i, degree), m, degree);
  int64_t* order = Auto_order(rot_idx);
  m = Q_modulus();
  for (uint32_t i = 0; i < lv; ++i, ++m) {
    Hw_rotate(Coeffs(&res._c0_poly, i, degree), Coeffs(d0, i, degree), order, m, degree);
    Hw_rotate(Coeffs(&res._c1_poly, i, degree), Coeffs(d1, i, degree), order, m, degree);
  }
  Free_poly(k0);
  Free_poly(k1);
  Free_poly(ext);
  Free_poly(tmp);
  Free_poly(d0);
  Free_poly(d1);
  return res;
}

/* res (+)= a * plain, limb by limb (res is zero-filled by Init when first) */
static void mul_plain_acc(CIPHER acc, CIPHER a, PLAIN pt, CIPHER scratch) {
  uint32_t degree = Degree();
  Init_ciph_up_scale_plain(scratch, a, pt);
  Init_ciph_same_scale(acc, acc, scratch);
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < Poly_level(&acc->_c0_poly); ++i, ++m) {
    Hw_modmul(Coeffs(&scratch->_c0_poly, i, degree), Coeffs(&a->_c0_poly, i, degree), Coeffs(&pt->_poly, i, degree), m, degree);
    Hw_modmul(Coeffs(&scratch->_c1_poly, i, degree), Coeffs(&a->_c1_poly, i, degree), Coeffs(&pt->_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&acc->_c0_poly, i, degree), Coeffs(&acc->_c0_poly, i, degree), Coeffs(&scratch->_c0_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&acc->_c1_poly, i, degree), Coeffs(&acc->_c1_poly, i, degree), Coeffs(&scratch->_c1_poly, i, degree), m, degree);
  }
}

static float g_w[TAPS][LEN];

bool Main_graph() {
  CIPHERTEXT in = Get_input_data("input", 0);
  CIPHERTEXT acc, scratch, rs, war, waw, pinned, sum, out;
  PLAINTEXT pt;
  memset(&acc, 0, sizeof(acc));
  memset(&scratch, 0, sizeof(scratch));
  memset(&rs, 0, sizeof(rs));
  memset(&war, 0, sizeof(war));
  memset(&waw, 0, sizeof(waw));
  memset(&pinned, 0, sizeof(pinned));
  memset(&sum, 0, sizeof(sum));
  memset(&out, 0, sizeof(out));
  memset(&pt, 0, sizeof(pt));
  uint32_t degree = Degree();
  /* 1: acc = sum_k w_k * rot_{k+1}(x); the rotation of tap k + 1 runs while tap k's products and accumulations wait */
  for (int k = 0; k < TAPS; ++k) {
    CIPHERTEXT r = rotate_like_generated_code(in, k + 1);
    Encode_plain_from_float(&pt, g_w[k], LEN, 1, Level(&r));
    mul_plain_acc(&acc, &r, &pt, &scratch);
    Free_poly_data(&r._c1_poly); /* freed while the product that reads it is still queued */
    Free_poly_data(&r._c0_poly);
  }
  /* 2 + 3: `war` = copy of acc made by queued ops (reads acc, writes war); Rescale then READS acc ... */
  Copy_ciph(&war, &acc);
  Init_ciph_down_scale(&rs, &acc);
  Rescale(&rs._c0_poly, &acc._c0_poly);
  Rescale(&rs._c1_poly, &acc._c1_poly);
  /* ... and now acc is REWRITTEN by a direct launch path while queued readers of it may still wait: acc2 = rot(war') lands in acc's
   * old memory when the pool hands it out again -- force it: free acc, allocate same-size blocks, fill them through a launch */
  Free_poly_data(&acc._c1_poly);
  Free_poly_data(&acc._c0_poly);
  /* 4: write-after-write on `waw`: queued writer, then a direct launch (Rescale) rewrites it, then a queued reader */
  Init_ciph_down_scale(&waw, &war);
  {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i + 1 < Poly_level(&war._c0_poly); ++i, ++m) { /* queued: waw = war + war on the limbs that survive */
      Hw_modadd(Coeffs(&waw._c0_poly, i, degree), Coeffs(&war._c0_poly, i, degree), Coeffs(&war._c0_poly, i, degree), m, degree);
      Hw_modadd(Coeffs(&waw._c1_poly, i, degree), Coeffs(&war._c1_poly, i, degree), Coeffs(&war._c1_poly, i, degree), m, degree);
    }
  }
  Rescale(&waw._c0_poly, &war._c0_poly); /* direct: overwrites what the queued ops wrote; waw = rescale(war) = rs */
  Rescale(&waw._c1_poly, &war._c1_poly);
  /* 5: a reader of `pinned` stays queued while pinned is freed and further launches allocate and write blocks */
  Copy_ciph(&pinned, &rs);
  Init_ciph_same_scale(&sum, &rs, &waw);
  {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(&sum._c0_poly); ++i, ++m) { /* sum = pinned + waw  (= 2 rs) */
      Hw_modadd(Coeffs(&sum._c0_poly, i, degree), Coeffs(&pinned._c0_poly, i, degree), Coeffs(&waw._c0_poly, i, degree), m, degree);
      Hw_modadd(Coeffs(&sum._c1_poly, i, degree), Coeffs(&pinned._c1_poly, i, degree), Coeffs(&waw._c1_poly, i, degree), m, degree);
    }
  }
  Free_poly_data(&pinned._c1_poly);
  Free_poly_data(&pinned._c0_poly);
  CIPHERTEXT r5 = rotate_like_generated_code(rs, 5); /* launches + allocations of pinned's size while `sum` waits */
  Init_ciph_same_scale(&out, &sum, &r5);
  {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(&out._c0_poly); ++i, ++m) { /* out = sum + rot5(rs) */
      Hw_modadd(Coeffs(&out._c0_poly, i, degree), Coeffs(&sum._c0_poly, i, degree), Coeffs(&r5._c0_poly, i, degree), m, degree);
      Hw_modadd(Coeffs(&out._c1_poly, i, degree), Coeffs(&sum._c1_poly, i, degree), Coeffs(&r5._c1_poly, i, degree), m, degree);
    }
  }
  Set_output_data("output", 0, &out);
  Free_ciph_poly(&in, 1);
  Free_ciph_poly(&scratch, 1);
  Free_ciph_poly(&rs, 1);
  Free_ciph_poly(&war, 1);
  Free_ciph_poly(&waw, 1);
  Free_ciph_poly(&sum, 1);
  Free_ciph_poly(&r5, 1);
  Free_plain(&pt);
  return true;
}

CKKS_PARAMS* Get_context_params() {
  static CKKS_PARAMS parm = {LIB_ANT, 16384, 0, 4, 60, 50, 2, 192, 5, {1, 2, 3, 4, 5}};
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
  for (int k = 0; k < TAPS; ++k)
    for (int i = 0; i < LEN; ++i) g_w[k][i] = (float)(0.25 + 0.5 * cos(0.11 * i + k));
  Prepare_context();
  double x[LEN];
  for (int i = 0; i < LEN; ++i) x[i] = sin(0.37 * i) * 0.9;
  TENSOR* t = Alloc_tensor(1, 1, 1, LEN, x);
  Prepare_input(t, "input");
  Free_tensor(t);
  Run_main_graph();
  double* r = Handle_output("output");
  Finalize_context();
  /* clear computation: a[i] = sum_k w_k[i] * x[i + k + 1]; out[i] = 2 a[i] + a[i + 5] */
  double a[LEN];
  for (int i = 0; i < LEN; ++i) {
    a[i] = 0;
    for (int k = 0; k < TAPS; ++k)
      if (i + k + 1 < LEN) a[i] += (double)g_w[k][i] * x[i + k + 1];
  }
  int bad = 0;
  double max_err = 0;
  for (int i = 0; i < LEN - 5 - TAPS; ++i) {
    double expect = 2 * a[i] + a[i + 5];
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
