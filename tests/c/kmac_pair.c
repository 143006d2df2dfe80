/* kmac_pair.c -- our own program against the rt_ant drop-in API (no reference sources involved): the hazards of running a Mod_down
 * pair on the raised digits and key parts themselves when the key inner product that feeds it is still queued (csrc/rt/rt_poly.cpp
 * keymac_pair_from_queue; N = 2^16, where the accumulators are then never stored).  Every rotation below is spelled the way ACE-generated
 * code spells it (resnet20_cifar10_pre.onnx.inc:6990-7060); the variants differ in what happens around the inner product, and each is built
 * so that taking the shortcut wrongly changes bits of the output:
 *  0. the plain generated form (the shortcut is expected to be taken; two digits at the top level, one lower down);
 *  1. the accumulators are READ after the Mod_down pair (queued ops add them to the output, and a further Mod_down reads one) -- the
 *     products and additions that were left queued must still reach memory;
 *  2. a raised digit limb is rewritten between the inner product and the Mod_down pair -- the sums saw the OLD value;
 *  3. the zero fills of the accumulators have already run when the inner product is queued (Acehip_rt_sync in between);
 *  4. an accumulator starts from a value (acc = acc + x before the inner product) instead of zero;
 *  5. a product is formed into the scratch limb, the scratch limb's operand is rewritten, THEN the product is added.
 * Output slots are printed with %.17g: the test compares runs with ACEHIP_KMAC_SHIM=1 / 0, under ACEHIP_POISON=1 and with three images per
 * launch under one ACEHIP_SEED bit for bit, and checks the clear computation to 1e-3. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#define LEN 64
void Acehip_rt_sync(void);

enum { AS_GENERATED = 0, READ_ACC = 1, REWRITE_DIGIT = 2, FILLS_RAN = 3, ACC_PRELOADED = 4, REWRITE_BEFORE_ADD = 5 };

/* rot_idx-rotation of ciph in the generated form, with the variant's disturbance.  extra (READ_ACC): receives
 * Mod_down(k0) once more, read AFTER the pair; the result of REWRITE_DIGIT / ACC_PRELOADED / REWRITE_BEFORE_ADD is not a rotation of
 * anything meaningful -- both builds of the run must simply agree on its bits */
static CIPHERTEXT rotate_variant(CIPHERTEXT ciph, int32_t rot_idx, int variant, POLY extra) {
  CIPHERTEXT res;
  uint32_t degree = Degree();
  memset(&res, 0, sizeof(res));
  Init_ciph_same_scale(&res, &ciph, 0);
  size_t lv = Poly_level(&ciph._c1_poly);
  POLY k0 = Alloc_poly(degree, lv, 1), k1 = Alloc_poly(degree, lv, 1), ext = Alloc_poly(degree, lv, 1);
  POLY tmp = Alloc_poly(degree, 1, 0), d0 = Alloc_poly(degree, lv, 0), d1 = Alloc_poly(degree, lv, 0);
  if (variant == FILLS_RAN) Acehip_rt_sync(); /* the zero fills of k0 / k1 are on the device now, not in the queue */
  if (variant == ACC_PRELOADED) {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < lv; ++i, ++m)
      Hw_modadd(Coeffs(k0, i, degree), Coeffs(k0, i, degree), Coeffs(&ciph._c0_poly, i, degree), m, degree);
  }
  SW_KEY swk = Swk(1, rot_idx);
  for (uint32_t part = 0; part < Num_decomp(&ciph._c1_poly); ++part) {
    Decomp_modup(ext, &ciph._c1_poly, part);
    POLY key0 = Pk0_at(swk, part), key1 = Pk1_at(swk, part);
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(ext); ++i, ++m) {
      Hw_modmul(Coeffs(tmp, 0, degree), Coeffs(key0, i, degree), Coeffs(ext, i, degree), m, degree);
      if (variant == REWRITE_BEFORE_ADD && part == 0 && i == 1)
        Hw_modadd(Coeffs(ext, i, degree), Coeffs(ext, i, degree), Coeffs(ext, i, degree), m, degree); /* after the product, before its addition */
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
    if (variant == REWRITE_DIGIT && part + 1 == Num_decomp(&ciph._c1_poly)) { /* the digit changes AFTER the sums have read it */
      MODULUS* mp = P_modulus();
      Hw_modadd(Coeffs(ext, p_ofst, degree), Coeffs(ext, p_ofst, degree), Coeffs(ext, p_ofst, degree), mp, degree);
      MODULUS* mq = Q_modulus();
      Hw_modadd(Coeffs(ext, 0, degree), Coeffs(ext, 0, degree), Coeffs(ext, 0, degree), mq, degree);
    }
  }
  Mod_down(d0, k0);
  Mod_down(d1, k1);
  if (variant == READ_ACC) {
    MODULUS* m = Q_modulus(); /* d1 += k0 (q-limbs of the accumulator itself), read by queued ops after the pair */
    for (uint32_t i = 0; i < lv; ++i, ++m) Hw_modadd(Coeffs(d1, i, degree), Coeffs(d1, i, degree), Coeffs(k0, i, degree), m, degree);
    Mod_down(extra, k1); /* an unpaired Mod_down that reads the other accumulator from memory */
  }
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

static void add_into(CIPHER acc, CIPHER x) {
  uint32_t degree = Degree();
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < Poly_level(&acc->_c0_poly); ++i, ++m) {
    Hw_modadd(Coeffs(&acc->_c0_poly, i, degree), Coeffs(&acc->_c0_poly, i, degree), Coeffs(&x->_c0_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&acc->_c1_poly, i, degree), Coeffs(&acc->_c1_poly, i, degree), Coeffs(&x->_c1_poly, i, degree), m, degree);
  }
}

static float g_w[LEN];

bool Main_graph() {
  CIPHERTEXT in = Get_input_data("input", 0);
  uint32_t degree = Degree();
  /* 0: plain rotations at the top level (two digits): out = rot1(x) + rot2(x), the second served from the digits raised for the first */
  CIPHERTEXT out = rotate_variant(in, 1, AS_GENERATED, NULL);
  CIPHERTEXT r2 = rotate_variant(in, 2, AS_GENERATED, NULL);
  add_into(&out, &r2);
  /* 3: the same rotation with the accumulators' fills already executed: out += rot3(x) */
  CIPHERTEXT r3 = rotate_variant(in, 3, FILLS_RAN, NULL);
  add_into(&out, &r3);
  /* one level down (one digit): y = rescale(x * w); out2 = rot1(y) */
  CIPHERTEXT prod, y;
  PLAINTEXT pt;
  memset(&prod, 0, sizeof(prod));
  memset(&y, 0, sizeof(y));
  memset(&pt, 0, sizeof(pt));
  Encode_plain_from_float(&pt, g_w, LEN, 1, Level(&in));
  Init_ciph_up_scale_plain(&prod, &in, &pt);
  {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(&prod._c0_poly); ++i, ++m) {
      Hw_modmul(Coeffs(&prod._c0_poly, i, degree), Coeffs(&in._c0_poly, i, degree), Coeffs(&pt._poly, i, degree), m, degree);
      Hw_modmul(Coeffs(&prod._c1_poly, i, degree), Coeffs(&in._c1_poly, i, degree), Coeffs(&pt._poly, i, degree), m, degree);
    }
  }
  Init_ciph_down_scale(&y, &prod);
  Rescale(&y._c0_poly, &prod._c0_poly);
  Rescale(&y._c1_poly, &prod._c1_poly);
  CIPHERTEXT out2 = rotate_variant(y, 1, AS_GENERATED, NULL);
  /* 1, 2, 4, 5: the disturbed variants; what they compute is not meaningful, both runs must agree on the bits.  Their results are folded
   * into a third output so that every bit of them reaches a printed slot */
  POLY extra = Alloc_poly(degree, Poly_level(&y._c1_poly), 0);
  CIPHERTEXT v1 = rotate_variant(y, 2, READ_ACC, extra);
  CIPHERTEXT v2 = rotate_variant(y, 3, REWRITE_DIGIT, NULL);
  CIPHERTEXT v4 = rotate_variant(y, 4, ACC_PRELOADED, NULL);
  CIPHERTEXT v5 = rotate_variant(y, 5, REWRITE_BEFORE_ADD, NULL);
  CIPHERTEXT out3;
  memset(&out3, 0, sizeof(out3));
  Copy_ciph(&out3, &v1);
  add_into(&out3, &v2);
  add_into(&out3, &v4);
  add_into(&out3, &v5);
  {
    MODULUS* m = Q_modulus();
    for (uint32_t i = 0; i < Poly_level(&out3._c0_poly); ++i, ++m)
      Hw_modadd(Coeffs(&out3._c0_poly, i, degree), Coeffs(&out3._c0_poly, i, degree), Coeffs(extra, i, degree), m, degree);
  }
  Set_output_data("output", 0, &out);
  Set_output_data("output2", 0, &out2);
  Set_output_data("output3", 0, &out3);
  Free_poly(extra);
  Free_ciph_poly(&in, 1);
  Free_ciph_poly(&r2, 1);
  Free_ciph_poly(&r3, 1);
  Free_ciph_poly(&prod, 1);
  Free_ciph_poly(&y, 1);
  Free_ciph_poly(&v1, 1);
  Free_ciph_poly(&v2, 1);
  Free_ciph_poly(&v4, 1);
  Free_ciph_poly(&v5, 1);
  Free_plain(&pt);
  return true;
}

CKKS_PARAMS* Get_context_params() {
  static CKKS_PARAMS parm = {LIB_ANT, 65536, 0, 4, 60, 50, 2, 192, 5, {1, 2, 3, 4, 5}};
  return &parm;
}
DATA_SCHEME* Get_encode_scheme(int idx) {
  static DATA_SCHEME scheme = {"input", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
DATA_SCHEME* Get_decode_scheme(int idx) {
  static DATA_SCHEME s0 = {"output", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  static DATA_SCHEME s1 = {"output2", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  static DATA_SCHEME s2 = {"output3", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return idx == 0 ? &s0 : idx == 1 ? &s1 : &s2;
}
RT_DATA_INFO* Get_rt_data_info() { return NULL; }
int Get_output_count() { return 3; }
int Get_input_count() { return 1; }

int main() {
  for (int i = 0; i < LEN; ++i) g_w[i] = (float)(0.25 + 0.5 * cos(0.11 * i));
  Prepare_context();
  double x[LEN];
  for (int i = 0; i < LEN; ++i) x[i] = sin(0.37 * i) * 0.9;
  TENSOR* t = Alloc_tensor(1, 1, 1, LEN, x);
  Prepare_input(t, "input");
  Free_tensor(t);
  Run_main_graph();
  double* r = Handle_output("output");
  double* r2 = Handle_output("output2");
  double* r3 = Handle_output("output3");
  Finalize_context();
  int bad = 0;
  double max_err = 0;
  for (int i = 0; i < LEN - 4; ++i) {
    double e1 = x[i + 1] + x[i + 2] + x[i + 3], e2 = (double)g_w[i + 1] * x[i + 1];
    double err = fmax(fabs(r[i] - e1), fabs(r2[i] - e2));
    if (err > max_err) max_err = err;
    if (err > 1e-3) {
      if (bad < 5) printf("index %d: %f != %f or %f != %f\n", i, r[i], e1, r2[i], e2);
      ++bad;
    }
    printf("slot %d = %.17g %.17g %.17g\n", i, r[i], r2[i], r3[i]);
  }
  free(r);
  free(r2);
  free(r3);
  printf("max_err = %.3e\n", max_err);
  printf(bad ? "FAILED!\n" : "SUCESS!\n");
  return bad ? 1 : 0;
}
