/* dropin_c1.c -- our own program against the rt_ant drop-in API (no reference sources involved).
 * BASELINE.json configs[0] ("C1"): single CKKS ciphertext HAdd + HMul(+relin) + Rescale at N=2^14, 4 RNS
 * limbs (q0=60, Delta=50, dnum=2), plus a rotation and a plaintext multiply, checked against the clear
 * computation with the reference's example tolerance (1e-3).  Written in the style of the generated code:
 * poly-level loops for HAdd, ciphertext-level API for the rest. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#define LEN 64

bool Main_graph() {
  CIPHERTEXT in = Get_input_data("input", 0);
  CIPHERTEXT sum, prod, resc, rot, out;
  memset(&sum, 0, sizeof(sum));
  memset(&prod, 0, sizeof(prod));
  memset(&resc, 0, sizeof(resc));
  memset(&rot, 0, sizeof(rot));
  memset(&out, 0, sizeof(out));
  uint32_t degree = Degree();
  /* HAdd spelled per limb like the POLY pass does: sum = in + in */
  Init_ciph_same_scale(&sum, &in, &in);
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < Poly_level(&sum._c0_poly); ++i, ++m) {
    Hw_modadd(Coeffs(&sum._c0_poly, i, degree), Coeffs(&in._c0_poly, i, degree), Coeffs(&in._c0_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&sum._c1_poly, i, degree), Coeffs(&in._c1_poly, i, degree), Coeffs(&in._c1_poly, i, degree), m, degree);
  }
  Mul_ciph(&prod, &sum, &in);          /* 2x * x, relinearised */
  Rescale_ciph(&resc, &prod);          /* back to one Delta */
  Rotate_ciph(&rot, &resc, 1);         /* slot i <- slot i+1 */
  /* plaintext multiply by 0.5 at the ciphertext's level, then rescale */
  PLAINTEXT pt;
  memset(&pt, 0, sizeof(pt));
  float half[LEN];
  for (int i = 0; i < LEN; ++i) half[i] = 0.5f;
  Encode_plain_from_float(&pt, half, LEN, Sc_degree(&rot), Level(&rot));
  CIPHERTEXT mp;
  memset(&mp, 0, sizeof(mp));
  Mul_plain(&mp, &rot, &pt);
  Rescale_ciph(&out, &mp);
  Free_plain(&pt);
  Set_output_data("output", 0, &out);
  Free_ciph_poly(&in, 1);
  Free_ciph_poly(&sum, 1);
  Free_ciph_poly(&prod, 1);
  Free_ciph_poly(&resc, 1);
  Free_ciph_poly(&rot, 1);
  Free_ciph_poly(&mp, 1);
  return true;
}

CKKS_PARAMS* Get_context_params() {
  static CKKS_PARAMS parm = {LIB_ANT, 16384, 0, 3, 60, 50, 2, 192, 1, {1}};
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
  for (int i = 0; i < LEN - 1; ++i) {
    double expect = 0.5 * (2.0 * x[i + 1] * x[i + 1]);
    double err = fabs(r[i] - expect);
    if (err > max_err) max_err = err;
    if (err > 1e-3) {
      if (bad < 5) printf("index %d: %f != %f\n", i, r[i], expect);
      ++bad;
    }
  }
  free(r);
  printf("max_err = %.3e\n", max_err);
  printf(bad ? "FAILED!\n" : "SUCESS!\n");
  return bad ? 1 : 0;
}
