/* lazy_fills.c -- our own program against the rt_ant drop-in API (no reference sources involved): the places where a zero
 * fill that the runtime defers (csrc/rt/rt_poly.cpp "lazy zero fills") could be seen late.
 *  1. the generated pattern: Init_ciph_same_scale zero-fills an accumulator, rotations (key-switches: direct launches) lie
 *     before its first addend;
 *  2. a zero-filled ciphertext is itself the INPUT of direct launches (key-switch of zeros, then used as an addend);
 *  3. a zero-filled ciphertext is freed while its fill is still waiting, and its memory is reused by a later result;
 *  4. per-limb Hw_modadd straight on a zero-filled polynomial after a rotation;
 *  5. zero-filled polynomials go straight into the direct launches of the polynomial-level API (Rescale, Mod_down,
 *     Decomp_modup): their results are zero polynomials, added limb by limb to the output they must not change it.
 * Every Init / Alloc gets memory that held other residues before (dirty_pool), so a fill that is missing shows.
 * Output = rot1(x) + rot2(x) + x + rot3(x) + x, checked against the clear computation; every slot is printed with %.17g so that
 * the test can compare runs with ACEHIP_LAZY_ZERO=0 and =1 under the same ACEHIP_SEED bit for bit. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#define LEN 64

/* leaves polynomial-sized blocks full of non-zero residues in the runtime's pool, so that the next Init_ciph_* gets memory
 * that is NOT zero by accident (a fresh device allocation usually is): a fill that comes late is then visible */
static void dirty_pool(CIPHER in) {
  CIPHERTEXT d1, d2, d3, r;
  memset(&d1, 0, sizeof(d1));
  memset(&d2, 0, sizeof(d2));
  memset(&d3, 0, sizeof(d3));
  memset(&r, 0, sizeof(r));
  Copy_ciph(&d1, in);
  Copy_ciph(&d2, in);
  Copy_ciph(&d3, in);
  Rotate_ciph(&r, in, 1); /* hands the queue over: the copies are written */
  Free_ciph_poly(&d1, 1);
  Free_ciph_poly(&d2, 1);
  Free_ciph_poly(&d3, 1);
  Rotate_ciph(&r, in, 1); /* ... and the freed blocks are back in the pool */
  Free_ciph_poly(&r, 1);
}

bool Main_graph() {
  CIPHERTEXT in = Get_input_data("input", 0);
  CIPHERTEXT acc, r1, r2, r3, z, zr, same, gone, reuse, limbwise, out;
  memset(&acc, 0, sizeof(acc));
  memset(&r1, 0, sizeof(r1));
  memset(&r2, 0, sizeof(r2));
  memset(&r3, 0, sizeof(r3));
  memset(&z, 0, sizeof(z));
  memset(&zr, 0, sizeof(zr));
  memset(&same, 0, sizeof(same));
  memset(&gone, 0, sizeof(gone));
  memset(&reuse, 0, sizeof(reuse));
  memset(&limbwise, 0, sizeof(limbwise));
  memset(&out, 0, sizeof(out));
  uint32_t degree = Degree();
  /* 1 */
  dirty_pool(&in);
  Init_ciph_same_scale(&acc, &in, &in); /* zero fill */
  Rotate_ciph(&r1, &in, 1);             /* the fill waits across the key-switch */
  Add_ciph(&acc, &acc, &r1);            /* first addend */
  Rotate_ciph(&r2, &in, 2);
  Add_ciph(&acc, &acc, &r2);            /* acc = rot1 + rot2 */
  /* 2 */
  dirty_pool(&in);
  Init_ciph_same_scale(&z, &in, &in);   /* zero */
  Rotate_ciph(&zr, &z, 1);              /* the key-switch READS the zero-filled c1: the fill has to be there */
  Add_ciph(&same, &in, &zr);            /* == in (up to the key-switch noise of a zero polynomial: none) */
  Add_ciph(&acc, &acc, &same);          /* acc = rot1 + rot2 + x */
  /* 3 */
  dirty_pool(&in);
  Init_ciph_same_scale(&gone, &in, &in);
  Rotate_ciph(&r3, &in, 3);             /* the fill of `gone` waits */
  Free_ciph_poly(&gone, 1);             /* ... and dies with its block */
  Copy_ciph(&reuse, &r3);               /* a later result may get that block: no stray fill may land on it */
  Rotate_ciph(&r1, &in, 1);             /* more hand-overs while `reuse` is live */
  Add_ciph(&acc, &acc, &reuse);         /* acc = rot1 + rot2 + x + rot3 */
  /* 4 */
  dirty_pool(&in);
  Init_ciph_same_scale(&limbwise, &in, &in);
  Rotate_ciph(&r2, &in, 2);
  MODULUS* m = Q_modulus();
  for (uint32_t i = 0; i < Poly_level(&limbwise._c0_poly); ++i, ++m) {
    Hw_modadd(Coeffs(&limbwise._c0_poly, i, degree), Coeffs(&limbwise._c0_poly, i, degree), Coeffs(&in._c0_poly, i, degree), m, degree);
    Hw_modadd(Coeffs(&limbwise._c1_poly, i, degree), Coeffs(&in._c1_poly, i, degree), Coeffs(&limbwise._c1_poly, i, degree), m, degree);
  }
  /* 5 */
  dirty_pool(&in);
  size_t lv = Poly_level(&in._c0_poly);
  POLY zq = Alloc_poly(degree, lv, false);      /* zero-filled */
  POLY rs = Alloc_poly(degree, lv, false);
  Rescale(rs, zq);                               /* READS zq: rs = 0 at level lv - 1 */
  POLY ze = Alloc_poly(degree, lv, true);       /* zero-filled, with the p-limbs */
  POLY md = Alloc_poly(degree, lv, false);
  Mod_down(md, ze);                              /* md = 0 */
  POLY ex = Alloc_poly(degree, lv, true);
  Decomp_modup(ex, zq, 0);                       /* digit 0 of the zero polynomial, raised: 0 */
  m = Q_modulus();
  for (uint32_t i = 0; i < lv; ++i, ++m) {
    int64_t* c0 = Coeffs(&limbwise._c0_poly, i, degree);
    if (i + 1 < lv) Hw_modadd(c0, c0, Coeffs(rs, i, degree), m, degree);
    Hw_modadd(c0, c0, Coeffs(md, i, degree), m, degree);
    Hw_modadd(c0, c0, Coeffs(ex, i, degree), m, degree);
  }
  Add_ciph(&out, &acc, &limbwise);      /* + x */
  Free_poly(zq);
  Free_poly(rs);
  Free_poly(ze);
  Free_poly(md);
  Free_poly(ex);
  Set_output_data("output", 0, &out);
  Free_ciph_poly(&in, 1);
  Free_ciph_poly(&acc, 1);
  Free_ciph_poly(&r1, 1);
  Free_ciph_poly(&r2, 1);
  Free_ciph_poly(&r3, 1);
  Free_ciph_poly(&z, 1);
  Free_ciph_poly(&zr, 1);
  Free_ciph_poly(&same, 1);
  Free_ciph_poly(&reuse, 1);
  Free_ciph_poly(&limbwise, 1);
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
    double expect = x[i + 1] + x[i + 2] + x[i + 3] + 2 * x[i];
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
