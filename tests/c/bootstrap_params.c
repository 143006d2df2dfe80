/* bootstrap_params.c -- Bootstrap at arbitrary CKKS parameters through the drop-in API (our own program):
 * encrypt, burn levels with (x * 1.0, rescale) down to 2 limbs, Bootstrap(level_after), decrypt, compare.
 * usage: bootstrap_params N mul_depth q0_bits sf_bits dnum hamming slots level_after
 * The reference's own bootstrap examples only cover q0=60/Delta=51 at N=16 (eg_fhertlib_bootstrap*.inc); the
 * generated ResNets run q0=51 with Delta=50 (ResNet-20/32) and Delta=48 (ResNet-110). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

static CKKS_PARAMS Parm;
static uint32_t Slots, Level_after;

bool Main_graph() {
  CIPHERTEXT cur = Get_input_data("input", 0);
  float* one = (float*)malloc(sizeof(float) * Slots);
  for (uint32_t i = 0; i < Slots; ++i) one[i] = 1.0f;
  while (Level(&cur) > 2) {
    PLAINTEXT pt;
    CIPHERTEXT mp, rs;
    memset(&pt, 0, sizeof(pt));
    memset(&mp, 0, sizeof(mp));
    memset(&rs, 0, sizeof(rs));
    Encode_plain_from_float(&pt, one, Slots, Sc_degree(&cur), Level(&cur));
    Mul_plain(&mp, &cur, &pt);
    Rescale_ciph(&rs, &mp);
    Free_poly_data(&pt._poly);
    Free_ciph_poly(&mp, 1);
    Free_ciph_poly(&cur, 1);
    cur = rs;
  }
  free(one);
  CIPHERTEXT out;
  memset(&out, 0, sizeof(out));
  Bootstrap(&out, &cur, Level_after);
  printf("level before %zu, after %zu (asked %u)\n", (size_t)Level(&cur), (size_t)Level(&out), Level_after);
  Set_output_data("output", 0, &out);
  Free_ciph_poly(&cur, 1);
  return true;
}

CKKS_PARAMS* Get_context_params() { return &Parm; }
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

int main(int argc, char** argv) {
  if (argc < 9) {
    fprintf(stderr, "usage: %s N mul_depth q0_bits sf_bits dnum hamming slots level_after\n", argv[0]);
    return 2;
  }
  memset(&Parm, 0, sizeof(Parm));
  Parm._provider = LIB_ANT;
  Parm._poly_degree = (uint32_t)atoi(argv[1]);
  Parm._mul_depth = (size_t)atoi(argv[2]);
  Parm._first_mod_size = (size_t)atoi(argv[3]);
  Parm._scaling_mod_size = (size_t)atoi(argv[4]);
  Parm._num_q_parts = (size_t)atoi(argv[5]);
  Parm._hamming_weight = (size_t)atoi(argv[6]);
  Slots = (uint32_t)atoi(argv[7]);
  Level_after = (uint32_t)atoi(argv[8]);
  Prepare_context();
  double* x = (double*)malloc(sizeof(double) * Slots);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = sin(0.37 * i) * 0.5;
  TENSOR* t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "input");
  Free_tensor(t);
  Run_main_graph();
  double* r = Handle_output("output");
  Finalize_context();
  int bad = 0;
  double max_err = 0;
  for (uint32_t i = 0; i < Slots; ++i) {
    double err = fabs(r[i] - x[i]);
    if (err > max_err) max_err = err;
    if (err > 1e-3) {
      if (bad < 5) printf("index %u: %f != %f\n", i, r[i], x[i]);
      ++bad;
    }
  }
  free(r);
  free(x);
  printf("max_err = %.3e\n", max_err);
  printf(bad ? "FAILED!\n" : "SUCESS!\n");
  return bad ? 1 : 0;
}
