/* api_extras.c -- the operators and helpers of the reference's rt_ant surface that no checked-in generated program calls
 * (include/ckks/cipher_eval.h Upscale_ciph / Downscale_ciph / Get_msg_with_imag / Print_cipher_*, include/ckks/plain_eval.h
 * Encode_plain_from_float_with_scale, include/ckks/cipher_valid.h <op>_msg), as a bit-level cross-check.  Test infrastructure, our own
 * program; ONE source for both runtimes, nothing but the public API:
 *   reference build (oracle/_ref/examples/refgen_api_extras: reference rtlib + tests/c/gen_parity_ref.c, keys of ACEHIP_SEED injected):
 *       writes every result ciphertext / plaintext and the text the diagnostics print;
 *   product build (libFHErt_ant.so): the same on the GPU; tests/test_gpu_gen_parity.py compares the sha256 of the files with the committed
 *       digests of the reference run and the diagnostic text line by line.
 * usage: api_extras DIR N mul_depth q0_bits sf_bits dnum
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"
#include "common/io_api.h"    /* the input / output table behind Prepare_input / Set_output_data */
#include "common/rt_config.h"

#ifdef REF_BUILD
#include "rtlib/context.h"
#include "util/ckks_key_generator.h"
#include "util/ckks_parameters.h"
#include "util/crt.h"
typedef unsigned long long u64;
#include "ref_containers.h"
#else
static void save_ciph(const char* path, CIPHER c) { Acehip_rt_save_ciph(path, c); }
static void save_plain(const char* path, PLAIN c) { Acehip_rt_save_plain(path, c); }
#endif

#define ZERO(x) memset(&(x), 0, sizeof(x))
static CKKS_PARAMS* Parm;
static const char*  Dir;
static uint32_t     Slots;
static FILE*        Txt;

static void out_ciph(const char* name, CIPHER c) {
  char p[1024];
  snprintf(p, sizeof p, "%s/%s.ct", Dir, name);
  save_ciph(p, c);
}
static void out_plain(const char* name, PLAIN c) {
  char p[1024];
  snprintf(p, sizeof p, "%s/%s.ct", Dir, name);
  save_plain(p, c);
}
static void out_vec(const char* name, const double* v, uint32_t n) { /* decoded messages: doubles, written to 12 significant digits */
  fprintf(Txt, "%s:", name);
  for (uint32_t i = 0; i < n; ++i) fprintf(Txt, " %.12e", v[i]);
  fprintf(Txt, "\n");
}

bool         Main_graph() { return true; }
CKKS_PARAMS* Get_context_params() { return Parm; }
DATA_SCHEME* Get_encode_scheme(int idx) {
  static DATA_SCHEME scheme_a = {"in_a", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  static DATA_SCHEME scheme_b = {"in_b", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return idx == 0 ? &scheme_a : &scheme_b;
}
DATA_SCHEME* Get_decode_scheme(int idx) {
  static DATA_SCHEME scheme = {"output", {0, 0, 0, 0}, 1, {NORMAL, 0, 0, 0, 0}};
  return &scheme;
}
RT_DATA_INFO* Get_rt_data_info() { return NULL; }
int           Get_output_count() { return 1; }
int           Get_input_count() { return 2; }

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s DIR N mul_depth q0_bits sf_bits dnum\n", argv[0]);
    return 2;
  }
  Dir  = argv[1];
  Parm = (CKKS_PARAMS*)calloc(1, sizeof(CKKS_PARAMS) + sizeof(int32_t) * 8);
  Parm->_provider         = LIB_ANT;
  Parm->_poly_degree      = (uint32_t)atoi(argv[2]);
  Parm->_mul_depth        = (size_t)atoi(argv[3]);
  Parm->_first_mod_size   = (size_t)atoi(argv[4]);
  Parm->_scaling_mod_size = (size_t)atoi(argv[5]);
  Parm->_num_q_parts      = (size_t)atoi(argv[6]);
  Parm->_num_rot_idx      = 2;
  Parm->_rot_idxs[0]      = 1;
  Parm->_rot_idxs[1]      = -3;
  Slots = Parm->_poly_degree / 2;
  const uint32_t sf_bits = (uint32_t)Parm->_scaling_mod_size;
  char p[1024];
  snprintf(p, sizeof p, "%s/text.txt", Dir);
  Txt = fopen(p, "w");
  if (!Txt) { perror(p); return 3; }
  Prepare_context();
  double* x = (double*)malloc(sizeof(double) * Slots);
  float*  w = (float*)malloc(sizeof(float) * Slots);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = sin(0.37 * i) * 0.5;
  TENSOR* t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_a");
  Free_tensor(t);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = cos(0.23 * i + 1.0) * 0.4;
  t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_b");
  Free_tensor(t);
  for (uint32_t i = 0; i < Slots; ++i) w[i] = (float)(cos(0.11 * i) * 0.75);
  /* the facilities a program sees through rt_ant.h besides the operators: the IO table, run-time switches, timing marks, assertions */
  fprintf(Txt, "io table: in_a %s, in_b %s\n", Io_get_input("in_a", 0) ? "set" : "empty", Io_get_input("in_b", 0) ? "set" : "empty");
  Set_rtlib_config(CONF_BTS_CLEAR_IMAG, 1);
  fprintf(Txt, "config: clear_imag %ld fusion %ld\n", (long)Get_rtlib_config(CONF_BTS_CLEAR_IMAG), (long)Get_rtlib_config(CONF_OP_FUSION_DECOMP_MODUP));
  Set_rtlib_config(CONF_BTS_CLEAR_IMAG, 0);
  RTLIB_TM_START(RTM_PT_GET, mark);
  FMT_ASSERT(RTM_PT_GET + 1 == RTM_LAST && RTM_MAIN_GRAPH == 8, "timing ids differ from the reference's order");
  RTLIB_TM_END(RTM_PT_GET, mark);
  IS_TRUE(Is_trace_on() == false, "trace is off by default");
  CIPHERTEXT a = Get_input_data("in_a", 0), b = Get_input_data("in_b", 0);
  const uint32_t level = (uint32_t)Level(&a);

  /* Encode_plain_from_float_with_scale: a vector and a single value, at a scale that is NOT a power of the scaling factor */
  PLAINTEXT pv, p1;
  ZERO(pv);
  ZERO(p1);
  const double odd_scale = ldexp(1.0, (int)sf_bits - 7) * 3.0;
  Encode_plain_from_float_with_scale(&pv, w, Slots, odd_scale, level);
  out_plain("plain_vec_scaled", &pv);
  float one_val = 0.8125f;
  Encode_plain_from_float_with_scale(&p1, &one_val, 1, ldexp(1.0, (int)sf_bits - 4), level - 1);
  out_plain("plain_val_scaled", &p1);
  CIPHERTEXT mp;
  ZERO(mp);
  Mul_plain(&mp, &a, &pv);
  out_ciph("mul_plain_scaled", &mp);
  fprintf(Txt, "mul_plain_scaled: sf_degree %u level %zu\n", Sc_degree(&mp), Level(&mp));

  /* Upscale_ciph / Downscale_ciph: raise the scale by 2^10, then bring a product back to waterline sf_bits - 5 */
  CIPHERTEXT up, dn;
  ZERO(up);
  ZERO(dn);
  Upscale_ciph(&up, &a, 10);
  out_ciph("upscale", &up);
  fprintf(Txt, "upscale: sf_degree %u level %zu\n", Sc_degree(&up), Level(&up));
  Downscale_ciph(&dn, &up, sf_bits - 5);
  out_ciph("downscale", &dn);
  fprintf(Txt, "downscale: sf_degree %u level %zu\n", Sc_degree(&dn), Level(&dn));

  /* message-level helpers (cipher_valid.h) and diagnostics: text */
  double* m = Add_msg(&a, &b, Slots - 1);
  out_vec("add_msg", m, 8);
  free(m);
  m = Mul_msg(&a, &b);
  out_vec("mul_msg", m, 8);
  free(m);
  m = Rotate_msg(&a, -3);
  out_vec("rotate_msg_m3", m, 8);
  free(m);
  m = Rotate_msg(&a, 1);
  out_vec("rotate_msg_1", m, 8);
  free(m);
  PLAINTEXT pw;
  ZERO(pw);
  Encode_plain_from_float(&pw, w, Slots, 1, level);
  m = Add_plain_msg(&a, &pw);
  out_vec("add_plain_msg", m, 8);
  free(m);
  m = Mul_plain_msg(&a, &pw);
  out_vec("mul_plain_msg", m, 8);
  free(m);
  m = Get_msg(&dn);
  out_vec("downscale_msg", m, 8);
  free(m);
  DCMPLX* z = Get_msg_with_imag(&a);
  fprintf(Txt, "msg_with_imag: %.12e %.12e\n", ((double*)z)[0], ((double*)z)[2]);
  free(z);
  z = Get_dcmplx_msg_from_plain(&pw);
  fprintf(Txt, "dcmplx_from_plain: %.12e %.12e\n", ((double*)z)[0], ((double*)z)[2]);
  free(z);
  Print_cipher_info(Txt, "a", &a);
  Print_cipher_info(Txt, "dn", &dn);
  fprintf(Txt, "key_level %zu q_parts %zu p_cnt %zu\n", Poly_level(Pk0_at(Swk(0, 0), 0)), Get_q_parts(), Get_p_cnt());
  double std_ok[4] = {sin(0.0) * 0.5, sin(0.37) * 0.5, sin(0.74) * 0.5, sin(1.11) * 0.5};
  fflush(Txt);
  Validate(&a, std_ok, 4, -4); /* stdout: "INFO: internal validation pass." */
  std_ok[2] += 1.0;
  Validate(&a, std_ok, 4, -4); /* stdout: "ERROR: internal validation fail." -- and the program goes on */
  fclose(Txt);
  free(x);
  free(w);
  Finalize_context();
  printf("SUCESS! api extras written\n");
  return 0;
}
