/* ksw_variants.c -- the reference's key-switch optimisation tests (rtlib/ant/unittest/ut_ksw_opt.cxx:115-660) as a bit-level cross-check.
 * Test infrastructure, our own program.  That unit test runs four computations twice -- a BASE form built from whole operators and an
 * OPT form that hoists or merges pieces of the key-switch -- and checks both against the clear result to 5e-3:
 *     1 modup_hoist            sum_i Rotate(ct, r_i)                         opt: ModUp of c1 once (Switch_key_precompute), Fast_rotate per r_i
 *     2 moddown_hoist          sum_i Rotate(ct_i, r_i)                       opt: rotations stay in the extended basis QP (Fast_rotate_ext), ONE ModDown
 *     3 moddown_rescale        Rescale(Mul(ct_a, ct_b))                      opt: relinearise in QP, ModDown from the coefficient domain, then Rescale
 *     4 moddown_rescale_modup  Rotate(Rescale(Rotate(ct, r1) * pt), r2)      opt: first rotation in QP, extended plaintext, ModDown, Rescale, Fast_rotate
 * A GPU runtime is tempted to make exactly these reorderings.  The contract here is the BASE form's bits:
 *   -DREF_BUILD (reference rtlib + tests/c/gen_parity_ref.c: keys and encryption randomness of ACEHIP_SEED injected): runs the base
 *       forms through the rt_ant operator API (cipher_eval.c: Rotate_ciph = Eval_fast_rotate, Add_ciph = Add_ciphertext, Mul_ciph =
 *       Mul_ciphertext with the relinearisation key, Rescale_ciph = Rescale_ciphertext, Mul_plain = Mul_plaintext -- the calls of the unit
 *       test) and writes every result; then runs the OPT forms with the evaluator's internals as the unit test does and reports, per
 *       computation, whether the opt result has the base result's bytes ("opt_vs_base <name> EQUAL|DIFFERENT").
 *   default (libFHErt_ant.so): runs the base forms through the same API on the GPU and writes every result; tests/test_gpu_gen_parity.py
 *       compares the sha256 of the files with the committed digests of the reference run (tests/golden/gen_parity.json "ksw_variants").
 * usage: ksw_variants DIR N mul_depth q0_bits sf_bits dnum        (ACEHIP_SEED / GEN_PARITY_SEED in the environment)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#ifdef REF_BUILD
#include "rtlib/context.h"
#include "util/ckks_encoder.h"
#include "util/ckks_evaluator.h"
#include "util/ckks_key_generator.h"
#include "util/ckks_parameters.h"
#include "util/crt.h"
typedef unsigned long long u64;
#include "ref_containers.h"
#else
static void save_ciph(const char* path, CIPHER c) { Acehip_rt_save_ciph(path, c); }
#endif

#define ZERO(x) memset(&(x), 0, sizeof(x))
static const int32_t Rots[3] = {1, 3, 5};
static CKKS_PARAMS*  Parm;
static const char*   Dir;
static uint32_t      Slots;

static void out_ciph(const char* name, CIPHER c) {
  char p[1024];
  snprintf(p, sizeof p, "%s/%s.ct", Dir, name);
  save_ciph(p, c);
}

/* ---------------------------------------------------------------- base forms: whole operators, both builds ---- */
static void base_modup_hoist(CIPHERTEXT* res, CIPHER ct) { /* ut_ksw_opt.cxx:115-170 */
  CIPHERTEXT acc, rot;
  ZERO(acc);
  for (int i = 0; i < 3; ++i) {
    ZERO(rot);
    Rotate_ciph(&rot, ct, Rots[i]);
    if (i == 0) {
      acc = rot;
    } else {
      Add_ciph(&acc, &rot, &acc);
      Free_ciph_poly(&rot, 1);
    }
  }
  *res = acc;
}
static void base_moddown_hoist(CIPHERTEXT* res, CIPHER cts[3]) { /* :230-310 */
  CIPHERTEXT acc, rot;
  ZERO(acc);
  for (int i = 0; i < 3; ++i) {
    ZERO(rot);
    Rotate_ciph(&rot, cts[i], Rots[i]);
    if (i == 0) {
      acc = rot;
    } else {
      Add_ciph(&acc, &rot, &acc);
      Free_ciph_poly(&rot, 1);
    }
  }
  *res = acc;
}
static void base_moddown_rescale(CIPHERTEXT* res, CIPHER a, CIPHER b) { /* :408-450 */
  CIPHERTEXT mul;
  ZERO(mul);
  ZERO(*res);
  Mul_ciph(&mul, a, b);
  Rescale_ciph(res, &mul);
  Free_ciph_poly(&mul, 1);
}
static void base_moddown_rescale_modup(CIPHERTEXT* res, CIPHER a, double* w) { /* :520-600 */
  CIPHERTEXT rot, mul, rs;
  PLAINTEXT  pt;
  ZERO(rot);
  ZERO(mul);
  ZERO(rs);
  ZERO(pt);
  ZERO(*res);
  Rotate_ciph(&rot, a, Rots[1]);
  Encode_plain_from_double(&pt, w, Slots, 1, (uint32_t)Level(&rot));
  Mul_plain(&mul, &rot, &pt);
  Rescale_ciph(&rs, &mul);
  Rotate_ciph(res, &rs, Rots[2]);
  Free_ciph_poly(&rot, 1);
  Free_ciph_poly(&mul, 1);
  Free_ciph_poly(&rs, 1);
  Free_poly_data(&pt._poly);
}

#ifdef REF_BUILD
/* ---------------------------------------------------------------- opt forms: the evaluator's internals, reference only ---- */
static CKKS_EVALUATOR*     eval(void) { return (CKKS_EVALUATOR*)Get_eval(Context); }
static CKKS_KEY_GENERATOR* keygen(void) { return (CKKS_KEY_GENERATOR*)Get_key_gen(Context); }
static CRT_CONTEXT*        crt(void) { return ((CKKS_PARAMETER*)Get_param(Context))->_crt_context; }
static SWITCH_KEY* rot_key(int32_t r) { return Get_auto_key(keygen(), Get_precomp_auto_idx(keygen(), r)); }

static int same_poly(POLYNOMIAL* a, POLYNOMIAL* b) {
  if (a->_num_primes != b->_num_primes || a->_num_primes_p != b->_num_primes_p || a->_is_ntt != b->_is_ntt) return 0;
  size_t n = a->_ring_degree;
  if (memcmp(a->_data, b->_data, a->_num_primes * n * 8) != 0) return 0;
  return a->_num_primes_p == 0 || memcmp(a->_data + (a->_num_alloc_primes - a->_num_primes_p) * n,
                                         b->_data + (b->_num_alloc_primes - b->_num_primes_p) * n, a->_num_primes_p * n * 8) == 0;
}
static void verdict(const char* name, CIPHER base, CIPHER opt) {
  int eq = same_poly(&base->_c0_poly, &opt->_c0_poly) && same_poly(&base->_c1_poly, &opt->_c1_poly) &&
           base->_scaling_factor == opt->_scaling_factor && base->_sf_degree == opt->_sf_degree;
  double *mb = Get_msg(base), *mo = Get_msg(opt), worst = 0;
  for (uint32_t i = 0; i < Slots; ++i) if (fabs(mb[i] - mo[i]) > worst) worst = fabs(mb[i] - mo[i]);
  printf("opt_vs_base %s %s (decrypted messages differ by at most %.3e)\n", name, eq ? "EQUAL" : "DIFFERENT", worst);
  free(mb);
  free(mo);
}
static void opt_modup_hoist(CIPHERTEXT* res, CIPHER ct) { /* :172-228 */
  CIPHERTEXT* acc = Alloc_ciphertext();
  CIPHERTEXT* rot = Alloc_ciphertext();
  VALUE_LIST* pre = Switch_key_precompute(Get_c1(ct), crt());
  for (int i = 0; i < 3; ++i) {
    Fast_rotate(i == 0 ? acc : rot, ct, Rots[i], rot_key(Rots[i]), eval(), pre);
    if (i) Add_ciphertext(acc, rot, acc, eval());
  }
  Free_switch_key_precomputed(pre);
  Free_ciphertext(rot);
  *res = *acc;
  free(acc);
}
static void opt_moddown_hoist(CIPHERTEXT* res, CIPHER cts[3]) { /* :312-400 */
  CIPHERTEXT* rot = Alloc_ciphertext();
  CIPHERTEXT* sum = Alloc_ciphertext();
  CIPHERTEXT* out = Alloc_ciphertext();
  for (int i = 0; i < 3; ++i) {
    VALUE_LIST* pre = Switch_key_precompute(Get_c1(cts[i]), crt());
    Fast_rotate_ext(rot, cts[i], Rots[i], rot_key(Rots[i]), eval(), pre, true);
    if (i == 0) Init_ciphertext_from_ciph(sum, rot, rot->_scaling_factor, rot->_sf_degree);
    Add_ciphertext(sum, sum, rot, eval()); /* in the extended basis */
    Free_switch_key_precomputed(pre);
  }
  Init_ciphertext_from_ciph(out, cts[0], cts[0]->_scaling_factor, cts[0]->_sf_degree);
  Reduce_rns_base(Get_c0(out), Get_c0(sum), crt()); /* the one ModDown */
  Reduce_rns_base(Get_c1(out), Get_c1(sum), crt());
  Free_ciphertext(rot);
  Free_ciphertext(sum);
  *res = *out;
  free(out);
}
static void opt_moddown_rescale(CIPHERTEXT* res, CIPHER a, CIPHER b) { /* :452-518 */
  CIPHERTEXT3* m3 = Alloc_ciphertext3();
  CIPHERTEXT*  m = Alloc_ciphertext();
  CIPHERTEXT*  red = Alloc_ciphertext();
  CIPHERTEXT*  out = Alloc_ciphertext();
  Mul_ciphertext3(m3, a, b, eval());
  Relinearize_ciph3_ext(m, m3, Get_relin_key(keygen()), eval());
  Conv_ntt2poly_inplace(Get_c0(m), crt());
  Conv_ntt2poly_inplace(Get_c1(m), crt());
  Init_ciphertext_from_ciph(red, a, Get_ciph3_sfactor(m3), Get_ciph3_sf_degree(m3));
  Reduce_rns_base(Get_c0(red), Get_c0(m), crt());
  Reduce_rns_base(Get_c1(red), Get_c1(m), crt());
  Rescale_ciphertext(out, red, eval());
  Conv_poly2ntt_inplace(Get_c0(out), crt());
  Conv_poly2ntt_inplace(Get_c1(out), crt());
  Free_ciphertext3(m3);
  Free_ciphertext(m);
  Free_ciphertext(red);
  *res = *out;
  free(out);
}
static void opt_moddown_rescale_modup(CIPHERTEXT* res, CIPHER a, double* w) { /* :602-660 */
  CIPHERTEXT* rot = Alloc_ciphertext();
  CIPHERTEXT* mul = Alloc_ciphertext();
  CIPHERTEXT* red = Alloc_ciphertext();
  CIPHERTEXT* rs = Alloc_ciphertext();
  CIPHERTEXT* out = Alloc_ciphertext();
  PLAINTEXT*  pt = Alloc_plaintext();
  VALUE_LIST* vec = Alloc_value_list(DCMPLX_TYPE, Slots);
  for (uint32_t i = 0; i < Slots; ++i) DCMPLX_VALUE_AT(vec, i) = w[i];
  Encode_ext_at_level(pt, (CKKS_ENCODER*)Context->_encoder, vec, Get_ciph_level(a), Get_ciph_slots(a), Get_crt_num_p(crt()));
  VALUE_LIST* pre = Switch_key_precompute(Get_c1(a), crt());
  Fast_rotate_ext(rot, a, Rots[1], rot_key(Rots[1]), eval(), pre, true);
  Mul_plaintext(mul, rot, pt, eval());
  Conv_ntt2poly_inplace(Get_c1(mul), crt());
  Init_ciphertext_from_ciph(red, a, Get_ciph_sfactor(mul), Get_ciph_sf_degree(mul));
  Reduce_rns_base(Get_c0(red), Get_c0(mul), crt());
  Reduce_rns_base(Get_c1(red), Get_c1(mul), crt());
  Rescale_ciphertext(rs, red, eval());
  VALUE_LIST* pre2 = Switch_key_precompute(Get_c1(rs), crt());
  Fast_rotate(out, rs, Rots[2], rot_key(Rots[2]), eval(), pre2);
  Free_switch_key_precomputed(pre);
  Free_switch_key_precomputed(pre2);
  Free_value_list(vec);
  Free_plaintext(pt);
  Free_ciphertext(rot);
  Free_ciphertext(mul);
  Free_ciphertext(red);
  Free_ciphertext(rs);
  *res = *out;
  free(out);
}
#endif

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
  Parm->_num_rot_idx      = 3;
  for (int i = 0; i < 3; ++i) Parm->_rot_idxs[i] = Rots[i];
  Slots = Parm->_poly_degree / 2;
  Prepare_context();
  double* x = (double*)malloc(sizeof(double) * Slots);
  double* w = (double*)malloc(sizeof(double) * Slots);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = sin(0.37 * i) * 0.5;
  TENSOR* t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_a");
  Free_tensor(t);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = cos(0.23 * i + 1.0) * 0.4;
  t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_b");
  Free_tensor(t);
  for (uint32_t i = 0; i < Slots; ++i) w[i] = cos(0.11 * i) * 0.75;
  CIPHERTEXT a = Get_input_data("in_a", 0), b = Get_input_data("in_b", 0), c;
  ZERO(c);
  Add_ciph(&c, &a, &b); /* a third ciphertext for the hoisted-ModDown sum */
  CIPHER     cts[3] = {&a, &b, &c};
  CIPHERTEXT r1, r2, r3, r4;
  base_modup_hoist(&r1, &a);
  out_ciph("modup_hoist", &r1);
  base_moddown_hoist(&r2, cts);
  out_ciph("moddown_hoist", &r2);
  base_moddown_rescale(&r3, &a, &b);
  out_ciph("moddown_rescale", &r3);
  base_moddown_rescale_modup(&r4, &a, w);
  out_ciph("moddown_rescale_modup", &r4);
#ifdef REF_BUILD
  CIPHERTEXT o;
  opt_modup_hoist(&o, &a);
  verdict("modup_hoist", &r1, &o);
  opt_moddown_hoist(&o, cts);
  verdict("moddown_hoist", &r2, &o);
  opt_moddown_rescale(&o, &a, &b);
  verdict("moddown_rescale", &r3, &o);
  opt_moddown_rescale_modup(&o, &a, w);
  verdict("moddown_rescale_modup", &r4, &o);
#endif
  free(x);
  free(w);
  Finalize_context();
  printf("SUCESS! four base forms written\n");
  return 0;
}
