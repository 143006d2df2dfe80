/* ref_dump.c -- golden-vector generator that RUNS THE REFERENCE rtlib.
 *
 * TEST INFRASTRUCTURE ONLY.  Our own code; it links against oracle/_ref/libref_rtlib.so (the
 * reference compiled from /root/reference by oracle/Makefile) and uses the reference's internal
 * headers.  Its JSON output is committed under tests/golden/ (data, not source).  It also has a
 * "bench" mode used by bench.py's cpu_baseline leg (kind "reference").
 *
 *   ref_dump params N L q0 sf dnum            -> primes, psi, CRT tables
 *   ref_dump ops    N L q0 sf dnum level seed -> per-op input/output vectors (full if N<=64,
 *                                                checksums otherwise)
 *   ref_dump bench  N L q0 sf dnum level ks_reps ntt_reps -> timings of the reference ops (JSON)
 *   ref_dump mix    N L q0 sf dnum level reps            -> seconds per call of every primitive family at this parameter set (JSON)
 *   ref_dump ptfile N L q0 sf dnum level out n_entries sc_degree seed -> a DE_PLAINTEXT data file ("!ANTFHE" container of
 *                   rt_data_def.h:90-111 whose entries are PLAINTEXT_BUFFERs made by the reference's Encode_plain_buffer,
 *                   plain_eval.c:107-130): the fixture of the pre-encoded weight path (Pt_get, pt_mgr.c:128-159)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "common/rt_config.h"
#include "common/rt_api.h"
#include "poly/poly_arith.h"
#include "util/ckks_parameters.h"
#include "util/crt.h"
#include "util/ntt.h"
#include "util/number_theory.h"
#include "util/polynomial.h"
#include "util/ckks_encoder.h"
#include "util/plaintext.h"

/* generated-code callbacks the rtlib expects from the program it is linked into */
CKKS_PARAMS*  Get_context_params() { return NULL; }
RT_DATA_INFO* Get_rt_data_info() { return NULL; }
int           Get_input_count() { return 0; }
int           Get_output_count() { return 0; }
DATA_SCHEME*  Get_encode_scheme(int idx) { return NULL; }
DATA_SCHEME*  Get_decode_scheme(int idx) { return NULL; }
bool          Main_graph() { return true; }

typedef unsigned long long u64;

static u64 splitmix64(u64 seed, u64 i) {
  u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static u64 sum64(const int64_t* v, size_t n) { u64 s = 0; for (size_t i = 0; i < n; i++) s += (u64)v[i]; return s; }
static u64 xorw(const int64_t* v, size_t n) { u64 s = 0; for (size_t i = 0; i < n; i++) s ^= (u64)v[i] * (2 * (u64)i + 1); return s; }

static int Full = 0;
static void emit_vec(const char* name, const int64_t* v, size_t n, int last) {
  printf("  \"%s\": {\"n\": %zu, \"sum64\": %llu, \"xorw\": %llu, \"first\": %llu, \"last\": %llu", name, n,
         sum64(v, n), xorw(v, n), (u64)v[0], (u64)v[n - 1]);
  if (Full) {
    printf(", \"data\": [");
    for (size_t i = 0; i < n; i++) printf("%s%llu", i ? "," : "", (u64)v[i]);
    printf("]");
  }
  printf("}%s\n", last ? "" : ",");
}
static void emit_list(const char* name, const int64_t* v, size_t n, int last) {
  printf("  \"%s\": [", name);
  for (size_t i = 0; i < n; i++) printf("%s%llu", i ? "," : "", (u64)v[i]);
  printf("]%s\n", last ? "" : ",");
}

static CKKS_PARAMETER* make_param(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum) {
  CKKS_PARAMETER* p = Alloc_ckks_parameter();
  Set_num_q_parts(p, dnum);
  Init_ckks_parameters_with_prime_size(p, N, HE_STD_NOT_SET, L, q0, sf, 192);
  return p;
}
static int64_t prime_val(CRT_CONTEXT* crt, size_t gi) {
  size_t L = Get_primes_cnt(Get_q(crt));
  return gi < L ? Get_modulus_val(Get_prime_at(Get_q(crt), gi)) : Get_modulus_val(Get_prime_at(Get_p(crt), gi - L));
}
static CRT_PRIME* prime_at(CRT_CONTEXT* crt, size_t gi) {
  size_t L = Get_primes_cnt(Get_q(crt));
  return gi < L ? Get_prime_at(Get_q(crt), gi) : Get_prime_at(Get_p(crt), gi - L);
}
static void fill_uniform(CRT_CONTEXT* crt, int64_t* out, size_t n_limbs, size_t level, uint32_t N, u64 seed) {
  /* limb l < level uses q_l; limbs >= level use p_{l-level} */
  size_t L = Get_primes_cnt(Get_q(crt));
  for (size_t l = 0; l < n_limbs; l++) {
    u64 q = (u64)prime_val(crt, l < level ? l : L + (l - level));
    for (uint32_t i = 0; i < N; i++) out[l * N + i] = (int64_t)(splitmix64(seed, (u64)l * N + i) % q);
  }
}

static int do_params(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CRT_CONTEXT*    crt = p->_crt_context;
  size_t          K = p->_num_p_primes;
  size_t          alpha = Get_per_part_size(Get_qpart(crt));
  size_t          dnum_req = dnum;
  dnum = p->_num_q_parts; /* 0 -> Get_default_num_q_parts */
  printf("{\n  \"N\": %u, \"L\": %zu, \"q0_bits\": %zu, \"sf_bits\": %zu, \"dnum\": %zu, \"dnum_req\": %zu, \"K\": %zu, \"alpha\": %zu,\n",
         N, L, q0, sf, dnum, dnum_req, K, alpha);
  int64_t* tmp = malloc(sizeof(int64_t) * (L + K) * (L + K + 4));
  for (size_t i = 0; i < L + K; i++) tmp[i] = prime_val(crt, i);
  emit_list("primes", tmp, L + K, 0);
  for (size_t i = 0; i < L + K; i++) { /* psi = rou[bitrev(1)] = rou[N/2] */
    NTT_CONTEXT* ntt = Get_ntt(prime_at(crt, i));
    tmp[i] = Get_i64_value_at(ntt->_rou, N >> 1);
  }
  emit_list("psi", tmp, L + K, 0);
  for (size_t i = 0; i < L + K; i++) tmp[i] = Get_ntt(prime_at(crt, i))->_degree_inv;
  emit_list("n_inv", tmp, L + K, 0);
  for (size_t i = 0; i < L + K; i++) tmp[i] = Get_ntt(prime_at(crt, i))->_degree_inv_prec;
  emit_list("n_inv_prec", tmp, L + K, 0);
  for (size_t i = 0; i < L + K; i++) tmp[i] = (int64_t)(uint64_t)Get_prec128(Get_modulus(prime_at(crt, i)));
  emit_list("prec128_lo", tmp, L + K, 0);
  for (size_t i = 0; i < L + K; i++) tmp[i] = (int64_t)(uint64_t)(Get_prec128(Get_modulus(prime_at(crt, i))) >> 64);
  emit_list("prec128_hi", tmp, L + K, 0);
  /* twiddle table checksums per prime (+ full tables for tiny N) */
  printf("  \"rou\": [\n");
  for (size_t i = 0; i < L + K; i++) {
    NTT_CONTEXT* ntt = Get_ntt(prime_at(crt, i));
    printf("   {\"sum64\": %llu, \"xorw\": %llu, \"prec_xorw\": %llu, \"inv_xorw\": %llu, \"inv_prec_xorw\": %llu}%s\n",
           sum64(Get_i64_values(ntt->_rou), N), xorw(Get_i64_values(ntt->_rou), N),
           xorw((int64_t*)Get_ui64_values(ntt->_rou_prec), N), xorw(Get_i64_values(ntt->_rou_inv), N),
           xorw((int64_t*)Get_ui64_values(ntt->_rou_inv_prec), N), i + 1 < L + K ? "," : "");
  }
  printf("  ],\n");
  /* ModDown tables */
  emit_list("phat_inv_modp", Get_i64_values(Get_phatinvmodp(Get_p(crt))), K, 0);
  emit_list("phat_inv_modp_prec", Get_i64_values(Get_phatinvmodp_prec(Get_p(crt))), K, 0);
  for (size_t i = 0; i < L; i++)
    for (size_t j = 0; j < K; j++) tmp[i * K + j] = Get_i64_value_at(Get_phatmodq_at(Get_p(crt), i), j);
  emit_list("phat_modq", tmp, L * K, 0);
  emit_list("pinv_modq", Get_i64_values(Get_pinvmodq(Get_p(crt))), L, 0);
  /* rescale tables: rows k = 0..L-2, columns i <= k */
  printf("  \"rescale\": [\n");
  for (size_t k = 0; k + 1 < L; k++) {
    printf("   {\"ql_inv\": [");
    for (size_t i = 0; i <= k; i++) printf("%s%llu", i ? "," : "", (u64)Get_i64_value_at(Get_ql_inv_mod_qi_at(Get_q(crt), k), i));
    printf("], \"ql_inv_prec\": [");
    for (size_t i = 0; i <= k; i++) printf("%s%llu", i ? "," : "", (u64)Get_i64_value_at(Get_ql_inv_mod_qi_prec_at(Get_q(crt), k), i));
    printf("], \"qlql\": [");
    for (size_t i = 0; i <= k; i++) printf("%s%llu", i ? "," : "", (u64)Get_i64_value_at(Get_ql_ql_inv_mod_ql_div_ql_mod_qi_at(Get_q(crt), k), i));
    printf("], \"qlql_prec\": [");
    for (size_t i = 0; i <= k; i++) printf("%s%llu", i ? "," : "", (u64)Get_i64_value_at(Get_ql_ql_inv_mod_ql_div_ql_mod_qi_prec_at(Get_q(crt), k), i));
    printf("]}%s\n", k + 2 < L ? "," : "");
  }
  printf("  ],\n");
  /* ModUp tables for every (level, digit) */
  printf("  \"modup\": [\n");
  int first = 1;
  for (size_t level = 1; level <= L; level++) {
    size_t nd = (level + alpha - 1) / alpha;
    if (nd > dnum) nd = dnum;
    for (size_t d = 0; d < nd; d++) {
      size_t       n2 = level - alpha * d < alpha ? level - alpha * d : alpha;
      VL_CRTPRIME* compl = Get_qpart_compl_at(Get_qpart_compl(crt), level - 1, d);
      VALUE_LIST*  hinv = VL_L2_VALUE_AT(Get_qlhatinvmodq(Get_qpart(crt)), d, n2 - 1);
      VL_VL_I64*   hmod = VL_L2_VALUE_AT(Get_qlhatmodp(Get_qpart(crt)), level - 1, d);
      size_t       nc = LIST_LEN(compl);
      printf("%s   {\"level\": %zu, \"digit\": %zu, \"n2\": %zu, \"hat_inv\": [", first ? "" : ",\n", level, d, n2);
      first = 0;
      for (size_t i = 0; i < n2; i++) printf("%s%llu", i ? "," : "", (u64)I64_VALUE_AT(hinv, i));
      printf("], \"compl\": [");
      for (size_t j = 0; j < nc; j++) printf("%s%llu", j ? "," : "", (u64)Get_modulus_val(Get_vlprime_at(compl, j)));
      if (L > 12) { /* large sets: checksums only (flat index i*nc + j) */
        u64 s = 0, x = 0;
        for (size_t i = 0; i < n2; i++)
          for (size_t j = 0; j < nc; j++) {
            u64 v = (u64)Get_i64_value_at(VL_VALUE_AT(hmod, i), j);
            s += v;
            x ^= v * (2 * (u64)(i * nc + j) + 1);
          }
        printf("], \"hat_mod_sum64\": %llu, \"hat_mod_xorw\": %llu}", s, x);
      } else {
        printf("], \"hat_mod\": [");
        for (size_t i = 0; i < n2; i++)
          for (size_t j = 0; j < nc; j++)
            printf("%s%llu", (i || j) ? "," : "", (u64)Get_i64_value_at(VL_VALUE_AT(hmod, i), j));
        printf("]}");
      }
    }
  }
  printf("\n  ]\n}\n");
  return 0;
}

/* the generated Rotate()/Relinearize() key-switch core, spelled with the reference's functions
 * exactly as the checked-in generated code does (dataset/resnet20_cifar10_pre.onnx.inc:6996-7040) */
static void ref_key_switch(CRT_CONTEXT* crt, POLYNOMIAL* out0, POLYNOMIAL* out1, POLYNOMIAL* in, int64_t* key,
                           size_t level, size_t L, size_t K, uint32_t N) {
  POLYNOMIAL swk0, swk1, ext, tmp;
  Alloc_poly_data(&swk0, N, level, K); Set_is_ntt(&swk0, TRUE);
  Alloc_poly_data(&swk1, N, level, K); Set_is_ntt(&swk1, TRUE);
  Alloc_poly_data(&ext, N, level, K);  Set_is_ntt(&ext, TRUE);
  Alloc_poly_data(&tmp, N, 1, 0);
  size_t nd = Get_num_decomp_poly(in, crt);
  for (size_t part = 0; part < nd; part++) {
    Decompose_modup(&ext, in, crt, nd, part);
    int64_t* key0 = key + (part * 2 + 0) * (L + K) * N;
    int64_t* key1 = key + (part * 2 + 1) * (L + K) * N;
    MODULUS* m = Get_q_modulus_head(crt);
    for (size_t i = 0; i < level; i++, m++) {
      Hw_modmul(tmp._data, key0 + i * N, ext._data + i * N, m, N);
      Hw_modadd(swk0._data + i * N, swk0._data + i * N, tmp._data, m, N);
      Hw_modmul(tmp._data, key1 + i * N, ext._data + i * N, m, N);
      Hw_modadd(swk1._data + i * N, swk1._data + i * N, tmp._data, m, N);
    }
    m = Get_p_modulus_head(crt);
    size_t p_ofst = Get_num_alloc_primes(&ext) - K; /* Num_alloc - Num_p */
    for (size_t i = 0; i < K; i++, m++) {
      size_t pi = i + p_ofst, ki = i + L; /* key P-limbs start at Poly_level(key) = L */
      Hw_modmul(tmp._data, key0 + ki * N, ext._data + pi * N, m, N);
      Hw_modadd(swk0._data + pi * N, swk0._data + pi * N, tmp._data, m, N);
      Hw_modmul(tmp._data, key1 + ki * N, ext._data + pi * N, m, N);
      Hw_modadd(swk1._data + pi * N, swk1._data + pi * N, tmp._data, m, N);
    }
  }
  Reduce_rns_base(out0, &swk0, crt);
  Reduce_rns_base(out1, &swk1, crt);
  Free_poly_data(&swk0); Free_poly_data(&swk1); Free_poly_data(&ext); Free_poly_data(&tmp);
}

static int do_ops(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum, size_t level, u64 seed) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CRT_CONTEXT*    crt = p->_crt_context;
  size_t          K = p->_num_p_primes;
  size_t          dnum_req = dnum;
  dnum = p->_num_q_parts;
  Full = N <= 64;
  Set_rtlib_config(CONF_OP_FUSION_DECOMP_MODUP, 1);
  printf("{\n  \"N\": %u, \"L\": %zu, \"q0_bits\": %zu, \"sf_bits\": %zu, \"dnum\": %zu, \"dnum_req\": %zu, \"K\": %zu, \"level\": %zu, \"seed\": %llu,\n",
         N, L, q0, sf, dnum, dnum_req, K, level, seed);

  /* input a: level q-limbs, uniform, flagged NTT-domain */
  POLYNOMIAL a;
  Alloc_poly_data(&a, N, level, 0);
  fill_uniform(crt, a._data, level, level, N, seed);
  Set_is_ntt(&a, TRUE);
  emit_vec("a", a._data, level * N, 0);

  /* NTT / iNTT of every limb of a (treated as raw data), incl. p-limbs of a second input */
  {
    int64_t* f = malloc(sizeof(int64_t) * (level + K) * N);
    int64_t* g = malloc(sizeof(int64_t) * (level + K) * N);
    fill_uniform(crt, f, level + K, level, N, seed + 1);
    emit_vec("x_ext", f, (level + K) * N, 0);
    memcpy(g, f, sizeof(int64_t) * (level + K) * N);
    size_t Lq = Get_primes_cnt(Get_q(crt));
    for (size_t l = 0; l < level + K; l++) {
      VALUE_LIST vl;
      Init_i64_value_list_no_copy(&vl, N, f + l * N);
      Ftt_fwd(&vl, Get_ntt(prime_at(crt, l < level ? l : Lq + (l - level))), &vl);
      Init_i64_value_list_no_copy(&vl, N, g + l * N);
      Ftt_inv(&vl, Get_ntt(prime_at(crt, l < level ? l : Lq + (l - level))), &vl);
    }
    emit_vec("ntt_fwd_x_ext", f, (level + K) * N, 0);
    emit_vec("ntt_inv_x_ext", g, (level + K) * N, 0);

    /* Mod_down of x_ext interpreted as an NTT-domain extended poly */
    POLYNOMIAL ext, dn;
    Alloc_poly_data(&ext, N, level, K);
    fill_uniform(crt, ext._data, level + K, level, N, seed + 1);
    Set_is_ntt(&ext, TRUE);
    Alloc_poly_data(&dn, N, level, 0);
    Reduce_rns_base(&dn, &ext, crt);
    emit_vec("mod_down_x_ext", dn._data, level * N, 0);
    Free_poly_data(&ext); Free_poly_data(&dn);
    free(f); free(g);
  }

  /* Hw_modadd / Hw_modmul / Hw_rotate on limbs of a and b */
  {
    POLYNOMIAL b, r;
    Alloc_poly_data(&b, N, level, 0);
    Alloc_poly_data(&r, N, level, 0);
    fill_uniform(crt, b._data, level, level, N, seed + 2);
    emit_vec("b", b._data, level * N, 0);
    MODULUS* m = Get_q_modulus_head(crt);
    for (size_t l = 0; l < level; l++, m++) Hw_modadd(r._data + l * N, a._data + l * N, b._data + l * N, m, N);
    emit_vec("hw_modadd_a_b", r._data, level * N, 0);
    m = Get_q_modulus_head(crt);
    for (size_t l = 0; l < level; l++, m++) Hw_modmul(r._data + l * N, a._data + l * N, b._data + l * N, m, N);
    emit_vec("hw_modmul_a_b", r._data, level * N, 0);
    int32_t rots[] = {1, -1, 5, (int32_t)(N / 4)};
    MODULUS two_n;
    Init_modulus(&two_n, 2 * (int64_t)N);
    printf("  \"rotate\": [\n");
    for (int ri = 0; ri < 5; ri++) {
      uint32_t k = ri < 4 ? Find_automorphism_index(rots[ri], &two_n) : 2 * N - 1; /* conj */
      VALUE_LIST* pre = Alloc_value_list(I64_TYPE, N);
      Precompute_automorphism_order(pre, k, N, TRUE);
      m = Get_q_modulus_head(crt);
      for (size_t l = 0; l < level; l++, m++) Hw_rotate(r._data + l * N, a._data + l * N, Get_i64_values(pre), m, N);
      printf("   {\"rot_idx\": %d, \"k\": %u, \"perm_sum64\": %llu, \"perm_xorw\": %llu, \"perm_head\": [%lld,%lld,%lld,%lld], \"out_sum64\": %llu, \"out_xorw\": %llu",
             ri < 4 ? rots[ri] : 0, k, sum64(Get_i64_values(pre), N), xorw(Get_i64_values(pre), N),
             (long long)I64_VALUE_AT(pre, 0), (long long)I64_VALUE_AT(pre, 1), (long long)I64_VALUE_AT(pre, 2),
             (long long)I64_VALUE_AT(pre, 3), sum64(r._data, level * N), xorw(r._data, level * N));
      if (Full) {
        printf(", \"perm\": [");
        for (size_t i = 0; i < N; i++) printf("%s%lld", i ? "," : "", (long long)I64_VALUE_AT(pre, i));
        printf("]");
      }
      printf("}%s\n", ri < 4 ? "," : "");
      Free_value_list(pre);
    }
    printf("  ],\n");
    Free_poly_data(&b); Free_poly_data(&r);
  }

  /* Decomp_modup for every digit */
  {
    size_t nd = Get_num_decomp_poly(&a, crt);
    printf("  \"num_decomp\": %zu,\n", nd);
    for (size_t d = 0; d < nd; d++) {
      POLYNOMIAL raised;
      Alloc_poly_data(&raised, N, level, K);
      Decompose_modup(&raised, &a, crt, nd, d);
      char name[64];
      snprintf(name, sizeof(name), "decomp_modup_%zu", d);
      emit_vec(name, raised._data, (level + K) * N, 0);
      Free_poly_data(&raised);
    }
  }

  /* Rescale_poly */
  if (level > 1) {
    POLYNOMIAL rs;
    Alloc_poly_data(&rs, N, level, 0);
    Rescale_poly(&rs, &a, crt);
    emit_vec("rescale_a", rs._data, (level - 1) * N, 0);
    Free_poly_data(&rs);
  }

  /* full key-switch core with a uniformly random "key" [dnum][2][L+K][N] */
  {
    size_t   key_n = dnum * 2 * (L + K) * N;
    int64_t* key = malloc(sizeof(int64_t) * key_n);
    for (size_t d = 0; d < dnum * 2; d++) fill_uniform(crt, key + d * (L + K) * N, L + K, L, N, seed + 100 + d);
    printf("  \"key_seed_base\": %llu,\n", seed + 100);
    POLYNOMIAL o0, o1;
    Alloc_poly_data(&o0, N, level, 0);
    Alloc_poly_data(&o1, N, level, 0);
    ref_key_switch(crt, &o0, &o1, &a, key, level, L, K, N);
    emit_vec("key_switch_c0", o0._data, level * N, 0);
    emit_vec("key_switch_c1", o1._data, level * N, 1);
    Free_poly_data(&o0); Free_poly_data(&o1);
    free(key);
  }
  printf("}\n");
  return 0;
}


/* Encode_at_level_with_sf / Encode_val_at_level (what Pt_from_msg / Encode_plain_from_float do,
 * src/ckks/plain_eval.c:17-42): message m[i] = (float)(((i*7 + seed) mod 17) - 8) / 16 for i < len */
static int do_encode(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum, size_t level, u64 seed) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CKKS_ENCODER*   enc = Alloc_ckks_encoder(p);
  Full = N <= 64;
  printf("{\n  \"N\": %u, \"L\": %zu, \"q0_bits\": %zu, \"sf_bits\": %zu, \"dnum_req\": %zu, \"level\": %zu, \"seed\": %llu,\n  \"cases\": [\n",
         N, L, q0, sf, dnum, level, seed);
  size_t lens[3] = {N / 2, N / 4, N / 8 > 1 ? N / 8 : 2};
  int first = 1;
  for (int li = 0; li < 3; li++) {
    for (uint32_t sfd = 1; sfd <= 2; sfd++) {
      size_t len = lens[li];
      if (sfd == 2 && level < 2) continue;
      VALUE_LIST* vals = Alloc_value_list(DCMPLX_TYPE, len);
      for (size_t i = 0; i < len; i++) DCMPLX_VALUE_AT(vals, i) = (double)((float)((int)((i * 7 + seed) % 17) - 8) / 16.0f);
      PLAINTEXT* pt = Alloc_plaintext();
      Encode_at_level_with_sf(pt, enc, vals, level, 0, sfd);
      printf("%s   {\"len\": %zu, \"sf_degree\": %u, \"slots\": %u, \"scale\": %.17g,\n", first ? "" : ",\n", len, sfd, pt->_slots, pt->_scaling_factor);
      first = 0;
      emit_vec("poly", Get_poly_coeffs(Get_plain_poly(pt)), level * N, 1);
      printf("   }");
      Free_plaintext(pt);
      Free_value_list(vals);
    }
  }
  printf("\n  ],\n  \"consts\": [\n");
  double cvals[4] = {0.5685231134608953462717, -1.0, 3.25e-7, 12345.678};
  for (int ci = 0; ci < 4; ci++) {
    for (uint32_t sfd = 1; sfd <= 2; sfd++) {
      PLAINTEXT* pt = Alloc_plaintext();
      Encode_val_at_level(pt, enc, cvals[ci], level, sfd);
      printf("   {\"value\": %.17g, \"sf_degree\": %u, \"limb0\": [", cvals[ci], sfd);
      for (size_t l = 0; l < level; l++) printf("%s%llu", l ? "," : "", (u64)Get_poly_coeffs(Get_plain_poly(pt))[l * N]);
      printf("]}%s\n", (ci == 3 && sfd == 2) ? "" : ",");
      Free_plaintext(pt);
    }
  }
  printf("  ]\n}\n");
  return 0;
}

static double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* CPU baseline: the reference rtlib timed on this host (single thread, as published: README.md:94) */
static int do_bench(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum, size_t level, int reps, int ntt_reps) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CRT_CONTEXT*    crt = p->_crt_context;
  size_t          K = p->_num_p_primes;
  dnum = p->_num_q_parts;
  Set_rtlib_config(CONF_OP_FUSION_DECOMP_MODUP, 1);
  POLYNOMIAL a, o0, o1;
  Alloc_poly_data(&a, N, level, 0);
  fill_uniform(crt, a._data, level, level, N, 1);
  Set_is_ntt(&a, TRUE);
  Alloc_poly_data(&o0, N, level, 0);
  Alloc_poly_data(&o1, N, level, 0);
  size_t   key_n = dnum * 2 * (L + K) * N;
  int64_t* key = malloc(sizeof(int64_t) * key_n);
  for (size_t d = 0; d < dnum * 2; d++) fill_uniform(crt, key + d * (L + K) * N, L + K, L, N, 101 + d);
  /* NTT fwd+inv on limb 0 */
  VALUE_LIST vl;
  int64_t*   buf = malloc(sizeof(int64_t) * N);
  memcpy(buf, a._data, sizeof(int64_t) * N);
  Init_i64_value_list_no_copy(&vl, N, buf);
  double t0 = now_s();
  for (int r = 0; r < ntt_reps; r++) Ftt_fwd(&vl, Get_ntt(prime_at(crt, 0)), &vl);
  double t_fwd = (now_s() - t0) / ntt_reps;
  t0 = now_s();
  for (int r = 0; r < ntt_reps; r++) Ftt_inv(&vl, Get_ntt(prime_at(crt, 0)), &vl);
  double t_inv = (now_s() - t0) / ntt_reps;
  t0 = now_s();
  for (int r = 0; r < reps; r++) ref_key_switch(crt, &o0, &o1, &a, key, level, L, K, N);
  double t_ks = reps > 0 ? (now_s() - t0) / reps : 0.0;
  printf("{\"kind\": \"reference\", \"N\": %u, \"L\": %zu, \"dnum\": %zu, \"K\": %zu, \"level\": %zu, \"reps\": %d, \"ntt_reps\": %d, "
         "\"ntt_fwd_s\": %.9f, \"ntt_inv_s\": %.9f, \"key_switch_s\": %.9f, \"ks_sum64\": %llu}\n",
         N, L, dnum, K, level, reps, ntt_reps, t_fwd, t_inv, t_ks, sum64(o0._data, level * N));
  return 0;
}

/* CPU baseline at the WORKLOAD's parameter set: seconds per call of every primitive family the generated ResNets spend their time
 * in (single thread), so that bench.py can price one image on this host from the per-image call statistics of the run
 * (acehip_stats: the same data-oblivious program makes the same calls on either runtime) instead of scaling one micro-op by a
 * constant measured elsewhere.  Families as in include/acehip.h acehip_stat_name(). */
static int do_mix(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum, size_t level, int reps) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CRT_CONTEXT*    crt = p->_crt_context;
  size_t          K = p->_num_p_primes;
  dnum = p->_num_q_parts;
  Set_rtlib_config(CONF_OP_FUSION_DECOMP_MODUP, 1);
  /* COLD operands.  In the generated program no polynomial is cache-resident when it is touched: a ciphertext at level 20 is
   * 20 MB, thousands of them pass between two uses of the same one.  Timing a primitive in a loop over ONE buffer measures it out
   * of L2 and underestimates the real run (measured against the full ResNet-20 run of the dev container, profiles/r04_ref_resnet20_
   * seeded.log: Hw_modadd 3.0x, Hw_rotate 6.9x, Hw_modmul 1.75x, the transforms 1.1-1.2x).  Every timed call below therefore takes
   * its operands from a rotating pool that is far larger than the caches (POOL polynomials of `level` limbs each, ~0.5 GB). */
  enum { POOL = 16 };
  POLYNOMIAL a[POOL], b[POOL], r[POOL], ext[POOL];
  for (int i = 0; i < POOL; i++) {
    Alloc_poly_data(&a[i], N, level, 0);
    Alloc_poly_data(&b[i], N, level, 0);
    Alloc_poly_data(&r[i], N, level, 0);
    Alloc_poly_data(&ext[i], N, level, K);
    if (i == 0) {
      fill_uniform(crt, a[0]._data, level, level, N, 1);
      fill_uniform(crt, b[0]._data, level, level, N, 2);
      fill_uniform(crt, ext[0]._data, level + K, level, N, 3);
    } else {
      memcpy(a[i]._data, a[0]._data, sizeof(int64_t) * level * N);
      memcpy(b[i]._data, b[0]._data, sizeof(int64_t) * level * N);
      memcpy(ext[i]._data, ext[0]._data, sizeof(int64_t) * (level + K) * N);
    }
    Set_is_ntt(&a[i], TRUE);
    Set_is_ntt(&b[i], TRUE);
    Set_is_ntt(&ext[i], TRUE);
  }
  POLYNOMIAL md, rs;
  Alloc_poly_data(&md, N, level, 0);
  Alloc_poly_data(&rs, N, level, 0);
  double t0;
  /* NTT / iNTT of one limb, the limbs of the pool one after the other (limb l of every polynomial uses prime l) */
  int        ntt_reps = reps * 100, done = 0;
  VALUE_LIST vl;
  t0 = now_s();
  for (int i = 0; done < ntt_reps; i++)
    for (size_t l = 1; l < level && done < ntt_reps; l++, done++) {
      Init_i64_value_list_no_copy(&vl, N, a[i % POOL]._data + l * N);
      Ftt_fwd(&vl, Get_ntt(prime_at(crt, l)), &vl);
    }
  double t_fwd = (now_s() - t0) / ntt_reps;
  done = 0;
  t0 = now_s();
  for (int i = 0; done < ntt_reps; i++)
    for (size_t l = 1; l < level && done < ntt_reps; l++, done++) {
      Init_i64_value_list_no_copy(&vl, N, a[i % POOL]._data + l * N);
      Ftt_inv(&vl, Get_ntt(prime_at(crt, l)), &vl);
    }
  double t_inv = (now_s() - t0) / ntt_reps;
  /* Hw_modmul / Hw_modadd / Hw_rotate of one limb (poly_arith.c:14-56) */
  MODULUS* mods = Get_q_modulus_head(crt);
  int      ew_reps = reps * 200;
  done = 0;
  t0 = now_s();
  for (int i = 0; done < ew_reps; i++)
    for (size_t l = 1; l < level && done < ew_reps; l++, done++)
      Hw_modmul(r[i % POOL]._data + l * N, a[i % POOL]._data + l * N, b[i % POOL]._data + l * N, mods + l, N);
  double t_mul = (now_s() - t0) / ew_reps;
  done = 0;
  t0 = now_s();
  for (int i = 0; done < ew_reps; i++)
    for (size_t l = 1; l < level && done < ew_reps; l++, done++)
      Hw_modadd(r[i % POOL]._data + l * N, a[i % POOL]._data + l * N, b[i % POOL]._data + l * N, mods + l, N);
  double t_add = (now_s() - t0) / ew_reps;
  MODULUS two_n_mod;
  Init_modulus(&two_n_mod, 2 * (int64_t)N);
  VALUE_LIST* order = Alloc_value_list(I64_TYPE, N);
  Precompute_automorphism_order(order, Find_automorphism_index(5, &two_n_mod), N, TRUE);
  done = 0;
  t0 = now_s();
  for (int i = 0; done < ew_reps; i++)
    for (size_t l = 1; l < level && done < ew_reps; l++, done++)
      Hw_rotate(r[i % POOL]._data + l * N, a[i % POOL]._data + l * N, Get_i64_values(order), mods + l, N);
  double t_rot = (now_s() - t0) / ew_reps;
  /* Decompose_modup of every digit (polynomial.c:1241-1335), Reduce_rns_base (:928-967), Rescale_poly (:1097-1163) */
  size_t nd = Get_num_decomp_poly(&a[0], crt);
  t0 = now_s();
  for (int i = 0; i < reps; i++)
    for (size_t part = 0; part < nd; part++) Decompose_modup(&ext[(i + 1) % POOL], &a[i % POOL], crt, nd, part);
  double t_modup = (now_s() - t0) / reps;
  for (int i = 1; i < POOL; i++) memcpy(ext[i]._data, ext[0]._data, sizeof(int64_t) * (level + K) * N);
  t0 = now_s();
  for (int i = 0; i < reps; i++) Reduce_rns_base(&md, &ext[i % POOL], crt);
  double t_md = (now_s() - t0) / reps;
  t0 = now_s();
  for (int i = 0; i < reps; i++) {
    Set_poly_level(&rs, level);
    Rescale_poly(&rs, &b[i % POOL], crt);
  }
  double t_rs = (now_s() - t0) / reps;
  /* Encode_at_level_with_sf of N/4 floats (the common weight-plaintext shape: Pt_from_msg, pt_mgr.c:182) */
  CKKS_ENCODER* enc = Alloc_ckks_encoder(p);
  size_t        len = N / 4;
  VALUE_LIST*   vals = Alloc_value_list(DCMPLX_TYPE, len);
  for (size_t i = 0; i < len; i++) DCMPLX_VALUE_AT(vals, i) = (double)((float)((int)((i * 7 + 3) % 17) - 8) / 16.0f);
  t0 = now_s();
  for (int i = 0; i < reps; i++) {
    PLAINTEXT* pt = Alloc_plaintext();
    Encode_at_level_with_sf(pt, enc, vals, level, 0, 1);
    Free_plaintext(pt);
  }
  double t_enc = (now_s() - t0) / reps;
  /* clearing and copying a polynomial, cold: what calloc'ed temporaries and Copy_poly cost the reference inside a run (the flat profile of a
   * whole ResNet-20 image, profiles/r04_ref_resnet20_profile.json, has 5.1 % of the image in libc's memset and 2.7 % in its memcpy) */
  const size_t poly_bytes = sizeof(int64_t) * level * N;
  int          mem_reps = reps * 16;
  t0 = now_s();
  for (int i = 0; i < mem_reps; i++) memset(r[i % POOL]._data, 0, poly_bytes);
  double set_GBs = (double)poly_bytes * mem_reps / (now_s() - t0) / 1e9;
  t0 = now_s();
  for (int i = 0; i < mem_reps; i++) memcpy(r[i % POOL]._data, a[(i + 5) % POOL]._data, poly_bytes);
  double cpy_GBs = (double)poly_bytes * mem_reps / (now_s() - t0) / 1e9;
  printf("{\"kind\": \"reference\", \"operands\": \"cold (rotating pool of %d polynomials)\", \"N\": %u, \"L\": %zu, \"dnum\": %zu, \"K\": %zu, "
         "\"level\": %zu, \"num_decomp\": %zu, \"reps\": %d, "
         "\"ntt_fwd_s\": %.9f, \"ntt_inv_s\": %.9f, \"hw_modmul_s\": %.9f, \"hw_modadd_s\": %.9f, \"hw_rotate_s\": %.9f, "
         "\"decomp_modup_all_digits_s\": %.9f, \"mod_down_s\": %.9f, \"rescale_s\": %.9f, \"encode_s\": %.9f, \"memset_GBs\": %.4f, \"memcpy_GBs\": %.4f}\n",
         POOL, N, L, dnum, K, level, nd, reps, t_fwd, t_inv, t_mul, t_add, t_rot, t_modup, t_md, t_rs, t_enc, set_GBs, cpy_GBs);
  return 0;
}

/* DE_PLAINTEXT data file written the way the compiler's RT_DATA_WRITER lays it out: header page, entries aligned to
 * 4096 bytes, lookup table at the end.  Entry e holds the message m[i] = ((splitmix64(seed + e, i) % 2001) - 1000) / 1024
 * (N/2 floats, exactly representable) encoded at `level` with scale degree sc_degree by the reference. */
#include "fhe/core/rt_data_def.h"
#include "fhe/core/rt_encode_api.h"
#include "fhe/core/rt_version.h"
#include "rtlib/context.h"
static int do_ptfile(uint32_t N, size_t L, size_t q0, size_t sf, size_t dnum, size_t level, const char* out, int n_entries,
                     uint32_t sc_degree, u64 seed) {
  CKKS_PARAMETER* p = make_param(N, L, q0, sf, dnum);
  CKKS_CONTEXT*   ctxt = (CKKS_CONTEXT*)calloc(1, sizeof(CKKS_CONTEXT));
  ctxt->_params = (PTR_TY)p;
  ctxt->_encoder = (PTR_TY)Alloc_ckks_encoder(p);
  Context = ctxt;
  FILE* f = fopen(out, "wb");
  if (!f) { perror(out); return 3; }
  struct DATA_FILE_HDR hdr;
  memset(&hdr, 0, sizeof(hdr));
  memcpy(hdr._magic, DATA_FILE_MAGIC, 8);
  hdr._rt_ver = RT_VERSION_FULL;
  hdr._ent_type = DE_PLAINTEXT;
  hdr._ent_align = 12;
  hdr._ent_count = n_entries;
  char page[DATA_FILE_PAGE_SIZE];
  memset(page, 0, sizeof(page));
  fwrite(page, 1, sizeof(page), f); /* header page, rewritten at the end */
  struct DATA_LUT_ENTRY* lut = calloc(n_entries, sizeof(struct DATA_LUT_ENTRY));
  size_t len = N / 2;
  float* msg = malloc(sizeof(float) * len);
  uint64_t ofst = DATA_FILE_PAGE_SIZE;
  for (int e = 0; e < n_entries; e++) {
    for (size_t i = 0; i < len; i++) msg[i] = (float)((double)((int64_t)(splitmix64(seed + e, i) % 2001) - 1000) / 1024.0);
    struct PLAINTEXT_BUFFER* pb = Encode_plain_buffer(msg, len, sc_degree, (uint32_t)level);
    uint64_t sz = Plain_buffer_length(pb);
    snprintf(lut[e]._name, sizeof(lut[e]._name), "pt_%d", e);
    lut[e]._index = e;
    lut[e]._size = (uint32_t)sz;
    lut[e]._ent_ofst = ofst;
    fwrite(pb, 1, sz, f);
    uint64_t padded = (sz + DATA_FILE_PAGE_SIZE - 1) & ~(uint64_t)(DATA_FILE_PAGE_SIZE - 1);
    fwrite(page, 1, padded - sz, f);
    ofst += padded;
    Free_plain_buffer(pb);
  }
  hdr._lut_ofst = ofst;
  fwrite(lut, sizeof(struct DATA_LUT_ENTRY), n_entries, f);
  fseek(f, 0, SEEK_SET);
  fwrite(&hdr, sizeof(hdr), 1, f);
  fclose(f);
  printf("{\"entries\": %d, \"level\": %zu, \"sc_degree\": %u, \"entry_bytes\": %u, \"lut_ofst\": %llu}\n", n_entries, level, sc_degree,
         lut[0]._size, (u64)hdr._lut_ofst);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s params|ops|bench N L q0 sf dnum [level seed|reps]\n", argv[0]);
    return 2;
  }
  uint32_t N = (uint32_t)atoi(argv[2]);
  size_t   L = atoi(argv[3]), q0 = atoi(argv[4]), sf = atoi(argv[5]), dnum = atoi(argv[6]);
  if (!strcmp(argv[1], "params")) return do_params(N, L, q0, sf, dnum);
  if (!strcmp(argv[1], "ops")) return do_ops(N, L, q0, sf, dnum, argc > 7 ? atoi(argv[7]) : L, argc > 8 ? strtoull(argv[8], 0, 10) : 1);
  if (!strcmp(argv[1], "encode")) return do_encode(N, L, q0, sf, dnum, argc > 7 ? atoi(argv[7]) : L, argc > 8 ? strtoull(argv[8], 0, 10) : 1);
  if (!strcmp(argv[1], "ptfile") && argc >= 12)
    return do_ptfile(N, L, q0, sf, dnum, atoi(argv[7]), argv[8], atoi(argv[9]), (uint32_t)atoi(argv[10]), strtoull(argv[11], 0, 10));
  if (!strcmp(argv[1], "mix")) return do_mix(N, L, q0, sf, dnum, argc > 7 ? atoi(argv[7]) : L, argc > 8 ? atoi(argv[8]) : 2);
  if (!strcmp(argv[1], "bench")) return do_bench(N, L, q0, sf, dnum, argc > 7 ? atoi(argv[7]) : L, argc > 8 ? atoi(argv[8]) : 1, argc > 9 ? atoi(argv[9]) : 20);
  return 2;
}
