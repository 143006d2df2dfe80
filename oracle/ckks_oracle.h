/* ckks_oracle.h -- CPU restatement of the ACE rt_ant RNS-CKKS polynomial layer.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (ace-compiler_amd/csrc) never
 * includes, links or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference/fhe-cmplr/rtlib/ant)
 * whose algorithm it restates.  Parity of this oracle is PINNED: tests/test_oracle_golden.py checks
 * it against tests/golden/ref_*.json, which oracle/ref_dump.c produced by running the reference
 * rtlib itself (oracle/_ref/libref_rtlib.so, compiled from /root/reference by oracle/Makefile),
 * and against the known-answer values of the reference's own unit tests
 * (unittest/ut_test_number_theory.cxx, unittest/ut_poly.cxx).
 */
#ifndef CKKS_ORACLE_H
#define CKKS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_u128;

/* Per-prime data: MODULUS (include/util/fhe_utils.h:27-32) + NTT_CONTEXT (include/util/ntt.h:32-48) */
typedef struct {
  uint64_t  q;
  uint64_t  br_k, br_m;       /* Init_modulus fhe_utils.h:66-71 */
  uint64_t  prec128_lo, prec128_hi; /* floor(2^128/q), Precompute_const_128 fhe_utils.h:385-401 */
  uint64_t  psi;              /* Root_of_unity(2N, q) number_theory.c:136-157 */
  uint64_t  n_inv, n_inv_prec;/* ntt.c:35-37 */
  uint64_t* rou;              /* rou[bitrev(i)] = psi^i,   ntt.c:90-97  */
  uint64_t* rou_prec;         /* Shoup companions,          ntt.c:115-126 */
  uint64_t* rou_inv;          /* rou_inv[bitrev(i)] = psi^-i ntt.c:99-113 */
  uint64_t* rou_inv_prec;
} ORC_PRIME;

typedef struct {
  uint32_t   N, logN;
  uint32_t   L;       /* number of q primes (mul_depth+1) */
  uint32_t   K;       /* number of p primes, crt.c:423 */
  uint32_t   dnum;    /* requested number of q parts */
  uint32_t   alpha;   /* per-part size, crt.c:386 */
  uint32_t   q0_bits, sf_bits;
  ORC_PRIME* prime;   /* [L+K]: q_0..q_{L-1}, p_0..p_{K-1} */
  /* ModDown tables (P -> Q), Appendix B-13 */
  uint64_t*  phat_inv_modp;      /* [K]      (P/p_j)^-1 mod p_j        crt.c:233-262 */
  uint64_t*  phat_inv_modp_prec; /* [K]      Shoup companion                          */
  uint64_t*  phat_modq;          /* [L][K]   (P/p_j) mod q_i           crt.c:330-381  */
  uint64_t*  pinv_modq;          /* [L]      P^-1 mod q_i              crt.c:378      */
  /* Rescale tables for level k+1 -> k ("k" = index of dropped limb - 1), crt.c:270-326 */
  uint64_t*  ql_inv_modqi;       /* [L][L]   [k][i] = q_{k+1}^-1 mod q_i              */
  uint64_t*  ql_inv_modqi_prec;
  uint64_t*  qlql;               /* [L][L]   [k][i] = ql_ql_inv_mod_ql_div_ql_mod_qi  */
  uint64_t*  qlql_prec;
} ORC_CTX;

/* ---- number theory ---- */
uint64_t orc_mul_mod(uint64_t a, uint64_t b, uint64_t m);              /* fhe_utils.h:176 */
uint64_t orc_pow_mod(uint64_t a, uint64_t e, uint64_t m);              /* number_theory.c:34 */
uint64_t orc_inv_mod_prime(uint64_t a, uint64_t m);                    /* number_theory.c:53 */
int      orc_is_prime(uint64_t n);                                     /* number_theory.c:160 */
uint64_t orc_find_generator(uint64_t q);                               /* number_theory.c:92 */
uint64_t orc_root_of_unity(uint64_t order, uint64_t q);                /* number_theory.c:136 */
uint32_t orc_reverse_bits(uint32_t v, uint32_t width);                 /* bit_operations.c:13 */
uint32_t orc_find_automorphism_index(int32_t rot_idx, uint32_t N);     /* number_theory.c:187 */
void     orc_automorphism_order(int64_t* out, uint32_t k, uint32_t N, int is_ntt); /* :201 */
void     orc_precompute_const_128(uint64_t q, uint64_t* lo, uint64_t* hi);
uint64_t orc_mod_barrett_128(orc_u128 v, const ORC_PRIME* p);          /* fhe_utils.h:241 */
uint64_t orc_shoup(uint64_t a, uint64_t w, uint64_t w_prec, uint64_t q);/* fhe_utils.h:311 */
uint64_t orc_precompute_const(uint64_t w, uint64_t q);                 /* fhe_utils.h:378 */
uint64_t orc_switch_modulus(uint64_t v, uint64_t old_q, uint64_t new_q);/* fhe_utils.h:349 */

/* ---- prime generation (crt.c:16-125) ---- */
void orc_generate_q_primes(uint64_t* out, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t N);
void orc_generate_p_primes(uint64_t* out, uint32_t K, uint32_t N, const uint64_t* q, uint32_t L);
uint32_t orc_num_p(const uint64_t* q, uint32_t L, uint32_t dnum);      /* crt.c:383-424 */

/* ---- context ---- */
ORC_CTX* orc_ctx_create(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum);
ORC_CTX* orc_ctx_create_from_primes(uint32_t N, const uint64_t* q, uint32_t L, uint32_t dnum);
void     orc_ctx_free(ORC_CTX* c);
int      orc_prime_init(ORC_PRIME* p, uint64_t q, uint32_t N);
void     orc_prime_free(ORC_PRIME* p);
uint32_t orc_num_decomp(const ORC_CTX* c, uint32_t level);             /* polynomial.h:158-168 */

/* ModUp tables for (level, digit): Appendix B-12 (crt.c:426-533).  Returns n2 (digit limbs).
 * compl_idx[j]: global prime index (q: 0..L-1, p: L..L+K-1) of complement basis entry j,
 * n_compl = level - n2 + K entries.  hat_inv[i] (i<n2), hat_mod[i*n_compl + j]. */
uint32_t orc_modup_tables(const ORC_CTX* c, uint32_t level, uint32_t digit,
                          uint64_t* hat_inv, uint32_t* compl_idx, uint64_t* hat_mod);

/* ---- NTT (ntt.c:190-353) ---- */
void orc_ntt_fwd(uint64_t* a, const ORC_PRIME* p, uint32_t N);
void orc_ntt_inv(uint64_t* a, const ORC_PRIME* p, uint32_t N);

/* ---- limb ops (src/poly/poly_arith.c) ---- */
void orc_hw_modadd(uint64_t* r, const uint64_t* a, const uint64_t* b, uint64_t q, uint32_t N);
void orc_hw_modmul(uint64_t* r, const uint64_t* a, const uint64_t* b, const ORC_PRIME* p, uint32_t N);
void orc_hw_modmul_faithful(uint64_t* r, const uint64_t* a, const uint64_t* b, const ORC_PRIME* p, uint32_t N);
void orc_hw_rotate(uint64_t* r, const uint64_t* a, const int64_t* perm, uint64_t q, uint32_t N);

/* ---- polynomial ops.  Layout: limb-major, limb l at data + l*N.
 * An "extended" poly at level l has l q-limbs followed directly by K p-limbs (l+K limbs). ---- */
/* Decompose_modup polynomial.c:1241-1335: in = level q-limbs (NTT domain), out = level+K limbs */
void orc_decomp_modup(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit);
/* Reduce_rns_base polynomial.c:928-967: in = level+K limbs (NTT), out = level limbs */
void orc_mod_down(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level);
/* Rescale_poly polynomial.c:1097-1163 (NTT branch): in = level limbs, out = level-1 limbs */
void orc_rescale(const ORC_CTX* c, uint64_t* out, const uint64_t* in, uint32_t level);
/* The generated Rotate()/Relinearize() key-switch core (resnet20_cifar10_pre.onnx.inc:6972-7146,
 * = Fast_switch_key ckks_evaluator.c:391-416):
 *   for each digit: ext = Decomp_modup(in, digit); acc0 += key0[digit] * ext; acc1 += key1[digit]*ext
 *   out0 = Mod_down(acc0); out1 = Mod_down(acc1)
 * key layout: [dnum][2][L+K][N] (key q-limbs at full level L, p-limbs at L..L+K-1). */
void orc_key_switch(const ORC_CTX* c, uint64_t* out0, uint64_t* out1, const uint64_t* in,
                    const uint64_t* key, uint32_t level);

/* ---- encoder (ckks_encode.c): Encode_impl ckks_encoder.c:199-297 with Embedding_inv ntt.c:713-753;
 * values = len complex numbers (re, im interleaved) zero padded to `slots` (0 = N/2); out_q = level limbs,
 * out_p = n_p limbs on the p primes, NTT domain.  -1 = the reference's "encode overflow" assert. ---- */
int orc_encode(const ORC_CTX* c, uint64_t* out_q, uint64_t* out_p, const double* values, size_t len, uint32_t slots,
               uint32_t sf_degree, uint32_t level, uint32_t n_p);
/* Encode_val_at_level ckks_encoder.c:464-530: residue of the constant on each of `level` limbs */
int orc_encode_value(const ORC_CTX* c, uint64_t* out_consts, double value, uint32_t sf_degree, uint32_t level);

/* checksums used by the golden fixtures */
uint64_t orc_sum64(const uint64_t* v, size_t n);
uint64_t orc_xorw(const uint64_t* v, size_t n);
uint64_t orc_splitmix64(uint64_t seed, uint64_t i);
/* deterministic test input: x[l*N+i] = splitmix64(seed, l*N+i) mod q_l  (prime_idx[l] selects q) */
void orc_fill_uniform(const ORC_CTX* c, uint64_t* out, const uint32_t* prime_idx, uint32_t n_limbs, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
