/* ref_containers.h -- REFERENCE-side writers of our container formats (test infrastructure; compiled only with -DREF_BUILD against
 * the reference rtlib's internal headers).  Shared by tests/c/ct_parity.c and tests/c/gen_parity_ref.c.
 *   ciphertext "ACEHCT01": u32 n_polys, N, level, num_p, is_ntt, slots, sf_degree, pad; f64 scaling_factor;
 *                          then per poly `level` q-limbs and `num_p` p-limbs of N u64
 *   keys       "ACEHKEY1": see write_keys() below
 * The including file defines `typedef unsigned long long u64;` and includes rtlib/context.h, util/ckks_key_generator.h,
 * util/ckks_parameters.h, util/crt.h first. */
#ifndef ACEHIP_TESTS_REF_CONTAINERS_H
#define ACEHIP_TESTS_REF_CONTAINERS_H
static void write_poly(FILE* f, POLYNOMIAL* p) {
  size_t n = p->_ring_degree;
  fwrite(p->_data, 8, p->_num_primes * n, f);
  if (p->_num_primes_p) fwrite(p->_data + (p->_num_alloc_primes - p->_num_primes_p) * n, 8, p->_num_primes_p * n, f);
}
static void save_polys(const char* path, POLYNOMIAL** polys, uint32_t n_polys, uint32_t slots, double sf, uint32_t sf_degree) {
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); exit(3); }
  uint32_t h[8] = {n_polys, polys[0]->_ring_degree, (uint32_t)polys[0]->_num_primes, (uint32_t)polys[0]->_num_primes_p,
                   polys[0]->_is_ntt, slots, sf_degree, 0};
  fwrite("ACEHCT01", 1, 8, f);
  fwrite(h, 4, 8, f);
  fwrite(&sf, 8, 1, f);
  for (uint32_t i = 0; i < n_polys; ++i) write_poly(f, polys[i]);
  fclose(f);
}
static void save_ciph(const char* path, CIPHER c) {
  POLYNOMIAL* p[2] = {&c->_c0_poly, &c->_c1_poly};
  save_polys(path, p, 2, c->_slots, c->_scaling_factor, c->_sf_degree);
}
static void save_ciph3(const char* path, CIPHER3 c) {
  POLYNOMIAL* p[3] = {&c->_c0_poly, &c->_c1_poly, &c->_c2_poly};
  save_polys(path, p, 3, c->_slots, c->_scaling_factor, c->_sf_degree);
}
static void save_plain(const char* path, PLAIN c) {
  POLYNOMIAL* p[1] = {&c->_poly};
  save_polys(path, p, 1, c->_slots, c->_scaling_factor, c->_sf_degree);
}
/* "ACEHKEY1": u32 version=1, N, L, K, dnum, n_rot, n_auto, pad; u64 primes[L+K];
 *             sk (NTT) [L+K][N]; pk0 [L][N]; pk1 [L][N]; relin [dnum][2][L+K][N] (b_j then a_j);
 *             n_rot x {i32 rotation, u32 auto_idx}; n_auto x {u32 auto_idx, u32 pad, [dnum][2][L+K][N]} */
static void write_swk(FILE* f, SWITCH_KEY* k, size_t dnum) {
  for (size_t j = 0; j < dnum; ++j) {
    PUBLIC_KEY* pk = Get_swk_at(k, j);
    write_poly(f, Get_pk0(pk));
    write_poly(f, Get_pk1(pk));
  }
}
static void write_keys(const char* path) {
  CKKS_KEY_GENERATOR* g = (CKKS_KEY_GENERATOR*)Get_key_gen(Context);
  CKKS_PARAMETER*     prm = (CKKS_PARAMETER*)Get_param(Context);
  CRT_CONTEXT*        crt = prm->_crt_context;
  uint32_t            L = Get_primes_cnt(Get_q(crt)), K = Get_primes_cnt(Get_p(crt));
  FILE*               f = fopen(path, "wb");
  if (!f) { perror(path); exit(3); }
  uint32_t n_rot = 0, n_auto = 0;
  PRECOMP_AUTO_IDX_MAP *ci, *ti;
  AUTO_KEY_MAP *        ck, *tk;
  HASH_ITER(HH, g->_precomp_auto_idx_map, ci, ti) n_rot++;
  HASH_ITER(HH, g->_auto_key_map, ck, tk) n_auto++;
  uint32_t h[8] = {1, prm->_poly_degree, L, K, (uint32_t)prm->_num_q_parts, n_rot, n_auto, 0};
  fwrite("ACEHKEY1", 1, 8, f);
  fwrite(h, 4, 8, f);
  for (uint32_t i = 0; i < L; ++i) { int64_t q = Get_modulus_val(Get_prime_at(Get_q(crt), i)); fwrite(&q, 8, 1, f); }
  for (uint32_t i = 0; i < K; ++i) { int64_t q = Get_modulus_val(Get_prime_at(Get_p(crt), i)); fwrite(&q, 8, 1, f); }
  write_poly(f, Get_ntt_sk(Get_sk(g)));
  write_poly(f, Get_pk0(Get_pk(g)));
  write_poly(f, Get_pk1(Get_pk(g)));
  write_swk(f, Get_relin_key(g), prm->_num_q_parts);
  HASH_ITER(HH, g->_precomp_auto_idx_map, ci, ti) {
    fwrite(&ci->_rot_idx, 4, 1, f);
    fwrite(&ci->_precomp_auto_idx, 4, 1, f);
  }
  HASH_ITER(HH, g->_auto_key_map, ck, tk) {
    uint32_t e[2] = {ck->_precomp_auto_idx, 0};
    fwrite(e, 4, 2, f);
    write_swk(f, ck->_auto_key, prm->_num_q_parts);
  }
  fclose(f);
  printf("keys: L=%u K=%u dnum=%u rot_map=%u auto_keys=%u\n", L, K, h[4], n_rot, n_auto);
}
#endif
