/* gen_parity_ref.c -- REFERENCE side of the bit-exact parity check for GENERATED (polynomial-level) programs.
 * Test infrastructure, our own code; compiled only against the reference rtlib (oracle/_ref/libref_rtlib.so) and linked into
 * an UNCHANGED generated program (rtlib/ant/example/eg_fhertlib_*.c + .inc, or tools/model_main.c + a ResNet .inc), see
 * `make -C oracle refgen`.  Nothing here is part of the product.
 *
 * The product derives every key from ACEHIP_SEED and the key's identity, and the encryption randomness from the calling
 * thread's stream (ace-compiler_amd/csrc/rt/rt_context.cpp key_rng, rt_encode.cpp encrypt).  This file restates that
 * derivation on the CPU -- std::mt19937_64, the counter-based uniform sampler of csrc/rt_kernels.hip, the triangle / ternary
 * samplers -- and INJECTS the result into the reference's structures by interposing on a handful of entry points the generated
 * program (not the library) calls:
 *     Prepare_context   the reference prepares its own random keys; every one is then overwritten by the seeded key
 *     Bootstrap         keys the reference creates lazily for a new slot count (cipher_eval.c:366-373) are overwritten before use
 *     Prepare_input     the reference encodes (its encoder is what ours is pinned to) and encrypts; c0 / c1 are then overwritten
 *                       by c0 = pk0*v + e1 + m, c1 = pk1*v + e2 with the seeded v, e1, e2 (ckks_encryptor.c:20-95)
 *     Encrypt           an encryption inside Main_graph (eg_fhertlib_bootstrap.inc): the same overwrite
 *     Set_output_data   every output ciphertext is written to <GEN_PARITY_OUT>.<call>.0 in the ACEHCT01 layout -- the file the
 *                       product writes under ACEHIP_DUMP_OUTPUT -- before the reference consumes it (rtlib.c:82-87)
 * Everything between input and output is deterministic (polynomial.c, ckks_evaluator.c, ckks_bootstrap_context.c), so the
 * product run with the same ACEHIP_SEED must produce the same bytes: tests/test_gpu_gen_parity.py compares the sha256 of the files
 * with the digests committed in tests/golden/gen_parity.json (made by tests/golden/gen_gen_parity.sh in the dev container).
 *
 * env: GEN_PARITY_SEED (required; = the product's ACEHIP_SEED), GEN_PARITY_OUT (output prefix), GEN_PARITY_BATCH / GEN_PARITY_ENC_SKIP (see
 *      Prepare_input), GEN_PARITY_KEYS (optional: write
 *      the injected key set as ACEHKEY1 -- comparable with the product's Acehip_rt_save_keys file), MODEL_ENC_SEED handled by
 *      the caller through Acehip_rt_seed_encryptor (defined here too).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"
#include "common/io_api.h"
#include "rtlib/context.h"
#include "util/ckks_bootstrap_context.h"
#include "util/ckks_encoder.h"
#include "util/ckks_encryptor.h"
#include "util/ckks_evaluator.h"
#include "util/ckks_key_generator.h"
#include "util/ckks_parameters.h"
#include "util/crt.h"

typedef unsigned long long u64;
typedef unsigned __int128  u128;
#define REF_BUILD 1
#include "ref_containers.h"

/* ---------------------------------------------------------------- std::mt19937_64 ---- */
typedef struct { u64 mt[312]; int i; } MT;
static void mt_seed(MT* r, u64 s) {
  r->mt[0] = s;
  for (int i = 1; i < 312; ++i) r->mt[i] = 6364136223846793005ull * (r->mt[i - 1] ^ (r->mt[i - 1] >> 62)) + (u64)i;
  r->i = 312;
}
static u64 mt_next(MT* r) {
  if (r->i >= 312) {
    for (int i = 0; i < 312; ++i) {
      u64 x = (r->mt[i] & 0xFFFFFFFF80000000ull) | (r->mt[(i + 1) % 312] & 0x7FFFFFFFull);
      r->mt[i] = r->mt[(i + 156) % 312] ^ (x >> 1) ^ ((x & 1) ? 0xB5026F5AA96619E9ull : 0);
    }
    r->i = 0;
  }
  u64 x = r->mt[r->i++];
  x ^= (x >> 29) & 0x5555555555555555ull;
  x ^= (x << 17) & 0x71D67FFFEDA60000ull;
  x ^= (x << 37) & 0xFFF7EEE000000000ull;
  x ^= x >> 43;
  return x;
}

/* ---------------------------------------------------------------- the product's derivation (rt_context.cpp) ---- */
static u64 Key_seed;
static MT  Enc_rng; /* the calling thread's encryption stream (single-threaded here) */
static int Seeded;
static u64 splitmix(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
#define KEY_TAG_SECRET 1ull
#define KEY_TAG_PUBLIC 2ull
#define KEY_TAG_RELIN 3ull
#define KEY_TAG_AUTO (1ull << 34)
static void key_rng(MT* r, u64 tag) { mt_seed(r, splitmix(Key_seed ^ splitmix(tag))); }
static void triangle(int64_t* v, size_t n, MT* r) {
  for (size_t i = 0; i < n; ++i) {
    u64 x = mt_next(r) & 3;
    v[i]  = x == 0 ? -1 : (x == 1 ? 1 : 0);
  }
}
static void ternary(int64_t* v, size_t n, size_t hw, MT* r) {
  if (hw == 0) {
    for (size_t i = 0; i < n; ++i) v[i] = (int64_t)(mt_next(r) % 3) - 1;
    return;
  }
  if (hw > n) hw = n;
  int64_t ones = -1000000;
  while (ones < (int64_t)hw / 2 - 1 || ones > (int64_t)hw / 2 + 1) {
    ones = 0;
    memset(v, 0, n * sizeof(*v));
    size_t weight = 0;
    while (weight < hw) {
      size_t idx = mt_next(r) % n;
      if (v[idx] == 0) {
        if (mt_next(r) & 1) { v[idx] = 1; ++ones; } else v[idx] = -1;
        ++weight;
      }
    }
  }
}
static u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static CRT_CONTEXT* crt_ctx(void) { return ((CKKS_PARAMETER*)Get_param(Context))->_crt_context; }
static u64 prime_at(uint32_t gi) { /* q primes then p primes */
  CRT_CONTEXT* crt = crt_ctx();
  uint32_t     L   = Get_primes_cnt(Get_q(crt));
  return gi < L ? (u64)Get_modulus_val(Get_prime_at(Get_q(crt), gi)) : (u64)Get_modulus_val(Get_prime_at(Get_p(crt), gi - L));
}
/* limb `pos` of a polynomial with `level` q-limbs followed by its p-limbs: its storage and its prime */
static u64* limb_of(POLYNOMIAL* p, uint32_t pos) {
  size_t n = p->_ring_degree;
  return (u64*)p->_data + (pos < p->_num_primes ? (size_t)pos : (size_t)(p->_num_alloc_primes - p->_num_primes_p) + (pos - p->_num_primes)) * n;
}
static u64 prime_of(POLYNOMIAL* p, uint32_t pos) {
  uint32_t L = Get_primes_cnt(Get_q(crt_ctx()));
  return pos < p->_num_primes ? prime_at(pos) : prime_at(L + (pos - (uint32_t)p->_num_primes));
}
/* sample_uniform_kernel (csrc/rt_kernels.hip): limb position `pos`, coefficient n -> 124 random bits mod q */
static void uniform_poly(POLYNOMIAL* p, uint32_t n_limbs, u64 seed) {
  size_t N = p->_ring_degree;
  for (uint32_t pos = 0; pos < n_limbs; ++pos) {
    u64 *d = limb_of(p, pos), q = prime_of(p, pos);
    for (size_t n = 0; n < N; ++n) {
      const u64 ctr = ((u64)pos << 32) | n;
      const u64 a   = mix64(seed + 0x9E3779B97F4A7C15ull * (2 * ctr + 1));
      const u64 b   = mix64(a ^ (seed * 0xD1342543DE82EF95ull + 2 * ctr + 2));
      d[n]          = (u64)((((u128)(b >> 4) << 64) | a) % q);
    }
  }
  p->_is_ntt = TRUE;
}
/* small signed values -> residues on every limb -> NTT domain (poly_from_small + poly_ntt) */
static void small_to_ntt(POLYNOMIAL* p, const int64_t* v) {
  size_t   N = p->_ring_degree;
  uint32_t T = (uint32_t)(p->_num_primes + p->_num_primes_p);
  for (uint32_t pos = 0; pos < T; ++pos) {
    u64 *d = limb_of(p, pos), q = prime_of(p, pos);
    for (size_t n = 0; n < N; ++n) d[n] = v[n] < 0 ? q - (u64)(-v[n]) : (u64)v[n];
  }
  p->_is_ntt = FALSE;
  Conv_poly2ntt_inplace(p, crt_ctx());
}
static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

/* the secret key of the seed, coefficient and NTT form over all L + K limbs */
static int64_t*   Sk_coef;
static POLYNOMIAL Sk_ntt;

/* make_switch_key (rt_context.cpp; Generate_switching_key ckks_key_generator.c:127-200): b_j = e_j + P*new[digit j] - a_j*old */
static void seeded_switch_key(SWITCH_KEY* k, POLYNOMIAL* new_ntt, POLYNOMIAL* old_ntt, u64 tag) {
  CKKS_PARAMETER* prm = (CKKS_PARAMETER*)Get_param(Context);
  CRT_CONTEXT*    crt = crt_ctx();
  const uint32_t  L = Get_primes_cnt(Get_q(crt)), K = Get_primes_cnt(Get_p(crt)), T = L + K;
  const size_t    N = prm->_poly_degree, dnum = prm->_num_q_parts, alpha = Get_per_part_size(Get_qpart(crt));
  MT              rng;
  key_rng(&rng, tag);
  int64_t*   tri = (int64_t*)malloc(N * sizeof(int64_t));
  POLYNOMIAL e;
  Alloc_poly_data(&e, N, L, K);
  for (size_t j = 0; j < dnum; ++j) {
    PUBLIC_KEY* pk = Get_swk_at(k, j);
    POLYNOMIAL *b = Get_pk0(pk), *a = Get_pk1(pk);
    if (b->_num_primes != L || b->_num_primes_p != K || a->_num_primes != L || a->_num_primes_p != K) {
      fprintf(stderr, "gen_parity_ref: switch key part of unexpected shape\n");
      exit(3);
    }
    const int a_ntt = a->_is_ntt, b_ntt = b->_is_ntt;
    uniform_poly(a, T, mt_next(&rng));
    triangle(tri, N, &rng);
    small_to_ntt(&e, tri);
    for (uint32_t i = 0; i < T; ++i) {
      const u64 q = prime_of(a, i);
      u64       pm_scale = 0;
      if (i < L && i / alpha == j) {
        pm_scale = 1;
        for (uint32_t t = 0; t < K; ++t) pm_scale = mulmod(pm_scale, prime_at(L + t) % q, q);
      }
      u64 *bd = limb_of(b, i), *ad = limb_of(a, i), *ed = limb_of(&e, i), *nd = limb_of(new_ntt, i), *od = limb_of(old_ntt, i);
      for (size_t n = 0; n < N; ++n) {
        u64 pm = pm_scale ? mulmod(nd[n], pm_scale, q) : 0;
        pm += ed[n];
        if (pm >= q) pm -= q;
        const u64 ao = mulmod(ad[n], od[n], q);
        bd[n] = pm >= ao ? pm - ao : pm + q - ao;
      }
    }
    a->_is_ntt = a_ntt;
    b->_is_ntt = b_ntt;
  }
  Free_poly_data(&e);
  free(tri);
}

/* injected automorphism keys so far (keys appear lazily with new bootstrap slot counts) */
static uint32_t* Done_auto;
static size_t    N_done_auto, Cap_done_auto;
static int auto_done(uint32_t k) {
  for (size_t i = 0; i < N_done_auto; ++i) if (Done_auto[i] == k) return 1;
  return 0;
}
static void inject_auto_keys(void) {
  CKKS_KEY_GENERATOR* g = (CKKS_KEY_GENERATOR*)Get_key_gen(Context);
  CRT_CONTEXT*        crt = crt_ctx();
  const uint32_t      L = Get_primes_cnt(Get_q(crt)), K = Get_primes_cnt(Get_p(crt));
  const size_t        N = ((CKKS_PARAMETER*)Get_param(Context))->_poly_degree;
  AUTO_KEY_MAP *      ck, *tk;
  size_t              n_new = 0;
  int64_t*            rot = (int64_t*)malloc(N * sizeof(int64_t));
  POLYNOMIAL          old;
  Alloc_poly_data(&old, N, L, K);
  HASH_ITER(HH, g->_auto_key_map, ck, tk) {
    const uint32_t k = ck->_precomp_auto_idx;
    if (auto_done(k)) continue;
    /* old key = sigma_{k^-1}(s): s(X) -> s(X^inv), inv = k^-1 mod 2N (ensure_auto_key; Generate_rot_key fast variant :238-266) */
    u64 inv = 1, base = k, e = N - 1, m = 2ull * N;
    for (; e; e >>= 1) {
      if (e & 1) inv = inv * base % m;
      base = base * base % m;
    }
    for (size_t i = 0; i < N; ++i) {
      const u64 idx = (u64)i * inv % m;
      if (idx < N) rot[idx] = Sk_coef[i];
      else rot[idx - N] = -Sk_coef[i];
    }
    small_to_ntt(&old, rot);
    seeded_switch_key(ck->_auto_key, &Sk_ntt, &old, KEY_TAG_AUTO + k);
    if (N_done_auto == Cap_done_auto) {
      Cap_done_auto = Cap_done_auto ? 2 * Cap_done_auto : 64;
      Done_auto     = (uint32_t*)realloc(Done_auto, Cap_done_auto * sizeof(uint32_t));
    }
    Done_auto[N_done_auto++] = k;
    ++n_new;
  }
  Free_poly_data(&old);
  free(rot);
  if (n_new) printf("[gen_parity_ref] %zu automorphism keys derived from seed %llu (%zu in all)\n", n_new, Key_seed, N_done_auto);
}

static void inject_keys(void) {
  CKKS_KEY_GENERATOR* g = (CKKS_KEY_GENERATOR*)Get_key_gen(Context);
  CKKS_PARAMETER*     prm = (CKKS_PARAMETER*)Get_param(Context);
  CRT_CONTEXT*        crt = crt_ctx();
  const uint32_t      L = Get_primes_cnt(Get_q(crt)), K = Get_primes_cnt(Get_p(crt));
  const size_t        N = prm->_poly_degree;
  MT                  rng;
  /* secret key (generate_keys: sample_ternary on the key's own generator) */
  Sk_coef = (int64_t*)malloc(N * sizeof(int64_t));
  key_rng(&rng, KEY_TAG_SECRET);
  ternary(Sk_coef, N, prm->_hamming_weight, &rng);
  Alloc_poly_data(&Sk_ntt, N, L, K);
  small_to_ntt(&Sk_ntt, Sk_coef);
  POLYNOMIAL *ref_s = Get_sk_poly(Get_sk(g)), *ref_ns = Get_ntt_sk(Get_sk(g));
  for (uint32_t pos = 0; pos < L + K; ++pos) {
    if (pos >= ref_ns->_num_primes + ref_ns->_num_primes_p) break;
    memcpy(limb_of(ref_ns, pos), limb_of(&Sk_ntt, pos), N * 8);
    u64 *d = limb_of(ref_s, pos), q = prime_of(ref_s, pos);
    for (size_t n = 0; n < N; ++n) d[n] = Sk_coef[n] < 0 ? q - 1 : (u64)Sk_coef[n];
  }
  /* public key: pk1 = a (uniform, NTT domain), pk0 = e - a*s */
  key_rng(&rng, KEY_TAG_PUBLIC);
  POLYNOMIAL *pk0 = Get_pk0(Get_pk(g)), *pk1 = Get_pk1(Get_pk(g));
  if (pk0->_num_primes != L || pk1->_num_primes != L) { fprintf(stderr, "gen_parity_ref: public key of unexpected shape\n"); exit(3); }
  const int pk0_ntt = pk0->_is_ntt, pk1_ntt = pk1->_is_ntt; /* (the flags stay what the reference's key generator left) */
  uniform_poly(pk1, L, mt_next(&rng));
  int64_t*   tri = (int64_t*)malloc(N * sizeof(int64_t));
  POLYNOMIAL e;
  Alloc_poly_data(&e, N, L, 0);
  triangle(tri, N, &rng);
  small_to_ntt(&e, tri);
  for (uint32_t i = 0; i < L; ++i) {
    const u64 q = prime_at(i);
    u64 *b = limb_of(pk0, i), *a = limb_of(pk1, i), *ed = limb_of(&e, i), *s = limb_of(&Sk_ntt, i);
    for (size_t n = 0; n < N; ++n) {
      const u64 as = mulmod(a[n], s[n], q);
      b[n] = ed[n] >= as ? ed[n] - as : ed[n] + q - as;
    }
  }
  pk0->_is_ntt = pk0_ntt;
  pk1->_is_ntt = pk1_ntt;
  Free_poly_data(&e);
  free(tri);
  /* relinearisation key: new = s^2 on the q-limbs (p-limbs 0), old = s */
  POLYNOMIAL s2;
  Alloc_poly_data(&s2, N, L, K);
  memset(s2._data, 0, (size_t)(L + K) * N * 8);
  for (uint32_t i = 0; i < L; ++i) {
    const u64 q = prime_at(i);
    u64 *d = limb_of(&s2, i), *s = limb_of(&Sk_ntt, i);
    for (size_t n = 0; n < N; ++n) d[n] = mulmod(s[n], s[n], q);
  }
  seeded_switch_key(Get_relin_key(g), &s2, &Sk_ntt, KEY_TAG_RELIN);
  Free_poly_data(&s2);
  inject_auto_keys();
}

/* ---------------------------------------------------------------- interposed entry points ---- */
static void* real(const char* name) {
  void* f = dlsym(RTLD_NEXT, name);
  if (!f) { fprintf(stderr, "gen_parity_ref: %s not found behind this program\n", name); exit(3); }
  return f;
}
static void setup_seed(void) {
  if (Seeded) return;
  const char* s = getenv("GEN_PARITY_SEED");
  if (!s) { fprintf(stderr, "gen_parity_ref: GEN_PARITY_SEED is not set\n"); exit(2); }
  Key_seed = strtoull(s, NULL, 10);
  mt_seed(&Enc_rng, Key_seed); /* Prepare_context: c->rng.seed(seed) */
  Seeded = 1;
}
void Acehip_rt_seed_encryptor(uint64_t seed) { mt_seed(&Enc_rng, seed); }
void Acehip_rt_set_batch(uint32_t n) { (void)n; }    /* (tools/model_main.c names these; one image at a time here) */
void Acehip_rt_select_image(uint32_t k) { (void)k; }

void Prepare_context(void) {
  int first = Context == NULL;
  ((void (*)(void))real("Prepare_context"))();
  if (!first) return;
  setup_seed();
  inject_keys();
}

CIPHER Bootstrap(CIPHER res, CIPHER ciph, uint32_t level_after_bts) {
  CKKS_BTS_CTX* bts = Get_bts_ctx((CKKS_EVALUATOR*)Get_eval(Context));
  uint32_t      slots = Get_ciph_slots(ciph);
  if (!Get_bts_precom(bts, slots)) { /* cipher_eval.c:370-373 would do this inside, with random keys */
    Bootstrap_precom(slots);
    inject_auto_keys();
  }
  return ((CIPHER(*)(CIPHER, CIPHER, uint32_t))real("Bootstrap"))(res, ciph, level_after_bts);
}

/* encrypt() of csrc/rt/rt_encode.cpp (Encrypt_msg ckks_encryptor.c:20-95) on a ciphertext the reference has just made from
 * the same plaintext: c0 = pk0*v + e1 + m, c1 = pk1*v + e2 with v, e1, e2 from the thread's stream */
static void seeded_encrypt(CIPHER ct, PLAINTEXT* plain) {
  POLYNOMIAL*     m = Get_plain_poly(plain);
  const uint32_t  l = (uint32_t)m->_num_primes;
  const size_t    N = m->_ring_degree;
  CKKS_KEY_GENERATOR* g = (CKKS_KEY_GENERATOR*)Get_key_gen(Context);
  POLYNOMIAL *pk0 = Get_pk0(Get_pk(g)), *pk1 = Get_pk1(Get_pk(g));
  POLYNOMIAL  v, e1, e2;
  int64_t*    tri = (int64_t*)malloc(N * sizeof(int64_t));
  Alloc_poly_data(&v, N, l, 0);
  Alloc_poly_data(&e1, N, l, 0);
  Alloc_poly_data(&e2, N, l, 0);
  triangle(tri, N, &Enc_rng); small_to_ntt(&v, tri);
  triangle(tri, N, &Enc_rng); small_to_ntt(&e1, tri);
  triangle(tri, N, &Enc_rng); small_to_ntt(&e2, tri);
  if (ct->_c0_poly._num_primes != l || ct->_c1_poly._num_primes != l || !m->_is_ntt) {
    fprintf(stderr, "gen_parity_ref: ciphertext / plaintext of unexpected shape\n");
    exit(3);
  }
  for (uint32_t i = 0; i < l; ++i) {
    const u64 q = prime_at(i);
    u64 *c0 = limb_of(&ct->_c0_poly, i), *c1 = limb_of(&ct->_c1_poly, i), *md = limb_of(m, i);
    u64 *p0 = limb_of(pk0, i), *p1 = limb_of(pk1, i), *vd = limb_of(&v, i), *a = limb_of(&e1, i), *b = limb_of(&e2, i);
    for (size_t n = 0; n < N; ++n) {
      u64 x = mulmod(p0[n], vd[n], q) + a[n];
      if (x >= q) x -= q;
      x += md[n];
      if (x >= q) x -= q;
      c0[n] = x;
      u64 y = mulmod(p1[n], vd[n], q) + b[n];
      if (y >= q) y -= q;
      c1[n] = y;
    }
  }
  Free_poly_data(&v);
  Free_poly_data(&e1);
  Free_poly_data(&e2);
  free(tri);
}
static void skip_encryptions(size_t k) {
  const size_t N = ((CKKS_PARAMETER*)Get_param(Context))->_poly_degree;
  for (size_t i = 0; i < 3 * N * k; ++i) (void)mt_next(&Enc_rng);
}
static size_t env_count(const char* name, size_t dflt) { return getenv(name) ? strtoul(getenv(name), NULL, 10) : dflt; }

VALUE_LIST* Pre_encode_scheme(TENSOR* image, DATA_SCHEME* scheme); /* rtlib.c:20-32: exported, in no header */
void        Ref_sampler_start(void);                                  /* tests/c/ref_sampler.c */
void        Ref_sampler_stop(void);
/* GEN_PARITY_BATCH=B, GEN_PARITY_ENC_SKIP=k: this run is image k of a batch of B under a program that knows nothing of batches --
 * the product's Prepare_input encrypts the tensor B times in a row from one stream (rt_io.cpp), image k takes the k-th */
void Prepare_input(TENSOR* input, const char* name) {
  ((void (*)(TENSOR*, const char*))real("Prepare_input"))(input, name);
  CIPHER ct = (CIPHER)Io_get_input(name, 0);
  /* the plaintext again (rtlib.c:41-48; deterministic) */
  VALUE_LIST* vec   = Pre_encode_scheme(input, Get_encode_scheme(0));
  PLAINTEXT*  plain = Alloc_plaintext();
  Encode_internal(plain, (CKKS_ENCODER*)Context->_encoder, vec, 0); /* ENCODE(plain, encoder, vec) of rtlib.c:47: slots = 0 */
  const size_t batch = env_count("GEN_PARITY_BATCH", 1), k = env_count("GEN_PARITY_ENC_SKIP", 0);
  skip_encryptions(k);
  seeded_encrypt(ct, plain);
  skip_encryptions(batch - 1 - k);
  Free_value_list(vec);
  Free_plaintext(plain);
  Ref_sampler_start(); /* REF_SAMPLER_OUT: profile from here to the first output (tests/c/ref_sampler.c) */
}

/* an encryption inside Main_graph (eg_fhertlib_bootstrap.inc): one for the whole batch on the product side */
CIPHER Encrypt(CIPHER res, PLAIN plain) {
  ((CIPHER(*)(CIPHER, PLAIN))real("Encrypt"))(res, plain);
  seeded_encrypt(res, plain);
  return res;
}

void Set_output_data(const char* name, size_t idx, CIPHER data) {
  static unsigned n_call = 0;
  Ref_sampler_stop();
  const char*     prefix = getenv("GEN_PARITY_OUT");
  if (prefix) {
    char path[1200];
    snprintf(path, sizeof(path), "%s.%u.0", prefix, n_call);
    save_ciph(path, data);
  }
  ++n_call;
  ((void (*)(const char*, size_t, CIPHER))real("Set_output_data"))(name, idx, data);
}

void Finalize_context(void) {
  const char* kpath = getenv("GEN_PARITY_KEYS");
  if (kpath && Context) write_keys(kpath);
  ((void (*)(void))real("Finalize_context"))();
}
