/* ct_parity.c -- ciphertext-level bit-exact parity with INJECTED keys (test infrastructure, our own program).
 *
 * One source, two builds:
 *   -DREF_BUILD : linked against the reference rtlib (oracle/_ref/libref_rtlib.so, make -C oracle refct).  Mode "dump":
 *                 Prepare_context (random keys), write every key to DIR/keys.bin, encrypt two messages, run the
 *                 operation script below and write every ciphertext to DIR/<name>.ct.
 *   default     : linked against OUR libFHErt_ant.so.  Mode "check": Prepare_context with ACEHIP_KEYS_FILE=DIR/keys.bin
 *                 (the keys of the reference run are loaded instead of generated), load the two input ciphertexts, run
 *                 THE SAME script, write DIR/got_<name>.ct and compare every output with the reference's file byte for
 *                 byte.  Every step is deterministic given keys and inputs (ckks_evaluator.c:45-600,
 *                 ckks_bootstrap_context.c:1584-1860), so anything but equality is a bug.
 *
 * File formats (little endian; the product side is include/rt_ant/rt_api.h Acehip_rt_save_ciph / Acehip_rt_load_keys):
 *   ciphertext "ACEHCT01": u32 n_polys, N, level, num_p, is_ntt, slots, sf_degree, pad; f64 scaling_factor;
 *                          then per poly `level` q-limbs and `num_p` p-limbs of N u64
 *   keys       "ACEHKEY1": see write_keys() below
 *
 * The REVERSE direction (does the reference accept what WE generate?):
 *   default build, mode "make": Prepare_context with OUR key generation (ACEHIP_SEED), encrypt the two messages and an all-zero
 *                 one with OUR encryptor, run the script, write keys.bin (Acehip_rt_save_keys), the inputs and every result.
 *   -DREF_BUILD, mode "load":  Prepare_context of the reference, then every key is REPLACED by ours from keys.bin (secret, public,
 *                 relinearisation, rotation keys), our input ciphertexts are read, the reference runs the same script and must
 *                 reproduce every one of our results byte for byte -- and its own decryptor must return the messages.
 *
 * usage: ct_parity dump|check|make|load DIR N mul_depth q0_bits sf_bits dnum hamming slots level_after [rot ...]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/rtlib.h"
#include "rt_ant/rt_ant.h"

#ifdef REF_BUILD
#include <time.h>
void Ref_sampler_start(void); /* tests/c/ref_sampler.c (inert without REF_SAMPLER_OUT) */
void Ref_sampler_stop(void);
#include "rtlib/context.h"
#include "util/ckks_key_generator.h"
#include "util/ckks_parameters.h"
#include "util/crt.h"
#endif

typedef unsigned long long u64;
static CKKS_PARAMS* Parm; /* ends in a flexible array of rotation indices */
static int32_t     Rot[64];
static uint32_t    Slots, Level_after, N_rot;
static const char* Dir;
static int         Check, Fail;

static void path_of(char* buf, const char* prefix, const char* name, const char* ext) { sprintf(buf, "%s/%s%s.%s", Dir, prefix, name, ext); }

/* ---------------------------------------------------------------- raw access to polynomials ---- */
#ifdef REF_BUILD
#include "ref_containers.h"
/* ---- the reverse direction: OUR containers into the reference's structures ---- */
static void read_poly(FILE* f, POLYNOMIAL* p, const char* what) {
  size_t n = p->_ring_degree;
  if (fread(p->_data, 8, p->_num_primes * n, f) != p->_num_primes * n) { fprintf(stderr, "short read: %s\n", what); exit(3); }
  if (p->_num_primes_p &&
      fread(p->_data + (p->_num_alloc_primes - p->_num_primes_p) * n, 8, p->_num_primes_p * n, f) != p->_num_primes_p * n) {
    fprintf(stderr, "short read: %s\n", what);
    exit(3);
  }
}
static void load_ciph_ref(const char* path, CIPHER c) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(3); }
  char     magic[8];
  uint32_t h[8];
  double   sf;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "ACEHCT01", 8) != 0 || fread(h, 4, 8, f) != 8 || fread(&sf, 8, 1, f) != 1 || h[0] != 2) {
    fprintf(stderr, "%s: not an ACEHCT01 ciphertext\n", path);
    exit(3);
  }
  memset(c, 0, sizeof(*c));
  Alloc_poly_data(&c->_c0_poly, h[1], h[2], h[3]);
  Alloc_poly_data(&c->_c1_poly, h[1], h[2], h[3]);
  c->_c0_poly._is_ntt = c->_c1_poly._is_ntt = h[4] != 0;
  c->_slots = h[5];
  c->_sf_degree = h[6];
  c->_scaling_factor = sf;
  read_poly(f, &c->_c0_poly, path);
  read_poly(f, &c->_c1_poly, path);
  fclose(f);
}
static void read_swk(FILE* f, SWITCH_KEY* k, size_t dnum, const char* what) {
  for (size_t j = 0; j < dnum; ++j) {
    PUBLIC_KEY* pk = Get_swk_at(k, j);
    read_poly(f, Get_pk0(pk), what);
    read_poly(f, Get_pk1(pk), what);
  }
}
/* every key of the prepared reference context is overwritten by the one OUR library generated ("ACEHKEY1" with a secret key) */
static void read_keys(const char* path) {
  CKKS_KEY_GENERATOR* g = (CKKS_KEY_GENERATOR*)Get_key_gen(Context);
  CKKS_PARAMETER*     prm = (CKKS_PARAMETER*)Get_param(Context);
  CRT_CONTEXT*        crt = prm->_crt_context;
  uint32_t            L = Get_primes_cnt(Get_q(crt)), K = Get_primes_cnt(Get_p(crt));
  FILE*               f = fopen(path, "rb");
  if (!f) { perror(path); exit(3); }
  char     magic[8];
  uint32_t h[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "ACEHKEY1", 8) != 0 || fread(h, 4, 8, f) != 8) { fprintf(stderr, "%s: not a key file\n", path); exit(3); }
  if (h[0] != 1 || h[1] != prm->_poly_degree || h[2] != L || h[3] != K || h[4] != prm->_num_q_parts || (h[7] & 1)) {
    fprintf(stderr, "%s: written for other parameters (or without a secret key)\n", path);
    exit(3);
  }
  for (uint32_t i = 0; i < L + K; ++i) {
    int64_t q, want = i < L ? Get_modulus_val(Get_prime_at(Get_q(crt), i)) : Get_modulus_val(Get_prime_at(Get_p(crt), i - L));
    if (fread(&q, 8, 1, f) != 1 || q != want) { fprintf(stderr, "%s: prime %u differs\n", path, i); exit(3); }
  }
  read_poly(f, Get_ntt_sk(Get_sk(g)), "secret key");
  read_poly(f, Get_pk0(Get_pk(g)), "public key");
  read_poly(f, Get_pk1(Get_pk(g)), "public key");
  read_swk(f, Get_relin_key(g), prm->_num_q_parts, "relinearisation key");
  uint32_t n_map = 0, n_keys = 0;
  for (uint32_t i = 0; i < h[5]; ++i) {
    int32_t  rot;
    uint32_t k;
    if (fread(&rot, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1) { fprintf(stderr, "short read: rotation map\n"); exit(3); }
    uint32_t ours = Get_precomp_auto_idx(g, rot);  /* the reference derives the same automorphism index for the rotation */
    if (ours != 0 && ours != k) { fprintf(stderr, "rotation %d: automorphism index %u here, %u in the file\n", rot, ours, k); exit(3); }
    n_map += ours != 0;
  }
  size_t key_bytes = (size_t)prm->_num_q_parts * 2 * (L + K) * prm->_poly_degree * 8;
  for (uint32_t i = 0; i < h[6]; ++i) {
    uint32_t e[2];
    if (fread(e, 4, 2, f) != 2) { fprintf(stderr, "short read: automorphism key header\n"); exit(3); }
    SWITCH_KEY* k = Get_auto_key(g, e[0]);
    if (k != NULL) {
      read_swk(f, k, prm->_num_q_parts, "automorphism key");
      n_keys++;
    } else if (fseek(f, (long)key_bytes, SEEK_CUR) != 0) {  /* a key this context never asked for */
      fprintf(stderr, "short read: automorphism key %u\n", e[0]);
      exit(3);
    }
  }
  fclose(f);
  printf("keys injected into the reference: L=%u K=%u dnum=%u, %u of %u rotations known here, %u of %u automorphism keys replaced\n", L, K, h[4],
         n_map, h[5], n_keys, h[6]);
}
#else
static void save_ciph(const char* path, CIPHER c) { Acehip_rt_save_ciph(path, c); }
static void save_ciph3(const char* path, CIPHER3 c) { Acehip_rt_save_ciph3(path, c); }
static void save_plain(const char* path, PLAIN c) { Acehip_rt_save_plain(path, c); }
#endif

/* ---------------------------------------------------------------- compare / record ---- */
static int same_file(const char* a, const char* b, char* why) {
  FILE *fa = fopen(a, "rb"), *fb = fopen(b, "rb");
  if (!fa || !fb) { sprintf(why, "cannot open %s", fa ? b : a); if (fa) fclose(fa); if (fb) fclose(fb); return 0; }
  size_t off = 0;
  int    ok = 1;
  static unsigned char ba[1 << 16], bb[1 << 16];
  for (;;) {
    size_t na = fread(ba, 1, sizeof(ba), fa), nb = fread(bb, 1, sizeof(bb), fb);
    if (na != nb) { sprintf(why, "length differs near byte %zu", off + (na < nb ? na : nb)); ok = 0; break; }
    if (na == 0) break;
    if (memcmp(ba, bb, na) != 0) {
      size_t i = 0;
      while (ba[i] == bb[i]) ++i;
      size_t at = off + i;
      if (at < 48) sprintf(why, "header differs at byte %zu", at);
      else sprintf(why, "first difference at word %zu of the payload", (at - 48) / 8);
      ok = 0;
      break;
    }
    off += na;
  }
  fclose(fa);
  fclose(fb);
  return ok;
}
static void verdict(const char* name) {
  if (!Check) return;
  char ref[1024], got[1024], why[1200];
  path_of(ref, "", name, "ct");
  path_of(got, "got_", name, "ct");
  if (same_file(ref, got, why)) {
    printf("MATCH %s\n", name);
  } else {
    printf("MISMATCH %s: %s\n", name, why);
    Fail++;
  }
}
/* CT_PARITY_TIMING_ONLY=1 (bench.py's measured CPU / GPU pair): the same script with nothing written or compared -- no ciphertext
 * or key files on either side, only the decrypted messages -- so that the span holds the operators alone; both builds print its
 * wall-clock seconds (script_wall_s; the product build after Acehip_rt_sync) */
static int Timing_only = 0;
static void out_ciph(const char* name, CIPHER c) {
  if (Timing_only) return;
  char p[1024];
  path_of(p, Check ? "got_" : "", name, "ct");
  save_ciph(p, c);
  verdict(name);
}
static void out_ciph3(const char* name, CIPHER3 c) {
  if (Timing_only) return;
  char p[1024];
  path_of(p, Check ? "got_" : "", name, "ct");
  save_ciph3(p, c);
  verdict(name);
}
static void out_plain(const char* name, PLAIN c) {
  if (Timing_only) return;
  char p[1024];
  path_of(p, Check ? "got_" : "", name, "ct");
  save_plain(p, c);
  verdict(name);
}
/* decoded message: doubles, compared bit for bit as well and reported separately (decode is FP64 after an exact CRT) */
static void out_msg(const char* name, CIPHER c) {
  double* m = Get_msg(c);
  char    p[1024];
  path_of(p, Check ? "got_" : "", name, "msg");
  FILE* f = fopen(p, "wb");
  fwrite(m, 8, Slots, f);
  fclose(f);
  if (Check) {
    char    r[1024];
    path_of(r, "", name, "msg");
    double* e = (double*)malloc(8 * Slots);
    FILE*   g = fopen(r, "rb");
    if (!g || fread(e, 8, Slots, g) != Slots) { printf("MISMATCH msg_%s: cannot read %s\n", name, r); Fail++; if (g) fclose(g); free(e); free(m); return; }
    fclose(g);
    double worst = 0;
    int    exact = memcmp(e, m, 8 * Slots) == 0;
    for (uint32_t i = 0; i < Slots; ++i) if (fabs(e[i] - m[i]) > worst) worst = fabs(e[i] - m[i]);
    printf("%s msg_%s: max |diff| %.3e\n", exact ? "MATCH" : "MISMATCH", name, worst);
    if (!exact) Fail++;
    free(e);
  } else {
    printf("msg_%s[0..3] = %.9f %.9f %.9f %.9f\n", name, m[0], m[1], m[2], m[3]);
  }
  free(m);
}

#define ZERO(x) memset(&(x), 0, sizeof(x))

/* ---------------------------------------------------------------- the operation script ---- */
static void script(CIPHERTEXT a, CIPHERTEXT b) {
  uint32_t level = (uint32_t)Level(&a);
  double*  w = (double*)malloc(8 * Slots);
  for (uint32_t i = 0; i < Slots; ++i) w[i] = cos(0.11 * i) * 0.75;
  out_ciph("in_a", &a);
  out_msg("in_a", &a);
  /* HAdd / HSub / plaintext add */
  CIPHERTEXT r;
  ZERO(r);
  Add_ciph(&r, &a, &b);
  out_ciph("add", &r);
  Free_ciph_poly(&r, 1);
  ZERO(r);
  Sub_ciph(&r, &a, &b);
  out_ciph("sub", &r);
  Free_ciph_poly(&r, 1);
  PLAINTEXT pt;
  ZERO(pt);
  Encode_plain_from_double(&pt, w, Slots, 1, level);
  out_plain("plain", &pt);
  ZERO(r);
  Add_plain(&r, &a, &pt);
  out_ciph("add_plain", &r);
  Free_ciph_poly(&r, 1);
  /* plaintext multiply + rescale */
  CIPHERTEXT mp, rs;
  ZERO(mp);
  ZERO(rs);
  Mul_plain(&mp, &a, &pt);
  out_ciph("mul_plain", &mp);
  Rescale_ciph(&rs, &mp);
  out_ciph("mul_plain_rescaled", &rs);
  Free_ciph_poly(&mp, 1);
  Free_poly_data(&pt._poly);
  /* HMul as tensor product, relinearise, rescale; and the fused Mul_ciph */
  CIPHERTEXT3 t3;
  ZERO(t3);
  Mul_ciph3(&t3, &a, &b);
  out_ciph3("mul3", &t3);
  CIPHERTEXT rl, rr, mc;
  ZERO(rl);
  ZERO(rr);
  ZERO(mc);
  Relin(&rl, &t3);
  out_ciph("relin", &rl);
  Rescale_ciph(&rr, &rl);
  out_ciph("relin_rescaled", &rr);
  out_msg("relin_rescaled", &rr);
  Mul_ciph(&mc, &a, &b);
  out_ciph("mul", &mc);
  Free_ciph_poly(&rl, 1);
  Free_ciph_poly(&mc, 1);
  Free_poly_data(&t3._c0_poly);
  Free_poly_data(&t3._c1_poly);
  Free_poly_data(&t3._c2_poly);
  /* rotations: at the top level and on the rescaled product (lower level: different ModUp tables) */
  for (uint32_t i = 0; i < N_rot; ++i) {
    char       name[64];
    CIPHERTEXT ro;
    ZERO(ro);
    Rotate_ciph(&ro, &a, Rot[i]);
    sprintf(name, "rot_%d", Rot[i]);
    out_ciph(name, &ro);
    Free_ciph_poly(&ro, 1);
    ZERO(ro);
    Rotate_ciph(&ro, &rr, Rot[i]);
    sprintf(name, "rot_low_%d", Rot[i]);
    out_ciph(name, &ro);
    Free_ciph_poly(&ro, 1);
  }
  /* ModSwitch: drop one limb without scaling */
  Modswitch_ciph(&rr);
  out_ciph("modswitch", &rr);
  Free_ciph_poly(&rr, 1);
  /* Bootstrap: burn levels down to 2 limbs, refresh, and decrypt */
  if (Level_after) {
    CIPHERTEXT cur = rs;
    for (uint32_t i = 0; i < Slots; ++i) w[i] = 1.0;
    while (Level(&cur) > 2) {
      PLAINTEXT  one;
      CIPHERTEXT m2, r2;
      ZERO(one);
      ZERO(m2);
      ZERO(r2);
      Encode_plain_from_double(&one, w, Slots, Sc_degree(&cur), Level(&cur));
      Mul_plain(&m2, &cur, &one);
      Rescale_ciph(&r2, &m2);
      Free_poly_data(&one._poly);
      Free_ciph_poly(&m2, 1);
      Free_ciph_poly(&cur, 1);
      cur = r2;
    }
    out_ciph("low", &cur);
    CIPHERTEXT bt;
    ZERO(bt);
    Bootstrap(&bt, &cur, Level_after);
    printf("bootstrap: level %zu -> %zu (asked %u), sf_degree %u\n", (size_t)Level(&cur), (size_t)Level(&bt), Level_after, Sc_degree(&bt));
    out_ciph("bootstrap", &bt);
    out_msg("bootstrap", &bt);
    /* a second bootstrap from a ciphertext with scale degree 2 (Eval_bootstrap rescales it first) */
    PLAINTEXT  one;
    CIPHERTEXT m2, b2;
    ZERO(one);
    ZERO(m2);
    ZERO(b2);
    Encode_plain_from_double(&one, w, Slots, 1, Level(&cur));
    Mul_plain(&m2, &cur, &one);
    Bootstrap(&b2, &m2, Level_after);
    out_ciph("bootstrap_deg2", &b2);
    Free_poly_data(&one._poly);
    Free_ciph_poly(&m2, 1);
    Free_ciph_poly(&b2, 1);
    Free_ciph_poly(&bt, 1);
    Free_ciph_poly(&cur, 1);
  } else {
    Free_ciph_poly(&rs, 1);
  }
  free(w);
}

bool Main_graph() { return true; }
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
int Get_output_count() { return 1; }
int Get_input_count() { return 2; }

int main(int argc, char** argv) {
  if (argc < 11) {
    fprintf(stderr, "usage: %s dump|check DIR N mul_depth q0_bits sf_bits dnum hamming slots level_after [rot ...]\n", argv[0]);
    return 2;
  }
  const int make_mode = strcmp(argv[1], "make") == 0, load_mode = strcmp(argv[1], "load") == 0;
  Check = strcmp(argv[1], "check") == 0 || load_mode;  /* "load": the reference checks itself against OUR outputs */
  Dir = argv[2];
  Parm = (CKKS_PARAMS*)calloc(1, sizeof(CKKS_PARAMS) + sizeof(int32_t) * 64);
  Parm->_provider = LIB_ANT;
  Parm->_poly_degree = (uint32_t)atoi(argv[3]);
  Parm->_mul_depth = (size_t)atoi(argv[4]);
  Parm->_first_mod_size = (size_t)atoi(argv[5]);
  Parm->_scaling_mod_size = (size_t)atoi(argv[6]);
  Parm->_num_q_parts = (size_t)atoi(argv[7]);
  Parm->_hamming_weight = (size_t)atoi(argv[8]);
  Slots = (uint32_t)atoi(argv[9]);
  Level_after = (uint32_t)atoi(argv[10]);
  N_rot = 0;
  for (int i = 11; i < argc && N_rot < 64; ++i) Rot[N_rot++] = atoi(argv[i]);
  Parm->_num_rot_idx = N_rot;
  for (uint32_t i = 0; i < N_rot; ++i) Parm->_rot_idxs[i] = Rot[i];
  char kpath[1024];
  path_of(kpath, "", "keys", "bin");
#ifdef REF_BUILD
  if (Check && !load_mode) { fprintf(stderr, "the reference build dumps, or loads what the product made\n"); return 2; }
  if (make_mode) { fprintf(stderr, "\"make\" is the product build's mode\n"); return 2; }
#else
  if (!Check && !make_mode) { fprintf(stderr, "the product build checks, or makes what the reference loads\n"); return 2; }
  if (load_mode) { fprintf(stderr, "\"load\" is the reference build's mode\n"); return 2; }
  if (!make_mode) {
    setenv("ACEHIP_KEYS_FILE", kpath, 1);
    setenv("ACEHIP_KEYS_STRICT", "1", 1); /* a key missing from the file is an error, not a reason to generate one */
  }
#endif
  Timing_only = getenv("CT_PARITY_TIMING_ONLY") != NULL && atoi(getenv("CT_PARITY_TIMING_ONLY")) != 0;
  if (Timing_only && Check) { fprintf(stderr, "CT_PARITY_TIMING_ONLY goes with dump (reference build) or make (product build)\n"); return 2; }
  Prepare_context();
  CIPHERTEXT a, b;
#ifdef REF_BUILD
  if (load_mode) {
    char p[1024];
    read_keys(kpath);
    path_of(p, "", "in_a", "ct");
    load_ciph_ref(p, &a);
    path_of(p, "", "in_b", "ct");
    load_ciph_ref(p, &b);
    script(a, b);
    /* the reference's own decryptor on a ciphertext OUR encryptor made: the message comes back */
    double* m = Get_msg(&a);
    double  worst = 0;
    for (uint32_t i = 0; i < Slots; ++i) if (fabs(m[i] - sin(0.37 * i) * 0.5) > worst) worst = fabs(m[i] - sin(0.37 * i) * 0.5);
    printf("reference decrypts our in_a: max |message error| %.3e\n", worst);
    if (!(worst < 1e-3)) Fail++;
    free(m);
    Free_ciph_poly(&a, 1);
    Free_ciph_poly(&b, 1);
    Finalize_context();
    printf(Fail ? "FAILED: %d mismatches\n" : "SUCESS! the reference reproduces every result of our keys and ciphertexts bit for bit (%d)\n", Fail);
    return Fail ? 1 : 0;
  }
  double* x = (double*)malloc(sizeof(double) * Slots);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = sin(0.37 * i) * 0.5;
  TENSOR* t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_a");
  Free_tensor(t);
  for (uint32_t i = 0; i < Slots; ++i) x[i] = cos(0.23 * i + 1.0) * 0.4;
  t = Alloc_tensor(1, 1, 1, Slots, x);
  Prepare_input(t, "in_b");
  Free_tensor(t);
  free(x);
  a = Get_input_data("in_a", 0);
  b = Get_input_data("in_b", 0);
  char p[1024];
  path_of(p, "", "in_b", "ct");
  if (!Timing_only) save_ciph(p, &b);
#else
  char p[1024];
  ZERO(a);
  ZERO(b);
  if (make_mode) {  /* our own encryptor: the two messages of the dump mode, and an all-zero one (noise alone) */
    double* x = (double*)calloc(Slots, sizeof(double));
    TENSOR* t = Alloc_tensor(1, 1, 1, Slots, x);
    Prepare_input(t, "in_a");
    Free_tensor(t);
    a = Get_input_data("in_a", 0);
    path_of(p, "", "zero", "ct");
    if (!Timing_only) save_ciph(p, &a);
    Free_ciph_poly(&a, 1);
    for (uint32_t i = 0; i < Slots; ++i) x[i] = sin(0.37 * i) * 0.5;
    t = Alloc_tensor(1, 1, 1, Slots, x);
    Prepare_input(t, "in_a");
    Free_tensor(t);
    for (uint32_t i = 0; i < Slots; ++i) x[i] = cos(0.23 * i + 1.0) * 0.4;
    t = Alloc_tensor(1, 1, 1, Slots, x);
    Prepare_input(t, "in_b");
    Free_tensor(t);
    free(x);
    a = Get_input_data("in_a", 0);
    b = Get_input_data("in_b", 0);
    path_of(p, "", "in_b", "ct");
    if (!Timing_only) save_ciph(p, &b);
  } else {
    path_of(p, "", "in_a", "ct");
    Acehip_rt_load_ciph(&a, p);
    path_of(p, "", "in_b", "ct");
    Acehip_rt_load_ciph(&b, p);
  }
#endif
#ifdef REF_BUILD
  /* the operator script as a timed span of the REFERENCE: CPU seconds of this process between here and the end of the script
   * (tools/cpu_model_check.py: a real program of the reference timed on the bench host, against what bench.py's CPU model predicts
   * for it); with REF_SAMPLER_OUT set the span is also sampled (tests/c/ref_sampler.c -> tools/ref_profile_report.py) */
  struct timespec ts0_, ts1_;
  clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts0_);
  Ref_sampler_start();
#else
  Acehip_rt_sync();
#endif
  struct timespec tw0_, tw1_;
  clock_gettime(CLOCK_MONOTONIC, &tw0_);
  script(a, b);
#ifndef REF_BUILD
  Acehip_rt_sync(); /* every launch of the script has finished */
#endif
  clock_gettime(CLOCK_MONOTONIC, &tw1_);
  printf("script_wall_s %.3f\n", (tw1_.tv_sec - tw0_.tv_sec) + 1e-9 * (tw1_.tv_nsec - tw0_.tv_nsec));
#ifdef REF_BUILD
  Ref_sampler_stop();
  clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts1_);
  printf("script_cpu_s %.3f\n", (ts1_.tv_sec - ts0_.tv_sec) + 1e-9 * (ts1_.tv_nsec - ts0_.tv_nsec));
  if (!Timing_only) write_keys(kpath); /* after the script: Bootstrap creates the keys of a new slot count on first use */
#endif
  Free_ciph_poly(&a, 1);
  Free_ciph_poly(&b, 1);
#ifndef REF_BUILD
  if (getenv("ACEHIP_CT_PARITY_RESAVE") && Acehip_rt_save_keys(getenv("ACEHIP_CT_PARITY_RESAVE")) != 0) Fail++;
  if (make_mode && !Timing_only && Acehip_rt_save_keys(kpath) != 0) { fprintf(stderr, "cannot write %s\n", kpath); Fail++; }
#endif
  Finalize_context();
  if (Check) printf(Fail ? "FAILED: %d mismatches\n" : "SUCESS! all outputs bit-identical to the reference (%d)\n", Fail);
  if (make_mode) printf(Fail ? "FAILED\n" : "made: keys.bin, in_a.ct, in_b.ct, zero.ct and the results of the script\n");
  return Fail ? 1 : 0;
}
