// rt_serial.cpp -- on-disk containers for ciphertexts, plaintexts and key sets.
//
// The reference has no such format beyond PLAINTEXT_BUFFER (fhe-cmplr/include/fhe/core/rt_encode_api.h:22-27): keys live
// and die with the process (context.c:29-138).  Here 227 switch keys x 135 MiB are generated once and can be kept:
//   * ACEHIP_KEYS_FILE=<path>: Prepare_context loads the key set from <path> when it exists instead of generating one;
//     when it does not exist the keys are generated and Finalize_context writes them (including every rotation key
//     that was created lazily in between).
//   * Acehip_rt_save_keys / Acehip_rt_load_keys: the same, explicitly.  Loading is also the injection hook of the
//     bit-exact ciphertext-level parity tests (tests/c/ct_parity.c): the reference's keys go in, every result must
//     equal the reference's.
//   * Acehip_rt_save_ciph / _ciph3 / _plain and Acehip_rt_load_ciph / _plain: the wire format that splits client
//     (encode + encrypt, decrypt + decode) from server (Main_graph).
//
// Formats, little endian, words are u64 residues in [0, q):
//   "ACEHCT01": u32 n_polys, N, level, num_p, is_ntt, slots, sf_degree, 0; f64 scaling_factor;
//               per polynomial `level` q-limbs then `num_p` p-limbs of N words
//   "ACEHKEY1": u32 version(1), N, L, K, dnum, n_rot, n_auto, 0; u64 primes[L+K];
//               sk (NTT domain) [L+K][N]; pk0 [L][N]; pk1 [L][N]; relin key [dnum][2][L+K][N] (b_j then a_j);
//               n_rot x {i32 rotation, u32 automorphism index}; n_auto x {u32 automorphism index, u32 0, key as above}
#include <cstring>

#include "rt_internal.hpp"

namespace rt {

namespace {
struct File {
  FILE* f = nullptr;
  const char* path;
  File(const char* p, const char* mode) : path(p) { f = fopen(p, mode); }
  ~File() {
    if (f) fclose(f);
  }
  bool put(const void* p, size_t bytes) { return fwrite(p, 1, bytes, f) == bytes; }
  bool get(void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }
};
constexpr size_t kChunkWords = 1u << 20;  // 8 MiB staging
thread_local std::vector<u64> g_stage;

bool put_device(File& f, const u64* d, size_t words) {
  g_stage.resize(std::min(words, kChunkWords));
  for (size_t off = 0; off < words; off += kChunkWords) {
    const size_t n = std::min(kChunkWords, words - off);
    HIPCHK(acehip_memcpy_d2h(g_stage.data(), d + off, n * 8, nullptr));
    if (!f.put(g_stage.data(), n * 8)) return false;
  }
  return true;
}
bool get_device(File& f, u64* d, size_t words) {
  g_stage.resize(std::min(words, kChunkWords));
  for (size_t off = 0; off < words; off += kChunkWords) {
    const size_t n = std::min(kChunkWords, words - off);
    if (!f.get(g_stage.data(), n * 8)) return false;
    HIPCHK(acehip_memcpy_h2d(d + off, g_stage.data(), n * 8, nullptr));
  }
  return true;
}

int save_polys(const char* path, POLYNOMIAL* const* polys, u32 n_polys, u32 slots, double sf, u32 sf_degree) {
  File f(path, "wb");
  if (!f.f) return -1;
  sync();
  const POLYNOMIAL& p0 = *polys[0];
  const u32 h[8] = {n_polys, p0._ring_degree, (u32)p0._num_primes, (u32)p0._num_primes_p, p0._is_ntt ? 1u : 0u, slots, sf_degree, 0};
  bool ok = f.put("ACEHCT01", 8) && f.put(h, sizeof(h)) && f.put(&sf, 8);
  for (u32 i = 0; ok && i < n_polys; ++i) {
    POLYNOMIAL* p = polys[i];
    RT_ASSERT(p->_num_primes == p0._num_primes && p->_num_primes_p == p0._num_primes_p, "save: polynomials of different shape");
    ok = put_device(f, q_limbs(p), p->_num_primes * (size_t)p->_ring_degree);
    if (ok && p->_num_primes_p) ok = put_device(f, p_limbs(p), p->_num_primes_p * (size_t)p->_ring_degree);
  }
  return ok ? 0 : -1;
}

int load_polys(const char* path, POLYNOMIAL* const* polys, u32 n_polys, u32* slots, double* sf, u32* sf_degree) {
  Context& c = ctx();
  File f(path, "rb");
  if (!f.f) return -1;
  char magic[8];
  u32 h[8];
  if (!f.get(magic, 8) || memcmp(magic, "ACEHCT01", 8) != 0 || !f.get(h, sizeof(h)) || !f.get(sf, 8)) return -2;
  if (h[0] != n_polys || h[1] != c.N || h[2] == 0 || h[2] > c.L || (h[3] != 0 && h[3] != c.K)) return -3;
  *slots = h[5];
  *sf_degree = h[6];
  for (u32 i = 0; i < n_polys; ++i) {
    POLYNOMIAL* p = polys[i];
    poly_free(p);
    poly_alloc(p, c.N, h[2], h[3], false);
    p->_is_ntt = h[4] != 0;
    if (!get_device(f, q_limbs(p), (size_t)h[2] * c.N)) return -2;
    if (h[3] && !get_device(f, p_limbs(p), (size_t)h[3] * c.N)) return -2;
  }
  return 0;
}

size_t key_words(const Context& c) { return (size_t)c.dnum * 2 * (c.L + c.K) * c.N; }

void adopt_key(Context& c, SwitchKeyStore* sk) {  // SWITCH_KEY shells over sk->data
  const u32 T = c.L + c.K;
  const size_t poly_words = (size_t)T * c.N;
  sk->parts.resize(c.dnum);
  for (u32 j = 0; j < c.dnum; ++j) {
    auto set = [&](POLYNOMIAL& p, u64* d) {
      p._ring_degree = c.N;
      p._num_alloc_primes = T;
      p._num_primes = c.L;
      p._num_primes_p = c.K;
      p._is_ntt = true;
      p._data = (int64_t*)d;
    };
    set(sk->parts[j]._pk0, sk->data + ((size_t)j * 2 + 0) * poly_words);
    set(sk->parts[j]._pk1, sk->data + ((size_t)j * 2 + 1) * poly_words);
  }
  sk->key._num_parts = c.dnum;
  sk->key._parts = sk->parts.data();
}
}  // namespace

int save_keys(const char* path) {
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  RT_ASSERT(g_primary != nullptr, "save_keys: no prepared context");
  Context& c = *g_primary;
  File f(path, "wb");
  if (!f.f) return -1;
  sync();
  const u32 T = c.L + c.K;
  const u32 h[8] = {1, c.N, c.L, c.K, c.dnum, (u32)c.rot2auto.size(), (u32)c.auto_keys.size(), 0};
  bool ok = f.put("ACEHKEY1", 8) && f.put(h, sizeof(h)) && f.put(c.primes.data(), (size_t)T * 8);
  ok = ok && put_device(f, c.sk_ntt, (size_t)T * c.N) && put_device(f, c.pk0, (size_t)c.L * c.N) && put_device(f, c.pk1, (size_t)c.L * c.N);
  ok = ok && put_device(f, c.relin.data, key_words(c));
  for (auto& kv : c.rot2auto) {
    const int32_t rot = kv.first;
    const u32 k = kv.second;
    ok = ok && f.put(&rot, 4) && f.put(&k, 4);
  }
  for (auto& kv : c.auto_keys) {
    const u32 e[2] = {kv.first, 0};
    ok = ok && f.put(e, 8) && put_device(f, kv.second->data, key_words(c));
  }
  return ok ? 0 : -1;
}

// Replaces (or provides) the key set of the prepared context.  Returns 0, -1 cannot open, -2 truncated / bad magic,
// -3 the file was written for other parameters.
int load_keys(const char* path) {
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  RT_ASSERT(g_primary != nullptr && g_ctx == g_primary, "load_keys: call from the thread that prepared the context");
  Context& c = *g_primary;
  File f(path, "rb");
  if (!f.f) return -1;
  char magic[8];
  u32 h[8];
  if (!f.get(magic, 8) || memcmp(magic, "ACEHKEY1", 8) != 0 || !f.get(h, sizeof(h))) return -2;
  const u32 T = c.L + c.K;
  if (h[0] != 1 || h[1] != c.N || h[2] != c.L || h[3] != c.K || h[4] != c.dnum) return -3;
  std::vector<u64> primes(T);
  if (!f.get(primes.data(), (size_t)T * 8)) return -2;
  if (primes != c.primes) return -3;
  sync();
  // drop what is there
  for (auto& kv : c.auto_keys) free_switch_key(kv.second);
  c.auto_keys.clear();
  c.rot2auto.clear();
  if (c.relin.data) dfree(c.relin.data);
  if (c.sk_ntt) dfree(c.sk_ntt);
  if (c.pk0) dfree(c.pk0);
  if (c.pk1) dfree(c.pk1);
  c.sk_coef.clear();
  c.sk_ntt = shared_alloc((size_t)T * c.N, false);
  c.pk0 = shared_alloc((size_t)c.L * c.N, false);
  c.pk1 = shared_alloc((size_t)c.L * c.N, false);
  c.relin.data = shared_alloc(key_words(c), false);
  bool ok = get_device(f, c.sk_ntt, (size_t)T * c.N) && get_device(f, c.pk0, (size_t)c.L * c.N) && get_device(f, c.pk1, (size_t)c.L * c.N) &&
            get_device(f, c.relin.data, key_words(c));
  if (!ok) return -2;
  adopt_key(c, &c.relin);
  for (u32 i = 0; i < h[5]; ++i) {
    int32_t rot;
    u32 k;
    if (!f.get(&rot, 4) || !f.get(&k, 4)) return -2;
    c.rot2auto[rot] = k;
  }
  for (u32 i = 0; i < h[6]; ++i) {
    u32 e[2];
    if (!f.get(e, 8)) return -2;
    auto* sk = new SwitchKeyStore();
    sk->data = shared_alloc(key_words(c), false);
    if (!get_device(f, sk->data, key_words(c))) {
      free_switch_key(sk);
      return -2;
    }
    adopt_key(c, sk);
    c.auto_keys[e[0]] = sk;
  }
  sync();
  c.keys_loaded = true;
  return 0;
}

}  // namespace rt

using namespace rt;

extern "C" {

int Acehip_rt_save_ciph(const char* path, CIPHER c) {
  POLYNOMIAL* p[2] = {&c->_c0_poly, &c->_c1_poly};
  return save_polys(path, p, 2, c->_slots, c->_scaling_factor, c->_sf_degree);
}
int Acehip_rt_save_ciph3(const char* path, CIPHER3 c) {
  POLYNOMIAL* p[3] = {&c->_c0_poly, &c->_c1_poly, &c->_c2_poly};
  return save_polys(path, p, 3, c->_slots, c->_scaling_factor, c->_sf_degree);
}
int Acehip_rt_save_plain(const char* path, PLAIN c) {
  POLYNOMIAL* p[1] = {&c->_poly};
  return save_polys(path, p, 1, c->_slots, c->_scaling_factor, c->_sf_degree);
}
int Acehip_rt_load_ciph(CIPHER c, const char* path) {
  POLYNOMIAL* p[2] = {&c->_c0_poly, &c->_c1_poly};
  return load_polys(path, p, 2, &c->_slots, &c->_scaling_factor, &c->_sf_degree);
}
int Acehip_rt_load_ciph3(CIPHER3 c, const char* path) {
  POLYNOMIAL* p[3] = {&c->_c0_poly, &c->_c1_poly, &c->_c2_poly};
  return load_polys(path, p, 3, &c->_slots, &c->_scaling_factor, &c->_sf_degree);
}
int Acehip_rt_load_plain(PLAIN c, const char* path) {
  POLYNOMIAL* p[1] = {&c->_poly};
  return load_polys(path, p, 1, &c->_slots, &c->_scaling_factor, &c->_sf_degree);
}
int Acehip_rt_save_keys(const char* path) { return save_keys(path); }
int Acehip_rt_load_keys(const char* path) { return load_keys(path); }

}  // extern "C"
