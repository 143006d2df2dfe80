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
//   "ACEHKEY1": u32 version(1), N, L, K, dnum, n_rot, n_auto, flags; u64 primes[L+K];
//               [sk (NTT domain) [L+K][N] unless flags & 1]; pk0 [L][N]; pk1 [L][N]; relin key [dnum][2][L+K][N] (b_j then a_j);
//               n_rot x {i32 rotation, u32 automorphism index}; n_auto x {u32 automorphism index, u32 0, key as above}
//               flags bit 0: EVALUATION key set -- no secret key inside (what a client hands to the server that runs
//               Main_graph: Acehip_rt_save_eval_keys).  Key files are created with mode 0600.
// Both formats are read defensively (sizes against the header before anything is replaced, header fields and residues
// range-checked): a truncated or corrupt file leaves the context as it was.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstring>

#include "rt_internal.hpp"

namespace rt {

namespace {
struct File {
  FILE* f = nullptr;
  const char* path;
  std::string tmp;       // key files are written under a temporary name and renamed into place by commit()
  bool discard = false;  // limb-sharded runs: every rank takes part in the gathers, rank 0 alone writes
  File(const char* p, const char* mode, bool secret = false, bool discard_ = false) : path(p), discard(discard_) {
    if (discard) {
      f = fopen("/dev/null", "wb");
    } else if (secret) {
      // key material: a new file of mode 0600 (never through a symbolic link, never an existing file), complete before it gets its
      // name -- an interrupted save leaves the old file (or none), not a truncated one
      tmp = std::string(p) + ".tmp." + std::to_string((long)getpid());
      unlink(tmp.c_str());
      const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
      if (fd >= 0) {
        (void)fchmod(fd, 0600);
        f = fdopen(fd, mode);
        if (!f) close(fd);
      }
    } else {
      f = fopen(p, mode);
    }
  }
  long size() const {
    struct stat st;
    return f && fstat(fileno(f), &st) == 0 ? (long)st.st_size : -1;
  }
  bool commit() {  // flush, reach the disk, take the final name
    if (!f) return false;
    bool ok = fflush(f) == 0 && (discard || fsync(fileno(f)) == 0);
    ok = fclose(f) == 0 && ok;
    f = nullptr;
    if (!tmp.empty()) {
      if (ok) ok = rename(tmp.c_str(), path) == 0;
      if (!ok) unlink(tmp.c_str());
      tmp.clear();
    }
    return ok;
  }
  ~File() {
    if (f) fclose(f);
    if (!tmp.empty()) unlink(tmp.c_str());  // never committed: nothing half-written stays behind
  }
  bool put(const void* p, size_t bytes) { return fwrite(p, 1, bytes, f) == bytes; }
  bool get(void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }
};
constexpr size_t kChunkWords = 1u << 20;  // 8 MiB staging
thread_local std::vector<u64> g_stage;

// `n_limbs` limbs at d; limb i has prime gi0 + i of a chain extended at `level` (limb-sharded execution: every limb is
// fetched from its owner first)
bool put_device(File& f, const u64* d, u32 n_limbs, u32 level, u32 pos0) {
  Context& c = ctx();
  const size_t words = (size_t)n_limbs * c.N;
  if (c.shard_world > 1) {
    // limb by limb: a rank's switch keys back only the limbs it owns, every other limb position is the same scratch limb
    // (acehip_malloc_limbs) -- a gathered limb must be written out before the next one arrives
    g_stage.resize(c.N);
    for (u32 l = 0; l < n_limbs; ++l) {
      HIPCHK(acehip_shard_gather(c.hip, const_cast<u64*>(d) - (size_t)pos0 * c.N, level, pos0 + l, 1, nullptr));
      HIPCHK(acehip_download(c.hip, g_stage.data(), d + (size_t)l * c.N, (size_t)c.N * 8, nullptr));
      if (!f.put(g_stage.data(), (size_t)c.N * 8)) return false;
    }
    return true;
  }
  g_stage.resize(std::min(words, kChunkWords));
  for (size_t off = 0; off < words; off += kChunkWords) {
    const size_t n = std::min(kChunkWords, words - off);
    HIPCHK(acehip_download(c.hip, g_stage.data(), d + off, n * 8, nullptr));
    if (!f.put(g_stage.data(), n * 8)) return false;
  }
  return true;
}
// the same limbs read back; every word must be a residue of its limb's prime.  false: short read or a word out of range
bool get_device(File& f, u64* d, u32 n_limbs, u32 level, u32 pos0) {
  Context& c = ctx();
  const size_t N = c.N;
  g_stage.resize(std::min((size_t)n_limbs * N, kChunkWords));
  for (u32 l = 0; l < n_limbs; ++l) {
    const u32 pos = pos0 + l;
    const u64 q = c.primes[pos < level ? pos : c.L + (pos - level)];
    for (size_t off = 0; off < N; off += kChunkWords) {
      const size_t n = std::min(kChunkWords, N - off);
      if (!f.get(g_stage.data(), n * 8)) return false;
      for (size_t i = 0; i < n; ++i)
        if (g_stage[i] >= q) return false;
      HIPCHK(acehip_upload(c.hip, d + (size_t)l * N + off, g_stage.data(), n * 8, nullptr));
    }
  }
  return true;
}

int save_polys(const char* path, POLYNOMIAL* const* polys, u32 n_polys, u32 slots, double sf, u32 sf_degree) {
  File f(path, "wb");
  if (!f.f) return -1;
  sync();
  ImageScope one_image(selected_image());  // (image batches: the selected image's copy)
  const POLYNOMIAL& p0 = *polys[0];
  const u32 h[8] = {n_polys, p0._ring_degree, (u32)p0._num_primes, (u32)p0._num_primes_p, p0._is_ntt ? 1u : 0u, slots, sf_degree, 0};
  bool ok = f.put("ACEHCT01", 8) && f.put(h, sizeof(h)) && f.put(&sf, 8);
  for (u32 i = 0; ok && i < n_polys; ++i) {
    POLYNOMIAL* p = polys[i];
    RT_ASSERT(p->_num_primes == p0._num_primes && p->_num_primes_p == p0._num_primes_p, "save: polynomials of different shape");
    ok = put_device(f, q_limbs(p), (u32)p->_num_primes, (u32)p->_num_primes, 0);
    if (ok && p->_num_primes_p) ok = put_device(f, p_limbs(p), (u32)p->_num_primes_p, 0, 0);
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
  // the header drives decode / rotate index arithmetic and the kernels' value ranges: nothing out of contract gets in
  double sfv;
  memcpy(&sfv, sf, 8);
  if (h[5] == 0 || h[5] > c.N / 2 || (h[5] & (h[5] - 1)) != 0 || h[6] == 0 || h[4] > 1 || !std::isfinite(sfv) || !(sfv > 0)) return -2;
  if (f.size() != (long)(8 + sizeof(h) + 8 + (size_t)n_polys * (h[2] + h[3]) * c.N * 8)) return -2;
  ImageScope one_image(selected_image());
  // into fresh blocks first: the caller's polynomials change only when the whole file was good
  std::vector<POLYNOMIAL> fresh(n_polys);
  bool ok = true;
  for (u32 i = 0; i < n_polys; ++i) {
    memset(&fresh[i], 0, sizeof(POLYNOMIAL));
    poly_alloc(&fresh[i], c.N, h[2], h[3], false);
    fresh[i]._is_ntt = h[4] != 0;
    ok = ok && get_device(f, q_limbs(&fresh[i]), h[2], h[2], 0);
    if (ok && h[3]) ok = get_device(f, p_limbs(&fresh[i]), h[3], 0, 0);
  }
  if (!ok) {
    for (auto& p : fresh) poly_free(&p);
    return -2;
  }
  *slots = h[5];
  *sf_degree = h[6];
  for (u32 i = 0; i < n_polys; ++i) {
    poly_free(polys[i]);
    *polys[i] = fresh[i];
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

static int save_keys_impl(const char* path, bool with_secret) {
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  RT_ASSERT(g_primary != nullptr, "save_keys: no prepared context");
  Context& c = *g_primary;
  if (with_secret && c.sk_ntt == nullptr) return -1;  // an evaluation-only context has no secret to write
  File f(path, "wb", true, /*discard=*/c.shard_world > 1 && !c.shard_sim && c.shard_rank != 0);
  if (!f.f) return -1;
  sync();
  UniformScope keys_are_shared;
  const u32 T = c.L + c.K;
  const u32 h[8] = {1, c.N, c.L, c.K, c.dnum, (u32)c.rot2auto.size(), (u32)c.auto_keys.size(), with_secret ? 0u : 1u};
  bool ok = f.put("ACEHKEY1", 8) && f.put(h, sizeof(h)) && f.put(c.primes.data(), (size_t)T * 8);
  auto put_key = [&](const u64* data) {  // [dnum][2] polynomials of L + K limbs
    bool good = true;
    for (u32 j = 0; good && j < 2 * c.dnum; ++j) good = put_device(f, data + (size_t)j * T * c.N, T, c.L, 0);
    return good;
  };
  if (with_secret) ok = ok && put_device(f, c.sk_ntt, T, c.L, 0);
  ok = ok && put_device(f, c.pk0, c.L, c.L, 0) && put_device(f, c.pk1, c.L, c.L, 0);
  ok = ok && put_key(c.relin.data);
  for (auto& kv : c.rot2auto) {
    const int32_t rot = kv.first;
    const u32 k = kv.second;
    ok = ok && f.put(&rot, 4) && f.put(&k, 4);
  }
  for (auto& kv : c.auto_keys) {
    const u32 e[2] = {kv.first, 0};
    ok = ok && f.put(e, 8) && put_key(kv.second->data);
  }
  return ok && f.commit() ? 0 : -1;
}
int save_keys(const char* path) { return save_keys_impl(path, true); }
int save_eval_keys(const char* path) { return save_keys_impl(path, false); }

// Replaces (or provides) the key set of the prepared context.  Returns 0, -1 cannot open, -2 truncated / bad magic / a word
// that is no residue, -3 the file was written for other parameters.  The context changes only when the whole file was good.
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
  if (h[0] != 1 || h[1] != c.N || h[2] != c.L || h[3] != c.K || h[4] != c.dnum || h[7] > 1) return -3;
  const bool has_sk = (h[7] & 1) == 0;
  {  // the size the header promises, before anything is touched
    const size_t poly = (size_t)c.N * 8;
    const size_t want = 8 + sizeof(h) + (size_t)T * 8 + (has_sk ? T * poly : 0) + 2 * c.L * poly + key_words(c) * 8 + (size_t)h[5] * 8 +
                        (size_t)h[6] * (8 + key_words(c) * 8);
    if (h[5] > (1u << 20) || h[6] > (1u << 20) || f.size() != (long)want) return -2;
  }
  std::vector<u64> primes(T);
  if (!f.get(primes.data(), (size_t)T * 8)) return -2;
  if (primes != c.primes) return -3;
  sync();
  UniformScope keys_are_shared;
  // everything into fresh buffers; the context's keys are exchanged for them at the very end
  struct Fresh {
    u64 *sk = nullptr, *pk0 = nullptr, *pk1 = nullptr, *relin = nullptr;
    std::map<int32_t, u32> rot2auto;
    std::map<u32, SwitchKeyStore*> auto_keys;
  } n;
  auto get_key = [&](u64* data) {
    bool good = true;
    for (u32 j = 0; good && j < 2 * c.dnum; ++j) good = get_device(f, data + (size_t)j * T * c.N, T, c.L, 0);
    return good;
  };
  auto drop = [&](Fresh& x) {
    for (u64* p : {x.sk, x.pk0, x.pk1, x.relin})
      if (p) dfree(p);
    for (auto& kv : x.auto_keys) free_switch_key(kv.second);
  };
  bool ok = true;
  if (has_sk) {
    n.sk = shared_alloc((size_t)T * c.N, false);
    ok = get_device(f, n.sk, T, c.L, 0);
  }
  n.pk0 = shared_alloc((size_t)c.L * c.N, false);
  n.pk1 = shared_alloc((size_t)c.L * c.N, false);
  n.relin = shared_alloc_key((size_t)c.dnum * 2);
  ok = ok && get_device(f, n.pk0, c.L, c.L, 0) && get_device(f, n.pk1, c.L, c.L, 0) && get_key(n.relin);
  for (u32 i = 0; ok && i < h[5]; ++i) {
    int32_t rot;
    u32 k;
    ok = f.get(&rot, 4) && f.get(&k, 4) && (k & 1) == 1 && k < 2 * c.N;
    if (ok) n.rot2auto[rot] = k;
  }
  for (u32 i = 0; ok && i < h[6]; ++i) {
    u32 e[2];
    ok = f.get(e, 8) && (e[0] & 1) == 1 && e[0] < 2 * c.N && n.auto_keys.find(e[0]) == n.auto_keys.end();
    if (!ok) break;
    auto* sk = new SwitchKeyStore();
    sk->data = shared_alloc_key((size_t)c.dnum * 2);
    n.auto_keys[e[0]] = sk;
    ok = get_key(sk->data);
    if (ok) adopt_key(c, sk);
  }
  if (!ok) {
    drop(n);
    return -2;
  }
  sync();
  // the exchange: from here on nothing can fail
  Fresh old;
  old.sk = c.sk_ntt;
  old.pk0 = c.pk0;
  old.pk1 = c.pk1;
  old.relin = c.relin.data;
  old.auto_keys.swap(c.auto_keys);
  c.rot2auto.swap(n.rot2auto);
  c.auto_keys.swap(n.auto_keys);
  c.sk_ntt = n.sk;
  c.pk0 = n.pk0;
  c.pk1 = n.pk1;
  c.relin.data = n.relin;
  c.sk_coef.clear();
  adopt_key(c, &c.relin);
  drop(old);
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
// the key set WITHOUT the secret key: what the party that only evaluates (Main_graph) needs; such a context cannot decrypt
int Acehip_rt_save_eval_keys(const char* path) { return save_eval_keys(path); }
int Acehip_rt_load_keys(const char* path) { return load_keys(path); }

}  // extern "C"
