// rt_context.cpp -- global context, key generation, key accessors
// (reference: src/rtlib/context.c:29-138, src/util/ckks_key_generator.c:69-335,
//  src/util/random_sample.c:78-150, include/rtlib/key_gen.h:28-75).
#include <cmath>
#include <cstring>
#include <ctime>

#include "rt_internal.hpp"

#include <cctype>
#include <cerrno>
#include <fcntl.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <unistd.h>
#include <execinfo.h>
#include <signal.h>

// Programs generated for the CKKS-level provider interface (rt_seal examples) do not define this callback: it is an
// optional (weak) reference here, a missing definition means "no weight data file".
extern "C" RT_DATA_INFO* Get_rt_data_info() __attribute__((weak));
static RT_DATA_INFO* rt_data_info() { return Get_rt_data_info ? Get_rt_data_info() : nullptr; }

namespace rt {

// Randomness layout (rt_rng.hpp).  Every KEY draws from a generator of its own, named by the key's identity (secret 1, public 2,
// relinearisation 3, automorphism key k: 2^34 + k) -- a key is the same whichever thread makes it and in whatever order keys are
// asked for.  ENCRYPTION draws from the calling thread's stream (Context::rng).  Normal operation: all of these are ChaCha20 streams
// of ONE 256-bit master key from getrandom(2) (Context::master_key; the reference: BLAKE2Xb seeded from /dev/urandom, prng.c:33-69).
// Test mode (ACEHIP_SEED): std::mt19937_64 generators derived from the 64-bit seed exactly as in rounds 1-4, so that a fixed seed
// names one key set (tests/c/gen_parity_ref.c derives the same set on the CPU and injects it into the reference);
// Acehip_rt_seed_encryptor puts a thread's encryption stream into test mode as well.
void Rng::abort_unseeded() { RT_ASSERT(false, "random generator used before Prepare_context seeded it"); }
static u64 splitmix(u64 z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
Rng key_rng(u64 tag) {
  RT_ASSERT(g_primary != nullptr, "no prepared context");
  Rng r;
  if (g_primary->drbg) r.key(g_primary->master_key, 'K', tag);
  else r.seed(splitmix(g_primary->key_seed ^ splitmix(tag)));  // test mode
  return r;
}
void sample_uniform_dev(u64* d, u32 level, u32 pos0, u32 n, Rng& rng) {
  Context& c = ctx();
  if (rng.test_mode()) {  // the 64-bit counter-based sampler the fixtures were made with
    HIPCHK(acehip_sample_uniform(c.hip, d, level, pos0, n, rng(), nullptr));
  } else {
    uint32_t k[8];
    rng.draw_key(k);
    HIPCHK(acehip_sample_uniform_keyed(c.hip, d, level, pos0, n, k, nullptr));
  }
}
// 256 bits from the operating system; no weaker fallback
static void os_random(void* out, size_t bytes) {
  unsigned char* p = (unsigned char*)out;
  size_t got = 0;
  while (got < bytes) {
    const ssize_t r = getrandom(p + got, bytes - got, 0);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    got += (size_t)r;
  }
  if (got < bytes) {  // (a kernel without the system call: the device it is a front end of)
    FILE* f = fopen("/dev/urandom", "rb");
    RT_ASSERT(f != nullptr && fread(p + got, 1, bytes - got, f) == bytes - got, "no entropy source: getrandom() and /dev/urandom both failed");
    fclose(f);
  }
}

// Sample_triangle random_sample.c:78-97: -1 w.p. 1/4, +1 w.p. 1/4, 0 w.p. 1/2
void sample_triangle(std::vector<int64_t>& v, Rng& rng) {
  for (auto& x : v) {
    const u64 r = rng() & 3;
    x = r == 0 ? -1 : (r == 1 ? 1 : 0);
  }
}

// Sample_ternary random_sample.c:99-150: exactly `hamming_weight` non-zeros, signs roughly balanced
void sample_ternary(std::vector<int64_t>& v, size_t hw, Rng& rng) {
  const size_t n = v.size();
  if (hw == 0) {
    for (auto& x : v) x = (int64_t)(rng() % 3) - 1;
    return;
  }
  if (hw > n) hw = n;
  int64_t ones = -1000000;
  while (ones < (int64_t)hw / 2 - 1 || ones > (int64_t)hw / 2 + 1) {
    ones = 0;
    std::fill(v.begin(), v.end(), 0);
    size_t weight = 0;
    while (weight < hw) {
      const size_t idx = rng() % n;
      if (v[idx] == 0) {
        if (rng() & 1) {
          v[idx] = 1;
          ++ones;
        } else {
          v[idx] = -1;
        }
        ++weight;
      }
    }
  }
}

static u64 p_mod(u64 q) {  // P mod q
  Context& c = ctx();
  unsigned __int128 r = 1;
  for (u32 j = 0; j < c.K; ++j) r = (r * (c.primes[c.L + j] % q)) % q;
  return (u64)r;
}

// Generate_switching_key ckks_key_generator.c:127-200:  b_j = -a_j*old + e_j + P*new [digit j limbs]
SwitchKeyStore* make_switch_key(const u64* new_key_ntt, const u64* old_key_ntt, u64 tag) {
  Context& c = ctx();
  Rng rng = key_rng(tag);
  const u32 T = c.L + c.K;
  const size_t N = c.N, poly_words = (size_t)T * N;
  auto* sk = new SwitchKeyStore();
  sk->data = shared_alloc_key((size_t)c.dnum * 2);  // outlives the generating thread; a sharded rank backs only the limbs it owns
  sk->parts.resize(c.dnum);
  POLYNOMIAL e{};
  poly_alloc(&e, c.N, c.L, c.K);
  u64* pm = dalloc(poly_words, false);
  std::vector<int64_t> tri(N);
  std::vector<u64> scal(T);
  for (u32 j = 0; j < c.dnum; ++j) {
    u64* b = sk->data + ((size_t)j * 2 + 0) * poly_words;
    u64* a = sk->data + ((size_t)j * 2 + 1) * poly_words;
    sample_uniform_dev(a, c.L, 0, T, rng);                        // a_j (NTT domain)
    q_ew(ACEHIP_HW_MUL, b, a, old_key_ntt, c.L, 0, T);            // a_j * old
    for (u32 i = 0; i < T; ++i) scal[i] = (i < c.L && i / c.alpha == j) ? p_mod(c.primes[i]) : 0;
    q_scalars(ACEHIP_HW_MULC, pm, new_key_ntt, scal.data(), c.L, 0, T);  // P*new on digit j
    sample_triangle(tri, rng);
    poly_from_small(&e, tri);
    poly_ntt(&e, false);
    q_ew(ACEHIP_HW_ADD, pm, pm, (u64*)e._data, c.L, 0, T);        // e + P*new
    q_ew(ACEHIP_HW_SUB, b, pm, b, c.L, 0, T);                     // b = e + P*new - a*old
    auto set = [&](POLYNOMIAL& p, u64* d) {
      p._ring_degree = c.N;
      p._num_alloc_primes = T;
      p._num_primes = c.L;
      p._num_primes_p = c.K;
      p._is_ntt = true;
      p._data = (int64_t*)d;
    };
    set(sk->parts[j]._pk0, b);
    set(sk->parts[j]._pk1, a);
  }
  poly_free(&e);
  dfree(pm);
  sk->key._num_parts = c.dnum;
  sk->key._parts = sk->parts.data();
  return sk;
}

void free_switch_key(SwitchKeyStore* k) {
  if (!k) return;
  dfree(k->data);  // shared allocation: dfree recognises it
  delete k;
}

// Generate_rot_key (fast variant) :238-266: key from sigma_{k^-1}(s) ... applied before the automorphism
SwitchKeyStore* ensure_auto_key(u32 auto_idx) {
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(shared_mu());  // keys are generated once, by whichever thread asks first
  Context& prim = *g_primary;
  auto it = prim.auto_keys.find(auto_idx);
  if (it != prim.auto_keys.end()) return it->second;
  RT_ASSERT(!(prim.keys_loaded && prim.keys_strict), "automorphism key %u is not in the loaded key file (ACEHIP_KEYS_STRICT)", auto_idx);
  RT_ASSERT(c.sk_ntt != nullptr, "automorphism key %u is not in the loaded evaluation-only key set (it cannot be made without the secret key)", auto_idx);
  const u32 T = c.L + c.K;
  UniformScope shared_by_all_images;  // (one key, whatever the batch: its blocks lie outside the arena)
  // k^-1 mod 2N (k odd): k^(N-1)
  u64 inv = 1, base = auto_idx, e = c.N - 1, m = 2ull * c.N;
  for (; e; e >>= 1) {
    if (e & 1) inv = inv * base % m;
    base = base * base % m;
  }
  u64* old_key = dalloc((size_t)T * c.N, false);
  const uint32_t* perm = acehip_auto_order(c.hip, (u32)inv);
  RT_ASSERT(perm, "automorphism table: %s", acehip_last_error());
  q_rotate(old_key, c.sk_ntt, perm, c.L, 0, T);
  SwitchKeyStore* k = make_switch_key(c.sk_ntt, old_key, KEY_TAG_AUTO + auto_idx);
  dfree(old_key);
  sync();  // complete in memory before another thread's stream may read it
  prim.auto_keys[auto_idx] = k;
  return k;
}

u32 ensure_rot_key(int32_t rotation) {
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  Context& prim = *g_primary;
  auto it = prim.rot2auto.find(rotation);
  if (it != prim.rot2auto.end()) return it->second;
  const u32 k = acehip_auto_index(c.hip, rotation);
  ensure_auto_key(k);
  prim.rot2auto[rotation] = k;
  return k;
}

// ACEHIP_SHARD=1: the processes of a launcher (torchrun, mpirun: RANK / WORLD_SIZE / LOCAL_RANK in the environment) are the ranks
// of ONE limb-sharded computation: each owns the limbs gi % WORLD_SIZE == RANK on its own GPU.  Rank 0 creates the RCCL id and
// publishes it in a file only this job's ranks look for; a caller that has its own channel passes the id with
// Acehip_rt_shard_connect instead.  All ranks must derive the same keys: the same ACEHIP_SEED (test mode), or -- without one --
// the master key rank 0 draws from the OS and puts into the record.
//
// The rendezvous file.  Name (unless ACEHIP_SHARD_ID_FILE names one): <dir>/acehip_rccl_<uid>_<job>.id with <dir> = $XDG_RUNTIME_DIR,
// $TMPDIR or /tmp and <job> = what the launcher gave every rank: its run id (TORCHELASTIC_RUN_ID) when there is one, MASTER_ADDR,
// MASTER_PORT and WORLD_SIZE (the parent's pid only when the launcher set none of these: ranks started through per-rank wrapper
// scripts have different parents) -- two jobs on one node look for different files, and a retry on the same port is told apart by
// the record's time stamp.  Content: magic, a token derived from the same job identity, rank 0's clock, the 128-byte id, the 256-bit master key of the
// job's randomness; a reader accepts only a regular file of its own user with the right token that is younger than five minutes and
// was not written long before the reader itself started.  Rank 0 creates it with O_CREAT | O_EXCL | O_NOFOLLOW, mode
// 0600, under a temporary name and renames it into place; it is removed once every rank has joined.  The join itself is bounded
// (acehip_ctx_shard_rccl, ACEHIP_RCCL_INIT_TIMEOUT_S): a rank whose peers never arrive exits with an error instead of hanging.
static void shard_connect(u32 rank, u32 world, const unsigned char* id, const u32* master_key) {
  Context& c = ctx();
  const int rc = acehip_ctx_shard_rccl(c.hip, rank, world, id, 128);
  RT_ASSERT(rc >= 0, "acehip_ctx_shard_rccl: %s", acehip_last_error());
  c.shard_world = world;
  c.shard_rank = rank;
  if (getenv("ACEHIP_SEED") == nullptr) {  // the same randomness on every rank: rank 0's master key, out of the 0600 record
    memcpy(c.master_key, master_key, sizeof c.master_key);
    c.drbg = true;
    c.rng.key(c.master_key, 'E', 0);
  }
}
static std::string shard_job_identity() {
  // what the LAUNCHER gave every rank: its run id, its rendezvous address and port, the world size.  (The parent's pid is the
  // same for all ranks only when no wrapper script sits between the launcher and the rank: it is used only when the launcher
  // identified the job by nothing else.)
  std::string job;
  const char* run = getenv("TORCHELASTIC_RUN_ID");
  const char *addr = getenv("MASTER_ADDR"), *port = getenv("MASTER_PORT"), *world = getenv("WORLD_SIZE");
  if (run && *run && strcmp(run, "none") != 0) job += std::string(run) + "_";
  // (a restarted worker group of the same elastic run is another launch: its ranks must not pick up the previous group's record)
  if (const char* rc = getenv("TORCHELASTIC_RESTART_COUNT")) if (*rc && strcmp(rc, "0") != 0) job += std::string("r") + rc + "_";
  if (port && *port) job += std::string(addr && *addr ? addr : "localhost") + "_" + port + "_w" + (world ? world : "1");
  else if (job.empty()) job = "ppid" + std::to_string((long)getppid());
  for (char& ch : job)
    if (!isalnum((unsigned char)ch) && ch != '_' && ch != '-') ch = '_';
  return job;
}
static u64 shard_job_token(const std::string& job) {
  u64 h = 0xCBF29CE484222325ull ^ (u64)geteuid();
  for (unsigned char ch : job) h = (h ^ ch) * 0x100000001B3ull;
  return h;
}
void shard_connect_if_asked() {
  const char* on = getenv("ACEHIP_SHARD");
  if (on == nullptr || atoi(on) == 0) return;
  const char* r = getenv("RANK");
  const char* w = getenv("WORLD_SIZE");
  RT_ASSERT(r != nullptr && w != nullptr, "ACEHIP_SHARD=1 needs RANK and WORLD_SIZE (torchrun / mpirun)");
  const u32 rank = (u32)atoi(r), world = (u32)atoi(w);
  RT_ASSERT(world >= 1 && rank < world, "bad RANK / WORLD_SIZE");
  const std::string job = shard_job_identity();
  std::string path;
  u64 token;
  if (const char* f = getenv("ACEHIP_SHARD_ID_FILE")) {  // the caller names the file (and with it the job): the token follows the name
    path = f;
    token = shard_job_token(path);
  } else {
    const char* dir = getenv("XDG_RUNTIME_DIR");
    if (!dir || !*dir) dir = getenv("TMPDIR");
    if (!dir || !*dir) dir = "/tmp";
    path = std::string(dir) + "/acehip_rccl_" + std::to_string((long)geteuid()) + "_" + job + ".id";
    token = shard_job_token(job);
  }
  struct Record {
    char magic[8];
    u64 token;
    int64_t created_s;           // rank 0's clock when it wrote the record: readers refuse one that predates their own start
    unsigned char id[128];
    u32 master_key[8];           // the randomness every rank derives its (identical) keys from; the file is 0600 and short-lived
  } rec;
  memset(&rec, 0, sizeof rec);
  unsigned char id[128];
  // when THIS process began (/proc/self/stat field 22 + the boot time of /proc/stat; the clock at first use -- Prepare_context -- where
  // those are unreadable).  The ranks of
  // one launch start within seconds of each other, so a record written more than a few seconds before a reader BEGAN belongs to an
  // earlier launch on the same address and port that died between publishing and joining: refused, the reader keeps waiting for its
  // own rank 0 (which unlinks the stale file first thing).
  static const time_t process_start = [] {
    const time_t now = time(nullptr);
    long long btime = -1, ticks = -1;
    if (FILE* f = fopen("/proc/stat", "r")) {
      char line[256];
      while (fgets(line, sizeof line, f))
        if (sscanf(line, "btime %lld", &btime) == 1) break;
      fclose(f);
    }
    if (FILE* f = fopen("/proc/self/stat", "r")) {
      char buf[2048];
      const size_t n = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      buf[n] = 0;
      if (const char* p = strrchr(buf, ')')) {  // (the command name may hold spaces: fields are counted behind its closing bracket)
        int field = 2;
        for (++p; *p && field < 22; ++p)
          if (*p == ' ' && p[1] != ' ') ++field;
        if (field == 22) ticks = atoll(p);
      }
    }
    const long hz = sysconf(_SC_CLK_TCK);
    if (btime <= 0 || ticks < 0 || hz <= 0) return now;
    const time_t t = (time_t)(btime + ticks / hz);
    return t > now || now - t > 7 * 24 * 3600 ? now : t;
  }();
  constexpr int64_t kRecordSlackS = 10;
  if (rank == 0) {
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    unlink(path.c_str());  // (left behind by a run that died between publishing and joining)
    unlink(tmp.c_str());
    const int n = acehip_rccl_unique_id(id, sizeof id);
    RT_ASSERT(n == 128, "acehip_rccl_unique_id: %s", acehip_last_error());
    memcpy(rec.magic, "ACEHRCC2", 8);
    rec.token = token;
    rec.created_s = (int64_t)time(nullptr);
    memcpy(rec.id, id, 128);
    os_random(rec.master_key, sizeof rec.master_key);
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    RT_ASSERT(fd >= 0, "cannot create %s: %s", tmp.c_str(), strerror(errno));
    const bool ok = write(fd, &rec, sizeof rec) == (ssize_t)sizeof rec && fsync(fd) == 0;
    close(fd);
    RT_ASSERT(ok, "cannot write %s", tmp.c_str());
    RT_ASSERT(rename(tmp.c_str(), path.c_str()) == 0, "cannot publish %s", path.c_str());
  } else {
    const char* t = getenv("ACEHIP_RCCL_INIT_TIMEOUT_S");
    const int limit_s = t && atoi(t) > 0 ? atoi(t) : 180;
    bool got = false;
    for (int tries = 0; tries < limit_s * 100 && !got; ++tries) {
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
      if (fd >= 0) {  // (an older file, another user's, or another job's token: not this run's -- keep waiting for rank 0)
        struct stat st;
        // (a record written more than kRecordSlackS before THIS process started belongs to an earlier launch that died before joining)
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && time(nullptr) - st.st_mtime < 300 &&
            read(fd, &rec, sizeof rec) == (ssize_t)sizeof rec && memcmp(rec.magic, "ACEHRCC2", 8) == 0 && rec.token == token &&
            rec.created_s >= (int64_t)process_start - kRecordSlackS) {
          memcpy(id, rec.id, 128);
          got = true;
        }
        close(fd);
      }
      if (!got) {
        struct timespec ts = {0, 10 * 1000 * 1000};
        nanosleep(&ts, nullptr);
      }
    }
    RT_ASSERT(got, "rank %u: no RCCL id for this job in %s within %d s", rank, path.c_str(), limit_s);
  }
  shard_connect(rank, world, id, rec.master_key);
  memset(&rec, 0, sizeof rec);
  if (rank == 0) {  // (every rank has joined: ncclCommInitRank returns only then)  The record held the job's master key: zeros before it goes
    const int fd = open(path.c_str(), O_WRONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd >= 0) {
      const bool wiped = write(fd, &rec, sizeof rec) == (ssize_t)sizeof rec && fsync(fd) == 0;  // rec is all zeros here
      (void)wiped;
      close(fd);
    }
    unlink(path.c_str());
  }
}

void generate_keys() {
  Context& c = ctx();
  const u32 T = c.L + c.K;
  const size_t N = c.N;
  // secret key (Generate_secret_key :69-83)
  c.sk_coef.assign(N, 0);
  Rng sk_rng = key_rng(KEY_TAG_SECRET), pk_rng = key_rng(KEY_TAG_PUBLIC);
  sample_ternary(c.sk_coef, c.hamming, sk_rng);
  POLYNOMIAL s{};
  poly_alloc(&s, c.N, c.L, c.K);
  poly_from_small(&s, c.sk_coef);
  poly_ntt(&s, false);
  c.sk_ntt = (u64*)s._data;  // ownership moves to the context
  // public key (Generate_public_key :85-125): pk1 = a, pk0 = -a*s + e
  c.pk0 = dalloc((size_t)c.L * N, false);
  c.pk1 = dalloc((size_t)c.L * N, false);
  sample_uniform_dev(c.pk1, c.L, 0, c.L, pk_rng);
  POLYNOMIAL e{};
  poly_alloc(&e, c.N, c.L, 0);
  std::vector<int64_t> tri(N);
  sample_triangle(tri, pk_rng);
  poly_from_small(&e, tri);
  poly_ntt(&e, false);
  q_ew(ACEHIP_HW_MUL, c.pk0, c.pk1, c.sk_ntt, c.L, 0, c.L);
  q_ew(ACEHIP_HW_SUB, c.pk0, (u64*)e._data, c.pk0, c.L, 0, c.L);
  poly_free(&e);
  // relinearisation key (Generate_relin_key :204-216): new = s^2 (q-limbs; p-limbs stay 0), old = s
  u64* s2 = dalloc((size_t)T * N, true, c.L);
  q_ew(ACEHIP_HW_MUL, s2, c.sk_ntt, c.sk_ntt, c.L, 0, c.L);
  SwitchKeyStore* rk = make_switch_key(s2, c.sk_ntt, KEY_TAG_RELIN);
  dfree(s2);
  c.relin = *rk;
  c.relin.key._parts = c.relin.parts.data();
  delete rk;
  // rotation keys for the compiler-provided index list (Generate_rot_maps :290-335)
  for (size_t i = 0; i < c.prm->_num_rot_idx; ++i) ensure_rot_key(c.prm->_rot_idxs[i]);
}

}  // namespace rt

using namespace rt;

extern "C" {

// A crash inside the runtime (or under it) must not be silent: a program whose stdout is a pipe loses everything it has printed when it
// dies of a signal, and an empty report is all a caller gets.  The first SIGSEGV / SIGBUS / SIGFPE / SIGILL of the process writes the
// signal and a backtrace to stderr (async-signal-safe calls only), flushes nothing else, and then dies of the same signal with the
// default action -- exit status and core behaviour are what they would have been.  ACEHIP_NO_CRASH_REPORT=1 leaves the handlers alone;
// a handler the program installed itself is never replaced.
static void crash_report(int sig) {
  static const char head[] = "\n[acehip] fatal signal -- backtrace of the faulting thread (libFHErt_ant / libacehip frames resolve with addr2line):\n";
  ssize_t w = write(2, head, sizeof head - 1);
  (void)w;
  void* frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
static void install_crash_report() {
  static bool done = false;
  if (done || getenv("ACEHIP_NO_CRASH_REPORT") != nullptr) return;
  done = true;
  void* warm[2];
  (void)backtrace(warm, 2);  // (loads libgcc's unwinder now: not from inside a signal handler)
  for (int sig : {SIGSEGV, SIGBUS, SIGFPE, SIGILL}) {
    struct sigaction old;
    if (sigaction(sig, nullptr, &old) != 0 || old.sa_handler != SIG_DFL || (old.sa_flags & SA_SIGINFO)) continue;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = crash_report;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_RESETHAND | SA_NODEFER;
    (void)sigaction(sig, &sa, nullptr);
  }
}

void Prepare_context() {
  if (g_ctx != nullptr) return;
  install_crash_report();
  RtmScope rtm(RTM_PREPARE_CONTEXT, false);
  std::lock_guard<std::recursive_mutex> prep_lock(shared_mu());
  if (g_primary != nullptr) {  // another thread prepared already: this one becomes a view of that context
    (void)ctx();
    return;
  }
  CKKS_PARAMS* prm = Get_context_params();
  RT_ASSERT(prm != nullptr, "Get_context_params() returned NULL");
  // both libraries carry the fingerprint of the sources they were built from (ace-compiler_amd/build.py): a shim next to a kernel
  // library of another revision is a build accident that must not run
  RT_ASSERT(strcmp(acehip_source_fingerprint(), acehip_rt_source_fingerprint()) == 0,
            "libFHErt_ant (sources %s) and libacehip (sources %s) were built from different revisions: rebuild both "
            "(python -m ace_compiler_amd.build --force)", acehip_rt_source_fingerprint(), acehip_source_fingerprint());
  RT_ASSERT(acehip_device_count() > 0, "no MI355X visible: the rt_ant HIP provider has no CPU fallback");
  auto* c = new Context();
  c->prm = prm;
  int dev = 0;
  if (const char* e = getenv("ACEHIP_DEVICE")) dev = atoi(e);
  else if (getenv("ACEHIP_SHARD") && atoi(getenv("ACEHIP_SHARD")) != 0 && getenv("LOCAL_RANK")) dev = atoi(getenv("LOCAL_RANK"));  // one GPU per rank
  c->hip = acehip_ctx_create(prm->_poly_degree, (uint32_t)prm->_mul_depth + 1, (uint32_t)prm->_first_mod_size,
                             (uint32_t)prm->_scaling_mod_size, (uint32_t)prm->_num_q_parts, dev);
  RT_ASSERT(c->hip != nullptr, "acehip_ctx_create failed: %s", acehip_last_error());
  c->N = prm->_poly_degree;
  c->L = acehip_num_q(c->hip);
  c->K = acehip_num_p(c->hip);
  c->dnum = acehip_num_q_parts(c->hip);
  c->alpha = acehip_part_size(c->hip);
  c->sf_bits = (u32)prm->_scaling_mod_size;
  c->q0_bits = (u32)prm->_first_mod_size;
  c->sf = (double)(1ull << c->sf_bits);
  c->hamming = prm->_hamming_weight;
  c->primes.resize(c->L + c->K);
  for (u32 i = 0; i < c->L + c->K; ++i) c->primes[i] = acehip_prime(c->hip, i);
  c->qmod.resize(c->L);
  c->pmod.resize(c->K ? c->K : 1);
  for (u32 i = 0; i < c->L; ++i) c->qmod[i] = MODULUS{(int64_t)c->primes[i], i, 0};
  for (u32 j = 0; j < c->K; ++j) c->pmod[j] = MODULUS{(int64_t)c->primes[c->L + j], c->L + j, 0};
  // 8 x 32 bits from the OS entropy source (the reference seeds rand() with the time of day, random_sample.c:23); a fixed
  // ACEHIP_SEED makes runs reproducible.  seed_rng only ever hands out seeds for attaching threads, under shared_mu.
  if (const char* e = getenv("ACEHIP_SEED")) {
    const u64 seed = strtoull(e, nullptr, 10);
    c->key_seed = seed;
    c->rng.seed(seed);
    c->seed_rng.seed(seed ^ 0x9E3779B97F4A7C15ull);
  } else {  // normal operation: 256 bits from the OS key every stream (limb-sharded ranks get rank 0's: shard_connect)
    os_random(c->master_key, sizeof c->master_key);
    c->drbg = true;
    c->rng.key(c->master_key, 'E', 0);
  }
  // canonical-embedding tables (Precompute_fft ntt.c:587-610), m = 2N
  const size_t m = 2ull * c->N;
  c->fft_rou.resize(m);
  for (size_t i = 0; i < m; ++i) {
    const double angle = 2 * M_PI * i / m;
    double sn, cs;
    sincos(angle, &sn, &cs);  // what gcc emits for the reference's cos()/sin() pair (see csrc/api_ops.cpp)
    c->fft_rou[i] = cplx(cs, sn);
  }
  c->rot_group.resize(c->N / 2);
  c->rot_group[0] = 1;
  for (size_t i = 1; i < c->N / 2; ++i) c->rot_group[i] = (u32)((5ull * c->rot_group[i - 1]) % m);
  c->profile = getenv("ACEHIP_PROFILE") != nullptr;
  if (const char* e = getenv("ACEHIP_BATCH")) c->batch = std::max(1, atoi(e));
  if (const char* e = getenv("ACEHIP_SHARD_SIM")) {  // G simulated ranks on this GPU (tests): rank r's limbs in replica r
    const int g = atoi(e);
    RT_ASSERT(g >= 1 && g <= 16, "ACEHIP_SHARD_SIM: 1..16 simulated ranks");
    if (g > 1) {
      c->shard_world = (u32)g;
      c->shard_sim = true;
      RT_ASSERT(c->batch == 1, "ACEHIP_SHARD_SIM and ACEHIP_BATCH exclude each other");
    }
  }
  g_ctx = c;
  g_primary = c;
  shard_connect_if_asked();  // ACEHIP_SHARD=1: this process is rank RANK of WORLD_SIZE (RCCL over xGMI)
  // first stdout line parsed by scripts/perf.py:266-276 (context.c:49-57)
  printf("ckks_param: _provider = %d, _poly_degree = %d, _sec_level = %ld, mul_depth = %ld, _first_mod_size = %ld, "
         "_scaling_mod_size = %ld, _num_q_parts = %ld, _num_p = %ld, _num_rot_idx = %ld,_hamming_wieght = %ld\n",
         prm->_provider, prm->_poly_degree, (long)prm->_sec_level, (long)prm->_mul_depth, (long)prm->_first_mod_size,
         (long)prm->_scaling_mod_size, (long)c->dnum, (long)c->K, (long)prm->_num_rot_idx, (long)prm->_hamming_weight);
  // key set: generated, or taken from / written to a key container (rt_serial.cpp)
  const char* kfile = getenv("ACEHIP_KEYS_FILE");
  c->keys_strict = getenv("ACEHIP_KEYS_STRICT") != nullptr && atoi(getenv("ACEHIP_KEYS_STRICT")) != 0;
  bool have_keys = false;
  if (kfile != nullptr && *kfile) {
    const int rc = load_keys(kfile);
    RT_ASSERT(rc == 0 || rc == -1, "ACEHIP_KEYS_FILE=%s: %s", kfile, rc == -3 ? "written for other CKKS parameters" : "truncated or not a key file");
    have_keys = rc == 0;
    if (!have_keys) c->keys_save_path = kfile;  // does not exist yet: written by Finalize_context
  }
  {
    UniformScope shared_by_all_images;  // keys, bootstrap tables, the weight file: one copy whatever the batch
    if (!have_keys) generate_keys();
    bootstrap_setup_if_needed();
    RT_DATA_INFO* di = rt_data_info();
    if (di != nullptr) {
      bool ok = Pt_mgr_init(di->_file_name);
      RT_ASSERT(ok, "Pt_mgr_init(%s) failed", di->_file_name);
    }
    rt::sync();
  }
  set_launch_mode(0, c->shard_sim ? 1 : c->batch);
}

void Finalize_context() {
  if (g_ctx == nullptr) return;
  if (g_ctx->secondary) {  // a worker thread leaves; the keys stay with the primary context
    thread_release();
    return;
  }
  std::lock_guard<std::recursive_mutex> fin_lock(shared_mu());
  const uint64_t fin_t0 = rtm_enabled() ? rtm_now() : 0;
  Context& c = *g_ctx;
  rt::sync();
  HIPCHK(acehip_encode_status(c.hip));  // the reference asserts on encode overflow; report it at the latest here
  if (!c.keys_save_path.empty()) RT_ASSERT(save_keys(c.keys_save_path.c_str()) == 0, "cannot write key file %s", c.keys_save_path.c_str());
  if (rt_data_info() != nullptr) Pt_mgr_fini();
  const size_t key_words = (size_t)c.dnum * 2 * (c.L + c.K) * c.N;
  const size_t rot_cnt = c.auto_keys.size();
  const size_t rot_bytes = rot_cnt * key_words * 8;
  const size_t total = rot_bytes + key_words * 8 + (size_t)(c.L + c.K) * c.N * 8 + 2ull * c.L * c.N * 8;
  printf("Total memory size for keys: rot_key_cnt = %ld, rot_key_size = %ld bytes, total_key_size = %ld bytes\n",
         (long)rot_cnt, (long)rot_bytes, (long)total);
  size_t wp_cnt = 0, wp_bytes = 0;
  weight_plain_totals(&wp_cnt, &wp_bytes, true);  // every image thread's encodes, once per image served (count_weight_plain)
  printf("Total memory size for weight plain: cnt = %ld, size = %ld bytes\n", (long)wp_cnt, (long)wp_bytes);
  if (c.profile) {
    printf("[ACEHIP] host seconds: encode (launch side) %.3f, Main_graph issue %.3f (until the last call returns), Main_graph %.3f\n",
           c.t_encode, c.t_issue, c.t_main);
    printf("[ACEHIP] %zu Bootstrap calls: %.3f s (synchronised on both sides while profiling)\n", c.n_bootstrap, c.t_bootstrap);
    printf("[ACEHIP] %zu encodes, %zu of them launched ahead of the per-limb queue\n", c.n_encode, c.n_encode_ahead);
    printf("[ACEHIP] pool arena: %.0f MB per replica, %.1f MB live at most; batch %u, limb-sharded world %u%s\n", arena_bytes() / 1048576.0,
           arena_live_peak_bytes() / 1048576.0, c.batch, c.shard_world, c.shard_sim ? " (simulated)" : "");
    {
      uint64_t backed = 0, addressed = 0;
      acehip_limb_memory(&backed, &addressed);
      if (addressed) printf("[ACEHIP] switch-key memory on this rank: %.1f MB of limbs backed of %.1f MB addressed%s\n", backed / 1e6, addressed / 1e6,
                            backed < addressed ? " (owner-only limbs: the other ranks' limb positions map one shared scratch limb; the driver rounds every "
                                                 "limb's handle up to 2 MiB)" : "");
    }
    if (c.shard_world > 1) {
      uint64_t steps[2] = {0, 0};
      const uint64_t b = acehip_shard_traffic(c.hip, steps, 0);
      printf("[ACEHIP] limb exchanges: %llu steps, %llu limbs, %.3f GB received, %llu collectives\n", (unsigned long long)steps[0],
             (unsigned long long)steps[1], b / 1e9, (unsigned long long)acehip_shard_collectives(c.hip));
    }
    hw_stats_print();
    hw_flush_sites_print();
    acehip_stat st[16];
    const int nf = acehip_stats(st, 16, 0);
    unsigned long long total = 0;
    for (int i = 0; i < nf && i < 16; ++i) {
      printf("[ACEHIP] %-18s calls %9llu units %10llu algorithmic GB %10.3f\n", acehip_stat_name(i),
             (unsigned long long)st[i].calls, (unsigned long long)st[i].units, st[i].bytes / 1e9);
      if (strcmp(acehip_stat_name(i), "zero_fill_executed") != 0 && strcmp(acehip_stat_name(i), "elementwise_mul") != 0 &&
          strcmp(acehip_stat_name(i), "ntt_launched") != 0)
        total += st[i].bytes;  // (those two are subsets of "elementwise")
    }
    printf("[ACEHIP] algorithmic bytes since process start: %.3f GB\n", total / 1e9);
  }
  for (auto& kv : c.auto_keys) free_switch_key(kv.second);
  c.auto_keys.clear();
  dfree(c.relin.data);
  c.relin.data = nullptr;
  dfree(c.sk_ntt);
  dfree(c.pk0);
  dfree(c.pk1);
  c.sk_ntt = c.pk0 = c.pk1 = nullptr;
  bootstrap_release();
  ev::clear_monomial_cache();
  stage_release();
  pool_release_all();
  acehip_ctx_destroy(c.hip);
  delete g_ctx;
  g_ctx = nullptr;
  g_primary = nullptr;
  if (fin_t0) rtm_add(RTM_FINALIZE_CONTEXT, rtm_now() - fin_t0);
  rtm_report();  // RTLIB_TM_REPORT() at the end of Finalize_context (context.c:130-133)
}
// Extension: a worker thread that used the API (it attached to the prepared context on first use) gives back
// its scratch context, pool and queue before it ends; Finalize_context from such a thread does the same.
void Acehip_rt_thread_release(void) { thread_release(); }

// Extension for callers that touch Coeffs() memory themselves (acehip_* / HIP calls on the raw device
// pointers): hands over everything the shim still holds back and waits for the device.
void Acehip_rt_sync(void) { rt::sync(); }
void Acehip_rt_next_input(void) { pt_image_boundary(); }
void Acehip_rt_seed_encryptor(uint64_t seed) { ctx().rng.seed(seed); }
// test hook: the block function every stream of rt_rng.hpp is made of (RFC 8439 2.3.2 has the known answer)
void acehip_rt_debug_chacha20_block(const uint32_t* key, uint32_t counter, const uint32_t* nonce, uint32_t* out) {
  rt::ChaCha20::block(key, counter, nonce, out);
}

// Extension: image batches.  B images run through every launch of this thread (Run_main_graph is called ONCE per batch): the
// GPU form of the reference's image-parallel loop (rtlib/ant/dataset/resnet_cifar.main.inc:77-116), where the threads share
// keys and weights (pt_mgr.c:182).  Call after Prepare_context and before the thread's first Prepare_input; then, per batch:
// for k < B { Acehip_rt_select_image(k); Prepare_input(...); }  Run_main_graph();  for k < B { Acehip_rt_select_image(k);
// Handle_output(...); }.  ACEHIP_BATCH=B sets the default.
void Acehip_rt_set_batch(uint32_t b) {
  Context& c = ctx();
  RT_ASSERT(b >= 1 && b <= 64, "Acehip_rt_set_batch: 1..64 images");
  RT_ASSERT(!c.shard_sim || b == 1, "simulated limb-sharded execution runs one image");
  set_batch_aware();
  if (b == c.batch) return;
  rt::sync();
  RT_ASSERT(arena_peak_bytes() == 0, "Acehip_rt_set_batch: call it before the thread's first input (its pool is in use already)");
  c.batch = b;
  set_launch_mode(0, b);
}
uint32_t Acehip_rt_batch(void) { return batch_size(); }
void Acehip_rt_select_image(uint32_t k) {
  set_batch_aware();
  select_image(k);
}
// Extension: limb-sharded execution (BASELINE configs[4]).  world / rank of this process and what the exchanges moved so far.
uint32_t Acehip_rt_shard_world(void) { return g_ctx ? g_ctx->shard_world : 1; }
uint32_t Acehip_rt_shard_rank(void) { return g_ctx ? g_ctx->shard_rank : 0; }
uint64_t Acehip_rt_shard_traffic(uint64_t* steps_limbs /* [2] or NULL */, int reset) {
  return g_ctx ? acehip_shard_traffic(g_ctx->hip, steps_limbs, reset) : 0;
}
size_t Acehip_rt_prefetched_count(void) { return g_ctx ? g_ctx->n_encode_prefetched : 0; }

// ---- key accessors (key_gen.h:28-75) ----
uint32_t Auto_idx(int32_t rot_idx) { return ensure_rot_key(rot_idx); }
int64_t* Auto_order(int32_t rot_idx) {
  const uint32_t* perm = acehip_auto_order(ctx().hip, Auto_idx(rot_idx));
  RT_ASSERT(perm != nullptr, "cannot get precompute automorphism order");
  return (int64_t*)perm;
}
SW_KEY Swk(bool is_rot, int32_t rot_idx) {
  if (!is_rot) return &ctx().relin.key;
  return &ensure_auto_key(Auto_idx(rot_idx))->key;
}
POLY Pk0_at(SW_KEY swk, uint32_t idx) {
  RT_ASSERT(idx < swk->_num_parts, "switch key part index out of range");
  return &swk->_parts[idx]._pk0;
}
POLY Pk1_at(SW_KEY swk, uint32_t idx) {
  RT_ASSERT(idx < swk->_num_parts, "switch key part index out of range");
  return &swk->_parts[idx]._pk1;
}

void Run_main_graph() {
  RtmScope rtm(RTM_MAIN_GRAPH);  // common/src/rt_lib.c:16-21
  const double t0 = wall_s();
  bool ok = Main_graph();
  const double t1 = wall_s();
  rt::sync();
  ctx().t_main += wall_s() - t0;
  ctx().t_issue += t1 - t0;
  RT_ASSERT(ok, "Main_graph() failed");
}

}  // extern "C"
