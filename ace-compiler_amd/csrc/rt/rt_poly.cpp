// rt_poly.cpp -- device memory pool, POLYNOMIAL helpers and the polynomial/ciphertext part of the
// rt_ant API (reference: include/poly/poly_eval.h, src/poly/{poly_eval,poly_arith}.c,
// src/ckks/cipher_eval.c:18-123, include/util/{polynomial,ciphertext}.h).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <ctime>
#include <set>

#include "rt_internal.hpp"

namespace rt {

// All runtime state is per host thread: a thread that calls Prepare_context owns a context, a memory pool, an
// operation queue and (through libacehip's per-thread default stream) a HIP stream of its own, so several
// threads can push images through the same GPU concurrently -- the reference runs one OpenMP thread per image
// (resnet_cifar.main.inc:77-116) on a shared context; here the keys are per thread.
thread_local Context* g_ctx = nullptr;
Context* g_primary = nullptr;
std::recursive_mutex& shared_mu() {
  static std::recursive_mutex mu;
  return mu;
}
// A thread that uses the API without having prepared a context itself (the OpenMP workers of the reference's
// main(), dataset/resnet_cifar.main.inc:77-116) attaches to the primary one: same parameters and keys (shared,
// read-only device memory), its own acehip_ctx (scratch, tables), pool, queue and HIP stream.
static void attach_thread() {
  std::lock_guard<std::recursive_mutex> lk(shared_mu());
  RT_ASSERT(g_primary != nullptr, "rt_ant context is not prepared (call Prepare_context first)");
  Context* p = g_primary;
  auto* c = new Context(*p);
  c->secondary = true;
  RT_ASSERT(p->shard_world <= 1 || p->shard_sim, "limb-sharded execution over RCCL runs one host thread per process");
  c->auto_keys.clear();  // the shared maps live in the primary context only
  c->rot2auto.clear();
  int dev = 0;
  if (const char* e = getenv("ACEHIP_DEVICE")) dev = atoi(e);
  c->hip = acehip_ctx_create(p->prm->_poly_degree, (uint32_t)p->prm->_mul_depth + 1, (uint32_t)p->prm->_first_mod_size,
                             (uint32_t)p->prm->_scaling_mod_size, (uint32_t)p->prm->_num_q_parts, dev);
  RT_ASSERT(c->hip != nullptr, "acehip_ctx_create failed: %s", acehip_last_error());
  // encryption noise of this thread: its own ChaCha20 stream of the master key (test mode: a seed from seed_rng; both under shared_mu)
  if (p->drbg) c->rng.key(p->master_key, 'E', ++p->enc_streams);
  else c->rng.seed(p->seed_rng() ^ (u64)(uintptr_t)c);
  c->t_encode = c->t_main = c->t_issue = c->t_bootstrap = 0;
  c->n_bootstrap = 0;
  g_ctx = c;
}
Context& ctx() {
  if (g_ctx == nullptr) attach_thread();
  return *g_ctx;
}

// shared device memory: plain hipMalloc, remembered so that any thread can free it
static std::mutex g_shared_alloc_mu;
static std::map<u64*, size_t> g_shared_allocs;
static thread_local int g_shared_scope = 0;
SharedAllocScope::SharedAllocScope() { ++g_shared_scope; }
SharedAllocScope::~SharedAllocScope() { --g_shared_scope; }
u64* shared_alloc(size_t words, bool zero) {
  if (words == 0) words = 1;
  u64* p = (u64*)acehip_malloc(words * sizeof(u64));
  RT_ASSERT(p != nullptr, "device allocation of %zu bytes failed: %s", words * sizeof(u64), acehip_last_error());
  {
    std::lock_guard<std::mutex> lk(g_shared_alloc_mu);
    g_shared_allocs[p] = words;
  }
  if (zero) HIPCHK(acehip_memset(p, 0, words * sizeof(u64), nullptr));
  return p;
}
// a switch key (n_polys polynomials of L + K limbs, reference layout): on a rank of limb-sharded execution over RCCL only the limbs it
// owns are physically backed (acehip_malloc_limbs, include/acehip.h); elsewhere a plain shared allocation
u64* shared_alloc_key(size_t n_polys) {
  Context& c = ctx();
  const u32 T = c.L + c.K;
  std::vector<uint32_t> gi(n_polys * T);
  for (size_t k = 0; k < gi.size(); ++k) gi[k] = (uint32_t)(k % T);
  u64* p = (u64*)acehip_malloc_limbs(c.hip, gi.data(), gi.size());
  RT_ASSERT(p != nullptr, "device allocation of a switch key (%zu limbs) failed: %s", gi.size(), acehip_last_error());
  std::lock_guard<std::mutex> lk(g_shared_alloc_mu);
  g_shared_allocs[p] = gi.size() * (size_t)c.N;
  return p;
}
static bool shared_free(u64* p) {
  {
    std::lock_guard<std::mutex> lk(g_shared_alloc_mu);
    auto it = g_shared_allocs.find(p);
    if (it == g_shared_allocs.end()) return false;
    g_shared_allocs.erase(it);
  }
  hw_flush_site(__FILE__, __LINE__);  // nothing queued may still name it
  acehip_free(p);
  return true;
}

// ---- pool: exact-size free lists.  All launches go to the default stream in program order, so a
// buffer released by Free_* can be handed out again immediately: later kernels are ordered after
// earlier ones.  Nothing is returned to the driver before Finalize_context (hipFree synchronises).
//
// Two kinds of blocks.  ARENA blocks are carved out of one slab of which `nrep` copies exist, a constant stride apart
// (acehip_ctx_set_arena): the images of a batch -- or the simulated ranks of limb-sharded execution -- each have their own
// copy of every such block and the kernels move pointers into the slab along with the replica they work on.  UNIFORM blocks
// are separate device allocations outside the slab: what every image shares (keys, bootstrap diagonals, weight plaintexts,
// anything made inside a UniformScope); they are written by launches that cover one replica.
enum BlockKind : int { BK_ARENA = 0, BK_UNIFORM = 1 };
struct LiveBlock {
  size_t words;
  int kind;
};
static thread_local std::mutex pool_mu;
static thread_local std::map<size_t, std::vector<u64*>> pool_free[2];
static thread_local std::map<u64*, LiveBlock> pool_live;
static thread_local size_t pool_live_bytes = 0;
// Blocks freed while per-limb ops are still queued may be named by those ops: they wait here until the queue has been
// handed to the device (hw_flush), and only then become reusable.  A block taken from pool_free is therefore never
// referenced by anything still queued -- which is what lets work that only writes a fresh block (Pt_from_msg's encode)
// be launched ahead of the queue instead of cutting it (rt_encode.cpp encode_device).
struct LimboBlock {
  u64* first;
  size_t second;
  int kind;
};
static thread_local std::vector<LimboBlock> pool_limbo;

namespace {
struct Arena {
  u64* base = nullptr;   // replica 0
  size_t words = 0;      // per replica (= stride)
  // free extents of the slab, by address: start (words) -> length.  Freed blocks wait in the exact-size lists (pool_free: the
  // program asks for the same few hundred sizes over and over) and are merged back into extents only when a request finds
  // neither a block of its size nor a large enough extent: the footprint stays near the live bytes instead of the sum of every
  // size class's high-water mark
  std::map<size_t, size_t> ext;
  size_t peak = 0;       // highest word ever handed out
  size_t live = 0, live_peak = 0;  // words in live blocks (now / most ever)
  u32 nrep = 0;
  bool exhausted_warned = false;
};
thread_local Arena g_arena;
thread_local int g_uniform_depth = 0;   // UniformScope nesting
thread_local int g_alloc_uniform = 0;   // allocations come from the uniform pool (UniformScope, shared plaintexts)
thread_local bool g_image_scope = false;
thread_local u32 g_mode_rep0 = 0, g_mode_nrep = 1;  // replicas the launches of this thread cover at the moment
thread_local u32 g_image = 0;                        // image Prepare_input / Handle_output address (Acehip_rt_select_image)

inline bool in_arena(const void* p) {
  return g_arena.base != nullptr && (const u64*)p >= g_arena.base && (const u64*)p < g_arena.base + g_arena.words;
}
// the slab: created on the first allocation that needs it, when the batch size / the simulated world are known
void arena_create() {
  Context& c = ctx();
  Arena& a = g_arena;
  const u32 nrep = std::max<u32>(c.batch, c.shard_sim ? c.shard_world : 1);
  size_t mb = 0;
  if (const char* e = getenv("ACEHIP_ARENA_MB")) mb = strtoull(e, nullptr, 10);
  if (mb == 0) {  // enough for one image stream of the generated ResNets at N = 2^16 (profile line "pool arena"), scaled down with N
    mb = (size_t)4096 * c.N / 65536;  // live peak of a ResNet-20 / -110 image: 1.1 GB
    if (mb < 64) mb = 64;
  }
  const size_t N = c.N, guard = (size_t)(c.L + c.K) * N, ws_words = acehip_workspace_words(c.hip);
  const size_t scratch_limbs = 512;
  size_t words = mb * (1u << 20) / 8;
  const size_t fixed = guard + ws_words + scratch_limbs * N;
  if (words < 2 * fixed) words = 2 * fixed;
  words = (words + 31) & ~(size_t)31;
  a.base = (u64*)acehip_malloc(words * 8 * nrep);
  RT_ASSERT(a.base != nullptr, "pool arena: %zu MB x %u replicas: %s (ACEHIP_ARENA_MB sets the size of one replica)", words * 8 >> 20, nrep,
            acehip_last_error());
  a.words = words;
  a.nrep = nrep;
  // the first limbs stay unused: kernels may form addresses a few limbs below a block (a limb position subtracted from a
  // base) and those must still fall inside the slab to be moved with their replica
  size_t bump = guard;
  acehip_arena_cfg cfg{};
  cfg.base = a.base;
  cfg.bytes = words * 8;
  cfg.stride_bytes = words * 8;
  cfg.n_replicas = nrep;
  cfg.workspace = a.base + bump;
  bump += (ws_words + 31) & ~(size_t)31;
  cfg.hw_scratch = a.base + bump;
  cfg.hw_scratch_limbs = scratch_limbs;
  bump += scratch_limbs * N;
  a.peak = bump;
  a.ext.clear();
  a.ext[bump] = words - bump;
  const int rc = acehip_ctx_set_arena(c.hip, &cfg);
  RT_ASSERT(rc >= 0, "acehip_ctx_set_arena: %s", acehip_last_error());
  if (c.shard_sim) {
    const int rs = acehip_ctx_shard_sim(c.hip, c.shard_world);
    RT_ASSERT(rs >= 0, "acehip_ctx_shard_sim: %s", acehip_last_error());
  }
  if (g_uniform_depth == 0 && !g_image_scope) {
    g_mode_rep0 = 0;
    g_mode_nrep = c.shard_sim ? 1 : c.batch;
  }
  const int rsel = acehip_ctx_select(c.hip, g_mode_rep0, g_mode_nrep);
  RT_ASSERT(rsel >= 0, "acehip_ctx_select: %s", acehip_last_error());
}
inline size_t granules(size_t words) { return (words + 31) & ~(size_t)31; }  // 256 bytes
u64* arena_take(size_t words) {  // first fit, lowest address
  Arena& a = g_arena;
  if (a.base == nullptr) arena_create();
  const size_t w = granules(words);
  for (auto it = a.ext.begin(); it != a.ext.end(); ++it) {
    if (it->second < w) continue;
    const size_t off = it->first, len = it->second;
    a.ext.erase(it);
    if (len > w) a.ext[off + w] = len - w;
    a.peak = std::max(a.peak, off + w);
    return a.base + off;
  }
  return nullptr;
}
void arena_give(u64* p, size_t words) {  // back into the extents, merged with its neighbours
  Arena& a = g_arena;
  size_t off = (size_t)(p - a.base), len = granules(words);
  auto nx = a.ext.lower_bound(off);
  if (nx != a.ext.end() && off + len == nx->first) {
    len += nx->second;
    nx = a.ext.erase(nx);
  }
  if (nx != a.ext.begin()) {
    auto pv = std::prev(nx);
    if (pv->first + pv->second == off) {
      pv->second += len;
      return;
    }
  }
  a.ext[off] = len;
}
}  // namespace
bool in_image_scope() { return g_image_scope; }
bool uniform_alloc_on() { return g_alloc_uniform > 0; }
u32 current_rep0() { return g_mode_rep0; }
u32 current_nrep() { return g_mode_nrep; }
u32 batch_size() { return g_ctx ? g_ctx->batch : 1; }
static std::atomic<size_t> g_weight_plain_cnt{0}, g_weight_plain_bytes{0};
void count_weight_plain(size_t bytes) {
  const size_t images = (g_ctx && !g_ctx->shard_sim) ? g_ctx->batch : 1;  // (simulated ranks are replicas of ONE image)
  g_weight_plain_cnt.fetch_add(images, std::memory_order_relaxed);
  g_weight_plain_bytes.fetch_add(images * bytes, std::memory_order_relaxed);
}
void weight_plain_totals(size_t* cnt, size_t* bytes, bool reset) {
  *cnt = reset ? g_weight_plain_cnt.exchange(0) : g_weight_plain_cnt.load();
  *bytes = reset ? g_weight_plain_bytes.exchange(0) : g_weight_plain_bytes.load();
}
u32 selected_image() { return g_image; }
static bool g_batch_aware = false;  // the program addresses images itself (Acehip_rt_set_batch / Acehip_rt_select_image were called)
bool batch_aware() { return g_batch_aware; }
void set_batch_aware() { g_batch_aware = true; }
void select_image(u32 k) {
  RT_ASSERT(k < batch_size(), "image %u outside the batch of %u", k, batch_size());
  g_image = k;
}
size_t arena_peak_bytes() { return g_arena.peak * 8; }
size_t arena_live_peak_bytes() { return g_arena.live_peak * 8; }
size_t arena_bytes() { return g_arena.words * 8; }
// replicas the launches that follow cover; the queue is handed over first when the selection changes (its ops were
// queued for the old one)
void set_launch_mode(u32 rep0, u32 nrep) {
  if (rep0 == g_mode_rep0 && nrep == g_mode_nrep) return;
  hw_flush_site(__FILE__, __LINE__);
  g_mode_rep0 = rep0;
  g_mode_nrep = nrep;
  if (g_ctx != nullptr && g_ctx->hip != nullptr && g_arena.base != nullptr) {
    const int rc = acehip_ctx_select(g_ctx->hip, rep0, nrep);
    RT_ASSERT(rc >= 0, "acehip_ctx_select(%u, %u): %s", rep0, nrep, acehip_last_error());
  }
}
UniformScope::UniformScope() : rep0(g_mode_rep0), nrep(g_mode_nrep) {
  set_launch_mode(0, 1);
  ++g_uniform_depth;
  ++g_alloc_uniform;
}
UniformScope::~UniformScope() {
  --g_uniform_depth;
  --g_alloc_uniform;
  set_launch_mode(rep0, nrep);
}
ImageScope::ImageScope(u32 k) : rep0(g_mode_rep0), nrep(g_mode_nrep), was(g_image_scope) {
  if (batch_size() > 1) set_launch_mode(k, 1);
  g_image_scope = true;
}
ImageScope::~ImageScope() {
  g_image_scope = was;
  set_launch_mode(rep0, nrep);
}
UniformAlloc::UniformAlloc(bool on) : on(on) { g_alloc_uniform += on; }
UniformAlloc::~UniformAlloc() { g_alloc_uniform -= on; }
SelectGuard::SelectGuard(u32 r0, u32 n) : rep0(g_mode_rep0), nrep(g_mode_nrep), active(r0 != g_mode_rep0 || n != g_mode_nrep) {
  if (active && g_arena.base != nullptr) {
    const int rc = acehip_ctx_select(ctx().hip, r0, n);
    RT_ASSERT(rc >= 0, "acehip_ctx_select: %s", acehip_last_error());
  }
}
SelectGuard::~SelectGuard() {
  if (active && g_arena.base != nullptr) acehip_ctx_select(ctx().hip, rep0, nrep);
}

// ---- deferred per-limb ops ----
// Generated code calls Hw_modadd / Hw_modmul / Hw_rotate once per RNS limb and component inside host loops
// (resnet20_cifar10_pre.onnx.inc:1492-1503).  They are queued here and handed to acehip_hw_batch when any other
// device work is issued (HIPCHK), so a whole loop nest becomes a few launches; zero fills and limb copies
// that sit between such loops ride in the same queue (one queue per host thread).
namespace {
// ---- ACEHIP_POISON=1: the safety net of the lazy execution below (debug mode, off by default).
// Everything the runtime DEFERS or GIVES UP is overwritten with a value no residue can have: the limb of a zero fill that is
// held back, every block that goes back to the pool.  A launch that reads an operand its HIPCHK_T list forgot (the fill is
// still waiting), or memory that was freed, then computes with garbage instead of with stale data that happens to be right --
// and the bit-exact comparisons of the test programs fail.  The declared lists themselves are checked against what the library's
// entry points report (acehip_debug_touches): a range of pool memory a launch touches without declaring it aborts at the site.
bool poison_on() {
  static const bool on = [] {
    const char* e = getenv("ACEHIP_POISON");
    return e != nullptr && atoi(e) != 0;
  }();
  return on;
}
void poison(const u64* p, size_t words) {
  // 0xA5A5...: above every prime of the chain (primes are below 2^61)
  const int rc = acehip_fill(ctx().hip, const_cast<u64*>(p), 0xA5, words * sizeof(u64), nullptr);
  RT_ASSERT(rc >= 0, "poison fill failed: %s", acehip_last_error());
}
thread_local std::vector<acehip_hw_op> g_hwq;
struct HwqStats {
  size_t flushes = 0, ops = 0, by_kind[9] = {}, hist[8] = {};  // hist: <=1, <=4, <=16, <=64, <=256, <=1024, <=4096, more
  size_t res_limbs = 0, res_limbs_freed = 0;  // distinct result limbs per flush; those whose block was already freed
};
thread_local HwqStats g_hwq_stats;
// ACEHIP_PROFILE: limbs a handed-over queue stored that a LATER queue loads again although no direct launch in between
// named them (what keeping such ops queued across declared launches can save at most)
struct ReloadStats {
  std::map<const u64*, std::pair<size_t, u32>> stored;  // limb -> (index of the hand-over that wrote it, kind of the op that did)
  size_t loads = 0, reloads = 0, reload_dist[6] = {};  // distance in hand-overs: 1, 2, <=4, <=8, <=16, more
  size_t pair[3][9][9] = {};  // [distance <= 2, <= 16, more][producer kind][consumer kind]
  size_t erased_by_touch = 0, cleared = 0;
};
thread_local ReloadStats g_reload;
bool reload_diag_on() {  // ACEHIP_PROFILE_RELOADS=1 with ACEHIP_PROFILE=1 (a std::map per limb: slow)
  static const bool on = getenv("ACEHIP_PROFILE_RELOADS") != nullptr;
  return on && ctx().profile;
}
// ---- lazy zero fills ----
// Generated code zero-fills a result (Init_ciph_*, Alloc_poly) long before the first per-limb op accumulates into it:
// a rotation with its key-switch lies in between, whose direct launches hand the queue over.  Issued there, the fill is
// written to memory and read back by its first consumer (200 k limbs = 100 GB of stores and as much again in loads per
// ResNet-20 image).  Instead, a fill that nothing queued behind it touches is taken out of the queue when the launch
// that forces the hand-over declares its operands (HIPCHK_T) and does not touch the limb; the limb waits here and the
// fill is put back right in front of the first queued op that names it -- where the batch kernel keeps it in registers
// (hw_batch_ew_kernel) -- or is issued as soon as an undeclared launch (HIPCHK) or one that touches it comes up.
thread_local std::set<const u64*> g_lazy;
struct LazyStats {
  size_t deferred = 0, met_consumer = 0, materialised = 0, dropped = 0;
};
thread_local LazyStats g_lazy_stats;
struct KeepStats {
  size_t launches = 0, kept_some = 0, ops_kept = 0, ops_submitted = 0, forced_full = 0, limbo_pinned = 0;
  double t_split = 0, t_limbo = 0, t_submit = 0;  // host seconds (ACEHIP_PROFILE)
};
thread_local KeepStats g_keep_stats;
}
static void muc_stats_print();
void hw_stats_print() {
  const HwqStats& s = g_hwq_stats;
  printf("[ACEHIP] hw queue: %zu flushes, %zu limb-ops (add %zu mul %zu rot %zu copy %zu zero %zu sub %zu muladd %zu mulc %zu addc %zu); "
         "ops per flush <=1:%zu <=4:%zu <=16:%zu <=64:%zu <=256:%zu <=1024:%zu <=4096:%zu more:%zu\n",
         s.flushes, s.ops, s.by_kind[0], s.by_kind[1], s.by_kind[2], s.by_kind[3], s.by_kind[4], s.by_kind[5], s.by_kind[6],
         s.by_kind[7], s.by_kind[8], s.hist[0], s.hist[1], s.hist[2], s.hist[3], s.hist[4], s.hist[5], s.hist[6], s.hist[7]);
  printf("[ACEHIP] hw queue: %zu distinct result limbs, %zu of them in blocks freed before the flush\n", s.res_limbs, s.res_limbs_freed);
  muc_stats_print();
  if (reload_diag_on()) {
    const ReloadStats& r = g_reload;
    printf("[ACEHIP] hw queue: %zu limb loads named by queued ops; %zu of them re-load a limb an EARLIER hand-over stored with no direct launch naming it in between "
           "(hand-overs apart 1:%zu 2:%zu <=4:%zu <=8:%zu <=16:%zu more:%zu); %zu stored limbs consumed by declared launches, %zu map resets by undeclared ones\n",
           r.loads, r.reloads, r.reload_dist[0], r.reload_dist[1], r.reload_dist[2], r.reload_dist[3], r.reload_dist[4], r.reload_dist[5],
           r.erased_by_touch, r.cleared);
    static const char* const kn[9] = {"add", "mul", "rot", "copy", "zero", "sub", "muladd", "mulc", "addc"};
    static const char* const dn[3] = {"<=2", "<=16", ">16"};
    for (int d = 0; d < 3; ++d)
      for (int a = 0; a < 9; ++a)
        for (int b = 0; b < 9; ++b)
          if (r.pair[d][a][b] * 200 > r.reloads)
            printf("[ACEHIP] reload %s hand-overs apart: stored by %s, loaded by %s: %zu\n", dn[d], kn[a], kn[b], r.pair[d][a][b]);
  }
  printf("[ACEHIP] queue kept open: %zu declared launches met a non-empty queue, %zu of them left ops queued (%zu limb-ops kept, %zu handed over); "
         "%zu took everything because of the size bounds; %zu freed blocks pinned by kept ops\n",
         g_keep_stats.launches, g_keep_stats.kept_some, g_keep_stats.ops_kept, g_keep_stats.ops_submitted, g_keep_stats.forced_full,
         g_keep_stats.limbo_pinned);
  printf("[ACEHIP] queue kept open, host seconds: choosing what a launch needs %.3f, pinned freed blocks %.3f, handing lists to the library %.3f\n",
         g_keep_stats.t_split, g_keep_stats.t_limbo, g_keep_stats.t_submit);
  const LazyStats& z = g_lazy_stats;
  printf("[ACEHIP] lazy zero fills: %zu limbs deferred; %zu met their first consumer in the queue, %zu issued for a launch, %zu dropped (block freed or rewritten)\n",
         z.deferred, z.met_consumer, z.materialised, z.dropped);
}
// ---- pairing of Mod_down / Rescale calls, speculative ModUp of all digits ----
// Generated code handles the two polynomials of a ciphertext with two consecutive calls (Mod_down(c0); Mod_down(c1),
// Rescale(c0); Rescale(c1)) and raises one digit per Decomp_modup call into the same buffer.  The first call of a
// pair is held back until the next API call shows whether its partner follows (then both run in the same
// launches: acehip_mod_down2 / acehip_rescale2); the first Decomp_modup of a polynomial raises every digit at once
// (acehip_modup_digits) and the following calls copy their digit out of that result.
namespace {
struct PendingPair {
  int kind = 0;  // 0 none, 1 Mod_down, 2 Rescale
  u64* out = nullptr;
  const u64* in = nullptr;
  u32 level = 0;
};
thread_local PendingPair g_pend;
struct ModupCache {
  const u64* src = nullptr;  // q-limbs of the polynomial the digits were raised from
  u32 level = 0;
  // one pool block of (level+K)*N words per digit: a digit is handed to the caller's polynomial by SWAPPING blocks (the
  // caller's old block becomes the cache's), so Decomp_modup moves no data (it used to copy level+K limbs per digit:
  // 176 k limb copies = 6 % of the memory traffic of a ResNet-20 image).
  // The digits outlive the rotation they were raised for: generated code rotates the same ciphertext by several steps in
  // a row (the taps of a convolution), each Rotate() raising the digits of the same c1 again -- 19 % of the all-digit
  // ModUps of a ResNet-20 image.  `ok` says that the blocks still hold the digits of (src, level) as it is NOW: every write
  // the runtime can make to the source or to a digit block clears it -- queued per-limb ops (hw_queue), freed blocks
  // (dfree), the operands of declared launches (HIPCHK_T; all taken as written) and any undeclared launch (HIPCHK).
  // holds[s] = the digit block s contains (-1: nothing of value); `lent` = the block handed out last, which comes back
  // untouched either with the next Decomp_modup into the same polynomial or when the caller frees it (dfree adopts it
  // and releases a valueless block instead).
  u64* blk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int holds[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  u32 n_blk = 0;
  size_t blk_words = 0;
  bool ok = false;
  u64* lent = nullptr;
  int lent_digit = -1;
  size_t n_raise = 0, n_reuse = 0;  // all-digit ModUps launched / Rotate()s served from digits raised earlier
  void forget() {
    ok = false;
    lent = nullptr;
    for (int& h : holds) h = -1;
  }
  // [p, p + words) may have been written
  void written(const u64* p, size_t words) {
    if (!ok || p == nullptr || words == 0) return;
    const u64* e = p + words;
    if (e > src && p < src + (size_t)level * g_ctx->N) {
      forget();
      return;
    }
    if (lent && e > lent && p < lent + blk_words) lent = nullptr;
    for (u32 s = 0; s < n_blk; ++s)
      if (holds[s] >= 0 && e > blk[s] && p < blk[s] + blk_words) holds[s] = -1;
  }
};
thread_local ModupCache g_muc;
}  // namespace
// ACEHIP_PROFILE: calls per level of the three pipelines generated code drives (which levels carry the transforms)
namespace {
struct LevelHist {
  size_t modup[72] = {}, modup_reuse[72] = {}, mod_down2[72] = {}, mod_down1[72] = {}, rescale2[72] = {}, rescale1[72] = {};
};
thread_local LevelHist g_lvl;
}
// Mod_down pairs whose key inner product was still queued (keymac_pair_from_queue below)
struct KmacStats {
  size_t tried = 0, fused = 0, no_zero = 0, shape = 0, other = 0;
};
thread_local KmacStats g_kmac_stats;
static void muc_stats_print() {
  {
    const KmacStats& k = g_kmac_stats;
    printf("[ACEHIP] Mod_down pairs behind a queued key inner product: %zu examined, %zu ran without storing the accumulators; not taken: %zu zero fill "
           "already executed, %zu irregular shape, %zu other\n", k.tried, k.fused, k.no_zero, k.shape, k.other);
  }
  printf("[ACEHIP] all-digit ModUp: %zu launched, %zu more rotations served from digits raised before\n", g_muc.n_raise, g_muc.n_reuse);
  for (u32 l = 0; l < 72; ++l)
    if (g_lvl.modup[l] + g_lvl.modup_reuse[l] + g_lvl.mod_down2[l] + g_lvl.mod_down1[l] + g_lvl.rescale2[l] + g_lvl.rescale1[l])
      printf("[ACEHIP] level %2u: ModUp %zu (+%zu reused) Mod_down pairs %zu singles %zu Rescale pairs %zu singles %zu\n", l, g_lvl.modup[l],
             g_lvl.modup_reuse[l], g_lvl.mod_down2[l], g_lvl.mod_down1[l], g_lvl.rescale2[l], g_lvl.rescale1[l]);
}
namespace {
void queue_submit(const Touch* touch = nullptr, size_t n_touch = 0, bool defer = false);
void cancel_fills(const u64* out, size_t n_limbs);
// The held-back op was called after everything that is queued now (a later hw_queue() would have issued it first): hand
// the queue over, then launch it.  Its output is rewritten completely, so zero fills still queued for it are dead.
void pending_flush() {
  if (!g_pend.kind) return;
  const PendingPair p = g_pend;
  g_pend.kind = 0;
  (p.kind == 1 ? g_lvl.mod_down1 : g_lvl.rescale1)[p.level < 72 ? p.level : 71]++;
  cancel_fills(p.out, p.kind == 1 ? p.level : p.level - 1);
  const size_t N = ctx().N;
  g_muc.written(p.out, (size_t)(p.kind == 1 ? p.level : p.level - 1) * N);
  const Touch touch[2] = {{p.out, (size_t)(p.kind == 1 ? p.level : p.level - 1) * N},
                          {p.in, (size_t)(p.kind == 1 ? p.level + ctx().K : p.level) * N}};
  queue_submit(touch, 2, true);
  if (poison_on()) (void)acehip_debug_touches(1, nullptr, nullptr, 0);
  const int rc = p.kind == 1 ? acehip_mod_down(ctx().hip, p.out, p.in, p.level, nullptr)
                             : acehip_rescale(ctx().hip, p.out, p.in, p.level, nullptr);
  RT_ASSERT(rc >= 0, "deferred %s failed: %s", p.kind == 1 ? "Mod_down" : "Rescale", acehip_last_error());
  check_declared(__FILE__, __LINE__, touch, 2);
}
}  // namespace
static void limbo_release() {
  if (pool_limbo.empty()) return;
  std::lock_guard<std::mutex> lk(pool_mu);
  for (auto& b : pool_limbo) {
    if (poison_on()) poison(b.first, b.second);  // (the queue that could name the block has been handed over: ordered behind it)
    pool_free[b.kind][b.second].push_back(b.first);
  }
  pool_limbo.clear();
}
namespace {
// Generated code initialises a result (Init_ciph_*: zero fill on reuse, polynomial.h:331-348) and then has Mod_down /
// Rescale / ... rewrite it completely; the fill sits in the queue when the direct kernel is about to be launched.
// Drops queued ZERO / COPY ops on limbs [out, out + n_limbs) that no later queued op reads.
void cancel_lazy(const u64* out, size_t n_limbs);
void cancel_fills(const u64* out, size_t n_limbs) {
  cancel_lazy(out, n_limbs);
  if (g_hwq.empty() || n_limbs == 0) return;
  const size_t N = ctx().N;
  const u64* end = out + n_limbs * N;
  auto inside = [&](const void* p) { return p != nullptr && (const u64*)p >= out && (const u64*)p < end; };
  std::vector<const u64*> read_later;
  bool any = false;
  for (size_t k = g_hwq.size(); k-- > 0;) {
    acehip_hw_op& o = g_hwq[k];
    if (inside(o.res) && (o.op == ACEHIP_HW_ZERO || o.op == ACEHIP_HW_COPY)) {
      bool live = false;
      for (const u64* r : read_later) live |= r == o.res;
      if (!live) {
        o.res = nullptr;  // marks the op as cancelled
        any = true;
        continue;
      }
    }
    // what this op reads: a always (except ZERO); b for the two-operand kinds; res for MULADD
    if (o.op != ACEHIP_HW_ZERO && inside(o.a)) read_later.push_back(o.a);
    if ((o.op == ACEHIP_HW_ADD || o.op == ACEHIP_HW_SUB || o.op == ACEHIP_HW_MUL || o.op == ACEHIP_HW_MULADD) && inside(o.b))
      read_later.push_back((const u64*)o.b);
    if (o.op == ACEHIP_HW_MULADD && inside(o.res)) read_later.push_back(o.res);
  }
  if (any) {
    size_t w = 0;
    for (size_t k = 0; k < g_hwq.size(); ++k)
      if (g_hwq[k].res != nullptr) g_hwq[w++] = g_hwq[k];
    g_hwq.resize(w);
  }
}
// a deferred fill of a limb in [out, out + n_limbs) that no queued op reads is dead as well
void cancel_lazy(const u64* out, size_t n_limbs) {
  if (g_lazy.empty() || n_limbs == 0) return;
  const u64* end = out + n_limbs * ctx().N;
  for (auto it = g_lazy.lower_bound(out); it != g_lazy.end() && *it < end;) {
    bool read = false;
    for (const acehip_hw_op& o : g_hwq) {
      read |= o.op != ACEHIP_HW_ZERO && o.a == *it;
      read |= (o.op == ACEHIP_HW_ADD || o.op == ACEHIP_HW_SUB || o.op == ACEHIP_HW_MUL || o.op == ACEHIP_HW_MULADD) && o.b == (const void*)*it;
      read |= o.op == ACEHIP_HW_MULADD && o.res == *it;
    }
    if (read) {
      ++it;
    } else {
      it = g_lazy.erase(it);
      g_lazy_stats.dropped++;
    }
  }
}
inline bool touches(const Touch* touch, size_t n_touch, const u64* limb, size_t N) {
  for (size_t i = 0; i < n_touch; ++i)
    if (touch[i].p && limb + N > (const u64*)touch[i].p && limb < (const u64*)touch[i].p + touch[i].words) return true;
  return false;
}
inline bool in_limbo(const u64* limb) {
  for (const auto& b : pool_limbo)
    if (limb >= b.first && limb < b.first + b.second) return true;
  return false;
}
// small pointer set for one pass over the queue (open addressing, cleared per use)
struct PtrSet {
  std::vector<const void*> slot;
  size_t mask = 0;
  void reset(size_t n) {
    size_t cap = 64;
    while (cap < 2 * n) cap <<= 1;
    slot.assign(cap, nullptr);
    mask = cap - 1;
  }
  bool insert(const void* p) {  // true: new
    for (size_t i = ((uintptr_t)p * 0x9E3779B97F4A7C15ull) >> 24;; ++i) {
      const void*& s = slot[i & mask];
      if (s == p) return false;
      if (s == nullptr) {
        s = p;
        return true;
      }
    }
  }
  bool has(const void* p) const {
    for (size_t i = ((uintptr_t)p * 0x9E3779B97F4A7C15ull) >> 24;; ++i) {
      const void* s = slot[i & mask];
      if (s == p) return true;
      if (s == nullptr) return false;
    }
  }
};
inline bool op_has_b(u32 op) { return op == ACEHIP_HW_ADD || op == ACEHIP_HW_SUB || op == ACEHIP_HW_MUL || op == ACEHIP_HW_MULADD; }
// touch: what the direct launch that follows reads or writes; defer = the list is complete (HIPCHK_T)
void lazy_meet_queue(const Touch* touch, size_t n_touch, bool defer) {
  const size_t N = ctx().N;
  // 1. deferred fills go back in front of the first queued op that names their limb
  if (!g_lazy.empty() && !g_hwq.empty()) {
    static thread_local std::vector<acehip_hw_op> out;
    out.clear();
    bool any = false;
    size_t k = 0;
    for (; k < g_hwq.size() && !g_lazy.empty(); ++k) {
      const acehip_hw_op& o = g_hwq[k];
      const void* named[3] = {o.res, o.op != ACEHIP_HW_ZERO ? o.a : nullptr, op_has_b(o.op) ? o.b : nullptr};
      for (const void* p : named) {
        if (!p) continue;
        const u64* lo = (const u64*)p - (N - 1);
        for (auto it = g_lazy.lower_bound(lo); it != g_lazy.end() && *it < (const u64*)p + N;) {
          // (an op that rewrites exactly this limb without reading it makes the deferred fill redundant; put in front of
          // a rotation the fill would also cut the list into more runs)
          const bool rewritten = o.res == *it && o.op != ACEHIP_HW_MULADD && (o.op == ACEHIP_HW_ZERO || o.a != *it) &&
                                 !(op_has_b(o.op) && o.b == (const void*)*it);
          if (!rewritten) out.push_back(acehip_hw_op{ACEHIP_HW_ZERO, 0, const_cast<u64*>(*it), nullptr, nullptr});
          it = g_lazy.erase(it);
          g_lazy_stats.met_consumer++;
          any = true;
        }
      }
      out.push_back(o);
    }
    if (any) {
      out.insert(out.end(), g_hwq.begin() + k, g_hwq.end());
      g_hwq.swap(out);
    }
  }
  // 2. the others: dead with their block, due now (the launch touches them / declares nothing), or kept waiting
  for (auto it = g_lazy.begin(); it != g_lazy.end();) {
    if (in_limbo(*it)) {
      g_lazy_stats.dropped++;
    } else if (!defer || touches(touch, n_touch, *it, N)) {
      g_hwq.push_back(acehip_hw_op{ACEHIP_HW_ZERO, 0, const_cast<u64*>(*it), nullptr, nullptr});
      g_lazy_stats.materialised++;
    } else {
      ++it;
      continue;
    }
    it = g_lazy.erase(it);
  }
  if (!defer) return;
  // 3. queued fills that nothing behind them names, and that the launch does not touch, start waiting
  size_t zeros = 0;
  for (const acehip_hw_op& o : g_hwq) zeros += o.op == ACEHIP_HW_ZERO;
  if (zeros == 0) return;
  static thread_local PtrSet later;
  later.reset(3 * g_hwq.size());
  bool any = false;
  for (size_t k = g_hwq.size(); k-- > 0 && zeros > 0;) {
    acehip_hw_op& o = g_hwq[k];
    if (o.op == ACEHIP_HW_ZERO) {
      --zeros;
      if (!later.has(o.res) && !touches(touch, n_touch, o.res, N)) {
        if (in_limbo(o.res)) {
          g_lazy_stats.dropped++;
        } else {
          g_lazy.insert(o.res);
          g_lazy_stats.deferred++;
          if (poison_on()) poison(o.res, N);  // (nothing queued names the limb and the launch does not touch it: safe to write now)
        }
        o.res = nullptr;  // taken out
        any = true;
        continue;
      }
    }
    later.insert(o.res);
    if (o.op != ACEHIP_HW_ZERO) later.insert(o.a);
    if (op_has_b(o.op)) later.insert(o.b);
  }
  if (any) {
    size_t w = 0;
    for (size_t k = 0; k < g_hwq.size(); ++k)
      if (g_hwq[k].res != nullptr) g_hwq[w++] = g_hwq[k];
    g_hwq.resize(w);
  }
}
// ---- keeping ops queued across declared launches (round 5) ----
// A direct launch used to take the WHOLE queue with it (every queued op had to be on the device before the launch, whose operands
// nobody knew).  With the operand list of HIPCHK_T the launch only needs the ops it depends on: walking the queue backwards,
// an op is SUBMITTED when it conflicts with the launch or with an op already marked (it writes a limb they read or write, or
// reads a limb they write: marked ops and the launch run BEFORE the ops that stay, so every such pair would be swapped); everything
// else STAYS queued, in program order, and runs with a later hand-over -- after the launch, which by construction neither reads nor
// writes anything those ops touch.  The tails of consecutive rotations (d0 + c0, gather, plaintext product, accumulation) then meet
// in ONE list, where the library sorts them into few long runs (api_hw_batch.cpp hw_stage_order) and accumulators stay in registers.
// Blocks the program freed while ops that name them are still queued stay in the pool's limbo (never reused, never given to the
// library as dead) until the last such op has been handed over.  Safety net: ACEHIP_POISON=1 overwrites the result limbs of kept
// pure overwrites with non-residues and compares what a launch touched with its declared list (check_declared).
// ACEHIP_HW_KEEP=0 restores the old behaviour (everything is submitted).
thread_local std::set<const u64*> g_keep_poisoned;  // result limbs of kept ops that hold poison right now (ACEHIP_POISON)
constexpr size_t kKeepMaxOps = 6000;
inline bool op_reads_res(u32 op) { return op == ACEHIP_HW_MULADD; }
// g_hwq -> (g_hwq = the ops to submit now, kept = the ops that stay), both in program order
void split_for_launch(const Touch* touch, size_t n_touch, std::vector<acehip_hw_op>& kept) {
  const size_t N = ctx().N, n = g_hwq.size();
  static thread_local PtrSet s_read, s_write;
  static thread_local std::vector<char> sub;
  sub.assign(n, 0);
  size_t n_sub = 0;
  // one interval around everything the launch names, one around everything the submitted ops name: most queued ops lie outside both
  const u64 *t_lo = (const u64*)~(uintptr_t)0, *t_hi = nullptr, *s_lo = (const u64*)~(uintptr_t)0, *s_hi = nullptr;
  for (size_t i = 0; i < n_touch; ++i)
    if (touch[i].p && touch[i].words) {
      t_lo = std::min(t_lo, (const u64*)touch[i].p);
      t_hi = std::max(t_hi, (const u64*)touch[i].p + touch[i].words);
    }
  auto hits_launch = [&](const u64* p) { return p + N > t_lo && p < t_hi && touches(touch, n_touch, p, N); };
  auto hits_launch_w = [&](const u64* p) {  // ... a range the launch may WRITE (a queued READ of a range the launch only reads is no conflict)
    if (!(p + N > t_lo && p < t_hi)) return false;
    for (size_t i = 0; i < n_touch; ++i)
      if (touch[i].p && !touch[i].ro && p + N > (const u64*)touch[i].p && p < (const u64*)touch[i].p + touch[i].words) return true;
    return false;
  };
  auto near_s = [&](const u64* p) { return p >= s_lo && p < s_hi; };
  for (size_t k = n; k-- > 0;) {
    const acehip_hw_op& o = g_hwq[k];
    const u64* a = o.op != ACEHIP_HW_ZERO ? o.a : nullptr;
    const u64* b = op_has_b(o.op) ? (const u64*)o.b : nullptr;
    // against the launch: every declared range counts as read AND written, unless it is marked read-only
    bool hit = hits_launch(o.res) || (a && hits_launch_w(a)) || (b && hits_launch_w(b));
    // against the ops behind it that are submitted
    if (!hit && n_sub)
      hit = (near_s(o.res) && (s_read.has(o.res) || s_write.has(o.res))) || (a && near_s(a) && s_write.has(a)) || (b && near_s(b) && s_write.has(b));
    if (!hit) continue;
    if (n_sub == 0) {
      s_read.reset(3 * n);
      s_write.reset(n);
    }
    sub[k] = 1;
    ++n_sub;
    s_write.insert(o.res);
    if (a) s_read.insert(a);
    if (b) s_read.insert(b);
    if (op_reads_res(o.op)) s_read.insert(o.res);
    for (const u64* p : {(const u64*)o.res, a, b})
      if (p) {
        s_lo = std::min(s_lo, p);
        s_hi = std::max(s_hi, p + 1);
      }
  }
  if (n_sub == n) return;
  kept.reserve(n - n_sub);
  if (n_sub == 0) {
    kept.swap(g_hwq);
    return;
  }
  size_t w = 0;
  for (size_t k = 0; k < n; ++k) {
    if (sub[k]) g_hwq[w++] = g_hwq[k];
    else kept.push_back(g_hwq[k]);
  }
  g_hwq.resize(w);
}
// limbo blocks that no op of `kept` names are released / may be declared dead; the others stay (returns how many stay)
size_t limbo_split(const std::vector<acehip_hw_op>& kept, std::vector<LimboBlock>& stay) {
  stay.clear();
  if (kept.empty() || pool_limbo.empty()) return 0;
  // blocks sorted by address; every pointer the kept ops name marks the block it lies in
  static thread_local std::vector<char> pinned;
  std::sort(pool_limbo.begin(), pool_limbo.end(), [](const LimboBlock& x, const LimboBlock& y) { return x.first < y.first; });
  pinned.assign(pool_limbo.size(), 0);
  const u64 *lo = pool_limbo.front().first, *hi = pool_limbo.back().first + pool_limbo.back().second;
  size_t n_pinned = 0;
  auto mark = [&](const u64* p) {
    if (p < lo || p >= hi) return;
    size_t a = 0, b = pool_limbo.size();
    while (a < b) {  // last block that starts at or below p
      const size_t m = (a + b) / 2;
      if (pool_limbo[m].first <= p) a = m + 1;
      else b = m;
    }
    if (a > 0 && p < pool_limbo[a - 1].first + pool_limbo[a - 1].second && !pinned[a - 1]) {
      pinned[a - 1] = 1;
      ++n_pinned;
    }
  };
  for (const acehip_hw_op& o : kept) {
    mark(o.res);
    if (o.op != ACEHIP_HW_ZERO) mark(o.a);
    if (op_has_b(o.op)) mark((const u64*)o.b);
    if (n_pinned == pool_limbo.size()) break;
  }
  size_t w = 0;
  for (size_t i = 0; i < pool_limbo.size(); ++i) {
    if (pinned[i]) stay.push_back(pool_limbo[i]);
    else pool_limbo[w++] = pool_limbo[i];
  }
  pool_limbo.resize(w);
  return stay.size();
}
// hands g_hwq (all of it) to the library
void submit_all() {
  if (reload_diag_on() && !g_hwq.empty()) {
    ReloadStats& rl = g_reload;
    const size_t idx = g_hwq_stats.flushes + 1;
    std::set<const u64*> seen;  // first reference inside this queue only
    auto load = [&](const void* p, u32 by) {
      const u64* x = (const u64*)p;
      if (!x || !seen.insert(x).second) return;
      rl.loads++;
      auto it = rl.stored.find(x);
      if (it == rl.stored.end()) return;
      rl.reloads++;
      const size_t d = idx - it->second.first;
      rl.reload_dist[d <= 1 ? 0 : d == 2 ? 1 : d <= 4 ? 2 : d <= 8 ? 3 : d <= 16 ? 4 : 5]++;
      rl.pair[d <= 2 ? 0 : d <= 16 ? 1 : 2][it->second.second][by]++;
    };
    for (const auto& o : g_hwq) {
      if (o.op != ACEHIP_HW_ZERO) load(o.a, o.op);
      if (op_has_b(o.op)) load(o.b, o.op);
      if (o.op == ACEHIP_HW_MULADD) load(o.res, o.op);
      seen.insert(o.res);
    }
    for (const auto& o : g_hwq) rl.stored[o.res] = {idx, o.op};
  }
  if (g_hwq.empty()) {
    limbo_release();
    return;
  }
  if (ctx().profile) {
    HwqStats& st = g_hwq_stats;
    st.flushes++;
    st.ops += g_hwq.size();
    for (const auto& o : g_hwq) st.by_kind[o.op]++;
    size_t b = 0, lim = 1;
    while (b < 7 && g_hwq.size() > lim) {
      ++b;
      lim *= 4;
    }
    st.hist[b]++;
    std::vector<const u64*> res;
    for (const auto& o : g_hwq) res.push_back(o.res);
    std::sort(res.begin(), res.end());
    res.erase(std::unique(res.begin(), res.end()), res.end());
    st.res_limbs += res.size();
    for (const u64* r : res)
      for (const auto& b : pool_limbo)
        if (r >= b.first && r < b.first + b.second) {
          ++st.res_limbs_freed;
          break;
        }
  }
  // blocks the program freed while these ops were queued (pool_limbo) are named by nothing but the queue: what the ops
  // leave in them is never read, which the library uses to skip stores and whole ops (ACEHIP_HW_DISCARD=0: plain list)
  static const bool discard = getenv("ACEHIP_HW_DISCARD") == nullptr || atoi(getenv("ACEHIP_HW_DISCARD")) != 0;
  static thread_local std::vector<acehip_hw_range> dead;
  dead.clear();
  if (discard)
    for (const auto& b : pool_limbo) dead.push_back(acehip_hw_range{b.first, b.second});
  const int rc = acehip_hw_batch_discard(ctx().hip, g_hwq.data(), g_hwq.size(), dead.data(), dead.size(), nullptr);
  if (!g_keep_poisoned.empty())
    for (const auto& o : g_hwq) g_keep_poisoned.erase(o.res);
  g_hwq.clear();
  RT_ASSERT(rc >= 0, "acehip_hw_batch failed: %s", acehip_last_error());
  limbo_release();
}
// the hand-over in front of a direct launch: all of the queue, or -- when the launch declares its operands -- the part it needs
void queue_submit(const Touch* touch, size_t n_touch, bool defer) {
  static const bool lazy_on = getenv("ACEHIP_LAZY_ZERO") == nullptr || atoi(getenv("ACEHIP_LAZY_ZERO")) != 0;
  static const bool keep_on = getenv("ACEHIP_HW_KEEP") == nullptr || atoi(getenv("ACEHIP_HW_KEEP")) != 0;
  // (limb-sharded execution: a deferred fill does not remember which rank owns its limb, so nothing is deferred there)
  // (with ops kept queued a zero fill that nothing needs yet simply stays in the queue like any other op -- with its owner, so also
  // under limb-sharded execution; the separate set of deferred fills is what ACEHIP_HW_KEEP=0 falls back to)
  lazy_meet_queue(touch, n_touch, defer && lazy_on && !keep_on && ctx().shard_world <= 1);
  if (reload_diag_on()) {
    ReloadStats& rl = g_reload;
    const size_t Nw = ctx().N;
    if (!defer) {
      rl.cleared += !rl.stored.empty();
      rl.stored.clear();
    } else {
      for (size_t i = 0; i < n_touch; ++i) {
        if (!touch[i].p || !touch[i].words) continue;
        const u64* lo = (const u64*)touch[i].p;
        for (auto it = rl.stored.lower_bound(lo - (Nw - 1)); it != rl.stored.end() && it->first < lo + touch[i].words;) {
          it = rl.stored.erase(it);
          rl.erased_by_touch++;
        }
      }
    }
  }
  static thread_local std::vector<acehip_hw_op> kept;
  static thread_local std::vector<LimboBlock> stay;
  kept.clear();
  stay.clear();
  if (defer && keep_on && !g_hwq.empty()) {
    g_keep_stats.launches++;
    // (a bound on what waits: the analysis is linear in the queue per launch, and pinned limbo blocks are arena memory)
    size_t limbo_words = 0;
    for (const auto& b : pool_limbo) limbo_words += b.second;
    // ... and on how long: an op that has waited through keep_run launches is re-examined by none of the following ones
    // (measured, profiles/r05h_*: per-limb kernel time of a 12-image batch 2.50 s without keeping, 2.35 / 2.30 / 2.27 s with runs of
    //  8 / 16 / unbounded; host time per launch grows with the run -- unbounded costs a single-image stream 70 % more wall time, a run
    //  of 8 costs it 2 % -- so single images use 8 and batches, whose launches are long enough to hide the host, 16)
    static const u32 keep_run_env = [] { const char* e = getenv("ACEHIP_HW_KEEP_RUN"); return e && atoi(e) > 0 ? (u32)atoi(e) : 0u; }();
    const u32 keep_run = keep_run_env ? keep_run_env : (batch_size() > 1 ? 16u : 8u);
    static thread_local u32 run = 0;
    if (g_hwq.size() > kKeepMaxOps || (g_arena.words && limbo_words > g_arena.words / 4) || ++run > keep_run) {
      g_keep_stats.forced_full++;
      run = 0;
    } else {
      const bool prof = ctx().profile;
      const double t0 = prof ? wall_s() : 0;
      split_for_launch(touch, n_touch, kept);
      const double t1 = prof ? wall_s() : 0;
      g_keep_stats.ops_kept += kept.size();
      g_keep_stats.ops_submitted += g_hwq.size();
      g_keep_stats.kept_some += !kept.empty();
      g_keep_stats.limbo_pinned += limbo_split(kept, stay);
      if (prof) {
        g_keep_stats.t_split += t1 - t0;
        g_keep_stats.t_limbo += wall_s() - t1;
      }
    }
  }
  {
    const bool prof = ctx().profile;
    const double t0 = prof ? wall_s() : 0;
    submit_all();
    if (prof) g_keep_stats.t_submit += wall_s() - t0;
  }
  if (!kept.empty()) {
    g_hwq.swap(kept);
    pool_limbo.insert(pool_limbo.end(), stay.begin(), stay.end());
    if (poison_on()) {  // the first op that names a limb overwrites it without reading it: until then the limb holds garbage on purpose
      static thread_local PtrSet seen;
      seen.reset(3 * g_hwq.size());
      const size_t N = ctx().N;
      for (const acehip_hw_op& o : g_hwq) {
        const u64* a = o.op != ACEHIP_HW_ZERO ? o.a : nullptr;
        const u64* b = op_has_b(o.op) ? (const u64*)o.b : nullptr;
        if (a) seen.insert(a);
        if (b) seen.insert(b);
        const bool reads_res = op_reads_res(o.op);
        if (seen.insert(o.res) && !reads_res && g_keep_poisoned.insert(o.res).second) poison(o.res, N);
      }
    }
  }
}
}  // namespace
void hw_flush() {
  pending_flush();  // it was called before everything queued since (there is nothing: see hw_queue) and after the rest
  g_muc.forget();   // some other device work follows, operands unknown: the raised digits may go stale
  queue_submit();
}
static void note_site(const char* file, int line);
// ACEHIP_POISON=1: what the launch just issued really touched (as reported by the library) against what its site declared
void check_declared(const char* file, int line, const Touch* touch, size_t n) {
  if (!poison_on()) return;
  static thread_local std::vector<const void*> ptrs(64);
  static thread_local std::vector<size_t> words(64);
  const size_t got = acehip_debug_touches(1, ptrs.data(), words.data(), ptrs.size());
  for (size_t i = 0; i < got && i < ptrs.size(); ++i) {
    const u64* p = (const u64*)ptrs[i];
    const u64* e = p + words[i];
    {  // only memory of this thread's pool can hold a deferred fill
      std::lock_guard<std::mutex> lk(pool_mu);
      auto it = pool_live.upper_bound(const_cast<u64*>(p));
      if (it == pool_live.begin()) continue;
      --it;
      if (p >= it->first + it->second.words) continue;
    }
    // covered by the declared ranges?  (ranges may split the operand between them)
    const u64* at = p;
    bool progress = true;
    while (at < e && progress) {
      progress = false;
      for (size_t k = 0; k < n; ++k) {
        const u64* tp = (const u64*)touch[k].p;
        if (tp && tp <= at && at < tp + touch[k].words) {
          at = tp + touch[k].words;
          progress = true;
        }
      }
    }
    RT_ASSERT(at >= e, "%s:%d: the launch touches %zu words at %p that its operand list does not declare (ACEHIP_POISON)", file, line,
              (size_t)(e - at), (const void*)at);
  }
}
void hw_flush_touching(const char* file, int line, const Touch* touch, size_t n) {
  note_site(file, line);
  if (poison_on()) (void)acehip_debug_touches(1, nullptr, nullptr, 0);  // start a fresh log for the launch that follows
  pending_flush();
  for (size_t i = 0; i < n; ++i)
    if (!touch[i].ro) g_muc.written((const u64*)touch[i].p, touch[i].words);  // (any operand not marked read-only may be an output)
  queue_submit(touch, n, true);
}
void hw_cancel_fills(const u64* out, size_t n_limbs) { cancel_fills(out, n_limbs); }
// ACEHIP_PROFILE: which call sites hand over how many queued limb-ops (finds what cuts accumulation chains short)
namespace {
thread_local std::map<std::pair<std::string, int>, std::pair<size_t, size_t>> g_flush_sites;
}
static void note_site(const char* file, int line) {
  if (g_ctx != nullptr && g_ctx->profile && !g_hwq.empty()) {
    auto& e = g_flush_sites[{file, line}];
    e.first++;
    e.second += g_hwq.size();
  }
}
void hw_flush_site(const char* file, int line) {
  note_site(file, line);
  hw_flush();
}
void hw_flush_sites_print() {
  for (auto& kv : g_flush_sites) {
    const char* f = strrchr(kv.first.first.c_str(), '/');
    printf("[ACEHIP] flush site %s:%d: %zu flushes, %zu limb-ops\n", f ? f + 1 : kv.first.first.c_str(), kv.first.second, kv.second.first,
           kv.second.second);
  }
}
bool hw_queue_empty() { return g_hwq.empty(); }
void hw_pending_flush() { pending_flush(); }
void hw_queue(u32 op, u32 prime_gi, u64* res, const u64* a, const void* b, size_t n_limbs) {
  const size_t N = ctx().N;
  pending_flush();
  // a batch of images shares what lies outside the arena: an op that covers the whole batch must not write there
  RT_ASSERT(g_mode_nrep == 1 || in_arena(res), "a per-limb op of an image batch writes memory all images share (a weight plaintext?)");
  g_muc.written(res, n_limbs * N);
  for (size_t l = 0; l < n_limbs; ++l)
    g_hwq.push_back(acehip_hw_op{op, prime_gi, res + l * N, a ? a + l * N : nullptr,
                                 b ? (const void*)((const u64*)b + l * N) : nullptr});
  if (g_hwq.size() >= 8192) {  // (nothing else follows: no operands to declare, the raised digits stay valid)
    note_site(__FILE__, __LINE__);
    queue_submit();
  }
}
static inline u32 limb_gi(u32 pos, u32 level) { return pos < level ? pos : ctx().L + (pos - level); }
void q_ew(u32 op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n) {
  const size_t N = ctx().N;
  for (u32 p = pos0; p < pos0 + n; ++p) hw_queue(op, limb_gi(p, level), r + p * N, a + p * N, b + p * N);
}
void q_scalars(u32 op, u64* r, const u64* a, const u64* scalars, u32 level, u32 pos0, u32 n) {
  const size_t N = ctx().N;
  for (u32 i = 0; i < n; ++i)
    hw_queue(op, limb_gi(pos0 + i, level), r + (pos0 + i) * N, a + (pos0 + i) * N, (const void*)(uintptr_t)scalars[i]);
}
void q_rotate(u64* r, const u64* a, const uint32_t* perm, u32 level, u32 pos0, u32 n) {
  const size_t N = ctx().N;
  for (u32 p = pos0; p < pos0 + n; ++p) hw_queue(ACEHIP_HW_ROTATE, limb_gi(p, level), r + p * N, a + p * N, perm);
}
// zero fill / copy of whole limbs through the queue (anything else goes the direct way).  nq: the block holds nq q-limbs
// (primes 0..nq-1) followed by p-limbs -- which names the owner of every limb under limb-sharded execution; NQ_ANY: not a
// polynomial of the chain (every rank runs the op)
static inline u32 map_gi(size_t i, u32 nq) { return nq == NQ_ANY ? ACEHIP_HW_ANY_RANK : (i < nq ? (u32)i : ctx().L + (u32)(i - nq)); }
void fill_zero(u64* p, size_t words, u32 nq) {
  const size_t N = ctx().N;
  if (words % N == 0) {
    for (size_t i = 0; i < words / N; ++i) hw_queue(ACEHIP_HW_ZERO, map_gi(i, nq), p + i * N, nullptr, nullptr);
  } else {
    HIPCHK(acehip_fill(ctx().hip, p, 0, words * sizeof(u64), nullptr));
  }
}
void copy_limbs(u64* dst, const u64* src, size_t words, u32 nq, u32 first) {
  const size_t N = ctx().N;
  if (dst == src) return;
  const size_t gap = dst < src ? src - dst : dst - src;
  if (words % N == 0 && gap >= words) {
    for (size_t i = 0; i < words / N; ++i) hw_queue(ACEHIP_HW_COPY, map_gi(first + i, nq), dst + i * N, src + i * N, nullptr);
  } else {
    HIPCHK(acehip_copy(ctx().hip, dst, src, words * sizeof(u64), nullptr));
  }
}

u64* dalloc(size_t words, bool zero, u32 nq) {
  if (g_shared_scope > 0) return shared_alloc(words, zero);
  if (words == 0) words = 1;
  int kind = g_alloc_uniform > 0 ? BK_UNIFORM : BK_ARENA;
  u64* p = nullptr;
  {
    std::lock_guard<std::mutex> lk(pool_mu);
    auto it = pool_free[kind].find(words);
    if (it != pool_free[kind].end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
    }
  }
  if (!p && kind == BK_ARENA) {
    p = arena_take(words);
    if (!p) {  // merge what waits in the exact-size lists back into extents and look again
      std::lock_guard<std::mutex> lk(pool_mu);
      for (auto& kv : pool_free[BK_ARENA])
        for (u64* q : kv.second) arena_give(q, kv.first);
      pool_free[BK_ARENA].clear();
      p = arena_take(words);
    }
    if (!p && (!g_hwq.empty() || !pool_limbo.empty())) {
      // blocks freed while ops that name them wait in the queue are pinned (pool_limbo): hand everything over and look again
      queue_submit();
      std::lock_guard<std::mutex> lk(pool_mu);
      auto it = pool_free[BK_ARENA].find(words);
      if (it != pool_free[BK_ARENA].end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
      } else {
        for (auto& kv : pool_free[BK_ARENA])
          for (u64* q : kv.second) arena_give(q, kv.first);
        pool_free[BK_ARENA].clear();
        p = arena_take(words);
      }
    }
    if (!p && g_arena.nrep == 1) {  // one replica: a separate allocation serves as well; reuse those first
      std::lock_guard<std::mutex> lk(pool_mu);
      auto it = pool_free[BK_UNIFORM].find(words);
      if (it != pool_free[BK_UNIFORM].end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
        kind = BK_UNIFORM;
      }
    }
    if (!p) {
      // the slab is full.  With one replica a block outside it works just as well (nothing moves with a replica);
      // with several there is no way out
      RT_ASSERT(g_arena.nrep == 1, "pool arena exhausted (%zu MB per replica): raise ACEHIP_ARENA_MB", g_arena.words * 8 >> 20);
      if (!g_arena.exhausted_warned && ctx().profile)
        fprintf(stderr, "[ACEHIP] pool arena of %zu MB exhausted: further blocks are separate allocations\n", g_arena.words * 8 >> 20);
      g_arena.exhausted_warned = true;
      kind = BK_UNIFORM;
    }
  }
  if (!p) {
    p = (u64*)acehip_malloc(words * sizeof(u64));
    RT_ASSERT(p != nullptr, "device allocation of %zu bytes failed: %s", words * sizeof(u64), acehip_last_error());
  }
  {
    std::lock_guard<std::mutex> lk(pool_mu);
    pool_live[p] = LiveBlock{words, kind};
    pool_live_bytes += words * sizeof(u64);
    if (kind == BK_ARENA) {
      g_arena.live += granules(words);
      g_arena.live_peak = std::max(g_arena.live_peak, g_arena.live);
    }
  }
  if (zero) fill_zero(p, words, nq);
  return p;
}

// words of the live block of this thread's pool that starts at p (0: not such a block)
static size_t pool_block_words(u64* p) {
  std::lock_guard<std::mutex> lk(pool_mu);
  auto it = pool_live.find(p);
  return it == pool_live.end() ? 0 : it->second.words;
}
// every image has its own copy of the block (false: one copy shared by the batch, or not a pool block)
bool block_is_replicated(const u64* p) {
  std::lock_guard<std::mutex> lk(pool_mu);
  auto it = pool_live.find(const_cast<u64*>(p));
  return it != pool_live.end() && it->second.kind == BK_ARENA;
}
bool block_is_uniform(const u64* p) {
  std::lock_guard<std::mutex> lk(pool_mu);
  auto it = pool_live.find(const_cast<u64*>(p));
  return it != pool_live.end() && it->second.kind == BK_UNIFORM;
}

void dfree(u64* p) {
  if (!p) return;
  // a Mod_down / Rescale that is still held back reads or writes this block: issue it first (the queue and the deferred
  // fills it depends on are still intact; a freed block would be handed over as dead memory)
  if (g_pend.kind) {
    const size_t N = ctx().N;
    const size_t out_w = (size_t)(g_pend.kind == 1 ? g_pend.level : g_pend.level - 1) * N;
    const size_t in_w = (size_t)(g_pend.kind == 1 ? g_pend.level + ctx().K : g_pend.level) * N;
    size_t bw = 0;
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      auto it = pool_live.find(p);
      if (it != pool_live.end()) bw = it->second.words;
    }
    const u64* e = p + bw;
    if (bw && ((g_pend.out < e && g_pend.out + out_w > p) || (g_pend.in < e && g_pend.in + in_w > p))) pending_flush();
  }
  if (p == g_muc.lent) {  // the caller is done with the digit it was handed last: keep it, release a valueless block instead
    g_muc.lent = nullptr;
    if (g_muc.ok)
      for (u32 s = 0; s < g_muc.n_blk; ++s)
        if (g_muc.holds[s] < 0) {
          std::swap(p, g_muc.blk[s]);
          g_muc.holds[s] = g_muc.lent_digit;
          break;
        }
  }
  {
    // (shared_free hands the queue over, which takes pool_mu itself: look the block up first, call it unlocked)
    bool mine;
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      mine = pool_live.find(p) != pool_live.end();
    }
    if (!mine) {
      const bool shared = shared_free(p);
      RT_ASSERT(shared, "free of a pointer the pool does not own");
      return;
    }
  }
  std::lock_guard<std::mutex> lk(pool_mu);
  auto it = pool_live.find(p);
  const size_t words = it->second.words;
  const int kind = it->second.kind;
  if (g_muc.ok && g_muc.src >= p && g_muc.src < p + words) g_muc.forget();
  pool_live_bytes -= words * sizeof(u64);
  if (kind == BK_ARENA) g_arena.live -= granules(words);
  if (g_hwq.empty()) {
    if (poison_on()) poison(p, words);
    pool_free[kind][words].push_back(p);
    for (auto z = g_lazy.lower_bound(p); z != g_lazy.end() && *z < p + words;) {  // fills nobody waits for any more
      z = g_lazy.erase(z);
      g_lazy_stats.dropped++;
    }
  } else {
    pool_limbo.push_back(LimboBlock{p, words, kind});  // queued ops may still name it (and deferred fills of it: queue_submit)
  }
  pool_live.erase(it);
}

// a secondary thread gives back what it owns (the primary thread does this in Finalize_context)
void thread_release() {
  if (g_ctx == nullptr || !g_ctx->secondary) return;
  sync();
  ev::clear_monomial_cache();
  stage_release();
  pool_release_all();
  acehip_ctx_destroy(g_ctx->hip);
  delete g_ctx;
  g_ctx = nullptr;
}

void pool_release_all() {
  std::lock_guard<std::mutex> lk(pool_mu);
  g_lazy.clear();
  g_muc = ModupCache{};
  for (auto& kv : pool_free[BK_UNIFORM])
    for (u64* p : kv.second) acehip_free(p);
  for (auto& b : pool_limbo)
    if (b.kind == BK_UNIFORM) acehip_free(b.first);
  for (auto& kv : pool_live)
    if (kv.second.kind == BK_UNIFORM) acehip_free(kv.first);
  pool_free[0].clear();
  pool_free[1].clear();
  pool_limbo.clear();
  pool_live.clear();
  pool_live_bytes = 0;
  if (g_arena.base != nullptr) {  // (the context that names it goes away right after: acehip_ctx_destroy)
    acehip_free(g_arena.base);
    g_arena = Arena{};
  }
  g_mode_rep0 = 0;
  g_mode_nrep = 1;
  g_image = 0;
}
size_t pool_bytes_in_use() { return pool_live_bytes; }

void sync() { HIPCHK(acehip_stream_sync(nullptr)); }
double wall_s() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

// ---- POLYNOMIAL helpers ----
void poly_alloc(POLYNOMIAL* p, u32 N, size_t nq, size_t np, bool zero) {
  p->_ring_degree = N;
  p->_num_primes = nq;
  p->_num_primes_p = np;
  p->_num_alloc_primes = nq + np;
  p->_is_ntt = false;
  p->_data = (int64_t*)dalloc((size_t)(nq + np) * N, zero, (u32)nq);
}
void poly_free(POLYNOMIAL* p) {
  if (p->_data) {
    dfree((u64*)p->_data);
    p->_data = nullptr;
  }
  p->_num_alloc_primes = 0;
}
// Init_poly polynomial.h:331-348: allocate, or reuse + zero-fill
void poly_init_like(POLYNOMIAL* res, POLYNOMIAL* like) {
  const u32 N = like->_ring_degree;
  const size_t nq = like->_num_primes, np = like->_num_primes_p;
  if (res->_data == nullptr) {
    poly_alloc(res, N, nq, np);
  } else {
    if (res->_num_alloc_primes * (size_t)res->_ring_degree < (nq + np) * (size_t)N) {
      poly_free(res);
      poly_alloc(res, N, nq, np);
    } else {
      fill_zero((u64*)res->_data, res->_num_alloc_primes * (size_t)res->_ring_degree, (u32)(res->_num_alloc_primes - res->_num_primes_p));
      res->_ring_degree = N;
      res->_num_primes = nq;
      res->_num_primes_p = np;
      if (res->_num_alloc_primes == 0) res->_num_alloc_primes = nq + np;
      res->_is_ntt = false;
    }
  }
}
u64* q_limbs(POLYNOMIAL* p) { return (u64*)p->_data; }
u64* p_limbs(POLYNOMIAL* p) { return (u64*)p->_data + (p->_num_alloc_primes - p->_num_primes_p) * (size_t)p->_ring_degree; }

// Copy_polynomial (polynomial.h): copies q part and p part
void poly_copy(POLYNOMIAL* res, POLYNOMIAL* src) {
  if (res == src) return;
  const size_t N = src->_ring_degree;
  RT_ASSERT(res->_data && res->_num_alloc_primes >= src->_num_primes + src->_num_primes_p, "Copy_poly: result too small");
  res->_num_primes = src->_num_primes;
  res->_num_primes_p = src->_num_primes_p;
  res->_is_ntt = src->_is_ntt;
  if (src->_num_primes) copy_limbs(q_limbs(res), q_limbs(src), src->_num_primes * N, (u32)src->_num_primes);
  if (src->_num_primes_p) copy_limbs(p_limbs(res), p_limbs(src), src->_num_primes_p * N, 0);
}

void poly_ew(Op op, POLYNOMIAL* res, POLYNOMIAL* a, POLYNOMIAL* b, bool with_p) {
  const u32 kind = op == Op::Add ? ACEHIP_HW_ADD : op == Op::Sub ? ACEHIP_HW_SUB : op == Op::Mul ? ACEHIP_HW_MUL : ACEHIP_HW_MULADD;
  const u32 l = (u32)res->_num_primes;
  if (l) q_ew(kind, q_limbs(res), q_limbs(a), q_limbs(b), l, 0, l);
  // level 0: position j -> prime p_j
  if (with_p && res->_num_primes_p) q_ew(kind, p_limbs(res), p_limbs(a), p_limbs(b), 0, 0, (u32)res->_num_primes_p);
}

void poly_ntt(POLYNOMIAL* p, bool inverse) {
  RtmScope rtm(inverse ? RTM_INTT : RTM_NTT);
  Context& c = ctx();
  auto fn = inverse ? acehip_ntt_inverse : acehip_ntt_forward;
  const u32 l = (u32)p->_num_primes;
  if (l) HIPCHK(fn(c.hip, q_limbs(p), l, 0, l, nullptr));
  if (p->_num_primes_p) HIPCHK(fn(c.hip, p_limbs(p), 0, 0, (u32)p->_num_primes_p, nullptr));
  p->_is_ntt = !inverse;
}

void poly_rotate(POLYNOMIAL* res, POLYNOMIAL* a, u32 auto_idx) {
  Context& c = ctx();
  const uint32_t* perm = acehip_auto_order(c.hip, auto_idx);
  RT_ASSERT(perm != nullptr, "automorphism table: %s", acehip_last_error());
  const u32 l = (u32)a->_num_primes;
  if (l) q_rotate(q_limbs(res), q_limbs(a), perm, l, 0, l);
  if (a->_num_primes_p) q_rotate(p_limbs(res), p_limbs(a), perm, 0, 0, (u32)a->_num_primes_p);
  res->_is_ntt = a->_is_ntt;
}

// small signed values -> residues on every limb of p (q then p)
void poly_from_small(POLYNOMIAL* p, const std::vector<int64_t>& vals) {
  Context& c = ctx();
  u64* tmp = dalloc(c.N, false);
  HIPCHK(acehip_upload(c.hip, tmp, vals.data(), c.N * 8, nullptr));
  const u32 l = (u32)p->_num_primes;
  if (l) HIPCHK(acehip_values_to_rns(c.hip, q_limbs(p), (const int64_t*)tmp, l, 0, l, nullptr));
  if (p->_num_primes_p) HIPCHK(acehip_values_to_rns(c.hip, p_limbs(p), (const int64_t*)tmp, 0, 0, (u32)p->_num_primes_p, nullptr));
  dfree(tmp);
  p->_is_ntt = false;
}

static u32 gi_of(MODULUS* m) { return m->_gi; }

}  // namespace rt

using namespace rt;

extern "C" {

// ---- context accessors (context.c:140-160) ----
uint32_t Degree() { return ctx().N; }
double Get_default_sc() { return ctx().sf; }
size_t Get_q_parts() { return ctx().dnum; }
size_t Get_p_cnt() { return ctx().K; }
MODULUS* Q_modulus() { return ctx().qmod.data(); }
MODULUS* P_modulus() { return ctx().pmod.data(); }

// ---- poly_eval.h:29-113 ----
POLY Alloc_poly(uint32_t degree, size_t q_primes, bool extend_p) {
  RT_ASSERT(q_primes > 0, "Alloc_poly: q primes should not be NULL");
  POLY p = (POLY)malloc(sizeof(POLYNOMIAL));
  poly_alloc(p, degree, q_primes, extend_p ? ctx().K : 0);
  p->_is_ntt = true;  // "hard code for now, ntt should be set by compiler" (poly_eval.h:34-35)
  return p;
}
void Free_poly_data(POLY poly) { poly_free(poly); }
void Free_poly(POLY poly) {
  poly_free(poly);
  free(poly);
}
void Copy_poly(POLY res, POLY poly) {
  RtmScope rtm(RTM_COPY_POLY, false);
  poly_copy(res, poly);
}
void Set_coeffs(POLY dst, uint32_t level, uint32_t degree, int64_t* src) {
  int64_t* d = Coeffs(dst, level, degree);
  if (d == src) return;  // generated code does self-copies (resnet20 .inc:1546)
  const size_t nq = dst->_num_alloc_primes - dst->_num_primes_p;
  copy_limbs((u64*)d, (const u64*)src, degree, (u32)nq, level);
}
size_t Num_decomp(POLY poly) { return acehip_num_decomp(ctx().hip, (uint32_t)poly->_num_primes); }

// ---- poly_arith.c:14-56: one limb per call, modulus = Q_modulus()+i or P_modulus()+i ----
int64_t* Hw_modadd(int64_t* res, int64_t* a, int64_t* b, MODULUS* m, uint32_t degree) {
  RtmScope rtm(RTM_HW_ADD, false);
  hw_queue(ACEHIP_HW_ADD, m->_gi, (u64*)res, (const u64*)a, b);
  return res + degree;
}
int64_t* Hw_modmul(int64_t* res, int64_t* a, int64_t* b, MODULUS* m, uint32_t degree) {
  RtmScope rtm(RTM_HW_MUL, false);
  hw_queue(ACEHIP_HW_MUL, m->_gi, (u64*)res, (const u64*)a, b);
  return res + degree;
}
int64_t* Hw_rotate(int64_t* res, int64_t* a, int64_t* rot_precomp, MODULUS* m, uint32_t degree) {
  RtmScope rtm(RTM_HW_ROT, false);
  if (res == a) {  // the reference loop would read overwritten data too; keep it well defined
    u64* tmp = dalloc(degree, false);
    hw_queue(ACEHIP_HW_COPY, m->_gi, tmp, (const u64*)a, nullptr);
    HIPCHK(acehip_hw_rotate(ctx().hip, (u64*)res, tmp, (const uint32_t*)rot_precomp, m->_gi, nullptr));
    dfree(tmp);
  } else {
    hw_queue(ACEHIP_HW_ROTATE, m->_gi, (u64*)res, (const u64*)a, rot_precomp);  // one limb: the table is not advanced
  }
  return res + degree;
}

// ---- poly_eval.c:11-49 ----
POLY Decomp(POLY res, POLY poly, uint32_t q_part_idx) {
  RtmScope rtm(RTM_DECOMP);
  Context& c = ctx();
  const u32 level = (u32)poly->_num_primes;
  const u32 start = c.alpha * q_part_idx;
  RT_ASSERT(q_part_idx < acehip_num_decomp(c.hip, level), "Decomp: part index out of range");
  const u32 n2 = std::min(c.alpha, level - start);
  if (res->_num_alloc_primes < n2) {  // Decompose_poly polynomial.c:860-866
    poly_free(res);
    poly_alloc(res, c.N, n2, 0);
  } else {
    res->_num_primes = n2;
    res->_num_primes_p = 0;
  }
  copy_limbs(q_limbs(res), q_limbs(poly) + (size_t)start * c.N, (size_t)n2 * c.N, level, start);  // = acehip_decomp, queued
  res->_is_ntt = poly->_is_ntt;
  return res;
}
POLY Mod_up(POLY new_poly, POLY old_poly, uint32_t q_part_idx) {
  RtmScope rtm(RTM_MOD_UP);
  Context& c = ctx();
  const u32 level = (u32)new_poly->_num_primes;  // Raise_rns_base_with_parts(.., Poly_level(new_poly), ..)
  RT_ASSERT(new_poly->_num_primes_p == c.K && new_poly->_num_alloc_primes - c.K == level,
            "raise_rns_base: result size not match");
  RT_ASSERT(old_poly->_is_ntt, "Mod_up: coefficient-domain input is not supported by the HIP path");
  HIPCHK(acehip_mod_up(c.hip, q_limbs(new_poly), q_limbs(old_poly), level, q_part_idx, nullptr));
  new_poly->_is_ntt = true;
  return new_poly;
}
POLY Decomp_modup(POLY res, POLY poly, uint32_t q_part_idx) {
  RtmScope rtm(RTM_DECOMP_MODUP);
  Context& c = ctx();
  const u32 level = (u32)poly->_num_primes;
  RT_ASSERT(res->_num_primes == level && res->_num_primes_p == c.K && res->_num_alloc_primes == level + c.K,
            "Decomp_modup: result must be allocated with Alloc_poly(degree, Poly_level(poly), 1)");
  RT_ASSERT(poly->_is_ntt, "Decomp_modup: coefficient-domain input is not supported by the HIP path");
  const u32 nd = acehip_num_decomp(c.hip, level);
  RT_ASSERT(q_part_idx < nd, "Decomp_modup: part index out of range");
  const size_t E = (size_t)(level + c.K) * c.N;
  const u64* src = q_limbs(poly);
  ModupCache& m = g_muc;
  auto slot_of = [&](int digit) {
    for (u32 s = 0; s < m.n_blk; ++s)
      if (m.holds[s] == digit) return (int)s;
    return -1;
  };
  // ACEHIP_MODUP_REUSE=0: raise the digits again for every Rotate() (digit 0 always misses), as the reference does
  static const bool reuse = getenv("ACEHIP_MODUP_REUSE") == nullptr || atoi(getenv("ACEHIP_MODUP_REUSE")) != 0;
  int slot = m.ok && m.src == src && m.level == level && m.blk_words == E && (reuse || q_part_idx != 0) ? slot_of((int)q_part_idx) : -1;
  if (slot < 0) {
    if (q_part_idx != 0 || nd > 8) {  // a digit asked for out of sequence
      HIPCHK(acehip_decomp_modup(c.hip, q_limbs(res), src, level, q_part_idx, nullptr));
      res->_is_ntt = true;
      return res;
    }
    m.forget();
    if (m.blk_words != E || m.n_blk != nd) {
      for (u32 d = 0; d < m.n_blk; ++d)
        if (m.blk[d]) dfree(m.blk[d]);
      for (u32 d = 0; d < 8; ++d) m.blk[d] = d < nd ? dalloc(E, false) : nullptr;
      m.n_blk = nd;
      m.blk_words = E;
    }
    // every digit in one go; operands: the nd cache blocks (rewritten completely) and the source's q-limbs
    for (u32 d = 0; d < nd; ++d) cancel_fills(m.blk[d], level + c.K);
    HIPCHK_T(acehip_modup_digits_to(c.hip, m.blk, src, level, nullptr), {src, (size_t)level * c.N}, {m.blk[0], E},
             {m.blk[1], nd > 1 ? E : 0}, {m.blk[2], nd > 2 ? E : 0}, {m.blk[3], nd > 3 ? E : 0}, {m.blk[4], nd > 4 ? E : 0},
             {m.blk[5], nd > 5 ? E : 0}, {m.blk[6], nd > 6 ? E : 0}, {m.blk[7], nd > 7 ? E : 0});
    m.src = src;
    m.level = level;
    m.ok = true;
    m.lent = nullptr;
    for (u32 d = 0; d < 8; ++d) m.holds[d] = d < nd ? (int)d : -1;
    m.n_raise++;
    g_lvl.modup[level < 72 ? level : 71]++;
    slot = 0;
  } else if (q_part_idx == 0) {
    m.n_reuse++;
    g_lvl.modup_reuse[level < 72 ? level : 71]++;
  }
  // hand the digit over by exchanging blocks: res keeps its size (asserted above: level + K limbs), the cache gets res's
  // old block, which queued ops may still read -- it is rewritten only by the next all-digit ModUp, a direct launch that
  // hands the queue over first.  Fills still queued for the old block are dead unless a queued op reads them.  If the old
  // block is the digit handed out last, untouched, it goes back into the cache as that digit.
  u64* old = (u64*)res->_data;
  if (pool_block_words(old) == E) {
    const bool comes_back = old == m.lent;
    const int back_digit = m.lent_digit;
    if (!comes_back) cancel_fills(old, level + c.K);
    res->_data = (int64_t*)m.blk[slot];
    m.blk[slot] = old;
    m.holds[slot] = comes_back ? back_digit : -1;
    m.lent = (u64*)res->_data;
    m.lent_digit = (int)q_part_idx;
  } else {  // res is not a pool block of its own (a view, foreign memory): copy
    copy_limbs(q_limbs(res), m.blk[slot], E, level);  // (queued: counts as a write of res, which is neither the source nor a digit)
  }
  res->_is_ntt = true;
  return res;
}
// ---- key inner product + Mod_down pair without the accumulators (round 5) ----
// Generated code forms the two accumulators of a key-switch limb by limb (resnet20_cifar10_pre.onnx.inc:7011-7034: per digit and
// limb  tmp = key0 * ext; swk_c0 += tmp; tmp = key1 * ext; swk_c1 += tmp), then calls Mod_down on both (:7035-7036) and frees them.
// When the queue still holds exactly that for the pair's inputs -- every limb of both accumulators: a zero fill, then one
// `acc += tmp` per digit whose tmp is the product of limb `pos` of ONE raised digit block and limb gi(pos) of ONE key part, nothing
// else written to those limbs, no operand rewritten afterwards -- the pair runs as acehip_keymac_mod_down2 on the digits and key
// parts themselves: the Mod_down passes form the sums where they would load them and the accumulators are never stored (4 (level + K)
// limb transfers per key-switch less).  The queued products and additions are NOT taken out: the launch does not touch what they
// touch (the digits are declared read-only), so they simply stay queued (split_for_launch) and die with the accumulators when
// generated code frees them a few calls later (the library drops ops whose results lie in freed blocks); a program that reads
// the accumulators afterwards instead finds the ops still queued and gets them executed: nothing changes for it but the order.
static inline u32 limb_gi_of(u32 pos, u32 level, u32 L) { return pos < level ? pos : L + (pos - level); }
namespace {
bool keymac_pair_from_queue(u64* out0, u64* out1, const u64* acc0, const u64* acc1, u32 level) {
  static const bool on = getenv("ACEHIP_KMAC_SHIM") == nullptr || atoi(getenv("ACEHIP_KMAC_SHIM")) != 0;
  static const bool keep_on = getenv("ACEHIP_HW_KEEP") == nullptr || atoi(getenv("ACEHIP_HW_KEEP")) != 0;
  Context& c = ctx();
  if (!on || !keep_on || c.shard_world > 1 || g_hwq.empty()) return false;
  const u32 nd = acehip_num_decomp(c.hip, level);
  if (!acehip_keymac_fusable(c.hip, level, nd)) return false;
  g_kmac_stats.tried++;
  const size_t N = c.N;
  const u32 E = level + c.K, T = c.L + c.K;
  const u64* base[2] = {acc0, acc1};
  struct Term {
    const u64 *a, *b;
    u32 gi;
  };
  static thread_local std::vector<Term> terms;          // [z][pos][8]
  static thread_local std::vector<unsigned char> cnt;   // [z][pos]: 0xff = not zero-filled yet, else the number of terms
  terms.assign((size_t)2 * E * 8, Term{nullptr, nullptr, 0});
  cnt.assign((size_t)2 * E, 0xff);
  // scratch limb -> the product it holds now (generated code funnels every product through ONE limb; a handful of entries, oldest dropped)
  struct Pending {
    const u64* tmp;
    Term t;
  };
  Pending pending[8];
  u32 n_pending = 0;
  auto pending_drop = [&](u32 i) {
    for (u32 j = i + 1; j < n_pending; ++j) pending[j - 1] = pending[j];
    --n_pending;
  };
  static thread_local PtrSet used;  // limbs that recorded products read
  used.reset(4 * (size_t)E * 8 + 64);
  auto slot_of = [&](const u64* p, u32& z, u32& pos) {
    for (u32 zz = 0; zz < 2; ++zz)
      if (p >= base[zz] && p < base[zz] + (size_t)E * N) {
        const size_t off = (size_t)(p - base[zz]);
        if (off % N) return -1;
        z = zz;
        pos = (u32)(off / N);
        return 1;
      }
    return 0;
  };
  for (const acehip_hw_op& o : g_hwq) {
    u32 z = 0, pos = 0;
    const int hit = slot_of(o.res, z, pos);
    if (hit < 0) return ++g_kmac_stats.other, false;
    if (hit == 0) {
      if (used.has(o.res)) return ++g_kmac_stats.other, false;  // an operand of a recorded product is rewritten later
      for (u32 i = n_pending; i-- > 0;)  // the limb gets a new value: products that read it or lie in it are history
        if (pending[i].tmp == o.res || pending[i].t.a == o.res || pending[i].t.b == o.res) pending_drop(i);
      if (o.op == ACEHIP_HW_MUL) {
        if (n_pending == 8) pending_drop(0);
        pending[n_pending++] = Pending{o.res, Term{o.a, (const u64*)o.b, o.prime_gi}};
      }
      continue;
    }
    unsigned char& n = cnt[(size_t)z * E + pos];
    if (o.op == ACEHIP_HW_ZERO) {
      if (n != 0xff && n != 0) return ++g_kmac_stats.other, false;
      n = 0;
      continue;
    }
    if (o.op != ACEHIP_HW_ADD || o.a != o.res) return ++g_kmac_stats.other, false;
    if (n == 0xff) return ++g_kmac_stats.no_zero, false;  // the fill ran earlier: what the limb holds is not in the queue
    u32 pi = n_pending;
    for (u32 i = 0; i < n_pending; ++i)
      if (pending[i].tmp == (const u64*)o.b) pi = i;
    const u32 gi = limb_gi_of(pos, level, c.L);
    if (pi == n_pending || n >= 8 || o.prime_gi != gi || pending[pi].t.gi != gi) return ++g_kmac_stats.other, false;
    terms[((size_t)z * E + pos) * 8 + n++] = pending[pi].t;
    used.insert(pending[pi].t.a);
    used.insert(pending[pi].t.b);
  }
  for (size_t i = 0; i < (size_t)2 * E; ++i)
    if (cnt[i] != nd) return ++(cnt[i] == 0xff ? g_kmac_stats.no_zero : g_kmac_stats.shape), false;
  // the regular shape: term d of limb pos = (digit block d) + pos*N times (key part d)[z] + gi(pos)*N, in either operand order
  const u64 *ext[8], *key[8];
  for (u32 d = 0; d < nd; ++d) {
    bool ok = false;
    for (int swap = 0; swap < 2 && !ok; ++swap) {
      const Term t0 = terms[d];
      const u64* e0 = swap ? t0.a : t0.b;
      const u64* k0 = swap ? t0.b : t0.a;
      ok = true;
      for (u32 z = 0; z < 2 && ok; ++z)
        for (u32 pos = 0; pos < E && ok; ++pos) {
          const Term t = terms[((size_t)z * E + pos) * 8 + d];
          const u64* e = swap ? t.a : t.b;
          const u64* k = swap ? t.b : t.a;
          ok = e == e0 + (size_t)pos * N && k == k0 + ((size_t)z * T + limb_gi_of(pos, level, c.L)) * N;
        }
      if (ok) {
        ext[d] = e0;
        key[d] = k0 - (size_t)limb_gi_of(0, level, c.L) * N;
      }
    }
    if (!ok) return ++g_kmac_stats.shape, false;
  }
  // the outputs must be strangers to everything the sums read
  for (u32 d = 0; d < nd; ++d)
    for (u64* o : {out0, out1})
      if (o + (size_t)level * N > ext[d] && o < ext[d] + (size_t)E * N) return ++g_kmac_stats.other, false;
  cancel_fills(out0, level);
  cancel_fills(out1, level);
  const size_t EW = (size_t)E * N;
  auto dig = [&](u32 d) { return Touch{d < nd ? ext[d] : nullptr, d < nd ? EW : 0, true}; };
  HIPCHK_T(acehip_keymac_mod_down2(c.hip, out0, out1, ext, key, nd, level, nullptr), {out0, (size_t)level * N}, {out1, (size_t)level * N}, dig(0),
           dig(1), dig(2), dig(3), dig(4), dig(5), dig(6), dig(7));
  g_kmac_stats.fused++;
  return true;
}
}  // namespace
POLY Mod_down(POLY res, POLY poly) {
  RtmScope rtm(RTM_MOD_DOWN);
  Context& c = ctx();
  const u32 level = (u32)poly->_num_primes;
  RT_ASSERT(res->_num_primes == level && poly->_num_primes_p == c.K, "reduce_rns_base: result size not match");
  RT_ASSERT(poly->_num_alloc_primes - c.K == level, "Mod_down: p-limbs must follow the q-limbs");
  u64* out = q_limbs(res);
  const u64* in = q_limbs(poly);
  if (g_pend.kind == 1 && g_pend.level == level && g_pend.out != out && g_pend.in != in && g_pend.out != in && g_pend.in != out) {
    const PendingPair p = g_pend;
    g_pend.kind = 0;
    g_lvl.mod_down2[level < 72 ? level : 71]++;
    if (keymac_pair_from_queue(p.out, out, p.in, in, level)) {
      res->_is_ntt = poly->_is_ntt;
      return res;
    }
    cancel_fills(p.out, level);
    cancel_fills(out, level);
    // (ACEHIP_POISON_SELFTEST=1 leaves one input out of the list on purpose: the check of ACEHIP_POISON must abort here -- tests only)
    static const bool selftest = getenv("ACEHIP_POISON_SELFTEST") != nullptr;
    HIPCHK_T(acehip_mod_down2(c.hip, p.out, out, p.in, in, level, nullptr), {p.out, (size_t)level * c.N}, {out, (size_t)level * c.N},
             {p.in, (size_t)(level + c.K) * c.N}, {selftest ? nullptr : in, (size_t)(level + c.K) * c.N});
  } else {
    pending_flush();  // an unpaired predecessor; this call is held back (the queue is handed over when it is issued)
    g_muc.written(out, (size_t)level * c.N);
    g_pend.kind = 1;
    g_pend.out = out;
    g_pend.in = in;
    g_pend.level = level;
  }
  res->_is_ntt = poly->_is_ntt;
  return res;
}
POLY Rescale(POLY res, POLY poly) {
  RtmScope rtm(RTM_RESCALE_POLY);
  Context& c = ctx();
  const u32 level = (u32)poly->_num_primes;
  RT_ASSERT(res->_num_primes == level, "Rescale_poly: primes not match");
  RT_ASSERT(level > 1, "Rescale_poly: level not enough after rescale");
  if (res == poly || res->_data == poly->_data) {
    u64* tmp = dalloc((size_t)(level - 1) * c.N, false);
    HIPCHK(acehip_rescale(c.hip, tmp, q_limbs(poly), level, nullptr));
    copy_limbs((u64*)q_limbs(res), (const u64*)tmp, (size_t)(level - 1) * c.N, level - 1);
    dfree(tmp);
  } else {
    u64* out = q_limbs(res);
    const u64* in = q_limbs(poly);
    if (g_pend.kind == 2 && g_pend.level == level && g_pend.out != out && g_pend.in != in && g_pend.out != in && g_pend.in != out) {
      const PendingPair p = g_pend;
      g_pend.kind = 0;
      cancel_fills(p.out, level - 1);
      cancel_fills(out, level - 1);
      g_lvl.rescale2[level < 72 ? level : 71]++;
      HIPCHK_T(acehip_rescale2(c.hip, p.out, out, p.in, in, level, nullptr), {p.out, (size_t)(level - 1) * c.N},
               {out, (size_t)(level - 1) * c.N}, {p.in, (size_t)level * c.N}, {in, (size_t)level * c.N});
    } else {
      pending_flush();
      g_muc.written(out, (size_t)(level - 1) * c.N);
      g_pend.kind = 2;
      g_pend.out = out;
      g_pend.in = in;
      g_pend.level = level;
    }
  }
  res->_is_ntt = true;
  res->_num_primes = level - 1;  // Mod_down_q_primes polynomial.h:300
  return res;
}

// ---- ciphertext metadata (cipher_eval.c:18-123, ciphertext.h) ----
static void init_ciphertext(CIPHER res, u32 N, size_t nq, size_t np, double sf, u32 sf_degree, u32 slots) {
  res->_scaling_factor = sf;  // Init_ciphertext ciphertext.h:179-205: allocate only if empty, never clears
  res->_sf_degree = sf_degree;
  res->_slots = slots;
  if (res->_c0_poly._data == nullptr) poly_alloc(&res->_c0_poly, N, nq, np);
  else RT_ASSERT(res->_c0_poly._ring_degree == N && res->_c0_poly._num_primes == nq, "unmatched ciphertxt");
  if (res->_c1_poly._data == nullptr) poly_alloc(&res->_c1_poly, N, nq, np);
  else RT_ASSERT(res->_c1_poly._ring_degree == N && res->_c1_poly._num_primes == nq, "unmatched ciphertxt");
}
static void set_level(CIPHER c, size_t level) {
  c->_c0_poly._num_primes = level;
  c->_c1_poly._num_primes = level;
}
// Adjust_level(ciph1, ciph2, resize=false) ciphertext.h:283-325: the operand with the smaller level
static CIPHER lower_level(CIPHER a, CIPHER b) {
  RT_ASSERT(a && b, "invalid ciph");
  if (a->_c0_poly._data == nullptr) return b;
  RT_ASSERT(b->_c0_poly._data != nullptr, "poly coeffs of input ciph is invalid");
  return a->_c0_poly._num_primes > b->_c0_poly._num_primes ? b : a;
}
// Init_cipher cipher_eval.c:18-30: Init_ciphertext_from_ciph (zero-fill on reuse unless res == ciph)
static void init_cipher_from(CIPHER res, CIPHER ciph, double sf, u32 sf_degree) {
  res->_scaling_factor = sf;
  res->_sf_degree = sf_degree;
  res->_slots = ciph->_slots;
  if (res != ciph) {
    poly_init_like(&res->_c0_poly, &ciph->_c0_poly);
    poly_init_like(&res->_c1_poly, &ciph->_c1_poly);
  }
  res->_c0_poly._is_ntt = true;
  res->_c1_poly._is_ntt = true;
  set_level(res, ciph->_c0_poly._num_primes);
}

void Init_ciph_same_scale(CIPHER res, CIPHER ciph1, CIPHER ciph2) {
  RtmScope rtm(RTM_INIT_CIPH_SM_SC, false);
  CIPHER c = ciph2 != nullptr ? lower_level(ciph1, ciph2) : ciph1;
  init_ciphertext(res, c->_c0_poly._ring_degree, c->_c0_poly._num_primes, c->_c0_poly._num_primes_p, c->_scaling_factor,
                  c->_sf_degree, c->_slots);
  res->_c0_poly._is_ntt = true;
  res->_c1_poly._is_ntt = true;
}
void Init_ciph_same_scale_plain(CIPHER res, CIPHER ciph, PLAIN) {
  RtmScope rtm(RTM_INIT_CIPH_SM_SC, false);
  init_cipher_from(res, ciph, ciph->_scaling_factor, ciph->_sf_degree);
}
void Init_ciph_up_scale(CIPHER res, CIPHER c1, CIPHER c2) {
  RtmScope rtm(RTM_INIT_CIPH_UP_SC, false);
  CIPHER c = lower_level(c1, c2);
  init_cipher_from(res, c, c1->_scaling_factor * c2->_scaling_factor, c1->_sf_degree + c2->_sf_degree);
}
void Init_ciph_up_scale_plain(CIPHER res, CIPHER ciph, PLAIN plain) {
  RtmScope rtm(RTM_INIT_CIPH_UP_SC, false);
  init_cipher_from(res, ciph, ciph->_scaling_factor * plain->_scaling_factor, ciph->_sf_degree + plain->_sf_degree);
}
void Init_ciph_down_scale(CIPHER res, CIPHER ciph) {
  RtmScope rtm(RTM_INIT_CIPH_DN_SC, false);
  init_cipher_from(res, ciph, ciph->_scaling_factor / ctx().sf, ciph->_sf_degree - 1);
}
static void init_from3(POLYNOMIAL* r0, POLYNOMIAL* r1, POLYNOMIAL* s0, POLYNOMIAL* s1) {
  poly_init_like(r0, s0);
  poly_init_like(r1, s1);
}
void Init_ciph_same_scale_ciph3(CIPHER res, CIPHER3 ciph) {
  RtmScope rtm(RTM_INIT_CIPH_SM_SC, false);
  RT_ASSERT(res, "invalid ciphertext");
  res->_scaling_factor = ciph->_scaling_factor;
  res->_sf_degree = ciph->_sf_degree;
  res->_slots = ciph->_slots;
  init_from3(&res->_c0_poly, &res->_c1_poly, &ciph->_c0_poly, &ciph->_c1_poly);
  res->_c0_poly._is_ntt = res->_c1_poly._is_ntt = true;
  set_level(res, ciph->_c0_poly._num_primes);
}
void Init_ciph3_same_scale_ciph3(CIPHER3 res, CIPHER3 c1, CIPHER3 c2) {
  RtmScope rtm(RTM_INIT_CIPH_SM_SC, false);
  RT_ASSERT(res, "invalid ciphertext");
  CIPHER3 c = c1;
  if (c2 != nullptr && c1->_c0_poly._data != nullptr && c2->_c0_poly._num_primes < c1->_c0_poly._num_primes) c = c2;
  if (c2 != nullptr && c1->_c0_poly._data == nullptr) c = c2;
  res->_scaling_factor = c->_scaling_factor;
  res->_sf_degree = c->_sf_degree;
  res->_slots = c->_slots;
  if (res != c) {
    poly_init_like(&res->_c0_poly, &c->_c0_poly);
    poly_init_like(&res->_c1_poly, &c->_c1_poly);
    poly_init_like(&res->_c2_poly, &c->_c2_poly);
  }
  res->_c0_poly._is_ntt = res->_c1_poly._is_ntt = res->_c2_poly._is_ntt = true;
  res->_c0_poly._num_primes = res->_c1_poly._num_primes = res->_c2_poly._num_primes = c->_c0_poly._num_primes;
}
void Init_ciph3_up_scale(CIPHER3 res, CIPHER c1, CIPHER c2) {
  RtmScope rtm(RTM_INIT_CIPH_UP_SC, false);
  RT_ASSERT(res, "invalid ciphertext");
  CIPHER c = lower_level(c1, c2);
  res->_scaling_factor = c1->_scaling_factor * c2->_scaling_factor;
  res->_sf_degree = c1->_sf_degree + c2->_sf_degree;
  res->_slots = c->_slots;
  RT_ASSERT(c->_c0_poly._num_primes_p == 0, "invalid num of p primes");
  poly_init_like(&res->_c0_poly, &c->_c0_poly);
  poly_init_like(&res->_c1_poly, &c->_c0_poly);
  poly_init_like(&res->_c2_poly, &c->_c0_poly);
  res->_c0_poly._is_ntt = res->_c1_poly._is_ntt = res->_c2_poly._is_ntt = true;
  res->_c0_poly._num_primes = res->_c1_poly._num_primes = res->_c2_poly._num_primes = c->_c0_poly._num_primes;
}
void Copy_ciph(CIPHER res, CIPHER ciph) {
  RtmScope rtm(RTM_COPY_CIPH, false);  // Copy_ciphertext ciphertext.h:247-253
  if (res == ciph) return;
  res->_scaling_factor = ciph->_scaling_factor;
  res->_sf_degree = ciph->_sf_degree;
  res->_slots = ciph->_slots;
  poly_init_like(&res->_c0_poly, &ciph->_c0_poly);
  poly_init_like(&res->_c1_poly, &ciph->_c1_poly);
  poly_copy(&res->_c0_poly, &ciph->_c0_poly);
  poly_copy(&res->_c1_poly, &ciph->_c1_poly);
}
size_t Level(CIPHER ciph) { return ciph->_c0_poly._num_primes; }
uint32_t Sc_degree(CIPHER ciph) { return ciph->_sf_degree; }
uint32_t Get_slots(CIPHER ciph) { return ciph->_slots; }
void Set_slots(CIPHER ciph, uint32_t slots) { ciph->_slots = slots; }
void Free_ciph_poly(CIPHER ciph, uint32_t cnt) {
  for (uint32_t i = 0; i < cnt; ++i) {
    poly_free(&ciph[i]._c0_poly);
    poly_free(&ciph[i]._c1_poly);
  }
}
void Zero_ciph(CIPHER ciph) {
  poly_free(&ciph->_c0_poly);
  poly_free(&ciph->_c1_poly);
  memset(ciph, 0, sizeof(*ciph));
}
void Free_cipher(CIPHER ciph) {
  if (!ciph) return;
  poly_free(&ciph->_c0_poly);
  poly_free(&ciph->_c1_poly);
  free(ciph);
}
void Free_plain(PLAIN plain) {
  if (!plain) return;
  poly_free(&plain->_poly);
}

}  // extern "C"
