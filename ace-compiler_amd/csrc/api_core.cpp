// api_core.cpp -- C ABI (include/acehip.h): context / table upload, device memory helpers, call statistics.
#include "api_internal.hpp"

std::string& acehip_err_slot() {
  static thread_local std::string e;
  return e;
}
acehip_stat* acehip_stat_slots() {
  static thread_local acehip_stat g[ST_COUNT];
  return g;
}
namespace {
struct TouchLog {
  bool on = false;
  std::vector<std::pair<const void*, size_t>> v;
};
TouchLog& touch_log() {
  static thread_local TouchLog t;
  return t;
}
}  // namespace
void dbg_touch(const void* p, size_t words) {
  TouchLog& t = touch_log();
  if (t.on && p != nullptr && words != 0) t.v.emplace_back(p, words);
}
namespace acehip {
void ntt_count(u64 limbs) {  // (launchers are called once per replica set: the multiplier is the launch's own)
  acehip_stat* g = acehip_stat_slots();
  g[ST_NTT_ALL].calls++;
  g[ST_NTT_ALL].units += limbs;
  g[ST_NTT_ALL].bytes += limbs * 16ull;  // x N below would need the ring size: bench.py multiplies
}
}  // namespace acehip
u32& acehip_stat_mult() {
  static thread_local u32 m = 1;
  return m;
}
bool& acehip_stat_mute() {
  static thread_local bool m = false;
  return m;
}
u32 replica_chunk() {
  static const u32 v = [] {
    const char* e = getenv("ACEHIP_REP_CHUNK");
    return e ? (u32)atoi(e) : 0u;
  }();
  return v;
}
static const char* const kStatName[ST_COUNT] = {"ntt", "elementwise", "rotate", "decomp_modup", "key_inner_product",
                                                "mod_down", "rescale", "key_switch", "encode", "zero_fill_executed", "elementwise_mul", "ntt_launched"};

extern "C" {

const char* acehip_last_error(void) { return acehip_err_slot().c_str(); }

int acehip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

acehip_ctx* acehip_ctx_create_host(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum) {
  std::unique_ptr<acehip_ctx> ctx;
  try {
    ctx.reset(new acehip_ctx());
    ctx->hp = make_params(N, L, q0_bits, sf_bits, dnum);
    return ctx.release();
  } catch (const std::exception& e) {  // the half-built context is released (found by `make -C oracle asan`)
    acehip_err_slot() = e.what();
    return nullptr;
  }
}

acehip_ctx* acehip_ctx_create(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum, int device) {
  if (acehip_device_count() <= device || device < 0) {
    acehip_err_slot() = "acehip_ctx_create: no such GPU device (the HIP path has no CPU fallback)";
    return nullptr;
  }
  acehip_ctx* ctx = acehip_ctx_create_host(N, L, q0_bits, sf_bits, dnum);
  if (!ctx) return nullptr;
  if (hipSetDevice(device) != hipSuccess) {
    acehip_err_slot() = "hipSetDevice failed";
    delete ctx;
    return nullptr;
  }
  ctx->device = device;
  const HostParams& hp = ctx->hp;
  const u32 T = hp.L + hp.K;
  std::vector<DevPrime> dp(T);
  std::memcpy(dp.data(), hp.primes.data(), T * sizeof(DevPrime));
  ctx->dc.primes = ctx->up(dp);
  {  // interleave {w, Shoup companion} so that a twiddle is one 16-byte load
    std::vector<ulong2> tw((size_t)T * hp.N);
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = ulong2{hp.rou[i], hp.rou_prec[i]};
    ctx->dc.tw_fwd = ctx->up(tw);
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = ulong2{hp.rou_inv[i], hp.rou_inv_prec[i]};
    ctx->dc.tw_inv = ctx->up(tw);
  }
  ctx->dc.N = hp.N;
  ctx->dc.logN = hp.logN;
  ctx->dc.L = hp.L;
  ctx->dc.K = hp.K;
  {
    u32 max_bits = 0;
    for (u32 i = 0; i < T; ++i) max_bits = std::max(max_bits, (u32)hp.primes[i].nbits);
    ctx->dc.split_bits = (max_bits + 1) / 2;  // <= 31: primes are below 2^61 (host_params)
  }
  ctx->phat_inv = ctx->up(hp.phat_inv_modp);
  ctx->phat_inv_prec = ctx->up(hp.phat_inv_modp_prec);
  // base_conv wants hat[i_src][j_dst]: transpose phat_modq[L][K] -> [K][L]
  std::vector<u64> tr((size_t)hp.K * hp.L);
  for (u32 i = 0; i < hp.L; ++i)
    for (u32 j = 0; j < hp.K; ++j) tr[(size_t)j * hp.L + i] = hp.phat_modq[(size_t)i * hp.K + j];
  ctx->phat_modq_t = ctx->up(tr);
  ctx->pinv = ctx->up(hp.pinv_modq);
  ctx->pinv_prec = ctx->up(hp.pinv_modq_prec);
  ctx->ql_inv = ctx->up(hp.ql_inv);
  ctx->ql_inv_prec = ctx->up(hp.ql_inv_prec);
  ctx->qlql = ctx->up(hp.qlql);
  ctx->qlql_prec = ctx->up(hp.qlql_prec);
  std::vector<u32> pgi(hp.K), qgi(hp.L);
  for (u32 j = 0; j < hp.K; ++j) pgi[j] = hp.L + j;
  for (u32 i = 0; i < hp.L; ++i) qgi[i] = i;
  ctx->p_gi = ctx->up(pgi);
  ctx->q_gi = ctx->up(qgi);
  ctx->q_pos = ctx->q_gi;
  if (hp.logN == 16) {  // companion-only twiddle tables (ntt_fast.hip Tp15): 8-byte twiddle stream in the contiguous passes
    // ACEHIP_NTT_TW8_POLYS = largest number of polynomials per launch that uses them (0: never).  Measured: ResNet-20 1.68 ->
    // 1.73 images/s, C3 key-switch 0.248 -> 0.237 ms, 1024-limb batch 0.555 -> 0.537 ms with every launch on the 8-byte stream
    const char* e = getenv("ACEHIP_NTT_TW8_POLYS");
    const u32 maxp = e ? (u32)strtoul(e, nullptr, 0) : 65535u;
    if (maxp) {
      ctx->dc.twp_fwd = ctx->up(hp.rou_prec);
      ctx->dc.twp_inv = ctx->up(hp.rou_inv_prec);
      ctx->dc.tw8_max_polys = (ctx->dc.twp_fwd && ctx->dc.twp_inv) ? maxp : 0;
      if (!ctx->dc.tw8_max_polys) ctx->dc.twp_fwd = ctx->dc.twp_inv = nullptr;
    }
  }
  if (hp.logN == 16) {  // FP64 butterflies for the 48..50-bit scaling primes (ntt_fp.hpp); ACEHIP_NTT_FP=0: integer classes only
    const char* e = getenv("ACEHIP_NTT_FP");
    bool any = false;
    for (u32 i = 0; i < T; ++i) any = any || hp.primes[i].q < ((1ull << 50) + (1ull << 43));  // kFpPrimeMax (ntt_fp.hpp)
    if ((!e || atoi(e) != 0) && any) {
      std::vector<double> twd((size_t)T * hp.N);
      for (size_t i = 0; i < twd.size(); ++i) twd[i] = (double)hp.rou[i];  // (exact wherever the class is used: w < q < 2^53)
      ctx->dc.twd_fwd = ctx->up(twd);
      for (size_t i = 0; i < twd.size(); ++i) twd[i] = (double)hp.rou_inv[i];
      ctx->dc.twd_inv = ctx->up(twd);
      if (!ctx->dc.twd_fwd || !ctx->dc.twd_inv) ctx->dc.twd_fwd = ctx->dc.twd_inv = nullptr;
    }
  }
  {  // ACEHIP_NTT_NARROW = largest launch (limb rows) that takes the narrow small-launch passes (0: never)
    const char* e = getenv("ACEHIP_NTT_NARROW");
    ctx->dc.ntt_narrow_max_rows = e ? (u32)strtoul(e, nullptr, 0) : 16u;
  }
  // workspace of the batched key-switch: coef (L) + ext[dnum] + two accumulators (L+K each) + tmp (2L)
  ctx->ws_words = ((size_t)hp.L * 3 + (size_t)(hp.dnum + 2) * T) * hp.N;
  if (hipMalloc(&ctx->ws, ctx->ws_words * sizeof(u64)) != hipSuccess || !ctx->dc.primes || !ctx->dc.tw_inv) {
    acehip_err_slot() = "acehip_ctx_create: device allocation/upload failed";
    acehip_ctx_destroy(ctx);
    return nullptr;
  }
  ctx->owned.push_back(ctx->ws);
  ctx->on_device = true;
  return ctx;
}

void acehip_ctx_destroy(acehip_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  shard_release(ctx);  // RCCL communicator, exchange stream, events (api_shard.cpp)
  for (void* p : ctx->owned) (void)hipFree(p);
  if (ctx->hw_scratch && !ctx->scratch_external) (void)hipFree(ctx->hw_scratch);
  delete ctx;
}

uint32_t acehip_degree(const acehip_ctx* c) { return c->hp.N; }
uint32_t acehip_num_q(const acehip_ctx* c) { return c->hp.L; }
uint32_t acehip_num_p(const acehip_ctx* c) { return c->hp.K; }
uint32_t acehip_num_q_parts(const acehip_ctx* c) { return c->hp.dnum; }
uint32_t acehip_part_size(const acehip_ctx* c) { return c->hp.alpha; }
uint32_t acehip_num_decomp(const acehip_ctx* c, uint32_t level) { return c->hp.num_decomp(level); }
uint64_t acehip_prime(const acehip_ctx* c, uint32_t gi) { return gi < c->hp.L + c->hp.K ? c->hp.primes[gi].q : 0; }

int64_t acehip_get_table(const acehip_ctx* c, int what, uint32_t gi, uint64_t* out, size_t cap) {
  const HostParams& hp = c->hp;
  const u32 T = hp.L + hp.K;
  auto copy = [&](const u64* src, size_t n) -> int64_t {
    if (cap < n) return fail(ACEHIP_EINVAL, "acehip_get_table: buffer too small");
    std::memcpy(out, src, n * sizeof(u64));
    return (int64_t)n;
  };
  if (what >= 0 && what <= 4) {
    if (cap < T) return fail(ACEHIP_EINVAL, "acehip_get_table: buffer too small");
    for (u32 i = 0; i < T; ++i) {
      const PrimeConsts& p = hp.primes[i];
      out[i] = what == 0 ? p.psi : what == 1 ? p.n_inv : what == 2 ? p.n_inv_prec : what == 3 ? p.prec128_lo : p.prec128_hi;
    }
    return T;
  }
  if (what >= 10 && what <= 13) {
    if (gi >= T) return fail(ACEHIP_EINVAL, "acehip_get_table: bad prime index");
    const std::vector<u64>& v = what == 10 ? hp.rou : what == 11 ? hp.rou_prec : what == 12 ? hp.rou_inv : hp.rou_inv_prec;
    return copy(v.data() + (size_t)gi * hp.N, hp.N);
  }
  switch (what) {
    case 20: return copy(hp.phat_inv_modp.data(), hp.K);
    case 21: return copy(hp.phat_inv_modp_prec.data(), hp.K);
    case 22: return copy(hp.phat_modq.data(), (size_t)hp.L * hp.K);
    case 23: return copy(hp.pinv_modq.data(), hp.L);
    case 30: return copy(hp.ql_inv.data(), (size_t)hp.L * hp.L);
    case 31: return copy(hp.ql_inv_prec.data(), (size_t)hp.L * hp.L);
    case 32: return copy(hp.qlql.data(), (size_t)hp.L * hp.L);
    case 33: return copy(hp.qlql_prec.data(), (size_t)hp.L * hp.L);
  }
  return fail(ACEHIP_EINVAL, "acehip_get_table: unknown table id");
}

int acehip_get_modup_tables(const acehip_ctx* c, uint32_t level, uint32_t digit, uint64_t* hat_inv,
                            uint32_t* compl_idx, uint64_t* hat_mod, uint32_t* nc_out) {
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "bad level/digit");
  HostParams::ModUp t = c->hp.modup(level, digit);
  std::memcpy(hat_inv, t.hat_inv.data(), t.n2 * sizeof(u64));
  std::memcpy(compl_idx, t.compl_idx.data(), t.nc * sizeof(u32));
  std::memcpy(hat_mod, t.hat_mod.data(), (size_t)t.n2 * t.nc * sizeof(u64));
  *nc_out = t.nc;
  return (int)t.n2;
}

uint32_t acehip_auto_index(const acehip_ctx* c, int32_t rot_idx) { return find_automorphism_index(rot_idx, c->hp.N); }

int acehip_auto_order_host(const acehip_ctx* c, uint32_t k, uint32_t* out_perm) {
  if ((k & 1) == 0 || k >= 2 * c->hp.N) return fail(ACEHIP_EINVAL, "automorphism index must be odd and < 2N");
  automorphism_order_ntt(out_perm, k, c->hp.N);
  return ACEHIP_OK;
}

const uint32_t* acehip_auto_order(acehip_ctx* c, uint32_t k) {
  if (!c->on_device) {
    acehip_err_slot() = "acehip_auto_order: context has no device";
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  auto it = c->auto_tabs.find(k);
  if (it != c->auto_tabs.end()) return it->second;
  std::vector<u32> perm(c->hp.N);
  if (acehip_auto_order_host(c, k, perm.data()) != ACEHIP_OK) return nullptr;
  (void)hipSetDevice(c->device);
  u32* d = c->up(perm);
  if (!d) {
    acehip_err_slot() = "acehip_auto_order: upload failed";
    return nullptr;
  }
  c->auto_tabs[k] = d;
  c->auto_tab_k[d] = k;
  return d;
}

// Debug aid of callers that defer work around launches whose operands they declare (the rt_ant shim's lazy zero fills,
// ACEHIP_POISON=1): with logging on, the pipeline entry points (key-switch, ModUp of all digits, Mod_down, Rescale, encode) record
// every range of caller memory they read or write; the call returns what was recorded since the last call.
size_t acehip_debug_touches(int enable, const void** ptrs, size_t* words, size_t cap) {
  TouchLog& t = touch_log();
  const size_t n = t.v.size();
  for (size_t i = 0; i < n && i < cap; ++i) {
    ptrs[i] = t.v[i].first;
    words[i] = t.v[i].second;
  }
  t.v.clear();
  t.on = enable != 0;
  return n;
}

// ---- replicas of the caller's arena ----
size_t acehip_workspace_words(const acehip_ctx* c) { return c ? c->ws_words : 0; }

int acehip_ctx_set_arena(acehip_ctx* c, const acehip_arena_cfg* cfg) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if (c->sh_world > 1 && c->rccl == nullptr) return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: the arena cannot change while simulated ranks occupy its replicas");
  if (cfg == nullptr || cfg->base == nullptr) {
    if (c->ws_external || c->scratch_external) return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: the workspace lives in the arena that is being removed");
    c->dc.rep_lo = c->dc.rep_span = c->dc.rep_stride = 0;
    c->n_replicas = 1;
    c->sel0 = 0;
    c->seln = 1;
    return ACEHIP_OK;
  }
  const u64 lo = (u64)cfg->base, span = cfg->bytes;
  auto inside = [&](const void* p, size_t bytes) { return (u64)p >= lo && (u64)p + bytes <= lo + span; };
  if (cfg->n_replicas == 0 || cfg->bytes == 0 || (cfg->n_replicas > 1 && cfg->stride_bytes < cfg->bytes) || (cfg->stride_bytes & 15))
    return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: bad geometry");
  if (cfg->n_replicas > 1 && (cfg->workspace == nullptr || cfg->hw_scratch == nullptr))
    return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: several replicas need a workspace and a hw scratch of their own inside the arena");
  if (cfg->workspace && !inside(cfg->workspace, c->ws_words * sizeof(u64)))
    return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: the workspace must lie inside replica 0 (acehip_workspace_words() words)");
  if (cfg->hw_scratch && (cfg->hw_scratch_limbs == 0 || !inside(cfg->hw_scratch, cfg->hw_scratch_limbs * c->hp.N * sizeof(u64))))
    return fail(ACEHIP_EINVAL, "acehip_ctx_set_arena: the hw scratch must lie inside replica 0");
  if (c->on_device) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();  // launches that still use the old workspace / scratch
  }
  if (cfg->workspace) {
    if (!c->ws_external && c->ws) {
      c->owned.erase(std::remove(c->owned.begin(), c->owned.end(), (void*)c->ws), c->owned.end());
      (void)hipFree(c->ws);
    }
    c->ws = (u64*)cfg->workspace;
    c->ws_external = true;
  }
  if (cfg->hw_scratch) {
    if (!c->scratch_external && c->hw_scratch) (void)hipFree(c->hw_scratch);
    c->hw_scratch = (u64*)cfg->hw_scratch;
    c->hw_scratch_limbs = cfg->hw_scratch_limbs;
    c->scratch_external = true;
  }
  c->dc.rep_lo = lo;
  c->dc.rep_span = span;
  c->dc.rep_stride = cfg->n_replicas > 1 ? cfg->stride_bytes : 0;
  c->n_replicas = cfg->n_replicas;
  c->sel0 = 0;
  c->seln = 1;
  return ACEHIP_OK;
}

int acehip_ctx_select(acehip_ctx* c, uint32_t rep0, uint32_t nrep) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if (nrep == 0 || rep0 + nrep > c->n_replicas) return fail(ACEHIP_EINVAL, "acehip_ctx_select: replicas outside the arena");
  if (c->sh_world > 1 && c->rccl == nullptr && (rep0 != 0 || nrep != 1))
    return fail(ACEHIP_EINVAL, "acehip_ctx_select: simulated ranks occupy the replicas");
  c->sel0 = rep0;
  c->seln = nrep;
  return ACEHIP_OK;
}

}  // extern "C"
namespace {
// every (replica, in-arena?) instance of a device address the current selection covers
template <class F>
int for_each_instance(acehip_ctx* c, const void* d_ptr, bool all, F f) {
  const u64 a = (u64)d_ptr;
  const bool in_arena = a - c->dc.rep_lo < c->dc.rep_span;
  if (!in_arena || c->dc.rep_stride == 0) return f((void*)a);
  const DcList dcs = launch_dcs(c);
  for (const DevCtx& dc : dcs)
    for (u32 r = dc.rep0; r < dc.rep0 + dc.nrep; ++r) {
      if (int e = f((void*)(a + (u64)r * c->dc.rep_stride))) return e;
      if (!all) return ACEHIP_OK;
    }
  return ACEHIP_OK;
}
}  // namespace
extern "C" {

int acehip_upload(acehip_ctx* c, void* d, const void* h, size_t n, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (int e = for_each_instance(c, d, true, [&](void* p) -> int {
        HIP_TRY(hipMemcpyAsync(p, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
        return ACEHIP_OK;
      }))
    return e;
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_download(acehip_ctx* c, void* h, const void* d, size_t n, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (int e = for_each_instance(c, d, false, [&](void* p) -> int {
        HIP_TRY(hipMemcpyAsync(h, p, n, hipMemcpyDeviceToHost, (hipStream_t)s));
        return ACEHIP_OK;
      }))
    return e;
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_fill(acehip_ctx* c, void* d, int v, size_t n, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  return for_each_instance(c, d, true, [&](void* p) -> int {
    HIP_TRY(hipMemsetAsync(p, v, n, (hipStream_t)s));
    return ACEHIP_OK;
  });
}
int acehip_copy(acehip_ctx* c, void* d, const void* src, size_t n, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u64 a = (u64)src, d0 = (u64)d;
  const bool rep_on = c->dc.rep_stride != 0;
  const bool src_in = rep_on && a - c->dc.rep_lo < c->dc.rep_span, dst_in = rep_on && d0 - c->dc.rep_lo < c->dc.rep_span;
  const u64 first = (u64)launch_dcs(c).d[0].rep0 * c->dc.rep_stride;
  return for_each_instance(c, d, true, [&](void* p) -> int {
    // replica r's copy reads replica r's instance of the source when that lies in the arena too; a destination outside the
    // arena gets the first selected replica's source
    const u64 from = !src_in ? a : dst_in ? a + ((u64)p - d0) : a + first;
    HIP_TRY(hipMemcpyAsync(p, (const void*)from, n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return ACEHIP_OK;
  });
}

// ---- owner-only limb memory (limb-sharded execution, one process per rank) -------------------------------------------------------------
// Switch keys are 77 % of a rank's HBM (227 keys x 135 MiB for the generated ResNet-20) and a rank of G only ever touches the limbs
// whose prime it owns (gi % G == rank) -- but generated code addresses key limbs itself (`Coeffs(key, i, degree)` = base + i * N), so the
// full reference layout must stay ADDRESSABLE.  HIP's virtual memory management gives exactly that: the whole block is one reserved
// address range, every owned limb gets physical memory of its own, and every other limb position is a mapping of ONE shared "sink"
// limb -- a stray access to a limb the rank does not own lands there (garbage, as in the simulated ranks' replicas, never a fault).
// Falls back to a plain allocation where the driver lacks the feature or a limb is not a multiple of its granularity.
// ACEHIP_SHARD_OWNER_LIMBS=1 switches it on (default: plain allocations; see owner_only_on for what the driver charges).
namespace {
struct LimbBlock {
  size_t bytes = 0, limb_bytes = 0, n_limbs = 0;
  std::vector<hipMemGenericAllocationHandle_t> own;
};
struct SinkKey {
  int dev;
  size_t bytes;
  bool operator<(const SinkKey& o) const { return dev != o.dev ? dev < o.dev : bytes < o.bytes; }
};
// The registries are never destroyed: acehip_free consults them, and callers free device memory from destructors that run at process
// exit in no particular order relative to this library's statics (a destroyed std::map here was a crash at exit, found by the GPU suite).
struct LimbRegistry {
  std::mutex mu;
  std::map<void*, LimbBlock> blocks;
  std::map<SinkKey, hipMemGenericAllocationHandle_t> sinks;  // one per device and limb size, for the life of the process
  std::map<void*, size_t> plain;                             // acehip_malloc_limbs blocks that are plain allocations (statistics only)
};
LimbRegistry& limb_reg() {
  static LimbRegistry* r = new LimbRegistry;
  return *r;
}
#define g_limb_mu (limb_reg().mu)
#define g_limb_blocks (limb_reg().blocks)
#define g_limb_sinks (limb_reg().sinks)
#define g_limb_plain (limb_reg().plain)
std::atomic<u64> g_limb_backed{0}, g_limb_addressed{0};
bool owner_only_on() {
  // opt-in: on ROCm 7.2 / MI355X a mapping cannot start inside a physical handle (hipMemMap with an offset: invalid argument), so every
  // owned limb needs a handle of its own, and a 512 KiB handle occupies 2 MiB and takes 0.2-0.6 ms to map (tools/ubench_vmm.hip,
  // profiles/r06m_vmm_probe.txt): a rank of 8 holds half of the key bytes instead of an eighth, a rank of 4 saves nothing
  static const bool on = [] { const char* e = getenv("ACEHIP_SHARD_OWNER_LIMBS"); return e && atoi(e) != 0; }();
  return on;
}
// true: p was one of ours and is gone
bool limb_block_free(void* p) {
  LimbBlock b;
  {
    std::lock_guard<std::mutex> lk(g_limb_mu);
    auto pl = g_limb_plain.find(p);
    if (pl != g_limb_plain.end()) {  // a plain allocation: only the statistics are ours
      g_limb_backed -= pl->second;
      g_limb_addressed -= pl->second;
      g_limb_plain.erase(pl);
      return false;
    }
    auto it = g_limb_blocks.find(p);
    if (it == g_limb_blocks.end()) return false;
    b = std::move(it->second);
    g_limb_blocks.erase(it);
  }
  (void)hipDeviceSynchronize();  // (hipFree synchronises too: nothing may still run on the range)
  for (size_t k = 0; k < b.n_limbs; ++k) (void)hipMemUnmap((char*)p + k * b.limb_bytes, b.limb_bytes);
  for (auto h : b.own) (void)hipMemRelease(h);
  (void)hipMemAddressFree(p, b.bytes);
  g_limb_backed -= (u64)b.own.size() * b.limb_bytes;
  g_limb_addressed -= b.bytes;
  return true;
}
void* limb_block_alloc(const acehip_ctx* c, const uint32_t* gi, size_t n_limbs) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  int vmm = 0;
  if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev) != hipSuccess || !vmm) return nullptr;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  const size_t limb = (size_t)c->hp.N * sizeof(u64);
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0 || limb % gran) return nullptr;
  std::lock_guard<std::mutex> lk(g_limb_mu);
  hipMemGenericAllocationHandle_t sink;
  auto st = g_limb_sinks.find(SinkKey{dev, limb});
  if (st != g_limb_sinks.end()) sink = st->second;
  else {
    if (hipMemCreate(&sink, limb, &prop, 0) != hipSuccess) return nullptr;
    g_limb_sinks[SinkKey{dev, limb}] = sink;
  }
  void* base = nullptr;
  const size_t bytes = limb * n_limbs;
  if (hipMemAddressReserve(&base, bytes, 0, nullptr, 0) != hipSuccess) return nullptr;
  LimbBlock b;
  b.bytes = bytes;
  b.limb_bytes = limb;
  b.n_limbs = n_limbs;
  size_t mapped = 0;
  bool ok = true;
  for (; mapped < n_limbs && ok; ++mapped) {
    void* at = (char*)base + mapped * limb;
    if (gi[mapped] % c->sh_world == c->sh_hosted[0]) {  // (one process per rank: the one hosted rank)
      hipMemGenericAllocationHandle_t h;
      ok = hipMemCreate(&h, limb, &prop, 0) == hipSuccess;
      if (ok) {
        b.own.push_back(h);
        ok = hipMemMap(at, limb, 0, h, 0) == hipSuccess;
      }
    } else {
      ok = hipMemMap(at, limb, 0, sink, 0) == hipSuccess;
    }
    if (!ok) break;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  ok = ok && hipMemSetAccess(base, bytes, &acc, 1) == hipSuccess;
  if (!ok) {  // undo: the caller falls back to a plain allocation
    (void)hipGetLastError();
    for (size_t k = 0; k < mapped; ++k) (void)hipMemUnmap((char*)base + k * limb, limb);
    for (auto h : b.own) (void)hipMemRelease(h);
    (void)hipMemAddressFree(base, bytes);
    return nullptr;
  }
  g_limb_backed += (u64)b.own.size() * limb;
  g_limb_addressed += bytes;
  g_limb_blocks[base] = std::move(b);
  return base;
}
}  // namespace

extern "C" void* acehip_malloc_limbs(acehip_ctx* c, const uint32_t* h_gi, size_t n_limbs) {
  if (!c || !h_gi || n_limbs == 0) {
    acehip_err_slot() = "acehip_malloc_limbs: bad arguments";
    return nullptr;
  }
  for (size_t k = 0; k < n_limbs; ++k)
    if (h_gi[k] >= c->hp.L + c->hp.K) {
      acehip_err_slot() = "acehip_malloc_limbs: prime index out of range";
      return nullptr;
    }
  const size_t bytes = (size_t)c->hp.N * sizeof(u64) * n_limbs;
  if (c->on_device && c->sh_world > 1 && c->rccl != nullptr && owner_only_on()) {
    if (void* p = limb_block_alloc(c, h_gi, n_limbs)) return p;
    static std::atomic<bool> told{false};
    if (!told.exchange(true)) fprintf(stderr, "[acehip] owner-only limb memory is not available here (HIP virtual memory management / granularity): full allocations\n");
  }
  void* p = acehip_malloc(bytes);
  if (p) {
    g_limb_backed += bytes;
    g_limb_addressed += bytes;
    std::lock_guard<std::mutex> lk(g_limb_mu);
    g_limb_plain[p] = bytes;
  }
  return p;
}
extern "C" void acehip_limb_memory(uint64_t* backed_bytes, uint64_t* addressed_bytes) {
  if (backed_bytes) *backed_bytes = g_limb_backed.load();
  if (addressed_bytes) *addressed_bytes = g_limb_addressed.load();
}

// ---- memory helpers ----
void* acehip_malloc(size_t bytes) {
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    acehip_err_slot() = std::string("hipMalloc failed: ") + hipGetErrorString(e) + " (" + std::to_string(fr >> 20) + " MiB free of " + std::to_string(tot >> 20) + ")";
    return nullptr;
  }
  return p;
}
int acehip_free(void* p) {
  if (limb_block_free(p)) return ACEHIP_OK;  // memory of acehip_malloc_limbs (below)
  HIP_TRY(hipFree(p));
  return ACEHIP_OK;
}
int acehip_memcpy_h2d(void* d, const void* h, size_t n, acehip_stream s) {
  HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return ACEHIP_OK;
}
void* acehip_malloc_host(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    acehip_err_slot() = "hipHostMalloc failed";
    return nullptr;
  }
  return p;
}
int acehip_free_host(void* p) {
  HIP_TRY(hipHostFree(p));
  return ACEHIP_OK;
}
int acehip_memcpy_h2d_async(void* d, const void* h, size_t n, acehip_stream s) {
  HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_event_sync(void* e) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)e));
  return ACEHIP_OK;
}
int acehip_memcpy_d2h(void* h, const void* d, size_t n, acehip_stream s) {
  HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_memcpy_d2d(void* d, const void* s_, size_t n, acehip_stream s) {
  HIP_TRY(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, (hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_memset(void* d, int v, size_t n, acehip_stream s) {
  HIP_TRY(hipMemsetAsync(d, v, n, (hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_stream_sync(acehip_stream s) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return ACEHIP_OK;
}
void* acehip_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    acehip_err_slot() = "hipEventCreate failed";
    return nullptr;
  }
  return (void*)e;
}
int acehip_event_record(void* e, acehip_stream s) {
  HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return ACEHIP_OK;
}
int acehip_event_elapsed_ms(void* a, void* b, float* ms) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)b));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return ACEHIP_OK;
}
int acehip_event_destroy(void* e) {
  HIP_TRY(hipEventDestroy((hipEvent_t)e));
  return ACEHIP_OK;
}

}  // extern "C"

int check_dev(acehip_ctx* c) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if (!c->on_device) return fail(ACEHIP_ENODEV, "context was created without a GPU; the HIP path has no CPU fallback");
  // every entry point launches on the context's own device, whichever device the calling thread last selected (a thread
  // may hold contexts on several GPUs; hipGetDevice is a thread-local read)
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != c->device) {
    if (hipSetDevice(c->device) != hipSuccess) return fail(ACEHIP_EHIP, "hipSetDevice failed");
  }
  acehip_stat_mult() = c->stat_reps ? c->stat_reps : c->seln;  // (stat_reps: the whole selection while for_replica_chunks narrows it)
  return ACEHIP_OK;
}
int check_range(acehip_ctx* c, uint32_t level, uint32_t pos0, uint32_t n) {
  if (int e = check_dev(c)) return e;
  if (level > c->hp.L) return fail(ACEHIP_EINVAL, "level exceeds the number of q primes");
  if (pos0 + n > level + c->hp.K) return fail(ACEHIP_EINVAL, "limb range exceeds level + K");
  return ACEHIP_OK;
}
int post_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ACEHIP_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return ACEHIP_OK;
}

// ---- replicas / limb ownership ----
DcList launch_dcs(const acehip_ctx* c) {
  DcList l;
  DevCtx d = c->dc;
  d.sh_world = c->sh_world;
  if (c->sh_world > 1 && c->rccl == nullptr) {  // simulated ranks: one launch set per hosted rank, replica h
    for (u32 h = 0; h < c->sh_hosted.size() && h < 16; ++h) {
      d.rep0 = h;
      d.nrep = 1;
      d.sh_rank = c->sh_hosted[h];
      l.d[l.n++] = d;
    }
    return l;
  }
  d.rep0 = c->sel0;
  d.nrep = c->seln;
  d.sh_rank = c->sh_world > 1 ? c->sh_hosted[0] : 0;
  l.d[l.n++] = d;
  return l;
}

void copy_limbs_dc(const DevCtx& dc, u64* dst, const u64* src, u32 n_limbs, u32 gi0, hipStream_t s) {
  HwBatchArgs cp;
  u32 n = 0;
  auto flush = [&] {
    if (n == 0) return;
    cp.seg_start[n] = (uint16_t)n;
    launch_hw_batch_ew(dc, cp, n, s);
    n = 0;
  };
  for (u32 i = 0; i < n_limbs; ++i) {
    if (!dc_owns(dc, gi0 + i)) continue;
    if (n == HW_BATCH_MAX) flush();
    cp.seg_start[n] = (uint16_t)n;
    cp.op[n++] = HwBatchOp{dst + (size_t)i * dc.N, src + (size_t)i * dc.N, nullptr, HW_OP_COPY, 0};
  }
  flush();
}

const DevModUp* get_modup(acehip_ctx* c, u32 level, u32 digit) {
  std::lock_guard<std::mutex> lk(c->mu);
  auto key = std::make_pair(level, digit);
  auto it = c->modup.find(key);
  if (it != c->modup.end()) return &it->second;
  HostParams::ModUp t = c->hp.modup(level, digit);
  DevModUp d;
  d.n2 = t.n2;
  d.nc = t.nc;
  d.start = t.start;
  std::vector<u32> src_gi(t.n2), pos(t.nc);
  for (u32 i = 0; i < t.n2; ++i) src_gi[i] = t.start + i;
  for (u32 j = 0; j < t.nc; ++j) pos[j] = t.compl_idx[j] < c->hp.L ? t.compl_idx[j] : level + (t.compl_idx[j] - c->hp.L);
  d.hat_inv = c->up(t.hat_inv);
  d.hat_inv_prec = c->up(t.hat_inv_prec);
  d.hat_mod = c->up(t.hat_mod);
  d.src_gi = c->up(src_gi);
  d.out_gi = c->up(t.compl_idx);
  d.out_pos = c->up(pos);
  if (!d.hat_inv || !d.hat_mod || !d.out_pos) return nullptr;
  return &(c->modup[key] = d);
}

extern "C" {

int acehip_stats(acehip_stat* out, int n, int reset) {
  acehip_stat* g_stat = acehip_stat_slots();
  for (int i = 0; i < n && i < ST_COUNT; ++i) out[i] = g_stat[i];
  if (reset) std::memset(g_stat, 0, sizeof(acehip_stat) * ST_COUNT);
  return ST_COUNT;
}
const char* acehip_stat_name(int i) { return i >= 0 && i < ST_COUNT ? kStatName[i] : nullptr; }

uint64_t acehip_key_switch_bytes(const acehip_ctx* c, uint32_t level) {
  const HostParams& hp = c->hp;
  const u64 b = hp.num_decomp(level);
  return 8ull * hp.N * (level + 2 * b * (level + hp.K) + 2 * level);
}

}  // extern "C"
