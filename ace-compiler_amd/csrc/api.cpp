// api.cpp -- C ABI (include/acehip.h) over the HIP kernels: context/table upload, workspace, and the
// host-side sequencing of Decomp_modup / Mod_down / Rescale / key-switch.
#include <unordered_map>
#include <atomic>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/acehip.h"
#include "host_params.hpp"
#include "kernels.hpp"

using namespace acehip;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(ACEHIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

static_assert(sizeof(PrimeConsts) == sizeof(DevPrime), "host/device prime layout mismatch");

namespace {

struct DevModUp {
  u32 n2 = 0, nc = 0, start = 0;
  u64 *hat_inv = nullptr, *hat_inv_prec = nullptr, *hat_mod = nullptr;
  u32 *src_gi = nullptr, *out_gi = nullptr, *out_pos = nullptr;
};

template <typename T>
T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (v.empty()) return nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

}  // namespace

// Per-level plan of the batched key-switch: one ConvDesc per digit (ModUp) and one for ModDown.
struct KsPlan {
  ConvDesc* d_descs = nullptr;  // [nd] ModUp problems, then [1] ModDown problem
  u32 nd = 0, max_nc = 0;
  // N = 2^16: the pre-factors of both base conversions ride in the last stage of the inverse NTT (NttFuse::inv_scale)
  u64* inv_up = nullptr;    // [level][4]
  u64* inv_down = nullptr;  // [K][4]
};

struct acehip_ctx {
  HostParams hp;
  bool on_device = false;
  int device = -1;
  DevCtx dc{};
  std::vector<void*> owned;  // device allocations freed at destroy
  // device CRT tables
  u64 *phat_inv = nullptr, *phat_inv_prec = nullptr, *phat_modq_t = nullptr, *pinv = nullptr, *pinv_prec = nullptr;
  u64 *ql_inv = nullptr, *ql_inv_prec = nullptr, *qlql = nullptr, *qlql_prec = nullptr;
  u32 *p_gi = nullptr;                     // [K] global indices of the p primes
  u32 *q_gi = nullptr, *q_pos = nullptr;   // [L] identity lists for ModDown targets
  std::mutex mu;
  std::map<std::pair<u32, u32>, DevModUp> modup;
  std::map<u32, u32*> auto_tabs;
  std::map<const void*, u32> auto_tab_k;   // device table -> automorphism index (hw_run_rotate: the kernel computes the index map)
  std::map<u32, KsPlan> ks_plans;
  // workspace (one per context; launches of one context are expected on one stream at a time)
  u64* ws = nullptr;
  size_t ws_words = 0;
  // encode (embed.hip): twiddles cos/sin(2 pi j / 2N) from the host libm, 5^i mod 2N, scratch, sticky overflow flag
  u64* hw_scratch = nullptr;       // acehip_hw_batch: private limbs for renamed intermediate versions
  size_t hw_scratch_limbs = 0;
  cd* emb_rou = nullptr;
  u32* emb_rot = nullptr;
  cd* emb_work = nullptr;
  int64_t* emb_msg = nullptr;
  int* emb_err = nullptr;
  std::map<std::pair<u64, u32>, u64*> enc_scales;  // (Delta, sf_degree) -> [L] Delta^(sf_degree-1) mod q_i

  template <typename T>
  T* up(const std::vector<T>& v) {
    T* d = upload(v);
    if (d) owned.push_back(d);
    return d;
  }
};

extern "C" {

const char* acehip_last_error(void) { return g_err.c_str(); }

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
    g_err = e.what();
    return nullptr;
  }
}

acehip_ctx* acehip_ctx_create(uint32_t N, uint32_t L, uint32_t q0_bits, uint32_t sf_bits, uint32_t dnum, int device) {
  if (acehip_device_count() <= device || device < 0) {
    g_err = "acehip_ctx_create: no such GPU device (the HIP path has no CPU fallback)";
    return nullptr;
  }
  acehip_ctx* ctx = acehip_ctx_create_host(N, L, q0_bits, sf_bits, dnum);
  if (!ctx) return nullptr;
  if (hipSetDevice(device) != hipSuccess) {
    g_err = "hipSetDevice failed";
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
  {  // ACEHIP_NTT_NARROW = largest launch (limb rows) that takes the narrow small-launch passes (0: never)
    const char* e = getenv("ACEHIP_NTT_NARROW");
    ctx->dc.ntt_narrow_max_rows = e ? (u32)strtoul(e, nullptr, 0) : 16u;
  }
  // workspace of the batched key-switch: coef (L) + ext[dnum] + two accumulators (L+K each) + tmp (2L)
  ctx->ws_words = ((size_t)hp.L * 3 + (size_t)(hp.dnum + 2) * T) * hp.N;
  if (hipMalloc(&ctx->ws, ctx->ws_words * sizeof(u64)) != hipSuccess || !ctx->dc.primes || !ctx->dc.tw_inv) {
    g_err = "acehip_ctx_create: device allocation/upload failed";
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
  for (void* p : ctx->owned) (void)hipFree(p);
  if (ctx->hw_scratch) (void)hipFree(ctx->hw_scratch);
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
    g_err = "acehip_auto_order: context has no device";
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
    g_err = "acehip_auto_order: upload failed";
    return nullptr;
  }
  c->auto_tabs[k] = d;
  c->auto_tab_k[d] = k;
  return d;
}

// ---- memory helpers ----
void* acehip_malloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    g_err = "hipMalloc failed";
    return nullptr;
  }
  return p;
}
int acehip_free(void* p) {
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
    g_err = "hipHostMalloc failed";
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
    g_err = "hipEventCreate failed";
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

// ---- argument checks shared by the launch entry points ----
// ---- call statistics: algorithmic bytes of SURVEY 8(d) per entry point (tables and scratch excluded) ----
enum { ST_NTT, ST_EW, ST_ROTATE, ST_MODUP, ST_KEYMAC, ST_MODDOWN, ST_RESCALE, ST_KEYSWITCH, ST_ENCODE, ST_ZERO_RUN, ST_COUNT };
static const char* const kStatName[ST_COUNT] = {"ntt", "elementwise", "rotate", "decomp_modup", "key_inner_product",
                                                "mod_down", "rescale", "key_switch", "encode", "zero_fill_executed"};
static thread_local acehip_stat g_stat[ST_COUNT];  // per host thread (= per image stream)
static inline void stat(int k, u64 units, u64 bytes) {
  g_stat[k].calls++;
  g_stat[k].units += units;
  g_stat[k].bytes += bytes;
}

static int check_dev(acehip_ctx* c) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if (!c->on_device) return fail(ACEHIP_ENODEV, "context was created without a GPU; the HIP path has no CPU fallback");
  // every entry point launches on the context's own device, whichever device the calling thread last selected (a thread
  // may hold contexts on several GPUs; hipGetDevice is a thread-local read)
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != c->device) {
    if (hipSetDevice(c->device) != hipSuccess) return fail(ACEHIP_EHIP, "hipSetDevice failed");
  }
  return ACEHIP_OK;
}
static int check_range(acehip_ctx* c, uint32_t level, uint32_t pos0, uint32_t n) {
  if (int e = check_dev(c)) return e;
  if (level > c->hp.L) return fail(ACEHIP_EINVAL, "level exceeds the number of q primes");
  if (pos0 + n > level + c->hp.K) return fail(ACEHIP_EINVAL, "limb range exceeds level + K");
  return ACEHIP_OK;
}
static int post_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ACEHIP_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return ACEHIP_OK;
}

static const DevModUp* get_modup(acehip_ctx* c, u32 level, u32 digit) {
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

// The base conversion can ride in the first pass of the following forward NTT (N = 2^16: ntt_fast.hip SRC_CONV*): the converted
// limbs are then never written / re-read in coefficient form (-104 MB of HBM traffic per C3 key-switch), but every output limb
// re-reads its alpha sources through L2 and the pass becomes VALU-bound.  Measured (MI355X, round 2): key-switch 0.256 ms
// either way, ResNet-20 1.51 images/s fused vs 1.56 unfused -- so it is OFF unless ACEHIP_CONV_FUSION=1 (bit-exact both ways,
// tests/test_gpu_parity.py::test_conv_fusion_matches).
static bool conv_fusable(const acehip_ctx* c, u32 n_in) {
  static const bool on = [] { const char* e = getenv("ACEHIP_CONV_FUSION"); return e && *e == '1'; }();
  return on && c->dc.logN == 16 && c->dc.split_bits <= 30 && n_in <= 12;
}

// workspace carving (in limbs of N words)
static u64* ws_at(acehip_ctx* c, size_t limb) { return c->ws + limb * c->hp.N; }
static const KsPlan* get_ks_plan(acehip_ctx* c, u32 level);
static int do_mod_down_n(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s);

static int do_decomp_modup(acehip_ctx* c, u64* out, const u64* in, u32 level, u32 digit, u64* scratch, hipStream_t s) {
  const HostParams& hp = c->hp;
  const DevModUp* t = get_modup(c, level, digit);
  if (!t) return fail(ACEHIP_EHIP, "ModUp table upload failed");
  const size_t N = hp.N;
  // digit limbs pass through unchanged (polynomial.c:1265-1273)
  HIP_TRY(hipMemcpyAsync(out + t->start * N, in + t->start * N, t->n2 * N * sizeof(u64), hipMemcpyDeviceToDevice, s));
  // iNTT of the digit limbs in scratch, scaled by (Q_d/q_i)^-1 mod q_i (polynomial.c:1276-1301)
  HIP_TRY(hipMemcpyAsync(scratch, in + t->start * N, t->n2 * N * sizeof(u64), hipMemcpyDeviceToDevice, s));
  // scratch limb i has prime start+i: run the iNTT as "positions [start, start+n2) of a level-L poly"
  launch_ntt(c->dc, scratch, hp.L, t->start, t->n2, true, s, t->start);
  launch_mul_const(c->dc, scratch, scratch, t->hat_inv, t->hat_inv_prec, t->src_gi, t->n2, s);
  // exact 128-bit sums + reduction into the complement limbs (polynomial.c:1302-1320)
  launch_base_conv(c->dc, out, scratch, t->hat_mod, t->out_gi, t->out_pos, t->n2, t->nc, t->nc, s);
  // NTT of the complement limbs (polynomial.c:1322-1329)
  launch_ntt(c->dc, out, level, 0, t->start, false, s);
  launch_ntt(c->dc, out, level, t->start + t->n2, level + hp.K - (t->start + t->n2), false, s);
  return post_launch();
}

static int do_mod_down(acehip_ctx* c, u64* out, const u64* in, u32 level, u64* scratch, hipStream_t s) {
  const HostParams& hp = c->hp;
  const size_t N = hp.N;
  // P part -> coefficient domain, times (P/p_j)^-1 mod p_j  (polynomial.c:941-945, 779-790)
  HIP_TRY(hipMemcpyAsync(scratch, in + level * N, hp.K * N * sizeof(u64), hipMemcpyDeviceToDevice, s));
  launch_ntt(c->dc, scratch, 0, 0, hp.K, true, s);  // level 0: position j -> prime p_j
  launch_mul_const(c->dc, scratch, scratch, c->phat_inv, c->phat_inv_prec, c->p_gi, hp.K, s);
  // conv P -> Q (polynomial.c:791-803); phat_modq_t is [K][L]: use its first `level` columns via n_out = L stride
  launch_base_conv(c->dc, out, scratch, c->phat_modq_t, c->q_gi, c->q_pos, hp.K, level, hp.L, s);
  launch_ntt(c->dc, out, level, 0, level, false, s);
  launch_moddown_tail(c->dc, out, in, c->pinv, c->pinv_prec, level, s);
  return post_launch();
}

extern "C" {

int acehip_ntt_forward(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  launch_ntt(c->dc, d, level, pos0, n, false, (hipStream_t)s);
  stat(ST_NTT, n, 16ull * c->hp.N * n);
  return post_launch();
}
int acehip_ntt_inverse(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  launch_ntt(c->dc, d, level, pos0, n, true, (hipStream_t)s);
  stat(ST_NTT, n, 16ull * c->hp.N * n);
  return post_launch();
}

int acehip_ntt_batch(acehip_ctx* c, uint64_t* d, size_t poly_stride, uint32_t n_polys, uint32_t level, uint32_t pos0,
                     uint32_t n, int inverse, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n_polys == 0) return ACEHIP_OK;
  if (n_polys > 65535) return fail(ACEHIP_EINVAL, "acehip_ntt_batch: at most 65535 polynomials per launch");
  launch_ntt(c->dc, d, level, pos0, n, inverse != 0, (hipStream_t)s, 0, n_polys, poly_stride);
  stat(ST_NTT, (u64)n * n_polys, 16ull * c->hp.N * n * n_polys);
  return post_launch();
}

static int ew(acehip_ctx* c, EwOp op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n, acehip_stream s) {
  if (c) stat(ST_EW, n, (op == EwOp::MulAdd ? 32ull : 24ull) * c->hp.N * n);
  if (int e = check_range(c, level, pos0, n)) return e;
  launch_ew(c->dc, op, r, a, b, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
int acehip_modadd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Add, r, a, b, level, pos0, n, s); }
int acehip_modsub(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Sub, r, a, b, level, pos0, n, s); }
int acehip_modmul(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::Mul, r, a, b, level, pos0, n, s); }
int acehip_modmuladd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) { return ew(c, EwOp::MulAdd, r, a, b, level, pos0, n, s); }

int acehip_rotate(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint32_t* perm, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (r == a) return fail(ACEHIP_EINVAL, "acehip_rotate: in-place rotation is not supported");
  launch_rotate(c->dc, r, a, perm, pos0, n, (hipStream_t)s);
  return post_launch();
}

int acehip_rotate_add2(acehip_ctx* c, uint64_t* r0, uint64_t* r1, const uint64_t* acc0, const uint64_t* acc1, const uint64_t* a0,
                       const uint64_t* a1, uint32_t auto_k, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (!r0 || !acc0 || !a0 || (r1 && (!acc1 || !a1))) return fail(ACEHIP_EINVAL, "acehip_rotate_add2: null operand");
  if (auto_k % 2 == 0 || auto_k >= 2 * c->hp.N) return fail(ACEHIP_EINVAL, "acehip_rotate_add2: automorphism index must be odd and below 2N");
  if (r0 == a0 || r0 == a1 || (r1 && (r1 == a0 || r1 == a1)))
    return fail(ACEHIP_EINVAL, "acehip_rotate_add2: the rotated operand must not alias a result");
  launch_rotate_add2(c->dc, r0, r1, acc0, acc1, a0, a1, auto_k, level, pos0, n, (hipStream_t)s);
  stat(ST_ROTATE, (r1 ? 2u : 1u) * n, (r1 ? 2ull : 1ull) * n * 24ull * c->hp.N);
  return post_launch();
}

// single-limb forms: the limb pointers are used directly; prime_gi selects the modulus
static int hw(acehip_ctx* c, EwOp op, u64* r, const u64* a, const u64* b, u32 gi, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 L = c->hp.L;
  if (gi >= L + c->hp.K) return fail(ACEHIP_EINVAL, "prime index out of range");
  if (gi < L) launch_ew(c->dc, op, r, a, b, L, gi, 1, (hipStream_t)s, gi);
  else launch_ew(c->dc, op, r, a, b, 0, gi - L, 1, (hipStream_t)s, gi - L);
  return post_launch();
}
int acehip_hw_modadd(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t gi, acehip_stream s) { return hw(c, EwOp::Add, r, a, b, gi, s); }
int acehip_hw_modmul(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* b, uint32_t gi, acehip_stream s) { return hw(c, EwOp::Mul, r, a, b, gi, s); }
int acehip_hw_rotate(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint32_t* perm, uint32_t gi, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (gi >= c->hp.L + c->hp.K) return fail(ACEHIP_EINVAL, "prime index out of range");
  if (r == a) return fail(ACEHIP_EINVAL, "acehip_hw_rotate: in-place rotation is not supported");
  launch_rotate(c->dc, r, a, perm, 0, 1, (hipStream_t)s);
  return post_launch();
}

// ---- acehip_hw_batch: a list of per-limb ops, executed as if issued one by one ----
namespace {
struct HwScratch {  // reused across calls: the shim flushes ~10k batches per ResNet-20 image
  struct Slot {
    u64 key;  // bucket + 1, 0 = empty
    u64 ptr;
    u32 node;
  };
  std::vector<Slot> table;
  std::vector<u32> parent, comp_of_node, ord, cnt, n_res, n_a, n_b, cur, last_pure;
  std::vector<u64> node_ptr;  // address of every node (limb)
  std::vector<char> written, dead, state, need, nostore;
  std::vector<u32> pos;  // place of every live op in the emission order
  std::vector<acehip_hw_op> sops;  // the list with ops on known-zero operands simplified
  std::vector<char> zero;
  std::vector<u32> kind, readers;  // effective op kind after fusion; reads of every node
};
thread_local HwScratch g_hw;

inline bool hw_has_a(u32 k) { return k != ACEHIP_HW_ZERO; }
inline bool hw_has_b(u32 k) {  // second operand is a limb in memory
  return k == ACEHIP_HW_ADD || k == ACEHIP_HW_MUL || k == ACEHIP_HW_SUB || k == ACEHIP_HW_MULADD;
}
inline bool hw_uses_prime(u32 k) { return hw_has_b(k) || k == ACEHIP_HW_MULC || k == ACEHIP_HW_ADDC; }

u32 uf_find(std::vector<u32>& p, u32 x) {
  while (p[x] != x) {
    p[x] = p[p[x]];
    x = p[x];
  }
  return x;
}

// node id of limb pointer `ptr` (one node per distinct limb), or UINT32_MAX if it partially overlaps a limb
// already seen (then the batch cannot be reordered and is issued op by op)
u32 hw_node(HwScratch& h, u64 ptr, u64 span, u64 mask) {
  const u64 b = ptr / span;
  u32 found = UINT32_MAX;
  for (int d = -1; d <= 1; ++d) {
    const u64 key = b + (u64)(int64_t)d + 1;
    if (key == 0) continue;
    for (u64 i = (key * 0x9E3779B97F4A7C15ull) >> 20;; ++i) {
      HwScratch::Slot& sl = h.table[i & mask];
      if (sl.key == 0) break;
      if (sl.key != key) continue;
      if (sl.ptr == ptr) found = sl.node;
      else if ((sl.ptr < ptr ? ptr - sl.ptr : sl.ptr - ptr) < span) return UINT32_MAX;
      break;  // at most one limb per bucket once partial overlaps are excluded
    }
  }
  if (found != UINT32_MAX) return found;
  const u32 node = (u32)h.parent.size();
  h.parent.push_back(node);
  h.node_ptr.push_back(ptr);
  for (u64 i = ((b + 1) * 0x9E3779B97F4A7C15ull) >> 20;; ++i) {
    HwScratch::Slot& sl = h.table[i & mask];
    if (sl.key == 0) {
      sl = HwScratch::Slot{b + 1, ptr, node};
      break;
    }
  }
  return node;
}

static u64* hw_scratch(acehip_ctx* c, size_t limbs);
// acehip_hw_batch_plan: the launches are recorded instead of issued (host-side test of the analysis, no GPU needed)
struct HwPlanSink {
  acehip_hw_op* ops;
  uint32_t* launch_id;
  uint32_t* seg_id;
  size_t cap, n;
  uint32_t launches;
  u64 scratch_base;
};
static thread_local HwPlanSink* g_plan = nullptr;
static void plan_append(const HwBatchOp& o, u32 seg) {
  HwPlanSink& p = *g_plan;
  if (p.n < p.cap) {
    p.ops[p.n] = acehip_hw_op{o.kind, o.gi, o.res, o.a, (const void*)o.b};
    p.launch_id[p.n] = p.launches;
    p.seg_id[p.n] = seg;
  }
  ++p.n;
}
// ACEHIP_HW_TRAFFIC=1: limb loads / stores the elementwise launches actually perform, per op kind (the kernel's forwarding
// rules replayed on the host), printed at exit -- where the bytes of the generated per-limb code go
static std::atomic<u64> g_hw_traffic[9][4];  // [kind][ops, limb loads, limb stores, segments started]
static bool hw_traffic_on() {
  static const bool on = [] {
    const bool v = getenv("ACEHIP_HW_TRAFFIC") != nullptr;
    if (v)
      atexit([] {
        static const char* const kn[9] = {"add", "mul", "rotate", "copy", "zero", "sub", "muladd", "mulc", "addc"};
        u64 tl = 0, ts = 0;
        for (int k = 0; k < 9; ++k) {
          const u64 o = g_hw_traffic[k][0], l = g_hw_traffic[k][1], w = g_hw_traffic[k][2];
          if (o) fprintf(stderr, "[hw traffic] %-7s ops %10llu  limb loads %10llu  limb stores %10llu\n", kn[k], (unsigned long long)o,
                         (unsigned long long)l, (unsigned long long)w);
          tl += l;
          ts += w;
        }
        fprintf(stderr, "[hw traffic] total limb loads %llu stores %llu\n", (unsigned long long)tl, (unsigned long long)ts);
      });
    return v;
  }();
  return on;
}
static void hw_traffic_count(const HwBatchArgs& args, u32 n_seg) {
  static std::atomic<u64> n_launch{0};
  static const u64 every = getenv("ACEHIP_HW_DUMP_EVERY") ? strtoull(getenv("ACEHIP_HW_DUMP_EVERY"), nullptr, 0) : 0;
  if (every && n_launch++ % every == 0) {  // a sample of launches, limbs numbered in order of appearance
    static const char* const kn[9] = {"add", "mul", "rot", "copy", "zero", "sub", "muladd", "mulc", "addc"};
    std::unordered_map<const void*, int> id;
    auto name = [&](const void* p) { return id.emplace(p, (int)id.size()).first->second; };
    std::string out = "[hw dump] launch " + std::to_string((u64)n_launch) + " segs " + std::to_string(n_seg) + "\n";
    for (u32 sgm = 0; sgm < n_seg; ++sgm) {
      out += "  seg:";
      for (u32 k = args.seg_start[sgm]; k < args.seg_start[sgm + 1]; ++k) {
        const HwBatchOp& o = args.op[k];
        char buf[96];
        const u32 kind = o.kind & HW_OP_KIND_MASK;
        const bool has_b = kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD;
        if (kind == HW_OP_ZERO) snprintf(buf, sizeof buf, " L%d=0", name(o.res));
        else if (has_b) snprintf(buf, sizeof buf, " L%d=%s(L%d,L%d)q%u", name(o.res), kn[kind], name(o.a), name(o.b), o.gi);
        else snprintf(buf, sizeof buf, " L%d=%s(L%d)q%u", name(o.res), kn[kind], name(o.a), o.gi);
        out += buf;
        if (o.kind & HW_OP_NOSTORE) out += "~";
      }
      out += "\n";
    }
    fputs(out.c_str(), stderr);
  }
  // what becomes of executed zero fills: the first later elementwise op that touches the limb (single-threaded runs only)
  static const bool zero_fate = getenv("ACEHIP_HW_ZERO_FATE") != nullptr;
  static std::unordered_map<const u64*, int> zeroed;
  static u64 fate_read[9], fate_over[9], fate_rezero;
  if (zero_fate) {
    static const bool reg = [] {
      atexit([] {
        static const char* const kn[9] = {"add", "mul", "rot", "copy", "zero", "sub", "muladd", "mulc", "addc"};
        for (int k = 0; k < 9; ++k)
          if (fate_read[k] || fate_over[k])
            fprintf(stderr, "[zero fate] first touched by %-7s: read %llu  overwritten %llu\n", kn[k], (unsigned long long)fate_read[k],
                    (unsigned long long)fate_over[k]);
        fprintf(stderr, "[zero fate] zeroed again %llu, never touched by an elementwise op %zu\n", (unsigned long long)fate_rezero, zeroed.size());
      });
      return true;
    }();
    (void)reg;
    for (u32 k = 0; k < args.seg_start[n_seg]; ++k) {
      const HwBatchOp& o = args.op[k];
      const u32 kind = o.kind & HW_OP_KIND_MASK;
      const bool has_b = kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD;
      bool rd = false;
      if (kind != HW_OP_ZERO && zeroed.erase(o.a)) rd = true;
      if (has_b && zeroed.erase(o.b)) rd = true;
      if (kind == HW_OP_MULADD && zeroed.erase(o.res)) rd = true;
      if (rd) fate_read[kind]++;
      if (kind == HW_OP_ZERO) {
        if (!zeroed.emplace(o.res, 1).second) fate_rezero++;
      } else if (zeroed.erase(o.res)) fate_over[kind]++;
    }
  }
  for (u32 sgm = 0; sgm < n_seg; ++sgm) {
    const u64* prev = nullptr;
    const u32 beg = args.seg_start[sgm], end = args.seg_start[sgm + 1];
    for (u32 k = beg; k < end; ++k) {
      const HwBatchOp& o = args.op[k];
      const u32 kind = o.kind & HW_OP_KIND_MASK;
      u64 loads = 0;
      if (kind != HW_OP_ZERO) {
        loads += o.a != prev;
        if (kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD) loads += o.b != prev;
        if (kind == HW_OP_MULADD) loads += o.res != prev;
      }
      const bool keep = (o.kind & HW_OP_NOSTORE) || (k + 1 < end && args.op[k + 1].res == o.res);
      g_hw_traffic[kind][0] += 1;
      g_hw_traffic[kind][1] += loads;
      g_hw_traffic[kind][2] += !keep;
      prev = o.res;
    }
  }
}
static void emit_ew(acehip_ctx* c, const HwBatchArgs& args, u32 n_seg, hipStream_t st) {
  if (hw_traffic_on()) hw_traffic_count(args, n_seg);
  if (!g_plan) {
    launch_hw_batch_ew(c->dc, args, n_seg, st);
    return;
  }
  for (u32 sgm = 0; sgm < n_seg; ++sgm)
    for (u32 k = args.seg_start[sgm]; k < args.seg_start[sgm + 1]; ++k) plan_append(args.op[k], sgm);
  ++g_plan->launches;
}
static void emit_rotate(acehip_ctx* c, const HwBatchArgs& args, u32 n_ops, hipStream_t st) {
  if (!g_plan) {
    launch_hw_batch_rotate(c->dc, args, n_ops, st);
    return;
  }
  if (n_ops == 0) return;
  for (u32 k = 0; k < n_ops; ++k) plan_append(args.op[k], k);  // every gather is its own segment
  ++g_plan->launches;
}
inline bool limbs_overlap(const void* x, const void* y, u64 span) {
  const u64 a = (u64)x, b = (u64)y;
  return (a < b ? b - a : a - b) < span;
}

// One op per launch, in the caller's order (lists with partially overlapping limbs -- nothing Coeffs() can produce).
// An operand that overlaps the result without being the same limb is read from a private copy taken before the op:
// the op sees the operand as it was, which is what the reference's ascending-index loop (poly_arith.c:14-39) sees
// whenever the operand lies above the result; an operand overlapping from below would be a loop-carried dependence
// there, which this interface does not reproduce (documented in include/acehip.h).
void hw_issue_one(acehip_ctx* c, const acehip_hw_op& o, hipStream_t st) {
  const u64 span = (u64)c->hp.N * 8;
  HwBatchArgs args;
  args.seg_start[0] = 0;
  args.seg_start[1] = 1;
  const u64* a = o.a;
  const u64* b = (const u64*)o.b;
  const bool has_a = o.op != ACEHIP_HW_ZERO;
  const bool has_b = o.op == ACEHIP_HW_ADD || o.op == ACEHIP_HW_MUL || o.op == ACEHIP_HW_SUB || o.op == ACEHIP_HW_MULADD;
  u64* scratch = nullptr;
  for (int which = 0; which < 2; ++which) {
    const u64*& src = which == 0 ? a : b;
    if (!(which == 0 ? has_a : has_b) || o.op == ACEHIP_HW_ROTATE) continue;
    if (src == o.res || !limbs_overlap(src, o.res, span)) continue;
    if (!scratch) scratch = hw_scratch(c, 2);
    if (!scratch) continue;  // no memory: run as is
    u64* priv = scratch + (size_t)which * c->hp.N;
    args.op[0] = HwBatchOp{priv, src, nullptr, HW_OP_COPY, 0};
    emit_ew(c, args, 1, st);
    src = priv;
  }
  args.op[0] = HwBatchOp{o.res, a, o.op == ACEHIP_HW_MULC || o.op == ACEHIP_HW_ADDC || o.op == ACEHIP_HW_ROTATE ? (const u64*)o.b : b,
                         o.op, o.prime_gi};
  if (o.op == ACEHIP_HW_ROTATE) emit_rotate(c, args, 1, st);
  else emit_ew(c, args, 1, st);
}

// scratch limbs for renamed intermediate versions (see hw_run_ew); grown on demand, owned by the context
static u64* hw_scratch(acehip_ctx* c, size_t limbs) {
  if (g_plan) return (u64*)g_plan->scratch_base;  // recording: addresses only
  if (limbs <= c->hw_scratch_limbs) return c->hw_scratch;
  size_t want = std::max<size_t>(256, c->hw_scratch_limbs);
  while (want < limbs) want *= 2;
  (void)hipDeviceSynchronize();  // launches that still use the old arena
  if (c->hw_scratch) (void)hipFree(c->hw_scratch);
  c->hw_scratch = nullptr;
  c->hw_scratch_limbs = 0;
  void* p = nullptr;
  if (hipMalloc(&p, want * c->hp.N * sizeof(u64)) != hipSuccess) return nullptr;
  c->hw_scratch = (u64*)p;
  c->hw_scratch_limbs = want;
  return c->hw_scratch;
}
constexpr size_t kHwScratchMaxLimbs = 2048;

// Memory the caller does not need after the batch (acehip_hw_batch_discard): sorted, disjoint [start, end) byte ranges,
// minus the limbs a later run of the same batch still reads (`keep`).
struct HwDead {
  const std::pair<u64, u64>* range;
  size_t n_range;
  const u64* keep;
  size_t n_keep;
  bool in_range(u64 ptr, u64 span) const {
    size_t lo = 0, hi = n_range;
    while (lo < hi) {  // last range that starts at or below ptr
      const size_t mid = (lo + hi) / 2;
      if (range[mid].first <= ptr) lo = mid + 1;
      else hi = mid;
    }
    return lo > 0 && ptr + span <= range[lo - 1].second;
  }
  bool limb_is_dead(u64 ptr, u64 span) const {
    if (!in_range(ptr, span)) return false;
    for (size_t i = 0; i < n_keep; ++i)
      if (keep[i] == ptr) return false;
    return true;
  }
};

// Elementwise run ops[0, m).  The list is executed as if one by one, but:
//  * an op whose result is completely rewritten later in the list before anything reads it is dropped (Alloc_poly and
//    Init_ciph_* zero-fill every result that the next Hw_* loop overwrites), and so is one whose result lies in memory
//    the caller has given up (`dead`: temporaries freed while the list was queued) and is read by nothing that follows;
//  * a result that only the next op of its chain reads, and that nobody needs afterwards, is not stored (HW_OP_NOSTORE);
//  * a limb that is purely overwritten several times (generated code funnels every limb of a key inner product
//    through ONE scratch limb: resnet20_cifar10_pre.onnx.inc:7011-7036) gets a private scratch limb for each
//    version but the last, which removes the false write-after-read / write-after-write dependencies;
//  * ops are then grouped into chains = connected components over limbs that some op writes, program order
//    kept inside a chain, and each chain segment runs in one blockIdx.y of hw_batch_ew_kernel.
void hw_run_ew(acehip_ctx* c, const acehip_hw_op* ops, size_t m, hipStream_t st, const HwDead* dead_mem) {
  HwScratch& h = g_hw;
  const u64 span = (u64)c->hp.N * 8;
  size_t cap = 64;
  while (cap < 6 * m) cap <<= 1;
  if (h.table.size() < cap) h.table.resize(cap);
  std::memset(h.table.data(), 0, cap * sizeof(HwScratch::Slot));
  const u64 mask = cap - 1;
  h.parent.clear();
  h.node_ptr.clear();
  h.n_res.resize(m);
  h.n_a.resize(m);
  h.n_b.resize(m);
  for (size_t k = 0; k < m; ++k) {
    const acehip_hw_op& o = ops[k];
    const u32 nr = hw_node(h, (u64)o.res, span, mask);
    const u32 na = hw_has_a(o.op) ? hw_node(h, (u64)o.a, span, mask) : 0;
    const u32 nb = hw_has_b(o.op) ? hw_node(h, (u64)o.b, span, mask) : 0;
    if (nr == UINT32_MAX || na == UINT32_MAX || nb == UINT32_MAX) {
      // partially overlapping limbs: keep the caller's order, one launch per op
      for (size_t j = 0; j < m; ++j) hw_issue_one(c, ops[j], st);
      return;
    }
    h.n_res[k] = nr;
    h.n_a[k] = na;
    h.n_b[k] = nb;
  }
  const u32 n_base = (u32)h.parent.size();
  // forwards: operands that a zero fill of this list has just cleared.  0 + x = x, 0 * x = 0, 0 + a*b = a*b on residues:
  // the op becomes a copy / fill / plain product, the fill loses its reader and usually dies in the backward pass below
  // (an accumulator is zero-filled by Init_ciph_* and meets its first addend much later: lazy fills of the runtime)
  {
    bool any_zero = false;
    for (size_t k = 0; k < m && !any_zero; ++k) any_zero = ops[k].op == ACEHIP_HW_ZERO;
    if (any_zero) {
      h.sops.assign(ops, ops + m);
      h.zero.assign(n_base, 0);
      for (size_t k = 0; k < m; ++k) {
        acehip_hw_op& o = h.sops[k];
        const u32 nr = h.n_res[k];
        const bool za = hw_has_a(o.op) && h.zero[h.n_a[k]], zb = hw_has_b(o.op) && h.zero[h.n_b[k]];
        auto to_copy_of = [&](const u64* src, u32 node) {  // res = src (left alone when that would be res = res)
          if (node == nr) return;
          o.op = ACEHIP_HW_COPY;
          o.a = src;
          o.b = nullptr;
          h.n_a[k] = node;
        };
        switch (o.op) {
          case ACEHIP_HW_COPY:
            if (za) o.op = ACEHIP_HW_ZERO;
            break;
          case ACEHIP_HW_ADD:
            if (za && zb) o.op = ACEHIP_HW_ZERO;
            else if (za) to_copy_of((const u64*)o.b, h.n_b[k]);
            else if (zb) to_copy_of(o.a, h.n_a[k]);
            break;
          case ACEHIP_HW_SUB:
            if (za && zb) o.op = ACEHIP_HW_ZERO;
            else if (zb) to_copy_of(o.a, h.n_a[k]);
            break;
          case ACEHIP_HW_MUL:
            if (za || zb) o.op = ACEHIP_HW_ZERO;
            break;
          case ACEHIP_HW_MULC:
            if (za) o.op = ACEHIP_HW_ZERO;
            break;
          case ACEHIP_HW_MULADD:
            if (!za && !zb && h.zero[nr]) o.op = ACEHIP_HW_MUL;
            break;
          default:
            break;
        }
        h.zero[nr] = o.op == ACEHIP_HW_ZERO;
      }
      ops = h.sops.data();
    }
  }
  auto pure_overwrite = [&](size_t k) {  // writes its result limb without reading it
    const u32 op = ops[k].op, nr = h.n_res[k];
    return op != ACEHIP_HW_MULADD && !(hw_has_a(op) && h.n_a[k] == nr) && !(hw_has_b(op) && h.n_b[k] == nr);
  };
  // backwards: dead stores, and the last pure overwrite of every limb (the version that stays in place)
  h.dead.assign(m, 0);
  h.state.assign(n_base, 0);  // 1 = overwritten by a later op (or given up by the caller) with no read in between
  if (dead_mem)
    for (u32 i = 0; i < n_base; ++i) h.state[i] = dead_mem->limb_is_dead(h.node_ptr[i], span);
  h.need.assign(h.state.begin(), h.state.end());  // kept for the store analysis below: 1 = not needed after the list
  h.last_pure.assign(n_base, UINT32_MAX);
  size_t live = m;
  bool rename_useful = false;
  for (size_t k = m; k-- > 0;) {
    const u32 op = ops[k].op, nr = h.n_res[k];
    if (h.state[nr]) {
      h.dead[k] = 1;
      --live;
      continue;
    }
    const bool pure = pure_overwrite(k);
    if (pure) {
      if (h.last_pure[nr] == UINT32_MAX) h.last_pure[nr] = (u32)k;
      else rename_useful = true;
    }
    h.state[nr] = pure;
    if (hw_has_a(op)) h.state[h.n_a[k]] = 0;
    if (hw_has_b(op)) h.state[h.n_b[k]] = 0;
  }
  if (!g_plan) {  // zero fills that survive (their overwrite, if any, is not in this list): units = limbs, a subset of "elementwise"
    u64 z = 0;
    for (size_t k = 0; k < m; ++k) z += !h.dead[k] && ops[k].op == ACEHIP_HW_ZERO;
    if (z) stat(ST_ZERO_RUN, z, z * span);
  }
  if (live == 0) return;
  // forwards: give intermediate versions private scratch limbs
  size_t n_scratch = 0;
  if (rename_useful) {
    h.cur.resize(n_base);
    for (u32 i = 0; i < n_base; ++i) h.cur[i] = i;
    h.state.assign(n_base, 0);  // reused: 1 = the limb was touched earlier in the list
    for (size_t k = 0; k < m; ++k) {
      if (h.dead[k]) continue;
      const u32 op = ops[k].op, nr0 = h.n_res[k];
      const bool pure = pure_overwrite(k);
      const u32 na = hw_has_a(op) ? h.cur[h.n_a[k]] : 0, nb = hw_has_b(op) ? h.cur[h.n_b[k]] : 0;
      if (hw_has_a(op)) h.state[h.n_a[k]] = 1;
      if (hw_has_b(op)) h.state[h.n_b[k]] = 1;
      if (pure) {
        if (h.last_pure[nr0] != k && h.state[nr0] && n_scratch < kHwScratchMaxLimbs) {
          h.cur[nr0] = (u32)h.parent.size();
          h.parent.push_back(h.cur[nr0]);
          h.node_ptr.push_back(n_scratch++);  // index into the scratch arena, resolved below
        } else {
          h.cur[nr0] = nr0;
        }
      }
      h.state[nr0] = 1;
      h.n_res[k] = h.cur[nr0];
      h.n_a[k] = na;
      h.n_b[k] = nb;
    }
    if (n_scratch) {
      u64* base = hw_scratch(c, n_scratch);
      if (!base) {  // no memory for the arena: run the list in order instead
        for (size_t j = 0; j < m; ++j) hw_issue_one(c, ops[j], st);
        return;
      }
      for (size_t i = n_base; i < h.node_ptr.size(); ++i) h.node_ptr[i] = (u64)base + h.node_ptr[i] * span;
    }
  }
  const u32 n_nodes = (u32)h.parent.size();
  h.kind.resize(m);
  for (size_t k = 0; k < m; ++k) h.kind[k] = ops[k].op;
  // fuse  t = a*b ; acc = acc + t  into  acc += a*b  when t is a private scratch version nobody else reads: the product
  // never goes to memory and runs of such pairs on one accumulator keep it in registers (hw_batch_ew_kernel)
  if (n_nodes > n_base) {
    h.readers.assign(n_nodes, 0);
    for (size_t k = 0; k < m; ++k) {
      if (h.dead[k]) continue;
      if (hw_has_a(ops[k].op)) h.readers[h.n_a[k]]++;
      if (hw_has_b(ops[k].op)) h.readers[h.n_b[k]]++;
      if (ops[k].op == ACEHIP_HW_MULADD) h.readers[h.n_res[k]]++;
    }
    size_t prev = SIZE_MAX;  // previous live op
    for (size_t k = 0; k < m; ++k) {
      if (h.dead[k]) continue;
      const size_t p = prev;
      prev = k;
      if (p != SIZE_MAX && ops[k].op == ACEHIP_HW_COPY && h.n_a[k] == h.n_res[p] && h.n_res[p] >= n_base && h.readers[h.n_res[p]] == 1 &&
          h.n_res[k] != h.n_res[p]) {
        // res = copy of a private version that only this copy reads: the producer writes res itself
        h.n_res[p] = h.n_res[k];
        h.dead[k] = 1;
        --live;
        prev = p;
        continue;
      }
      if (p == SIZE_MAX || ops[k].op != ACEHIP_HW_ADD || ops[p].op != ACEHIP_HW_MUL) continue;
      const u32 t = h.n_res[p], acc = h.n_res[k];
      if (t < n_base || h.readers[t] != 1 || ops[p].prime_gi != ops[k].prime_gi) continue;
      const bool acc_a = h.n_a[k] == acc && h.n_b[k] == t, acc_b = h.n_b[k] == acc && h.n_a[k] == t;
      if (!(acc_a || acc_b) || acc == t) continue;
      h.kind[k] = ACEHIP_HW_MULADD;
      h.n_a[k] = h.n_a[p];
      h.n_b[k] = h.n_b[p];
      h.dead[p] = 1;
      --live;
    }
  }
  h.written.assign(n_nodes, 0);
  for (size_t k = 0; k < m; ++k)
    if (!h.dead[k]) h.written[h.n_res[k]] = 1;
  for (size_t k = 0; k < m; ++k) {
    if (h.dead[k]) continue;
    const u32 r = uf_find(h.parent, h.n_res[k]);
    if (hw_has_a(h.kind[k]) && h.written[h.n_a[k]]) h.parent[uf_find(h.parent, h.n_a[k])] = r;
    if (hw_has_b(h.kind[k]) && h.written[h.n_b[k]]) h.parent[uf_find(h.parent, h.n_b[k])] = uf_find(h.parent, r);
  }
  // chains numbered by first appearance; ops of a chain keep their program order
  h.comp_of_node.assign(n_nodes, UINT32_MAX);
  h.cnt.clear();
  for (size_t k = 0; k < m; ++k) {
    if (h.dead[k]) continue;
    const u32 root = uf_find(h.parent, h.n_res[k]);
    if (h.comp_of_node[root] == UINT32_MAX) {
      h.comp_of_node[root] = (u32)h.cnt.size();
      h.cnt.push_back(0);
    }
    h.cnt[h.comp_of_node[root]]++;
  }
  u32 run = 0;
  for (auto& x : h.cnt) {  // counts -> start offsets
    const u32 t = x;
    x = run;
    run += t;
  }
  h.ord.resize(live);
  for (size_t k = 0; k < m; ++k)
    if (!h.dead[k]) h.ord[h.cnt[h.comp_of_node[uf_find(h.parent, h.n_res[k])]]++] = (u32)k;
  // after the scatter cnt[j] = end offset of chain j
  // Which results have to reach memory: walking the list backwards, need[x] = the version of limb x that is current here is
  // loaded by a later op (an operand is taken from registers only when the op right before its reader, in the same
  // chain and launch, produced it: hw_batch_ew_kernel) or outlives the list.
  h.pos.resize(m);
  for (size_t t = 0; t < live; ++t) h.pos[h.ord[t]] = (u32)t;
  {
    const size_t nb0 = h.need.size();  // base limbs: needed afterwards unless the caller gave them up; scratch versions: never
    for (size_t i = 0; i < nb0; ++i) h.need[i] = !h.need[i];
    h.need.resize(n_nodes, 0);
  }
  h.nostore.assign(m, 0);
  {
    std::vector<u32>& chain_of = h.cur;  // reused: chain of every emission place
    chain_of.resize(live);
    u32 ch = 0;
    for (size_t t = 0; t < live; ++t) {
      while (t >= h.cnt[ch]) ++ch;
      chain_of[t] = ch;
    }
    for (size_t k = m; k-- > 0;) {
      if (h.dead[k]) continue;
      const u32 kind = h.kind[k], nr = h.n_res[k], t = h.pos[k];
      h.nostore[k] = !h.need[nr];
      h.need[nr] = 0;
      const bool fwd = t > 0 && t % HW_BATCH_MAX != 0 && chain_of[t - 1] == chain_of[t];
      const u32 prev_res = fwd ? h.n_res[h.ord[t - 1]] : UINT32_MAX;
      if (hw_has_a(kind) && h.n_a[k] != prev_res) h.need[h.n_a[k]] = 1;
      if (hw_has_b(kind) && h.n_b[k] != prev_res) h.need[h.n_b[k]] = 1;
      if (kind == ACEHIP_HW_MULADD && nr != prev_res) h.need[nr] = 1;
    }
  }
  HwBatchArgs args;
  u32 n_ops = 0, n_seg = 0, chain = 0, prev_chain = UINT32_MAX;
  args.seg_start[0] = 0;
  for (size_t t = 0; t < live; ++t) {
    while (t >= h.cnt[chain]) ++chain;
    if (n_ops == HW_BATCH_MAX) {  // a chain cut here continues in the next launch, which is ordered after this one
      args.seg_start[++n_seg] = (uint16_t)n_ops;
      emit_ew(c, args, n_seg, st);
      n_ops = 0;
      n_seg = 0;
      prev_chain = UINT32_MAX;
    }
    if (n_ops && chain != prev_chain) args.seg_start[++n_seg] = (uint16_t)n_ops;
    prev_chain = chain;
    const u32 k = h.ord[t];
    const acehip_hw_op& o = ops[k];
    const u32 kind = h.kind[k];
    args.op[n_ops++] = HwBatchOp{(u64*)h.node_ptr[h.n_res[k]], hw_has_a(kind) ? (const u64*)h.node_ptr[h.n_a[k]] : nullptr,
                                 hw_has_b(kind) ? (const u64*)h.node_ptr[h.n_b[k]] : (const u64*)o.b,
                                 kind | (h.nostore[k] ? HW_OP_NOSTORE : 0u), o.prime_gi};
  }
  args.seg_start[++n_seg] = (uint16_t)n_ops;
  emit_ew(c, args, n_seg, st);
}

// rotation run: gathers are independent unless a result aliases a source or result of the same launch
void hw_run_rotate(acehip_ctx* c, const acehip_hw_op* ops, size_t m, hipStream_t st) {
  const u64 span = (u64)c->hp.N * 8;
  HwBatchArgs args;
  u32 n_ops = 0;
  size_t first = 0;
  for (size_t k = 0; k < m; ++k) {
    bool cut = n_ops == HW_BATCH_MAX;
    for (size_t j = first; j < k && !cut; ++j)
      cut = limbs_overlap(ops[k].res, ops[j].res, span) || limbs_overlap(ops[k].res, ops[j].a, span) ||
            limbs_overlap(ops[k].a, ops[j].res, span);
    if (cut) {
      emit_rotate(c, args, n_ops, st);
      n_ops = 0;
      first = k;
    }
    // a table this context built is a known automorphism k: the kernel computes perm[i] = rev(((2 rev(i) + 1) k mod 2N) / 2)
    // itself instead of loading 4 bytes per coefficient (gi carries k; 0 = load the caller's table)
    u32 auto_k = 0;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      auto it = c->auto_tab_k.find(ops[k].b);
      if (it != c->auto_tab_k.end()) auto_k = it->second;
    }
    args.op[n_ops++] = HwBatchOp{ops[k].res, ops[k].a, (const u64*)ops[k].b, ops[k].op, auto_k};
  }
  emit_rotate(c, args, n_ops, st);
}
}  // namespace

static int hw_batch_run(acehip_ctx* c, const acehip_hw_op* ops, size_t n, hipStream_t st, const acehip_hw_range* dead = nullptr,
                        size_t n_dead = 0) {
  if (n == 0) return ACEHIP_OK;
  if (!ops) return fail(ACEHIP_EINVAL, "acehip_hw_batch: null op list");
  if (n_dead && !dead) return fail(ACEHIP_EINVAL, "acehip_hw_batch_discard: null range list");
  const u32 T = c->hp.L + c->hp.K;
  const u64 span = (u64)c->hp.N * 8;
  // limbs moved per op (24N per add/mul, 16N per rotate; copy 16N, zero 8N, muladd 32N, scalar forms 16N)
  static const u64 kHwWords[9] = {3, 3, 2, 2, 1, 3, 4, 2, 2};
  u64 alg_words = 0, n_rot = 0;
  for (size_t k = 0; k < n; ++k) {
    const acehip_hw_op& o = ops[k];
    if (o.op > ACEHIP_HW_ADDC) return fail(ACEHIP_EINVAL, "acehip_hw_batch: unknown op");
    alg_words += kHwWords[o.op];
    n_rot += o.op == ACEHIP_HW_ROTATE;
    if (!o.res || (hw_has_a(o.op) && !o.a) || ((hw_has_b(o.op) || o.op == ACEHIP_HW_ROTATE) && !o.b))
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: null operand");
    if (hw_uses_prime(o.op) && o.prime_gi >= T) return fail(ACEHIP_EINVAL, "prime index out of range");
    if ((o.op == ACEHIP_HW_MULC || o.op == ACEHIP_HW_ADDC) && (u64)(uintptr_t)o.b >= c->hp.primes[o.prime_gi].q)
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: scalar operand is not a residue of the prime");
    if (o.op == ACEHIP_HW_ROTATE && limbs_overlap(o.res, o.a, span))
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: in-place rotation is not supported");
  }
  // the list runs as alternating rotation / elementwise runs.  Memory the caller gave up is dead for a run only where no
  // later run reads it: walking the runs backwards, keep[0, run.n_keep) = the given-up limbs that runs after it read
  static thread_local std::vector<std::pair<u64, u64>> ranges;
  static thread_local std::vector<u64> keep;
  struct Run {
    size_t i, j, n_keep;
  };
  static thread_local std::vector<Run> runs;
  runs.clear();
  for (size_t i = 0; i < n;) {
    size_t j = i;
    const bool rot = ops[i].op == ACEHIP_HW_ROTATE;
    while (j < n && (ops[j].op == ACEHIP_HW_ROTATE) == rot) ++j;
    runs.push_back(Run{i, j, 0});
    i = j;
  }
  HwDead dm{nullptr, 0, nullptr, 0};
  if (n_dead) {
    ranges.clear();
    for (size_t r = 0; r < n_dead; ++r)
      if (dead[r].ptr && dead[r].words) ranges.emplace_back((u64)dead[r].ptr, (u64)dead[r].ptr + (u64)dead[r].words * 8);
    std::sort(ranges.begin(), ranges.end());
    for (size_t r = 1; r < ranges.size(); ++r)
      if (ranges[r].first < ranges[r - 1].second) return fail(ACEHIP_EINVAL, "acehip_hw_batch_discard: overlapping ranges");
    dm.range = ranges.data();
    dm.n_range = ranges.size();
    keep.clear();
    for (size_t r = runs.size(); r-- > 0;) {
      runs[r].n_keep = keep.size();
      if (r == 0) break;
      for (size_t k = runs[r].i; k < runs[r].j; ++k) {
        const acehip_hw_op& o = ops[k];
        if (hw_has_a(o.op) && dm.in_range((u64)o.a, span)) keep.push_back((u64)o.a);
        if (hw_has_b(o.op) && dm.in_range((u64)o.b, span)) keep.push_back((u64)o.b);
        if (o.op == ACEHIP_HW_MULADD && dm.in_range((u64)o.res, span)) keep.push_back((u64)o.res);
      }
    }
  }
  if (hw_traffic_on() && getenv("ACEHIP_HW_ROT_FATE")) {  // diagnostic: who reads the result of a queued rotation
    static std::atomic<u64> cnt[6];  // rotations; result given up; readers in this list: 0, 1, 2+; source rewritten later in the list
    static const bool reg = [] {
      atexit([] {
        fprintf(stderr, "[rot fate] rotations %llu, result in given-up memory %llu; read in the same list by 0 / 1 / 2+ ops: %llu / %llu / %llu; source written later in the list %llu\n",
                (unsigned long long)cnt[0], (unsigned long long)cnt[1], (unsigned long long)cnt[2], (unsigned long long)cnt[3],
                (unsigned long long)cnt[4], (unsigned long long)cnt[5]);
      });
      return true;
    }();
    (void)reg;
    for (size_t k = 0; k < n; ++k) {
      if (ops[k].op != ACEHIP_HW_ROTATE) continue;
      cnt[0]++;
      cnt[1] += dm.n_range && dm.in_range((u64)ops[k].res, span);
      u32 readers = 0;
      bool src_written = false;
      for (size_t j = k + 1; j < n; ++j) {
        const acehip_hw_op& o = ops[j];
        readers += hw_has_a(o.op) && o.a == ops[k].res;
        readers += hw_has_b(o.op) && o.b == (const void*)ops[k].res;
        readers += o.op == ACEHIP_HW_MULADD && o.res == ops[k].res;
        src_written |= o.res == ops[k].a;
        if (o.res == ops[k].res && o.op != ACEHIP_HW_MULADD) break;
      }
      cnt[2 + std::min(readers, 2u)]++;
      cnt[5] += src_written;
    }
  }
  for (const Run& r : runs) {
    if (ops[r.i].op == ACEHIP_HW_ROTATE) {
      hw_run_rotate(c, ops + r.i, r.j - r.i, st);
      continue;
    }
    // (a long keep list would make the per-limb lookup slow: such a run is analysed without the caller's hint)
    const bool hint = dm.n_range && r.n_keep <= 256;
    dm.keep = keep.data();
    dm.n_keep = r.n_keep;
    hw_run_ew(c, ops + r.i, r.j - r.i, st, hint ? &dm : nullptr);
  }
  if (!g_plan) {
    stat(ST_EW, n - n_rot, (alg_words - 2 * n_rot) * span);
    if (n_rot) {
      stat(ST_ROTATE, n_rot, 2 * n_rot * span);
      g_stat[ST_ROTATE].calls--;  // one entry point call, counted under elementwise
    }
  }
  return ACEHIP_OK;
}

int acehip_hw_batch(acehip_ctx* c, const acehip_hw_op* ops, size_t n, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  (void)hipSetDevice(c->device);
  if (int e = hw_batch_run(c, ops, n, (hipStream_t)s)) return e;
  return post_launch();
}

int acehip_hw_batch_discard(acehip_ctx* c, const acehip_hw_op* ops, size_t n, const acehip_hw_range* dead, size_t n_dead, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  (void)hipSetDevice(c->device);
  if (int e = hw_batch_run(c, ops, n, (hipStream_t)s, dead, n_dead)) return e;
  return post_launch();
}

long acehip_hw_batch_plan(acehip_ctx* c, const acehip_hw_op* ops, size_t n, acehip_hw_op* out_ops, uint32_t* out_launch,
                          uint32_t* out_segment, size_t cap, uint64_t scratch_base) {
  return acehip_hw_batch_plan_discard(c, ops, n, nullptr, 0, out_ops, out_launch, out_segment, cap, scratch_base);
}

long acehip_hw_batch_plan_discard(acehip_ctx* c, const acehip_hw_op* ops, size_t n, const acehip_hw_range* dead, size_t n_dead,
                                  acehip_hw_op* out_ops, uint32_t* out_launch, uint32_t* out_segment, size_t cap, uint64_t scratch_base) {
  if (!c) return fail(ACEHIP_EINVAL, "null context");
  if ((cap && (!out_ops || !out_launch || !out_segment)) || !scratch_base) return fail(ACEHIP_EINVAL, "acehip_hw_batch_plan: bad output arguments");
  HwPlanSink sink{out_ops, out_launch, out_segment, cap, 0, 0, scratch_base};
  g_plan = &sink;
  const int e = hw_batch_run(c, ops, n, nullptr, dead, n_dead);
  g_plan = nullptr;
  if (e) return e;
  return (long)sink.n;
}

int acehip_decomp_modup(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_decomp_modup: bad level/digit");
  (void)hipSetDevice(c->device);
  stat(ST_MODUP, 1, 8ull * c->hp.N * (std::min(c->hp.alpha, level - c->hp.alpha * digit) + level + c->hp.K));
  return do_decomp_modup(c, out, in, level, digit, ws_at(c, 0), (hipStream_t)s);
}

int acehip_mod_down(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_mod_down: bad level");
  if (out == in) return fail(ACEHIP_EINVAL, "acehip_mod_down: out must not alias in");
  if (c->dc.logN == 16) return do_mod_down_n(c, out, nullptr, in, nullptr, level, (hipStream_t)s);
  stat(ST_MODDOWN, 1, 8ull * c->hp.N * (2 * level + c->hp.K));
  return do_mod_down(c, out, in, level, ws_at(c, 0), (hipStream_t)s);
}

// Mod_down of one or two extended polynomials (the two accumulators of a key-switch) in the same launches
static int do_mod_down_n(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s) {
  const HostParams& hp = c->hp;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const u32 np = in1 ? 2 : 1;
  const size_t N = hp.N, PK = (size_t)hp.K * N, QL = (size_t)level * N;
  u64* pc = c->ws;            // [2][K][N] p-limbs in the coefficient domain
  u64* tmp = pc + 2 * PK;     // [2][level][N]
  if (c->dc.logN == 16) {
    NttFuse fi;
    fi.src0 = in0 + QL;
    fi.src1 = in1 ? in1 + QL : nullptr;
    fi.inv_scale = plan->inv_down;  // (P/p_j)^-1 folded into the last inverse stage
    launch_ntt_fused(c->dc, pc, 0, 0, hp.K, true, s, 0, np, PK, 0, fi);  // level 0: position j -> prime p_j
  } else {
    HIP_TRY(hipMemcpyAsync(pc, in0 + QL, PK * sizeof(u64), hipMemcpyDeviceToDevice, s));
    if (in1) HIP_TRY(hipMemcpyAsync(pc + PK, in1 + QL, PK * sizeof(u64), hipMemcpyDeviceToDevice, s));
    launch_ntt(c->dc, pc, 0, 0, hp.K, true, s, 0, np, PK);
  }
  // the ModDown descriptor reads source limbs at positions level.. : hand it a base `level` limbs below pc
  const bool conv_in_ntt = conv_fusable(c, hp.K);
  if (!conv_in_ntt) launch_base_conv_batch(c->dc, tmp, QL, pc - QL, PK, plan->d_descs + plan->nd, 0, np, level, s, hp.K);
  if (c->dc.logN == 16) {
    NttFuse fo;
    if (conv_in_ntt) {  // the conversion P -> Q rides in the first pass of the NTT
      fo.conv = plan->d_descs + plan->nd;
      fo.conv_step = 0;
      fo.conv_max_in = hp.K;
      fo.conv_src = pc - QL;
      fo.conv_src_stride = PK;
    }
    fo.epi = 2;
    fo.out0 = out0;
    fo.out1 = out1;
    fo.x0 = in0;
    fo.x1 = in1;
    fo.w = c->pinv;
    fo.wp = c->pinv_prec;
    launch_ntt_fused(c->dc, tmp, level, 0, level, false, s, 0, np, QL, 0, fo);
  } else {
    launch_ntt(c->dc, tmp, level, 0, level, false, s, 0, np, QL);
    launch_moddown_tail2(c->dc, out0, out1 ? out1 : out0, in0, in1 ? in1 : in0, tmp, tmp + (in1 ? QL : 0), c->pinv, c->pinv_prec,
                         level, s, np);
  }
  stat(ST_MODDOWN, np, 8ull * np * N * (2 * level + hp.K));
  return post_launch();
}
int acehip_mod_down2(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1, uint32_t level,
                     acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_mod_down2: bad level");
  if (!out0 || !out1 || !in0 || !in1 || out0 == in0 || out1 == in1 || out0 == in1 || out1 == in0 || out0 == out1)
    return fail(ACEHIP_EINVAL, "acehip_mod_down2: outputs must not alias inputs or each other");
  return do_mod_down_n(c, out0, out1, in0, in1, level, (hipStream_t)s);
}

// ModRaise of bootstrapping (Transform_values_from_level0 ckks_bootstrap_context.c:1527-1551): limb 0 of each
// polynomial (NTT domain) -> coefficient domain -> centred lift -> residues on `level_out` limbs -> NTT domain
int acehip_mod_raise(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1,
                     uint32_t level_out, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level_out == 0 || level_out > c->hp.L || !out0 || !in0 || (in1 != nullptr) != (out1 != nullptr))
    return fail(ACEHIP_EINVAL, "acehip_mod_raise: bad arguments");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 np = in1 ? 2 : 1;
  u64* last = ws_at(c, 0);  // [np][N]
  if (c->dc.logN == 16) {
    NttFuse fi;
    fi.src0 = in0;
    fi.src1 = in1;
    fi.center_out = true;
    launch_ntt_fused(c->dc, last, hp.L, 0, 1, true, s, 0, np, N, 0, fi);
    NttFuse fo;
    fo.msg = (const int64_t*)last;
    fo.msg_stride = N;
    launch_ntt_fused(c->dc, out0, level_out, 0, level_out, false, s, 0, np, (size_t)(out1 - out0), 0, fo);
  } else {
    for (u32 z = 0; z < np; ++z) {
      HIP_TRY(hipMemcpyAsync(last + z * N, z ? in1 : in0, N * sizeof(u64), hipMemcpyDeviceToDevice, s));
      launch_ntt(c->dc, last + z * N, hp.L, 0, 1, true, s);
      launch_center(c->dc, (int64_t*)(last + z * N), last + z * N, 0, s);
      u64* out = z ? out1 : out0;
      launch_values_to_rns(c->dc, out, (const int64_t*)(last + z * N), level_out, 0, level_out, s);
      launch_ntt(c->dc, out, level_out, 0, level_out, false, s);
    }
  }
  stat(ST_RESCALE, np, 8ull * N * (1 + level_out) * np);
  return post_launch();
}

// Base conversion onto a SUBSET of the target limbs (limb-sharded execution, SURVEY 8e: every GPU converts only the
// limbs it owns once the source limbs have been gathered).  which = digit index (ModUp of that digit at `level`,
// sources = the digit's limbs in the coefficient domain, NOT yet scaled) or ACEHIP_CONV_MODDOWN (sources = the K
// p-limbs in the coefficient domain).  d_in: the n_in source limbs, contiguous; h_out_pos: target limb positions (in the
// polynomial extended at `level`); d_out: n_out limbs, output k is the limb at h_out_pos[k], coefficient domain.
int acehip_base_conv(acehip_ctx* c, uint64_t* d_out, const uint64_t* d_in, uint32_t level, int which, const uint32_t* h_out_pos,
                     uint32_t n_out, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_out || !d_in || !h_out_pos || n_out == 0 || n_out > hp.L + hp.K)
    return fail(ACEHIP_EINVAL, "acehip_base_conv: bad arguments");
  std::vector<u32> gi(n_out), col(n_out), pos(n_out);
  ConvDesc cd{};
  if (which == ACEHIP_CONV_MODDOWN) {
    for (u32 k = 0; k < n_out; ++k) {
      if (h_out_pos[k] >= level) return fail(ACEHIP_EINVAL, "acehip_base_conv: ModDown targets are q-limbs below the level");
      gi[k] = col[k] = h_out_pos[k];
      pos[k] = k;
    }
    cd.hat = c->phat_modq_t;
    cd.scale = c->phat_inv;
    cd.scale_prec = c->phat_inv_prec;
    cd.src_gi = c->p_gi;
    cd.n_in = hp.K;
    cd.hat_ld = hp.L;
  } else {
    if (which < 0 || (u32)which >= hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_base_conv: bad digit");
    const DevModUp* t = get_modup(c, level, (u32)which);
    if (!t) return fail(ACEHIP_EHIP, "ModUp table upload failed");
    HostParams::ModUp hm = hp.modup(level, (u32)which);
    for (u32 k = 0; k < n_out; ++k) {
      const u32 p = h_out_pos[k];
      const u32 want = p < level ? p : hp.L + (p - level);  // global prime index of the target
      u32 j = 0;
      while (j < hm.nc && hm.compl_idx[j] != want) ++j;
      if (p >= level + hp.K || j == hm.nc) return fail(ACEHIP_EINVAL, "acehip_base_conv: target is not a complement limb of the digit");
      gi[k] = want;
      col[k] = j;
      pos[k] = k;
    }
    cd.hat = t->hat_mod;
    cd.scale = t->hat_inv;
    cd.scale_prec = t->hat_inv_prec;
    cd.src_gi = t->src_gi;
    cd.n_in = t->n2;
    cd.hat_ld = t->nc;
  }
  // the three index lists and the descriptor travel in one small upload; freed after the launch has been ordered
  std::vector<u32> blob;
  blob.insert(blob.end(), gi.begin(), gi.end());
  blob.insert(blob.end(), col.begin(), col.end());
  blob.insert(blob.end(), pos.begin(), pos.end());
  u32* d_blob = nullptr;
  ConvDesc* d_desc = nullptr;
  HIP_TRY(hipMalloc((void**)&d_blob, blob.size() * sizeof(u32)));
  HIP_TRY(hipMalloc((void**)&d_desc, sizeof(ConvDesc)));
  HIP_TRY(hipMemcpy(d_blob, blob.data(), blob.size() * sizeof(u32), hipMemcpyHostToDevice));
  cd.out_gi = d_blob;
  cd.col = d_blob + n_out;
  cd.out_pos = d_blob + 2 * n_out;
  cd.src_pos0 = 0;
  cd.n_out = n_out;
  HIP_TRY(hipMemcpy(d_desc, &cd, sizeof(ConvDesc), hipMemcpyHostToDevice));
  hipStream_t s = (hipStream_t)s_;
  launch_base_conv_batch(c->dc, d_out, 0, d_in, 0, d_desc, 0, 1, n_out, s, cd.n_in);
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(d_blob);
  (void)hipFree(d_desc);
  return post_launch();
}

// one or two polynomials (c0, c1 of a ciphertext) through Rescale_poly in the same launches
static int do_rescale(acehip_ctx* c, u64* out0, u64* out1, const u64* in0, const u64* in1, u32 level, hipStream_t s) {
  const HostParams& hp = c->hp;
  const size_t N = hp.N;
  const u32 np = in1 ? 2 : 1;
  u64* last = ws_at(c, 0);   // [np][N]
  u64* t = ws_at(c, 2);      // [np][level-1][N]
  const size_t t_stride = (size_t)(level - 1) * N;
  HwBatchArgs cp;            // the last limbs into scratch, one launch
  for (u32 z = 0; z < np; ++z) {
    cp.op[z] = HwBatchOp{last + z * N, (z ? in1 : in0) + (size_t)(level - 1) * N, nullptr, HW_OP_COPY, 0};
    cp.seg_start[z] = (uint16_t)z;
  }
  cp.seg_start[np] = (uint16_t)np;
  const size_t row = (size_t)(level - 2) * hp.L;
  if (c->dc.logN == 16) {
    // fused: the iNTT reads the last limbs where they lie and leaves their centred lift; the forward NTT of the
    // remaining limbs starts from that lift (modulus switch and constant folded into its first pass) and applies
    // the Rescale tail in its last pass: 4 launches, no intermediate polynomial in memory
    NttFuse fi;
    fi.src0 = in0 + (size_t)(level - 1) * N;
    fi.src1 = in1 ? in1 + (size_t)(level - 1) * N : nullptr;
    fi.center_out = true;
    launch_ntt_fused(c->dc, last, hp.L, level - 1, 1, true, s, level - 1, np, N, 0, fi);
    NttFuse fo;
    fo.msg = (const int64_t*)last;
    fo.msg_stride = N;
    fo.msg_scale = c->qlql + row;
    fo.epi = 1;
    fo.out0 = out0;
    fo.out1 = out1;
    fo.x0 = in0;
    fo.x1 = in1;
    fo.w = c->ql_inv + row;
    fo.wp = c->ql_inv_prec + row;
    launch_ntt_fused(c->dc, t, hp.L, 0, level - 1, false, s, 0, np, t_stride, 0, fo);
  } else {
    launch_hw_batch_ew(c->dc, cp, np, s);
    launch_ntt(c->dc, last, hp.L, level - 1, 1, true, s, level - 1, np, N);
    launch_rescale_spread(c->dc, t, t_stride, last, N, c->qlql + row, c->qlql_prec + row, level, np, s);
    launch_ntt(c->dc, t, hp.L, 0, level - 1, false, s, 0, np, t_stride);
    launch_rescale_tail(c->dc, out0, out1, in0, in1, t, t_stride, c->ql_inv + row, c->ql_inv_prec + row, level, np, s);
  }
  stat(ST_RESCALE, np, 8ull * N * (2 * level - 1) * np);
  return post_launch();
}
int acehip_rescale(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_rescale: level must be in [2, L]");
  return do_rescale(c, out, nullptr, in, nullptr, level, (hipStream_t)s);
}
int acehip_rescale2(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in0, const uint64_t* in1, uint32_t level,
                    acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_rescale2: level must be in [2, L]");
  if (!out0 || !out1 || !in0 || !in1) return fail(ACEHIP_EINVAL, "acehip_rescale2: null polynomial");
  return do_rescale(c, out0, out1, in0, in1, level, (hipStream_t)s);
}

static const KsPlan* get_ks_plan(acehip_ctx* c, u32 level) {
  {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->ks_plans.find(level);
    if (it != c->ks_plans.end()) return &it->second;
  }
  const HostParams& hp = c->hp;
  KsPlan plan;
  plan.nd = hp.num_decomp(level);
  const bool fold = c->dc.logN == 16;
  std::vector<u64> inv_up(4 * (size_t)level, 0), inv_down(4 * (size_t)hp.K, 0);
  std::vector<ConvDesc> descs;
  for (u32 d = 0; d < plan.nd; ++d) {
    const DevModUp* t = get_modup(c, level, d);
    if (!t) return nullptr;
    ConvDesc cd{};
    cd.hat = t->hat_mod;
    cd.scale = fold ? nullptr : t->hat_inv;
    cd.scale_prec = fold ? nullptr : t->hat_inv_prec;
    if (fold) {
      HostParams::ModUp hm = hp.modup(level, d);
      for (u32 i = 0; i < hm.n2; ++i) {
        const PrimeConsts& P = hp.primes[hm.start + i];
        const u64 tn = mul_mod(P.n_inv, hm.hat_inv[i], P.q), tw = mul_mod(P.inv_w1_ninv, hm.hat_inv[i], P.q);
        u64* o = &inv_up[4 * (size_t)(hm.start + i)];
        o[0] = tn;
        o[1] = shoup_prec(tn, P.q);
        o[2] = tw;
        o[3] = shoup_prec(tw, P.q);
      }
    }
    cd.src_gi = t->src_gi;
    cd.out_gi = t->out_gi;
    cd.out_pos = t->out_pos;
    cd.src_pos0 = t->start;
    cd.n_in = t->n2;
    cd.n_out = t->nc;
    cd.hat_ld = t->nc;
    plan.max_nc = std::max(plan.max_nc, t->nc);
    descs.push_back(cd);
  }
  ConvDesc md{};  // ModDown: K p-limbs at positions level.. -> level q-limbs (polynomial.c:755-807)
  md.hat = c->phat_modq_t;
  md.scale = fold ? nullptr : c->phat_inv;
  md.scale_prec = fold ? nullptr : c->phat_inv_prec;
  if (fold)
    for (u32 j = 0; j < hp.K; ++j) {
      const PrimeConsts& P = hp.primes[hp.L + j];
      const u64 tn = mul_mod(P.n_inv, hp.phat_inv_modp[j], P.q), tw = mul_mod(P.inv_w1_ninv, hp.phat_inv_modp[j], P.q);
      inv_down[4 * j + 0] = tn;
      inv_down[4 * j + 1] = shoup_prec(tn, P.q);
      inv_down[4 * j + 2] = tw;
      inv_down[4 * j + 3] = shoup_prec(tw, P.q);
    }
  md.src_gi = c->p_gi;
  md.out_gi = c->q_gi;
  md.out_pos = c->q_pos;
  md.src_pos0 = level;
  md.n_in = hp.K;
  md.n_out = level;
  md.hat_ld = hp.L;
  descs.push_back(md);
  std::lock_guard<std::mutex> lk(c->mu);
  plan.d_descs = c->up(descs);
  if (!plan.d_descs) return nullptr;
  if (fold) {
    plan.inv_up = c->up(inv_up);
    plan.inv_down = c->up(inv_down);
    if (!plan.inv_up || !plan.inv_down) return nullptr;
  }
  return &(c->ks_plans[level] = plan);
}

int acehip_key_switch(acehip_ctx* c, uint64_t* out0, uint64_t* out1, const uint64_t* in, const uint64_t* key,
                      uint32_t level, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_switch: bad level");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const size_t N = hp.N, E = (size_t)(level + hp.K) * N;  // words per extended polynomial
  const u32 nd = plan->nd;
  // workspace: coef (level limbs) | ext[nd] | acc0 | acc1 | tmp[2] (level limbs each)
  u64* coef = c->ws;
  u64* ext = coef + (size_t)level * N;
  u64* acc0 = ext + nd * E;
  u64* acc1 = acc0 + E;
  u64* tmp = acc1 + E;
  // 1. all digit limbs to the coefficient domain in one launch (polynomial.c:1276-1283 for every part)
  const bool fused = c->dc.logN == 16;
  if (fused) {
    NttFuse fi;
    fi.src0 = in;
    fi.inv_scale = plan->inv_up;  // (Q_d/q_i)^-1 folded into the last inverse stage
    launch_ntt_fused(c->dc, coef, hp.L, 0, level, true, s, 0, 1, 0, 0, fi);
  } else {
    HIP_TRY(hipMemcpyAsync(coef, in, (size_t)level * N * sizeof(u64), hipMemcpyDeviceToDevice, s));
    launch_ntt(c->dc, coef, hp.L, 0, level, true, s);
  }
  // 2. every digit's base conversion (scaling by (Q_d/q_i)^-1 folded into the inverse NTT) and
  // 3. the NTT of every digit's complement limbs (own digit limbs are skipped): at N = 2^16 one pipeline, the conversion
  //    is computed by the first NTT pass while it loads its input
  const u32 n_ext_rows = level + hp.K - std::min(hp.alpha, level - hp.alpha * (nd - 1));
  if (conv_fusable(c, hp.alpha)) {
    NttFuse fc;
    fc.conv = plan->d_descs;
    fc.conv_step = 1;
    fc.conv_max_in = hp.alpha;
    fc.conv_src = coef;
    fc.conv_src_stride = 0;
    launch_ntt_fused(c->dc, ext, level, 0, n_ext_rows, false, s, 0, nd, E, hp.alpha, fc);
  } else {
    launch_base_conv_batch(c->dc, ext, E, coef, 0, plan->d_descs, 1, nd, plan->max_nc, s, hp.alpha);
    launch_ntt(c->dc, ext, level, 0, n_ext_rows, false, s, 0, nd, E, hp.alpha);
  }
  // 4. key inner product fused over digits; a digit's own limbs are read from `in` directly
  launch_key_mac_fused(c->dc, acc0, acc1, key, ext, E, in, level, nd, hp.alpha, s);
  // 5. ModDown of both accumulators together (polynomial.c:928-967)
  if (fused) {
    NttFuse fa;
    fa.inv_scale = plan->inv_down - 4 * (size_t)level;  // the p-limbs sit at positions level .. level+K-1
    launch_ntt_fused(c->dc, acc0, level, level, hp.K, true, s, 0, 2, E, 0, fa);
  } else {
    launch_ntt(c->dc, acc0, level, level, hp.K, true, s, 0, 2, E);
  }
  const bool conv_in_ntt = fused && conv_fusable(c, hp.K);
  if (!conv_in_ntt) launch_base_conv_batch(c->dc, tmp, (size_t)level * N, acc0, E, plan->d_descs + nd, 0, 2, level, s, hp.K);
  if (fused) {  // the ModDown tail rides in the last NTT pass, the conversion P -> Q in the first
    NttFuse fo;
    if (conv_in_ntt) {
      fo.conv = plan->d_descs + nd;
      fo.conv_step = 0;
      fo.conv_max_in = hp.K;
      fo.conv_src = acc0;
      fo.conv_src_stride = E;
    }
    fo.epi = 2;
    fo.out0 = out0;
    fo.out1 = out1;
    fo.x0 = acc0;
    fo.x1 = acc1;
    fo.w = c->pinv;
    fo.wp = c->pinv_prec;
    launch_ntt_fused(c->dc, tmp, level, 0, level, false, s, 0, 2, (size_t)level * N, 0, fo);
  } else {
    launch_ntt(c->dc, tmp, level, 0, level, false, s, 0, 2, (size_t)level * N);
    launch_moddown_tail2(c->dc, out0, out1, acc0, acc1, tmp, tmp + (size_t)level * N, c->pinv, c->pinv_prec, level, s);
  }
  stat(ST_KEYSWITCH, 1, acehip_key_switch_bytes(c, level));
  return post_launch();
}

int acehip_values_to_rns(acehip_ctx* c, uint64_t* d, const int64_t* vals, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  launch_values_to_rns(c->dc, d, vals, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
int acehip_sample_uniform(acehip_ctx* c, uint64_t* d, uint32_t level, uint32_t pos0, uint32_t n, uint64_t seed, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  launch_sample_uniform(c->dc, d, level, pos0, n, seed, (hipStream_t)s);
  return post_launch();
}
int acehip_mul_scalars(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n > 64) return fail(ACEHIP_EINVAL, "acehip_mul_scalars: at most 64 limbs per call");
  LimbConsts w{};
  for (u32 i = 0; i < n; ++i) w.w[i] = h_scalars[i];
  launch_mul_scalars(c->dc, r, a, w, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
// Encode_at_level_with_sf ckks_encoder.c:395 -> Encode_impl :199-297 (64-bit path)
static int ensure_embed_tables(acehip_ctx* c) {
  std::lock_guard<std::mutex> g(c->mu);
  if (c->emb_rou) return 0;
  const size_t N = c->hp.N, m = 2 * N;
  std::vector<double> rou(2 * m);
  for (size_t i = 0; i < m; ++i) {  // Precompute_fft ntt.c:587-610
    // glibc's sincos(), which is what gcc makes of the reference's cos(angle) + sin(angle) pair: it differs from
    // separate cos()/sin() calls in the last bit for ~0.1% of the entries, and clang would emit the latter
    const double angle = 2 * M_PI * i / m;
    sincos(angle, &rou[2 * i + 1], &rou[2 * i]);
  }
  std::vector<u32> rot(N / 2 ? N / 2 : 1, 1);
  for (size_t i = 1; i < N / 2; ++i) rot[i] = (u32)((5ull * rot[i - 1]) % m);
  // per-stage layout of the twiddles Embedding_inv uses: stage logm, butterfly i < 2^(logm-1)
  std::vector<double> tws(2 * (N / 2 ? N / 2 : 1), 0.0);
  for (u32 logm = 1; (1ull << logm) <= N / 2; ++logm) {
    const size_t idx_mod = 1ull << (logm + 2), gap = m / idx_mod, half = 1ull << (logm - 1);
    for (size_t i = 0; i < half; ++i) {
      const size_t k = (idx_mod - (rot[i] % idx_mod)) * gap;
      tws[2 * (half - 1 + i)] = rou[2 * k];
      tws[2 * (half - 1 + i) + 1] = rou[2 * k + 1];
    }
  }
  rou.swap(tws);  // the device gets the per-stage table (the flat one is only needed to build it)
  u32* d_rot = c->up(rot);
  double* d_rou = c->up(rou);
  void *work = nullptr, *msg = nullptr, *err = nullptr;
  // scratch for a batch of EMB_BATCH_MAX messages (acehip_encode_batch)
  if (!d_rot || !d_rou || hipMalloc(&work, EMB_BATCH_MAX * (N / 2 * 16) + 16) != hipSuccess ||
      hipMalloc(&msg, (size_t)EMB_BATCH_MAX * N * 8) != hipSuccess ||
      hipMalloc(&err, 64) != hipSuccess)
    return fail(ACEHIP_EHIP, "acehip_encode: table allocation failed");
  c->owned.push_back(work);
  c->owned.push_back(msg);
  c->owned.push_back(err);
  if (hipMemset(err, 0, 64) != hipSuccess) return fail(ACEHIP_EHIP, "acehip_encode: memset failed");
  c->emb_rot = d_rot;
  c->emb_work = (cd*)work;
  c->emb_msg = (int64_t*)msg;
  c->emb_err = (int*)err;
  c->emb_rou = (cd*)d_rou;
  return 0;
}

// [L] Delta^(sf_degree-1) mod q_i on the device (ckks_encoder.c:270-285), cached per (Delta, sf_degree)
static const u64* encode_scale_table(acehip_ctx* c, u64 sfi, u32 sf_degree) {
  std::lock_guard<std::mutex> g(c->mu);
  u64*& tab = c->enc_scales[{sfi, sf_degree}];
  if (!tab) {
    std::vector<u64> w(c->hp.L);
    for (u32 i = 0; i < c->hp.L; ++i) {
      const u64 q = c->hp.primes[i].q;
      u64 pw = sfi % q;
      for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
      w[i] = pw;
    }
    tab = c->up(w);
  }
  return tab;
}

int acehip_encode_batch(acehip_ctx* c, uint64_t* const* h_q, const void* const* h_vals, uint32_t n_batch, int kind, size_t len,
                        uint32_t slots, double sf, uint32_t sf_degree, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (n_batch == 0 || n_batch > EMB_BATCH_MAX || !h_q || !h_vals || kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) ||
      len > slots || len == 0 || sf_degree < 1 || level == 0 || level > c->hp.L)
    return fail(ACEHIP_EINVAL, "acehip_encode_batch: bad arguments");
  for (u32 b = 0; b < n_batch; ++b)
    if (!h_q[b] || !h_vals[b]) return fail(ACEHIP_EINVAL, "acehip_encode_batch: null pointer in the batch");
  if (c->dc.logN != 16 || n_batch == 1) {  // no batched form below N = 2^16: one encode after the other
    for (u32 b = 0; b < n_batch; ++b)
      if (int e = acehip_encode(c, h_q[b], nullptr, h_vals[b], kind, len, slots, sf, sf_degree, level, 0, s)) return e;
    return ACEHIP_OK;
  }
  if (int e = ensure_embed_tables(c)) return e;
  hipStream_t st = (hipStream_t)s;
  EmbBatch eb{};
  for (u32 b = 0; b < n_batch; ++b) eb.vals[b] = h_vals[b];
  launch_embed_inv_batch(c->emb_msg, c->emb_work, eb, n_batch, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, st);
  NttFuse f;
  f.msg = c->emb_msg;
  f.msg_stride = N;
  for (u32 b = 0; b < n_batch; ++b) f.polyz[b] = h_q[b];
  if (sf_degree > 1) {
    f.msg_scale = encode_scale_table(c, (u64)sf, sf_degree);
    if (!f.msg_scale) return fail(ACEHIP_EHIP, "acehip_encode: scale table upload failed");
  }
  launch_ntt_fused(c->dc, h_q[0], level, 0, level, false, st, 0, n_batch, 0, 0, f);
  stat(ST_ENCODE, n_batch, n_batch * (8ull * N * level + len * (kind == 0 ? 4 : kind == 1 ? 8 : 16)));
  return post_launch();
}

int acehip_encode(acehip_ctx* c, uint64_t* d_q, uint64_t* d_p, const void* d_vals, int kind, size_t len, uint32_t slots,
                  double sf, uint32_t sf_degree, uint32_t level, uint32_t n_p, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) || len > slots || sf_degree < 1 || level == 0 ||
      level > c->hp.L || n_p > c->hp.K || (n_p && !d_p) || !d_q || (!d_vals && len))
    return fail(ACEHIP_EINVAL, "acehip_encode: bad arguments");
  if (int e = ensure_embed_tables(c)) return e;
  hipStream_t st = (hipStream_t)s;
  launch_embed_inv(c->emb_msg, c->emb_work, d_vals, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, st);
  const u64 sfi = (u64)sf;
  if (c->dc.logN == 16) {  // the first NTT pass reduces (and scales) the message itself: no residue pass over memory
    NttFuse f;
    f.msg = c->emb_msg;
    if (sf_degree > 1) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1) on the q limbs
      f.msg_scale = encode_scale_table(c, sfi, sf_degree);
      if (!f.msg_scale) return fail(ACEHIP_EHIP, "acehip_encode: scale table upload failed");
    }
    launch_ntt_fused(c->dc, d_q, level, 0, level, false, st, 0, 1, 0, 0, f);
    if (n_p) {
      f.msg_scale = nullptr;
      launch_ntt_fused(c->dc, d_p, 0, 0, n_p, false, st, 0, 1, 0, 0, f);
    }
  } else {
    launch_values_to_rns(c->dc, d_q, c->emb_msg, level, 0, level, st);
    if (n_p) launch_values_to_rns(c->dc, d_p, c->emb_msg, 0, 0, n_p, st);
    if (sf_degree > 1) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1) on the q limbs
      for (u32 l0 = 0; l0 < level; l0 += 64) {
        LimbConsts w{};
        const u32 n = std::min(64u, level - l0);
        for (u32 i = 0; i < n; ++i) {
          const u64 q = c->hp.primes[l0 + i].q;
          u64 pw = sfi % q;
          for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
          w.w[i] = pw;
        }
        launch_mul_scalars(c->dc, d_q, d_q, w, level, l0, n, st);
      }
    }
    launch_ntt(c->dc, d_q, level, 0, level, false, st);
    if (n_p) launch_ntt(c->dc, d_p, 0, 0, n_p, false, st);
  }
  stat(ST_ENCODE, 1, 8ull * N * (level + n_p) + len * (kind == 0 ? 4 : kind == 1 ? 8 : 16));
  return post_launch();
}

int acehip_encode_status(acehip_ctx* c) {
  if (int e = check_dev(c)) return e;
  if (!c->emb_err) return 0;
  int flag = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&flag, c->emb_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIP_TRY(hipMemset(c->emb_err, 0, sizeof(int)));
    return fail(ACEHIP_EINVAL, "encode overflow, please choose a smaller scaling factor");
  }
  return 0;
}

int acehip_decomp(acehip_ctx* c, uint64_t* out, const uint64_t* in, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_decomp: bad level/digit");
  const u32 start = c->hp.alpha * digit, n2 = std::min(c->hp.alpha, level - start);
  HIP_TRY(hipMemcpyAsync(out, in + (size_t)start * c->hp.N, (size_t)n2 * c->hp.N * sizeof(u64), hipMemcpyDeviceToDevice, (hipStream_t)s));
  return (int)n2;
}
int acehip_mod_up(acehip_ctx* c, uint64_t* out, const uint64_t* digit_limbs, uint32_t level, uint32_t digit, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || digit >= c->hp.num_decomp(level)) return fail(ACEHIP_EINVAL, "acehip_mod_up: bad level/digit");
  const u32 start = c->hp.alpha * digit, n2 = std::min(c->hp.alpha, level - start);
  // same pipeline as Decomp_modup with the digit limbs supplied separately
  if (int e = do_decomp_modup(c, out, digit_limbs - (size_t)start * c->hp.N, level, digit, ws_at(c, 0), (hipStream_t)s)) return e;
  return (int)n2;
}

int acehip_add_scalars(acehip_ctx* c, uint64_t* r, const uint64_t* a, const uint64_t* h_scalars, uint32_t level, uint32_t pos0, uint32_t n, acehip_stream s) {
  if (int e = check_range(c, level, pos0, n)) return e;
  if (n > 64) return fail(ACEHIP_EINVAL, "acehip_add_scalars: at most 64 limbs per call");
  LimbConsts w{};
  for (u32 i = 0; i < n; ++i) w.w[i] = h_scalars[i];
  launch_add_scalars(c->dc, r, a, w, level, pos0, n, (hipStream_t)s);
  return post_launch();
}
// Switch_key_precompute (polynomial.c:1224-1239, 1337-1343): every digit of d_in raised to level+K limbs.
// h_ext[d] = output polynomial of digit d (separate blocks: the rt_ant shim hands them to the caller's polynomials without a copy)
static int modup_digits_to(acehip_ctx* c, uint64_t* const* h_ext, const uint64_t* in, uint32_t level, acehip_stream s_) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_modup_digits: bad level");
  const HostParams& hp = c->hp;
  hipStream_t s = (hipStream_t)s_;
  const KsPlan* plan = get_ks_plan(c, level);
  if (!plan) return fail(ACEHIP_EHIP, "key-switch plan upload failed");
  const size_t N = hp.N;
  const u32 nd = plan->nd;
  if (nd > 8) return fail(ACEHIP_EINVAL, "acehip_modup_digits: more than 8 digits");
  PtrTab8 outz;
  for (u32 d = 0; d < nd; ++d) {
    if (!h_ext[d]) return fail(ACEHIP_EINVAL, "acehip_modup_digits: null output");
    outz.p[d] = h_ext[d];
  }
  u64* coef = c->ws;
  if (c->dc.logN == 16) {
    NttFuse fi;
    fi.src0 = in;
    fi.inv_scale = plan->inv_up;  // (Q_d/q_i)^-1 folded into the last inverse stage
    launch_ntt_fused(c->dc, coef, hp.L, 0, level, true, s, 0, 1, 0, 0, fi);
  } else {
    HIP_TRY(hipMemcpyAsync(coef, in, (size_t)level * N * sizeof(u64), hipMemcpyDeviceToDevice, s));
    launch_ntt(c->dc, coef, hp.L, 0, level, true, s);
  }
  const u32 n_ext_rows = level + hp.K - std::min(hp.alpha, level - hp.alpha * (nd - 1));
  if (conv_fusable(c, hp.alpha)) {  // the conversions ride in the first pass of the NTT
    NttFuse fc;
    fc.conv = plan->d_descs;
    fc.conv_step = 1;
    fc.conv_max_in = hp.alpha;
    fc.conv_src = coef;
    fc.conv_src_stride = 0;
    for (u32 d = 0; d < nd; ++d) fc.polyz[d] = outz.p[d];
    launch_ntt_fused(c->dc, outz.p[0], level, 0, n_ext_rows, false, s, 0, nd, 0, hp.alpha, fc);
  } else {
    launch_base_conv_batch(c->dc, outz.p[0], 0, coef, 0, plan->d_descs, 1, nd, plan->max_nc, s, hp.alpha, outz);
    if (c->dc.logN == 16) {
      NttFuse fz;
      for (u32 d = 0; d < nd; ++d) fz.polyz[d] = outz.p[d];
      launch_ntt_fused(c->dc, outz.p[0], level, 0, n_ext_rows, false, s, 0, nd, 0, hp.alpha, fz);
    } else {
      for (u32 d = 0; d < nd; ++d) {  // the generic passes address polynomials by stride: one digit at a time
        const u32 start = hp.alpha * d, n2 = std::min(hp.alpha, level - start);
        if (start) launch_ntt(c->dc, outz.p[d], level, 0, start, false, s);
        launch_ntt(c->dc, outz.p[d], level, start + n2, level + hp.K - (start + n2), false, s);
      }
    }
  }
  {  // digit limbs pass through (polynomial.c:1265-1273): `level` limb copies in one launch
    HwBatchArgs cp;
    u32 n_ops = 0;
    for (u32 pos = 0; pos < level; ++pos) {
      if (n_ops == HW_BATCH_MAX) return fail(ACEHIP_EINVAL, "acehip_modup_digits: too many limbs");
      cp.seg_start[n_ops] = (uint16_t)n_ops;
      cp.op[n_ops++] = HwBatchOp{outz.p[pos / hp.alpha] + (size_t)pos * N, in + (size_t)pos * N, nullptr, HW_OP_COPY, 0};
    }
    cp.seg_start[n_ops] = (uint16_t)n_ops;
    launch_hw_batch_ew(c->dc, cp, n_ops, s);
  }
  stat(ST_MODUP, nd, 8ull * N * (level + (u64)nd * (level + hp.K)));
  return post_launch();
}
int acehip_modup_digits(acehip_ctx* c, uint64_t* ext, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (!c || level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_modup_digits: bad level");
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  uint64_t* tab[8];
  const u32 nd = c->hp.num_decomp(level);
  if (nd > 8) return fail(ACEHIP_EINVAL, "acehip_modup_digits: more than 8 digits");
  for (u32 d = 0; d < nd; ++d) tab[d] = ext + d * E;
  return modup_digits_to(c, tab, in, level, s);
}
int acehip_modup_digits_to(acehip_ctx* c, uint64_t* const* h_ext, const uint64_t* in, uint32_t level, acehip_stream s) {
  if (!c || !h_ext) return fail(ACEHIP_EINVAL, "acehip_modup_digits_to: null argument");
  return modup_digits_to(c, h_ext, in, level, s);
}
// Fast_switch_key_ext (ckks_evaluator.c:418-460): acc{0,1} = sum_d key{0,1}[d] * ext[d] over level+K limbs, no ModDown
int acehip_key_inner_product(acehip_ctx* c, uint64_t* acc0, uint64_t* acc1, const uint64_t* key, const uint64_t* ext, uint32_t level, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_inner_product: bad level");
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  launch_key_mac_fused(c->dc, acc0, acc1, key, ext, E, nullptr, level, c->hp.num_decomp(level), c->hp.alpha, (hipStream_t)s);
  stat(ST_KEYMAC, 1, 8ull * E * (3ull * c->hp.num_decomp(level) + 2));
  return post_launch();
}

int acehip_key_inner_product_add(acehip_ctx* c, uint64_t* acc0, uint64_t* acc1, const uint64_t* key, const uint64_t* ext, uint32_t level,
                                 const uint64_t* add0, const uint64_t* h_scalars, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: bad level");
  if (!add0 || !h_scalars) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: null addend");
  if (level > 64) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: at most 64 q-limbs");
  LimbConsts w{};
  for (u32 i = 0; i < level; ++i) {
    if (h_scalars[i] >= c->hp.primes[i].q) return fail(ACEHIP_EINVAL, "acehip_key_inner_product_add: scalar is not a residue of its prime");
    w.w[i] = h_scalars[i];
  }
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  launch_key_mac_fused(c->dc, acc0, acc1, key, ext, E, nullptr, level, c->hp.num_decomp(level), c->hp.alpha, (hipStream_t)s, add0, &w);
  stat(ST_KEYMAC, 1, 8ull * E * (3ull * c->hp.num_decomp(level) + 2) + 8ull * level * c->hp.N);
  return post_launch();
}

// Rotate_iteration's inner loop (ckks_bootstrap_context.c:1326-1341): out_i = sum_j rot_j (*) pt_{i,j} in the PQ basis
int acehip_bsgs_inner(acehip_ctx* c, uint64_t* const* out0, uint64_t* const* out1, const uint64_t* const* in0, const uint64_t* const* in1,
                      const uint64_t* const* pt, uint32_t g, uint32_t b, uint32_t pt_q_limbs, uint32_t level, acehip_stream s) {
  return acehip_bsgs_inner_rot(c, out0, out1, in0, in1, nullptr, pt, g, b, pt_q_limbs, level, s);
}

int acehip_bsgs_inner_rot(acehip_ctx* c, uint64_t* const* out0, uint64_t* const* out1, const uint64_t* const* in0, const uint64_t* const* in1,
                          const uint32_t* in_auto, const uint64_t* const* pt, uint32_t g, uint32_t b, uint32_t pt_q_limbs, uint32_t level,
                          acehip_stream s) {
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || pt_q_limbs < level || pt_q_limbs > c->hp.L) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: bad level");
  if (g == 0 || b == 0 || g > BSGS_MAX_G || b > BSGS_MAX_B || g * b > BSGS_MAX_PT) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: g, b out of range");
  BsgsArgs a{};
  u32 n_pt = 0;
  for (u32 j = 0; j < g; ++j) {
    if (!in0[j] || !in1[j]) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: null input");
    a.in0[j] = in0[j];
    a.in1[j] = in1[j];
    a.in_auto[j] = in_auto ? in_auto[j] : 0;
    if (a.in_auto[j] != 0 && (a.in_auto[j] % 2 == 0 || a.in_auto[j] >= 2 * c->hp.N))
      return fail(ACEHIP_EINVAL, "acehip_bsgs_inner_rot: automorphism index must be odd and below 2N");
    for (u32 i = 0; i < b && a.in_auto[j] != 0; ++i)  // a gathered input is read at other lanes' positions: it cannot be an output
      if (out0[i] == in0[j] || out0[i] == in1[j] || out1[i] == in0[j] || out1[i] == in1[j])
        return fail(ACEHIP_EINVAL, "acehip_bsgs_inner_rot: a rotated input aliases an output");
  }
  for (u32 i = 0; i < b; ++i) {
    if (!out0[i] || !out1[i]) return fail(ACEHIP_EINVAL, "acehip_bsgs_inner: null output");
    a.out0[i] = out0[i];
    a.out1[i] = out1[i];
    for (u32 j = 0; j < g; ++j) {
      a.pt[i * g + j] = pt[i * g + j];
      n_pt += pt[i * g + j] != nullptr;
    }
  }
  a.g = g;
  a.b = b;
  a.pt_q_alloc = pt_q_limbs;
  launch_bsgs_inner(c->dc, a, level, (hipStream_t)s);
  const size_t E = (size_t)(level + c->hp.K) * c->hp.N;
  stat(ST_EW, n_pt, 8ull * E * (2ull * g + n_pt + 2ull * b));
  return post_launch();
}

// ------------------------------------------------------------------------------------------------
// Limb-sharded execution (BASELINE configs[4]): rank r of `world` holds the limbs gi with gi % world == r
// of every polynomial and key (q_i: gi = i, p_j: gi = L + j), packed in ascending gi.  NTT, limb-wise arithmetic and
// the key inner product are local; the two base conversions of a key-switch and the last limb of a rescale need the
// other ranks' limbs: the caller moves those with one all-gather / broadcast each (RCCL over xGMI on a node) between
// the phases below.  Every phase is a handful of batched launches on the caller's stream and never synchronises.
// The arithmetic is that of acehip_key_switch / acehip_rescale: gathering every rank's result reproduces them bit
// for bit (tests/test_gpu_shard.py runs world simulated ranks on one GPU).
// ------------------------------------------------------------------------------------------------
struct ShardLevelPlan {
  u32 nq = 0, nd = 0;
  ConvDesc* d_up = nullptr;             // [nd] ModUp onto the owned complement limbs of each digit
  ConvDesc* d_down = nullptr;           // [1]  ModDown onto the owned q-limbs
  std::vector<u32> n_tgt;               // per digit: converted limbs
  std::vector<std::vector<u32>> tgt_y;  // per digit: packed index (among the owned limbs at this level) of each target
  std::vector<u32*> d_tgt_gi;           // per digit: primes of the targets (NTT of the raised limbs)
  u32 max_tgt = 0;
  u64 *d_pinv = nullptr, *d_pinv_prec = nullptr;                      // [nq]  P^-1 mod q (ModDown tail)
  u64 *d_rs_c1 = nullptr, *d_rs_c1p = nullptr, *d_rs_inv = nullptr, *d_rs_invp = nullptr;  // rescale constants of the owned limbs < level-1
  u32 nq_rs = 0;
};
struct acehip_shard {
  acehip_ctx* c = nullptr;
  u32 rank = 0, world = 1;
  std::vector<u32> q_own, p_own;        // owned q indices i (ascending), owned p indices j
  u32 *d_q_gi = nullptr, *d_p_gi = nullptr, *d_own_gi_full = nullptr;  // device: gi of owned q-limbs / p-limbs / both
  u64 *full = nullptr, *ext = nullptr, *acc = nullptr, *pfull = nullptr, *conv = nullptr, *tmp = nullptr;
  std::map<u32, ShardLevelPlan> plans;
  std::mutex mu;
};

namespace {
using PtrTab = PackedPtrs;
u32 shard_nq(const acehip_shard* sh, u32 level) {
  u32 n = 0;
  while (n < sh->q_own.size() && sh->q_own[n] < level) ++n;
  return n;
}
u32 shard_pad(const acehip_shard* sh, u32 n_total) {  // most limbs any rank owns out of gi in [0, n_total)
  return (n_total + sh->world - 1) / sh->world;
}
// NTT of n packed limbs of primes h_gi[0..n) (device copy d_gi), n_polys polynomials `stride` words apart, optionally
// reading the input out of place (inverse, N = 2^16) from src0 / src1
void ntt_packed(acehip_ctx* c, u64* data, const u64* src0, const u64* src1, const u32* d_gi, const u32* h_gi, u32 n, u32 n_polys,
                size_t stride, bool inverse, hipStream_t s) {
  if (n == 0) return;
  const size_t N = c->hp.N;
  if (c->dc.logN == 16) {
    NttFuse f;
    f.gi_tab = d_gi;
    f.src0 = src0;
    f.src1 = src1;
    launch_ntt_fused(c->dc, data, 0, 0, n, inverse, s, 0, n_polys, stride, 0, f);
    return;
  }
  for (u32 z = 0; z < n_polys; ++z) {
    const u64* src = z ? src1 : src0;
    if (src) (void)hipMemcpyAsync(data + z * stride, src, (size_t)n * N * sizeof(u64), hipMemcpyDeviceToDevice, s);
    for (u32 k = 0; k < n; ++k) {
      const u32 gi = h_gi[k];
      u64* ptr = data + z * stride + (size_t)k * N;
      if (gi < c->hp.L) launch_ntt(c->dc, ptr, c->hp.L, gi, 1, inverse, s, gi);
      else              launch_ntt(c->dc, ptr, 0, gi - c->hp.L, 1, inverse, s, gi - c->hp.L);
    }
  }
}
const ShardLevelPlan* shard_plan(acehip_shard* sh, u32 level) {
  std::lock_guard<std::mutex> g(sh->mu);
  auto it = sh->plans.find(level);
  if (it != sh->plans.end()) return &it->second;
  acehip_ctx* c = sh->c;
  const HostParams& hp = c->hp;
  ShardLevelPlan pl;
  pl.nq = shard_nq(sh, level);
  pl.nd = hp.num_decomp(level);
  // owned limbs at this level, in packed order: (position, prime)
  std::vector<std::pair<u32, u32>> own;
  for (u32 k = 0; k < pl.nq; ++k) own.push_back({sh->q_own[k], sh->q_own[k]});
  for (u32 j : sh->p_own) own.push_back({level + j, hp.L + j});
  std::vector<ConvDesc> up(pl.nd);
  pl.n_tgt.resize(pl.nd);
  pl.tgt_y.resize(pl.nd);
  pl.d_tgt_gi.resize(pl.nd, nullptr);
  for (u32 d = 0; d < pl.nd; ++d) {
    const DevModUp* t = get_modup(c, level, d);
    if (!t) return nullptr;
    HostParams::ModUp hm = hp.modup(level, d);
    std::vector<u32> gi, col, pos;
    for (u32 y = 0; y < own.size(); ++y) {
      const u32 p = own[y].first, want = own[y].second;
      if (p >= hm.start && p < hm.start + hm.n2) continue;  // the digit's own limbs pass through
      u32 j = 0;
      while (j < hm.nc && hm.compl_idx[j] != want) ++j;
      if (j == hm.nc) return nullptr;
      gi.push_back(want);
      col.push_back(j);
      pos.push_back((u32)pl.tgt_y[d].size());
      pl.tgt_y[d].push_back(y);
    }
    pl.n_tgt[d] = (u32)gi.size();
    pl.max_tgt = std::max(pl.max_tgt, pl.n_tgt[d]);
    ConvDesc cd{};
    cd.hat = t->hat_mod;
    cd.scale = t->hat_inv;
    cd.scale_prec = t->hat_inv_prec;
    cd.src_gi = t->src_gi;
    cd.n_in = t->n2;
    cd.hat_ld = t->nc;
    cd.src_pos0 = hm.start;
    cd.n_out = pl.n_tgt[d];
    if (cd.n_out) {
      cd.out_gi = pl.d_tgt_gi[d] = c->up(gi);
      cd.col = c->up(col);
      cd.out_pos = c->up(pos);
      if (!cd.out_gi || !cd.col || !cd.out_pos) return nullptr;
    }
    up[d] = cd;
  }
  pl.d_up = c->up(up);
  if (pl.nq) {
    std::vector<u32> gi(pl.nq), pos(pl.nq);
    std::vector<u64> pinv(pl.nq), pinvp(pl.nq);
    for (u32 k = 0; k < pl.nq; ++k) {
      gi[k] = sh->q_own[k];
      pos[k] = k;
      pinv[k] = hp.pinv_modq[gi[k]];
      pinvp[k] = hp.pinv_modq_prec[gi[k]];
    }
    ConvDesc md{};
    md.hat = c->phat_modq_t;
    md.scale = c->phat_inv;
    md.scale_prec = c->phat_inv_prec;
    md.src_gi = c->p_gi;
    md.n_in = hp.K;
    md.hat_ld = hp.L;
    md.src_pos0 = 0;
    md.n_out = pl.nq;
    md.out_gi = c->up(gi);
    md.col = md.out_gi;  // column of phat_modq_t = the q index
    md.out_pos = c->up(pos);
    std::vector<ConvDesc> one(1, md);
    pl.d_down = c->up(one);
    pl.d_pinv = c->up(pinv);
    pl.d_pinv_prec = c->up(pinvp);
    if (!pl.d_down || !pl.d_pinv || !pl.d_pinv_prec) return nullptr;
  }
  if (level > 1) {  // rescale constants (crt.c:270-326) of the owned limbs below the last one
    const size_t row = (size_t)(level - 2) * hp.L;
    std::vector<u64> c1, c1p, inv, invp;
    for (u32 k = 0; k < pl.nq && sh->q_own[k] < level - 1; ++k) {
      const u32 i = sh->q_own[k];
      c1.push_back(hp.qlql[row + i]);
      c1p.push_back(hp.qlql_prec[row + i]);
      inv.push_back(hp.ql_inv[row + i]);
      invp.push_back(hp.ql_inv_prec[row + i]);
    }
    pl.nq_rs = (u32)c1.size();
    if (pl.nq_rs) {
      pl.d_rs_c1 = c->up(c1);
      pl.d_rs_c1p = c->up(c1p);
      pl.d_rs_inv = c->up(inv);
      pl.d_rs_invp = c->up(invp);
    }
  }
  return &(sh->plans[level] = pl);
}
}  // namespace

acehip_shard* acehip_shard_create(acehip_ctx* c, uint32_t rank, uint32_t world) {
  if (!c) {
    fail(ACEHIP_EINVAL, "null context");
    return nullptr;
  }
  if (world == 0 || rank >= world) {
    fail(ACEHIP_EINVAL, "acehip_shard_create: rank outside [0, world)");
    return nullptr;
  }
  const HostParams& hp = c->hp;
  auto* sh = new acehip_shard();
  sh->c = c;
  sh->rank = rank;
  sh->world = world;
  std::vector<u32> qg, pg, all;
  for (u32 i = 0; i < hp.L; ++i)
    if (i % world == rank) sh->q_own.push_back(i), qg.push_back(i), all.push_back(i);
  for (u32 j = 0; j < hp.K; ++j)
    if ((hp.L + j) % world == rank) sh->p_own.push_back(j), pg.push_back(hp.L + j), all.push_back(hp.L + j);
  if (!c->on_device) return sh;  // host-only context: ownership / exchange-layout queries only, every phase returns ENODEV
  sh->d_q_gi = c->up(qg);
  sh->d_p_gi = c->up(pg);
  sh->d_own_gi_full = c->up(all);
  const size_t N = hp.N, nown = std::max<size_t>(all.size(), 1);
  auto dev = [&](size_t limbs) {
    u64* p = nullptr;
    if (hipMalloc((void**)&p, std::max<size_t>(limbs, 1) * N * sizeof(u64)) != hipSuccess) return (u64*)nullptr;
    c->owned.push_back(p);
    return p;
  };
  sh->full = dev(hp.L);
  sh->ext = dev((size_t)hp.dnum * nown);
  sh->acc = dev(2 * nown);
  sh->pfull = dev(2ull * hp.K);
  sh->conv = dev(2 * nown);
  sh->tmp = dev(2 * nown);
  if (!sh->full || !sh->ext || !sh->acc || !sh->pfull || !sh->conv || !sh->tmp) {
    fail(ACEHIP_EHIP, "acehip_shard_create: device allocation failed");
    delete sh;
    return nullptr;
  }
  return sh;
}
void acehip_shard_destroy(acehip_shard* sh) { delete sh; }  // device memory belongs to the context
uint32_t acehip_shard_num_q(const acehip_shard* sh, uint32_t level) { return sh ? shard_nq(sh, level) : 0; }
uint32_t acehip_shard_num_p(const acehip_shard* sh) { return sh ? (u32)sh->p_own.size() : 0; }
uint32_t acehip_shard_pad_q(const acehip_shard* sh, uint32_t level) { return sh ? shard_pad(sh, level) : 0; }
uint32_t acehip_shard_pad_p(const acehip_shard* sh) {
  if (!sh) return 0;
  u32 m = 0;  // p-limbs start at gi = L: count per rank
  for (u32 r = 0; r < sh->world; ++r) {
    u32 n = 0;
    for (u32 j = 0; j < sh->c->hp.K; ++j) n += (sh->c->hp.L + j) % sh->world == r;
    m = std::max(m, n);
  }
  return m;
}
uint32_t acehip_shard_owned(const acehip_shard* sh, uint32_t level, uint32_t* q_out, uint32_t* p_out) {
  if (!sh) return 0;
  const u32 nq = shard_nq(sh, level);
  if (q_out) std::copy(sh->q_own.begin(), sh->q_own.begin() + nq, q_out);
  if (p_out) std::copy(sh->p_own.begin(), sh->p_own.end(), p_out);
  return nq;
}

// phase 1: the owned q-limbs of the key-switch input (NTT domain, packed) to the coefficient domain, written into the
// send buffer of the first exchange ([pad_q][N]; the caller all-gathers it rank-major)
int acehip_shard_ks_phase1(acehip_shard* sh, uint64_t* d_send, const uint64_t* d_x_own, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level == 0 || level > c->hp.L || !d_send || !d_x_own) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase1: bad arguments");
  const u32 nq = shard_nq(sh, level);
  ntt_packed(c, d_send, d_x_own, nullptr, sh->d_q_gi, sh->q_own.data(), nq, 1, 0, true, (hipStream_t)s_);
  stat(ST_NTT, 1, 16ull * c->hp.N * nq);
  return post_launch();
}

// phase 2: from the gathered coefficient-domain limbs ([world][pad_q][N], rank-major): ModUp of every digit onto the owned
// limbs, NTT, key inner product over the owned limbs (d_key_own: [dnum][2][n_own][N], this rank's limbs of the switch
// key in packed order: owned q-limbs of the full chain, then owned p-limbs), inverse NTT of the owned p-limbs of both
// accumulators into the send buffer of the second exchange ([2][pad_p][N])
int acehip_shard_ks_phase2(acehip_shard* sh, uint64_t* d_send2, const uint64_t* d_gath, const uint64_t* d_x_own, const uint64_t* d_key_own,
                           uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_send2 || !d_gath || !d_x_own || !d_key_own) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase2: bad arguments");
  if (hp.L > 256 || hp.dnum * (sh->q_own.size() + sh->p_own.size()) > 256) return fail(ACEHIP_EINVAL, "acehip_shard: pointer table too large");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 G = sh->world, pad_q = shard_pad(sh, level), nq = pl->nq, np = (u32)sh->p_own.size(), nown = nq + np;
  const u32 nq_full = (u32)sh->q_own.size(), nown_full = nq_full + np;
  // (a) every q-limb in position order: limb i is the (i / G)-th owned limb of rank i % G
  PtrTab gt{};
  for (u32 i = 0; i < level; ++i) gt.p[i] = d_gath + ((size_t)(i % G) * pad_q + i / G) * N;
  launch_packed_gather(c->dc, sh->full, gt, level, s);
  // (b) ModUp: all digits in one launch onto the owned complement limbs; ext digit d at ext + d*nown_full*N
  const size_t ext_stride = (size_t)nown_full * N;
  if (pl->max_tgt) launch_base_conv_batch(c->dc, sh->ext, ext_stride, sh->full, 0, pl->d_up, 1, pl->nd, pl->max_tgt, s, hp.alpha);
  // (c) NTT of the raised limbs
  for (u32 d = 0; d < pl->nd; ++d) {
    std::vector<u32> h_gi(pl->n_tgt[d]);
    for (u32 k = 0; k < pl->n_tgt[d]; ++k) {
      const u32 y = pl->tgt_y[d][k];
      h_gi[k] = y < nq ? sh->q_own[y] : hp.L + sh->p_own[y - nq];
    }
    ntt_packed(c, sh->ext + d * ext_stride, nullptr, nullptr, pl->d_tgt_gi[d], h_gi.data(), pl->n_tgt[d], 1, 0, false, s);
  }
  // (d) key inner product over the owned limbs; e_d[y] = the input's own limb (y in digit d) or the raised limb
  PtrTab qt{}, pt{};
  for (u32 d = 0; d < pl->nd; ++d) {
    const u32 start = hp.alpha * d, n2 = std::min(hp.alpha, level - start);
    std::vector<const u64*> of_y(nown, nullptr);
    for (u32 k = 0; k < pl->n_tgt[d]; ++k) of_y[pl->tgt_y[d][k]] = sh->ext + d * ext_stride + (size_t)k * N;
    for (u32 y = 0; y < nq; ++y)
      if (sh->q_own[y] >= start && sh->q_own[y] < start + n2) of_y[y] = d_x_own + (size_t)y * N;
    for (u32 y = 0; y < nq; ++y) qt.p[d * nq + y] = of_y[y];
    for (u32 y = 0; y < np; ++y) pt.p[d * np + y] = of_y[nq + y];
  }
  u64 *acc0 = sh->acc, *acc1 = sh->acc + (size_t)nown_full * N;
  const size_t key_stride = (size_t)nown_full * N;
  launch_packed_key_mac(c->dc, acc0, acc1, d_key_own, key_stride, qt, sh->d_q_gi, pl->nd, nq, s);
  launch_packed_key_mac(c->dc, acc0 + (size_t)nq * N, acc1 + (size_t)nq * N, d_key_own + (size_t)nq_full * N, key_stride, pt,
                        sh->d_p_gi, pl->nd, np, s);
  // (e) owned p-limbs of both accumulators to the coefficient domain, into the send buffer [2][pad_p][N]
  const u32 pad_p = acehip_shard_pad_p(sh);
  std::vector<u32> h_pgi(np);
  for (u32 k = 0; k < np; ++k) h_pgi[k] = hp.L + sh->p_own[k];
  ntt_packed(c, d_send2, acc0 + (size_t)nq * N, acc1 + (size_t)nq * N, sh->d_p_gi, h_pgi.data(), np, 2, (size_t)pad_p * N, true, s);
  stat(ST_KEYMAC, 1, 8ull * N * nown * (3ull * pl->nd + 2));
  return post_launch();
}

// phase 3: from the gathered coefficient-domain p-limbs of both accumulators ([world][2][pad_p][N]): conversion P -> owned
// q-limbs, NTT, out = (acc - conv) * P^-1 on the owned q-limbs (packed)
int acehip_shard_ks_phase3(acehip_shard* sh, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_gath2, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || !d_out0 || !d_out1 || !d_gath2) return fail(ACEHIP_EINVAL, "acehip_shard_ks_phase3: bad arguments");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = hp.N;
  const u32 G = sh->world, pad_p = acehip_shard_pad_p(sh), nq = pl->nq, nq_full = (u32)sh->q_own.size(), np = (u32)sh->p_own.size();
  const u32 nown_full = nq_full + np;
  if (nq == 0) return 0;
  // p-limb j of accumulator z: rank (L + j) % G, its k-th owned p-limb
  PtrTab gt{};
  for (u32 z = 0; z < 2; ++z)
    for (u32 j = 0; j < hp.K; ++j) {
      const u32 r = (hp.L + j) % G;
      u32 k = 0;
      for (u32 jj = 0; jj < j; ++jj) k += (hp.L + jj) % G == r;
      gt.p[z * hp.K + j] = d_gath2 + (((size_t)r * 2 + z) * pad_p + k) * N;
    }
  launch_packed_gather(c->dc, sh->pfull, gt, 2 * hp.K, s);
  // both accumulators: P -> owned q-limbs (problem z reads pfull + z*K*N, writes conv + z*nq_full*N)
  launch_base_conv_batch(c->dc, sh->conv, (size_t)nq_full * N, sh->pfull, (size_t)hp.K * N, pl->d_down, 0, 2, nq, s, hp.K);
  ntt_packed(c, sh->conv, nullptr, nullptr, sh->d_q_gi, sh->q_own.data(), nq, 2, (size_t)nq_full * N, false, s);
  u64 *acc0 = sh->acc, *acc1 = sh->acc + (size_t)nown_full * N;
  launch_packed_moddown_tail(c->dc, d_out0, d_out1, acc0, acc1, sh->conv, (size_t)nq_full * N, sh->d_q_gi, pl->d_pinv, pl->d_pinv_prec, nq, 2, s);
  stat(ST_MODDOWN, 2, 8ull * N * (2ull * hp.K + 2ull * nq));
  return post_launch();
}

// Rescale (Rescale_poly polynomial.c:1097-1163), exchange = broadcast of the last limb of c0 and c1 in the coefficient
// domain.  _send: on the rank that owns limb level-1, writes it for both polynomials into d_send ([2][N]) and returns 1;
// the other ranks return 0 and receive the broadcast.  _apply: every rank, owned limbs below level-1 (packed, in and out).
int acehip_shard_rescale_send(acehip_shard* sh, uint64_t* d_send, const uint64_t* d_c0_own, const uint64_t* d_c1_own, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L || !d_send || !d_c0_own || !d_c1_own) return fail(ACEHIP_EINVAL, "acehip_shard_rescale_send: bad arguments");
  if ((level - 1) % sh->world != sh->rank) return 0;
  const size_t N = c->hp.N;
  const u32 y = shard_nq(sh, level) - 1;  // the last owned limb is limb level-1
  const u32 gi = level - 1;
  ntt_packed(c, d_send, d_c0_own + (size_t)y * N, d_c1_own + (size_t)y * N, sh->d_q_gi + y, &gi, 1, 2, N, true, (hipStream_t)s_);
  if (int e = post_launch()) return e;
  return 1;
}
int acehip_shard_rescale_apply(acehip_shard* sh, uint64_t* d_out0, uint64_t* d_out1, const uint64_t* d_c0_own, const uint64_t* d_c1_own,
                               const uint64_t* d_last, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  if (level < 2 || level > c->hp.L || !d_out0 || !d_out1 || !d_c0_own || !d_c1_own || !d_last) return fail(ACEHIP_EINVAL, "acehip_shard_rescale_apply: bad arguments");
  const ShardLevelPlan* pl = shard_plan(sh, level);
  if (!pl) return fail(ACEHIP_EHIP, "acehip_shard: plan upload failed");
  hipStream_t s = (hipStream_t)s_;
  const size_t N = c->hp.N;
  const u32 n = pl->nq_rs, nq_full = (u32)sh->q_own.size();
  if (n == 0) return 0;
  u64* t = sh->tmp;  // [2][nq_full][N]
  launch_packed_rescale_spread(c->dc, t, (size_t)nq_full * N, d_last, N, sh->d_q_gi, level - 1, pl->d_rs_c1, pl->d_rs_c1p, n, 2, s);
  ntt_packed(c, t, nullptr, nullptr, sh->d_q_gi, sh->q_own.data(), n, 2, (size_t)nq_full * N, false, s);
  launch_packed_rescale_tail(c->dc, d_out0, d_out1, d_c0_own, d_c1_own, t, (size_t)nq_full * N, sh->d_q_gi, pl->d_rs_inv, pl->d_rs_invp, n, 2, s);
  stat(ST_RESCALE, 2, 8ull * N * (2ull * n + 1) * 2);
  return post_launch();
}

// Encode (SURVEY 8e collective 4): the integer message of a plaintext (rounded, scaled inverse embedding: N signed words) is
// computed once (acehip_encode_message, any rank) and broadcast; every rank then reduces it into its own limbs and
// transforms them (acehip_shard_encode_limbs).  Ranks may also each compute the message themselves: no exchange at all.
int acehip_encode_message(acehip_ctx* c, int64_t* d_msg, const void* d_vals, int kind, size_t len, uint32_t slots, double sf, acehip_stream s) {
  if (int e = check_dev(c)) return e;
  const u32 N = c->hp.N;
  if (slots == 0) slots = N / 2;
  if (kind < 0 || kind > 2 || slots > N / 2 || (slots & (slots - 1)) || len > slots || !d_msg || (!d_vals && len))
    return fail(ACEHIP_EINVAL, "acehip_encode_message: bad arguments");
  if (int e = ensure_embed_tables(c)) return e;
  launch_embed_inv(d_msg, c->emb_work, d_vals, kind, len, slots, N, c->emb_rou, c->emb_rot, sf, c->emb_err, (hipStream_t)s);
  return post_launch();
}
int acehip_shard_encode_limbs(acehip_shard* sh, uint64_t* d_q_own, const int64_t* d_msg, double sf, uint32_t sf_degree, uint32_t level, acehip_stream s_) {
  if (!sh) return fail(ACEHIP_EINVAL, "null shard");
  acehip_ctx* c = sh->c;
  if (int e = check_dev(c)) return e;
  const HostParams& hp = c->hp;
  if (level == 0 || level > hp.L || sf_degree < 1 || !d_q_own || !d_msg) return fail(ACEHIP_EINVAL, "acehip_shard_encode_limbs: bad arguments");
  hipStream_t s = (hipStream_t)s_;
  const u32 nq = shard_nq(sh, level);
  if (nq == 0) return 0;
  const size_t N = hp.N;
  const u64 sfi = (u64)sf;
  std::vector<u64> w(nq, 1);
  for (u32 k = 0; k < nq && sf_degree > 1; ++k) {  // ckks_encoder.c:270-285: times Delta^(sf_degree-1)
    const u64 q = hp.primes[sh->q_own[k]].q;
    u64 pw = sfi % q;
    for (u32 d = 2; d < sf_degree; ++d) pw = (u64)(((unsigned __int128)pw * (sfi % q)) % q);
    w[k] = pw;
  }
  if (c->dc.logN == 16) {
    NttFuse f;
    f.gi_tab = sh->d_q_gi;
    f.msg = d_msg;
    if (sf_degree > 1) {
      u64* tab = sh->tmp;  // [nq] words at the head of tmp (stream ordered)
      HIP_TRY(hipMemcpyAsync(tab, w.data(), nq * sizeof(u64), hipMemcpyHostToDevice, s));
      f.msg_scale = tab;
    }
    launch_ntt_fused(c->dc, d_q_own, 0, 0, nq, false, s, 0, 1, 0, 0, f);
  } else {
    for (u32 k = 0; k < nq; ++k) {
      const u32 gi = sh->q_own[k];
      u64* ptr = d_q_own + (size_t)k * N;
      launch_values_to_rns(c->dc, ptr - (size_t)gi * N, d_msg, hp.L, gi, 1, s);
      if (sf_degree > 1) {
        LimbConsts lc{};
        lc.w[0] = w[k];
        launch_mul_scalars(c->dc, ptr - (size_t)gi * N, ptr - (size_t)gi * N, lc, hp.L, gi, 1, s);
      }
      launch_ntt(c->dc, ptr, hp.L, gi, 1, false, s, gi);
    }
  }
  stat(ST_ENCODE, 1, 8ull * N * (nq + 1));
  return post_launch();
}

int acehip_stats(acehip_stat* out, int n, int reset) {
  for (int i = 0; i < n && i < ST_COUNT; ++i) out[i] = g_stat[i];
  if (reset) std::memset(g_stat, 0, sizeof(g_stat));
  return ST_COUNT;
}
const char* acehip_stat_name(int i) { return i >= 0 && i < ST_COUNT ? kStatName[i] : nullptr; }

uint64_t acehip_key_switch_bytes(const acehip_ctx* c, uint32_t level) {
  const HostParams& hp = c->hp;
  const u64 b = hp.num_decomp(level);
  return 8ull * hp.N * (level + 2 * b * (level + hp.K) + 2 * level);
}

}  // extern "C"
