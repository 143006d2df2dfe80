// api_internal.hpp -- shared between the translation units of the C ABI (include/acehip.h):
//   api_core.cpp      context, tables, memory helpers, statistics
//   api_hw_batch.cpp  acehip_hw_batch: analysis of a per-limb op list and its launches
//   api_ops.cpp       the launch entry points and the Decomp_modup / Mod_down / Rescale / key-switch / encode pipelines
//   api_shard.cpp     packed-list phases of limb-sharded execution (acehip_shard_*)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/acehip.h"
#include "host_params.hpp"
#include "kernels.hpp"

using namespace acehip;

std::string& acehip_err_slot();  // this thread's last error message
inline int fail(int code, const std::string& msg) {
  acehip_err_slot() = msg;
  return code;
}
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(ACEHIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

static_assert(sizeof(PrimeConsts) == sizeof(DevPrime), "host/device prime layout mismatch");

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

// Per-level plan of the batched key-switch: one ConvDesc per digit (ModUp) and one for ModDown.
struct KsPlan {
  ConvDesc* d_descs = nullptr;  // [nd] ModUp problems, then [1] ModDown problem
  u32 nd = 0, max_nc = 0;
  // N = 2^16: the pre-factors of both base conversions ride in the last stage of the inverse NTT (NttFuse::inv_scale)
  u64* inv_up = nullptr;    // [level][4]
  u64* inv_down = nullptr;  // [K][4]
  // k-steps of the matrix-core conversion (ConvDesc::bfrag) prepared for the ModUp / the ModDown problems; 0: not prepared
  u32 mfma_up = 0, mfma_down = 0;
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
  // ---- replicas of the caller's polynomial arena (acehip_ctx_set_arena): image batches, simulated ranks.  dc.rep_lo / rep_span /
  // rep_stride describe the arena; launches cover replicas [sel0, sel0 + seln) (acehip_ctx_select)
  u32 n_replicas = 1, sel0 = 0, seln = 1;
  u32 stat_reps = 0;  // replicas one call counts for in the statistics while its launches go out in chunks (0: seln)
  bool ws_external = false, scratch_external = false;  // workspace / hw scratch handed in by the caller (inside the arena)
  // ---- limb-sharded execution (api_shard.cpp): world ranks, limb gi belongs to rank gi % world.  hosted[h] = rank whose limbs
  // live in replica h of the arena: every rank of a simulation (ACEHIP_SHARD_SIM: exchanges are copies between replicas), or
  // the one rank of this process (exchanges are RCCL broadcasts from the owner)
  u32 sh_world = 1;
  std::vector<u32> sh_hosted;
  void* rccl = nullptr;            // RcclComm* (api_shard.cpp) when the ranks are processes
  u64 xchg_bytes = 0, xchg_calls = 0;  // bytes this process received through exchanges / exchange steps
  u64 xchg_collectives = 0;            // RCCL collectives issued for them (one per packed exchange, one per limb otherwise)

  template <typename T>
  T* up(const std::vector<T>& v) {
    T* d = upload(v);
    if (d) owned.push_back(d);
    return d;
  }
};

// ---- call statistics: algorithmic bytes of SURVEY 8(d) per entry point (tables and scratch excluded) ----
enum { ST_NTT, ST_EW, ST_ROTATE, ST_MODUP, ST_KEYMAC, ST_MODDOWN, ST_RESCALE, ST_KEYSWITCH, ST_ENCODE, ST_ZERO_RUN, ST_EW_MUL, ST_NTT_ALL, ST_COUNT };
acehip_stat* acehip_stat_slots();  // this thread's counters [ST_COUNT] (one host thread = one image stream)
u32& acehip_stat_mult();  // replicas the current call covers (set by check_dev): an op on B images counts B times
bool& acehip_stat_mute();  // replica chunks after the first of one call (for_replica_chunks): already counted
inline void stat(int k, u64 units, u64 bytes) {
  if (acehip_stat_mute()) return;
  acehip_stat* g = acehip_stat_slots();
  const u32 m = acehip_stat_mult();
  g[k].calls++;
  g[k].units += units * m;
  g[k].bytes += bytes * m;
}

// ---- argument checks shared by the launch entry points ----
int check_dev(acehip_ctx* c);
int check_range(acehip_ctx* c, uint32_t level, uint32_t pos0, uint32_t n);
int post_launch();
const DevModUp* get_modup(acehip_ctx* c, u32 level, u32 digit);
const KsPlan* get_ks_plan(acehip_ctx* c, u32 level);
bool conv_fusable(const acehip_ctx* c, u32 n_in);
// workspace carving (in limbs of N words)
inline u64* ws_at(acehip_ctx* c, size_t limb) { return c->ws + limb * c->hp.N; }
int ensure_embed_tables(acehip_ctx* c);
// debug aid (acehip_debug_touches): the caller's memory a pipeline entry point reads or writes, as the entry point itself sees it
void dbg_touch(const void* p, size_t words);

// ---- replicas / limb ownership ----
// The launch sets one call has to issue: ONE DevCtx covering the selected replicas, or -- simulated limb-sharded execution --
// one per hosted rank (replica h, owner filter hosted[h]).
struct DcList {
  DevCtx d[16];
  u32 n = 0;
  const DevCtx* begin() const { return d; }
  const DevCtx* end() const { return d + n; }
};
DcList launch_dcs(const acehip_ctx* c);
// A pipeline of several launches over the images of a batch, issued chunk by chunk: body() runs once per group of
// ACEHIP_REP_CHUNK replicas (the selection narrowed to the group) instead of once over all selected replicas, so that what one
// launch writes is still in the 256 MiB Infinity Cache when the next launch of the pipeline reads it -- the hand-off between the
// two passes of a transform, the coefficient-domain limbs between an inverse transform, a conversion and the forward transform.
// Replicas are independent, so the results are the same bits in any grouping.  0 / unset: one group (off); never when limb-sharded.
// The narrowing is written into the context (sel0 / seln / stat_reps) for the duration of the loop: like acehip_ctx_select itself it
// assumes ONE host thread per acehip_ctx, which is how every caller uses a context (the rt_ant shim gives each thread its own;
// include/acehip.h "Threads").  An error of a group ends the loop and is returned.
u32 replica_chunk();
template <class F>
int for_replica_chunks(acehip_ctx* c, F&& body) {
  const u32 chunk = replica_chunk();
  if (chunk == 0 || c->sh_world > 1 || c->seln <= chunk) return body();
  const u32 s0 = c->sel0, sn = c->seln;
  int rc = 0;
  c->stat_reps = sn;
  acehip_stat_mult() = sn;
  for (u32 r = 0; r < sn && rc >= 0; r += chunk) {
    c->sel0 = s0 + r;
    c->seln = sn - r < chunk ? sn - r : chunk;
    acehip_stat_mute() = r != 0;
    rc = body();
  }
  acehip_stat_mute() = false;
  c->stat_reps = 0;
  c->sel0 = s0;
  c->seln = sn;
  return rc;
}
inline bool sharded(const acehip_ctx* c) { return c->sh_world > 1; }
inline bool dc_owns(const DevCtx& dc, u32 gi) { return dc.sh_world <= 1 || gi % dc.sh_world == dc.sh_rank; }
// dst limb i = src limb i for i < n_limbs, limb i having prime gi0 + i (only the limbs the DevCtx owns; every replica it covers):
// the d2d copies of the pipelines
void copy_limbs_dc(const DevCtx& dc, u64* dst, const u64* src, u32 n_limbs, u32 gi0, hipStream_t s);
// Limbs that are valid on their owner only become valid on every rank (no-op when not sharded).  Addresses as the caller sees
// them (replica 0 of the arena / shared memory); root = owning rank.
struct XItem {
  u64* ptr;
  u32 root;
};
int shard_exchange(acehip_ctx* c, const XItem* items, size_t n, hipStream_t s);
int shard_exchange_begin(acehip_ctx* c, const XItem* items, size_t n, hipStream_t s);  // enqueue, ordered after s so far; s does not wait
int shard_exchange_end(acehip_ctx* c, hipStream_t s);                                   // s waits for every exchange begun since
void shard_release(acehip_ctx* c);  // RCCL communicator + exchange stream / events of a context (acehip_ctx_destroy)
