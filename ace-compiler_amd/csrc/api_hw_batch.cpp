// api_hw_batch.cpp -- acehip_hw_batch: a list of per-limb ops (Hw_modadd / Hw_modmul / Hw_rotate calls of generated code,
// poly_arith.c:14-56) analysed on the host and executed as a few launches, as if issued one by one.
#include "api_internal.hpp"

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
  std::vector<uint8_t> from_mem;
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
// the launch set the list is being issued for (replicas of the arena, or one simulated rank): hw_batch_run
static thread_local const DevCtx* g_dc = nullptr;
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
static std::atomic<u64> g_hw_traffic[9][4];  // [kind][ops, limb loads, limb stores, loads of memory all replicas share]
static std::atomic<u64> g_hw_what_if[3];     // limb loads the same launches would make with 3, 4, 8 register entries instead of 2
static bool hw_traffic_on() {
  static const bool on = [] {
    const bool v = getenv("ACEHIP_HW_TRAFFIC") != nullptr;
    if (v)
      atexit([] {
        static const char* const kn[9] = {"add", "mul", "rotate", "copy", "zero", "sub", "muladd", "mulc", "addc"};
        u64 tl = 0, ts = 0;
        for (int k = 0; k < 9; ++k) {
          const u64 o = g_hw_traffic[k][0], l = g_hw_traffic[k][1], w = g_hw_traffic[k][2];
          if (o) fprintf(stderr, "[hw traffic] %-7s ops %10llu  limb loads %10llu (shared by the replicas: %llu)  limb stores %10llu\n", kn[k],
                         (unsigned long long)o, (unsigned long long)l, (unsigned long long)g_hw_traffic[k][3], (unsigned long long)w);
          tl += l;
          ts += w;
        }
        fprintf(stderr, "[hw traffic] total limb loads %llu stores %llu\n", (unsigned long long)tl, (unsigned long long)ts);
        fprintf(stderr, "[hw traffic] what if the kernel kept more results in registers: limb loads with 3 entries %llu, 4 entries %llu, 8 entries %llu\n",
                (unsigned long long)g_hw_what_if[0], (unsigned long long)g_hw_what_if[1], (unsigned long long)g_hw_what_if[2]);
      });
    return v;
  }();
  return on;
}
static void hw_traffic_count(const HwBatchArgs& args, u32 n_seg, const DevCtx& dc) {
  auto shared = [&](const void* p) { return !((u64)p - dc.rep_lo < dc.rep_span); };  // outside the replicated arena
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
  for (int w = 0; w < 3; ++w) {  // the same rule (a result replaces its own limb's entry, otherwise pushes the others down) with more entries
    const int cap = w == 0 ? 3 : (w == 1 ? 4 : 8);
    u64 loads = 0;
    for (u32 sgm = 0; sgm < n_seg; ++sgm) {
      const u64* e[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      auto miss = [&](const u64* x) {
        for (int i = 0; i < cap; ++i)
          if (e[i] == x) return false;
        return true;
      };
      for (u32 k = args.seg_start[sgm]; k < args.seg_start[sgm + 1]; ++k) {
        const HwBatchOp& o = args.op[k];
        const u32 kind = o.kind & HW_OP_KIND_MASK;
        if (kind != HW_OP_ZERO) {
          loads += miss(o.a);
          if (kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD) loads += miss(o.b);
          if (kind == HW_OP_MULADD) loads += miss(o.res);
        }
        int at = cap - 1;
        for (int i = 0; i < cap; ++i)
          if (e[i] == o.res) at = i;
        for (int i = at; i > 0; --i) e[i] = e[i - 1];
        e[0] = o.res;
      }
    }
    g_hw_what_if[w] += loads;
  }
  for (u32 sgm = 0; sgm < n_seg; ++sgm) {
    const u64 *r0 = nullptr, *r1 = nullptr, *rb = nullptr;  // the kernel's two result entries and its operand entry
    const u32 beg = args.seg_start[sgm], end = args.seg_start[sgm + 1];
    for (u32 k = beg; k < end; ++k) {
      const HwBatchOp& o = args.op[k];
      const u32 kind = o.kind & HW_OP_KIND_MASK;
      auto miss = [&](const u64* x) { return x != r0 && x != r1 && x != rb; };
      u64 loads = 0, sh = 0;
      if (kind != HW_OP_ZERO) {
        loads += miss(o.a);
        sh += miss(o.a) && shared(o.a);
        if (kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD) {
          loads += miss(o.b);
          sh += miss(o.b) && shared(o.b);
          if (o.b != r0 && o.b != r1) rb = o.b;
        }
        if (kind == HW_OP_MULADD) loads += miss(o.res);
      }
      if (o.res == rb) rb = nullptr;
      g_hw_traffic[kind][3] += sh;
      const bool keep = (o.kind & HW_OP_NOSTORE) || (k + 1 < end && args.op[k + 1].res == o.res);
      g_hw_traffic[kind][0] += 1;
      g_hw_traffic[kind][1] += loads;
      g_hw_traffic[kind][2] += !keep;
      if (o.res != r0) r1 = r0;
      r0 = o.res;
    }
  }
}
static void emit_ew(acehip_ctx* c, const HwBatchArgs& args, u32 n_seg, hipStream_t st) {
  if (hw_traffic_on()) hw_traffic_count(args, n_seg, *g_dc);
  if (!g_plan) {
    launch_hw_batch_ew(*g_dc, args, n_seg, st);
    return;
  }
  for (u32 sgm = 0; sgm < n_seg; ++sgm)
    for (u32 k = args.seg_start[sgm]; k < args.seg_start[sgm + 1]; ++k) plan_append(args.op[k], sgm);
  ++g_plan->launches;
}
static void emit_rotate(acehip_ctx* c, const HwBatchArgs& args, u32 n_ops, hipStream_t st) {
  if (!g_plan) {
    launch_hw_batch_rotate(*g_dc, args, n_ops, st);
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
  if (c->scratch_external) return nullptr;  // the caller's block inside its arena: fixed size (the analysis stays within it)
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
  const size_t scratch_cap = c->scratch_external ? std::min(c->hw_scratch_limbs, kHwScratchMaxLimbs) : kHwScratchMaxLimbs;
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
        // (the first version too, although nothing earlier in the list can collide with it: a private version is what the
        // product + accumulate fusions below look for)
        if (h.last_pure[nr0] != k && n_scratch < scratch_cap) {
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
  // The product + accumulate fusion again, this time on neighbours WITHIN a chain: generated code multiplies both polynomials of
  // a ciphertext before it accumulates them (mul c0, mul c1, add c0, add c1 per limb: Mul_plain + Add_ciph of a convolution
  // tap), so in list order a product is never next to its accumulation, but in its chain it is.  Between two neighbours of a
  // chain no other op touches a limb the chain writes, and every limb written in the list belongs to the chain of its
  // readers, so the product's operands are the same at the accumulation's place.  Runs of taps then keep the accumulator in
  // registers (hw_batch_ew_kernel) instead of storing and reloading it per tap.
  static const bool chain_fuse = [] { const char* e = getenv("ACEHIP_HW_CHAIN_FUSE"); return !e || atoi(e) != 0; }();  // (measurement knob)
  if (n_nodes > n_base && chain_fuse) {
    bool fused_any = false;
    u32 ch = 0;
    for (size_t t = 1; t < live; ++t) {
      while (t >= h.cnt[ch]) ++ch;
      const size_t beg = ch ? h.cnt[ch - 1] : 0;
      if (t == beg) continue;  // first op of its chain
      const u32 k = h.ord[t], p = h.ord[t - 1];
      if (h.dead[p] || h.kind[k] != ACEHIP_HW_ADD || h.kind[p] != ACEHIP_HW_MUL) continue;
      const u32 tv = h.n_res[p], acc = h.n_res[k];
      if (tv < n_base || h.readers[tv] != 1 || ops[p].prime_gi != ops[k].prime_gi) continue;
      const bool acc_a = h.n_a[k] == acc && h.n_b[k] == tv, acc_b = h.n_b[k] == acc && h.n_a[k] == tv;
      if (!(acc_a || acc_b) || acc == tv) continue;
      h.kind[k] = ACEHIP_HW_MULADD;
      h.n_a[k] = h.n_a[p];
      h.n_b[k] = h.n_b[p];
      h.dead[p] = 1;
      fused_any = true;
    }
    if (fused_any) {  // drop the absorbed products from the emission order
      size_t w = 0, t = 0;
      for (size_t j = 0; j < h.cnt.size(); ++j) {
        for (; t < h.cnt[j]; ++t)
          if (!h.dead[h.ord[t]]) h.ord[w++] = h.ord[t];
        h.cnt[j] = (u32)w;
      }
      live = w;
      h.ord.resize(live);
    }
  }
  // Siblings (round 5): chains that read the SAME read-only limb as second operand at neighbouring places of the program -- the c0 and
  // the c1 chain of a plaintext product (both multiply by the plaintext's limb), the two accumulators of a key inner product (both
  // multiply by the raised digit's limb) -- are put into ONE segment, ops in program order, so that the kernel's operand entry
  // (hw_batch_ew_kernel) loads the shared limb once.  Chains are independent (no written limb in common), so any merge that keeps program
  // order inside the segment computes the same values; the accumulators of two siblings fit the kernel's two result entries.
  static const bool siblings_on = [] { const char* e = getenv("ACEHIP_HW_SIBLINGS"); return !e || atoi(e) != 0; }();  // (measurement knob)
  if (siblings_on && h.cnt.size() > 1) {
    const u32 n_ch = (u32)h.cnt.size();
    static thread_local std::vector<u32> ch_of, ch_parent, ch_size, last_reader, grp_start;
    ch_of.assign(m, UINT32_MAX);
    ch_parent.resize(n_ch);
    ch_size.resize(n_ch);
    {
      u32 ch = 0;
      for (size_t t = 0; t < live; ++t) {
        while (t >= h.cnt[ch]) ++ch;
        ch_of[h.ord[t]] = ch;
      }
      for (u32 j = 0; j < n_ch; ++j) {
        ch_parent[j] = j;
        ch_size[j] = h.cnt[j] - (j ? h.cnt[j - 1] : 0);
      }
    }
    // only chains that (but for one op) write ONE limb (an accumulator, a plain result) are merged: two of them fill the kernel's two result entries;
    // a chain that alternates between a temporary and an accumulator needs both entries for itself (merging such chains measured
    // 3 % MORE stores: the temporaries evicted each other)
    static thread_local std::vector<char> one_limb;
    one_limb.assign(n_ch, 1);
    {
      static thread_local std::vector<u32> first_res, other_writes;
      first_res.assign(n_ch, UINT32_MAX);
      other_writes.assign(n_ch, 0);
      for (size_t t = 0; t < live; ++t) {
        const u32 k = h.ord[t], ch = ch_of[k], nr = h.n_res[k];
        if (first_res[ch] == UINT32_MAX) first_res[ch] = nr;
        else if (first_res[ch] != nr && ++other_writes[ch] > 1) one_limb[ch] = 0;  // (one stray write is fine: the last, visible product of a tap loop)
      }
    }
    last_reader.assign(n_nodes, UINT32_MAX);  // per read-only limb: the live op that read it last as second operand
    bool merged = false;
    constexpr u32 kWindow = 6, kMaxSegment = HW_BATCH_MAX / 2;
    u32 live_seen = 0;
    static thread_local std::vector<u32> live_idx;
    live_idx.assign(m, 0);
    for (size_t k = 0; k < m; ++k) {
      if (h.dead[k]) continue;
      live_idx[k] = live_seen++;
      if (!hw_has_b(h.kind[k])) continue;
      const u32 nb = h.n_b[k];
      if (h.written[nb]) continue;
      const u32 p = last_reader[nb];
      last_reader[nb] = (u32)k;
      if (p == UINT32_MAX || live_idx[k] - live_idx[p] > kWindow || ops[p].prime_gi != ops[k].prime_gi) continue;
      if (!one_limb[ch_of[k]] || !one_limb[ch_of[p]]) continue;
      const u32 ca = uf_find(ch_parent, ch_of[k]), cb = uf_find(ch_parent, ch_of[p]);
      if (ca == cb || ch_size[ca] + ch_size[cb] > kMaxSegment) continue;
      if (ch_size[ca] != (h.cnt[ca] - (ca ? h.cnt[ca - 1] : 0)) || ch_size[cb] != (h.cnt[cb] - (cb ? h.cnt[cb - 1] : 0))) continue;  // pairs only
      ch_parent[cb] = ca;
      ch_size[ca] += ch_size[cb];
      merged = true;
    }
    if (merged) {  // new emission order: groups by first appearance, ops of a group in program order
      static thread_local std::vector<u32> grp_of_root, new_cnt, new_ord;
      grp_of_root.assign(n_ch, UINT32_MAX);
      new_cnt.clear();
      for (size_t k = 0; k < m; ++k) {
        if (h.dead[k]) continue;
        const u32 root = uf_find(ch_parent, ch_of[k]);
        if (grp_of_root[root] == UINT32_MAX) {
          grp_of_root[root] = (u32)new_cnt.size();
          new_cnt.push_back(0);
        }
        new_cnt[grp_of_root[root]]++;
      }
      u32 run2 = 0;
      for (auto& x : new_cnt) {
        const u32 t = x;
        x = run2;
        run2 += t;
      }
      new_ord.resize(live);
      for (size_t k = 0; k < m; ++k)
        if (!h.dead[k]) new_ord[new_cnt[grp_of_root[uf_find(ch_parent, ch_of[k])]]++] = (u32)k;
      h.ord.swap(new_ord);
      h.cnt.assign(new_cnt.begin(), new_cnt.end());
    }
  }
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
    // Forwards: the kernel's register cache (hw_batch_ew_kernel: the two most recent results of a segment, keyed by limb; a
    // result replaces its own limb's entry, otherwise the newer entry becomes the older one), replayed per segment -- a
    // segment is a chain's run inside one launch -- to know which operand reads come from memory.  Entries are compared by
    // address, as the kernel does.
    std::vector<uint8_t>& from_mem = h.from_mem;  // per op: bit 0 operand a, bit 1 operand b, bit 2 the accumulator of a multiply-add
    from_mem.assign(m, 0);
    {
      u32 ch = 0;
      u64 r0 = 0, r1 = 0;
      for (size_t t = 0; t < live; ++t) {
        const u32 before = ch;
        while (t >= h.cnt[ch]) ++ch;
        if (t == 0 || ch != before || t % HW_BATCH_MAX == 0) r0 = r1 = 0;  // a new segment starts with empty registers
        const u32 k = h.ord[t], kind = h.kind[k];
        auto cached = [&](u32 node) { const u64 a = h.node_ptr[node]; return a == r0 || a == r1; };
        uint8_t f = 0;
        if (hw_has_a(kind) && !cached(h.n_a[k])) f |= 1;
        if (hw_has_b(kind) && !cached(h.n_b[k])) f |= 2;
        if (kind == ACEHIP_HW_MULADD && !cached(h.n_res[k])) f |= 4;
        from_mem[k] = f;
        const u64 res = h.node_ptr[h.n_res[k]];
        if (res != r0) r1 = r0;
        r0 = res;
      }
    }
    // Backwards: need[x] = the version of limb x that is current here is loaded from memory by a later op or outlives the list
    for (size_t k = m; k-- > 0;) {
      if (h.dead[k]) continue;
      const u32 kind = h.kind[k], nr = h.n_res[k];
      h.nostore[k] = !h.need[nr];
      h.need[nr] = 0;
      if (hw_has_a(kind) && (h.from_mem[k] & 1)) h.need[h.n_a[k]] = 1;
      if (hw_has_b(kind) && (h.from_mem[k] & 2)) h.need[h.n_b[k]] = 1;
      if (kind == ACEHIP_HW_MULADD && (h.from_mem[k] & 4)) h.need[nr] = 1;
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

// ---- stage order: elementwise ops and gathers of one list sorted into as few alternating runs as their dependencies allow ----
// A list runs as alternating elementwise / rotation runs (below).  In program order every Hw_rotate cuts the elementwise run it
// sits in, and with it every accumulation chain that continues behind it -- although the chain usually has nothing to do with that
// gather (the queue of the rt_ant shim now stays open across the direct launches of a key-switch, csrc/rt/rt_poly.cpp "keeping ops
// queued", so one list holds the tails of several rotations: add d0 + c0, gather, multiply by a plaintext, accumulate).  Every op
// gets a stage: elementwise ops even, gathers odd, at least the stage of every earlier op it depends on through memory (read after
// write, write after write, write after read), rounded up to its own parity.  Ops are then issued stage by stage, program order kept
// inside a stage: two ops that conflict are never swapped (equal stages keep their order inside one run, whose analysis -- hw_run_ew,
// hw_run_rotate -- sees them in program order), everything else is independent.  Lists with partially overlapping limbs are left alone.
static bool hw_stage_order(acehip_ctx* c, const acehip_hw_op* ops, size_t n, std::vector<acehip_hw_op>& out) {
  static thread_local HwScratch hs;
  static thread_local std::vector<int> last_w, max_r, stage;
  static thread_local std::vector<u32> count;
  const u64 span = (u64)c->hp.N * 8;
  size_t cap = 64;
  while (cap < 6 * n) cap <<= 1;
  if (hs.table.size() < cap) hs.table.resize(cap);
  std::memset(hs.table.data(), 0, cap * sizeof(HwScratch::Slot));
  const u64 mask = cap - 1;
  hs.parent.clear();
  hs.node_ptr.clear();
  last_w.clear();
  max_r.clear();
  stage.resize(n);
  int top = 0;
  bool in_order = true;
  for (size_t k = 0; k < n; ++k) {
    const acehip_hw_op& o = ops[k];
    const bool rot = o.op == ACEHIP_HW_ROTATE;
    const u32 nr = hw_node(hs, (u64)o.res, span, mask);
    const u32 na = hw_has_a(o.op) ? hw_node(hs, (u64)o.a, span, mask) : UINT32_MAX - 1;
    const u32 nb = hw_has_b(o.op) ? hw_node(hs, (u64)o.b, span, mask) : UINT32_MAX - 1;
    if (nr == UINT32_MAX || na == UINT32_MAX || nb == UINT32_MAX) return false;
    last_w.resize(hs.parent.size(), -1);
    max_r.resize(hs.parent.size(), -1);
    int s = rot ? 1 : 0;
    if (na < UINT32_MAX - 1) s = std::max(s, last_w[na]);
    if (nb < UINT32_MAX - 1) s = std::max(s, last_w[nb]);
    s = std::max(s, std::max(last_w[nr], max_r[nr]));
    if ((s & 1) != (rot ? 1 : 0)) ++s;
    stage[k] = s;
    if (na < UINT32_MAX - 1) max_r[na] = std::max(max_r[na], s);
    if (nb < UINT32_MAX - 1) max_r[nb] = std::max(max_r[nb], s);
    last_w[nr] = s;
    max_r[nr] = o.op == ACEHIP_HW_MULADD ? s : -1;  // (a multiply-add reads the limb it rewrites)
    if (k && s < stage[k - 1]) in_order = false;
    top = std::max(top, s);
  }
  if (in_order) return false;  // program order is stage order already
  count.assign((size_t)top + 2, 0);
  for (size_t k = 0; k < n; ++k) count[(size_t)stage[k] + 1]++;
  for (size_t i = 1; i < count.size(); ++i) count[i] += count[i - 1];
  out.resize(n);
  for (size_t k = 0; k < n; ++k) out[count[(size_t)stage[k]]++] = ops[k];
  return true;
}

static int hw_batch_run_one(acehip_ctx* c, const acehip_hw_op* ops, size_t n, hipStream_t st, const acehip_hw_range* dead, size_t n_dead);
// Limb-sharded execution: every rank gets the list (SPMD) and runs the ops on the limbs it owns -- prime_gi names the limb
// of every op (ACEHIP_HW_ANY_RANK: an op on memory that is not a limb of the chain, run by everyone).
static int hw_batch_run(acehip_ctx* c, const acehip_hw_op* ops, size_t n, hipStream_t st, const acehip_hw_range* dead = nullptr,
                        size_t n_dead = 0) {
  const DcList dcs = launch_dcs(c);
  int rc = ACEHIP_OK;
  static thread_local std::vector<acehip_hw_op> mine;
  for (const DevCtx& dc : dcs) {
    g_dc = &dc;
    if (dc.sh_world > 1) {
      mine.clear();
      for (size_t k = 0; k < n; ++k)
        if (ops[k].prime_gi == ACEHIP_HW_ANY_RANK || ops[k].prime_gi % dc.sh_world == dc.sh_rank) mine.push_back(ops[k]);
      rc = hw_batch_run_one(c, mine.data(), mine.size(), st, dead, n_dead);
    } else {
      rc = hw_batch_run_one(c, ops, n, st, dead, n_dead);
    }
    if (rc) break;
  }
  g_dc = nullptr;
  return rc;
}
static int hw_batch_run_one(acehip_ctx* c, const acehip_hw_op* ops, size_t n, hipStream_t st, const acehip_hw_range* dead, size_t n_dead) {
  if (n == 0) return ACEHIP_OK;
  if (!ops) return fail(ACEHIP_EINVAL, "acehip_hw_batch: null op list");
  if (n_dead && !dead) return fail(ACEHIP_EINVAL, "acehip_hw_batch_discard: null range list");
  const u32 T = c->hp.L + c->hp.K;
  const u64 span = (u64)c->hp.N * 8;
  // limbs moved per op (24N per add/mul, 16N per rotate; copy 16N, zero 8N, muladd 32N, scalar forms 16N)
  static const u64 kHwWords[9] = {3, 3, 2, 2, 1, 3, 4, 2, 2};
  u64 alg_words = 0, n_rot = 0, n_mul = 0;
  for (size_t k = 0; k < n; ++k) {
    const acehip_hw_op& o = ops[k];
    if (o.op > ACEHIP_HW_ADDC) return fail(ACEHIP_EINVAL, "acehip_hw_batch: unknown op");
    alg_words += kHwWords[o.op];
    n_rot += o.op == ACEHIP_HW_ROTATE;
    n_mul += o.op == ACEHIP_HW_MUL || o.op == ACEHIP_HW_MULADD || o.op == ACEHIP_HW_MULC;
    if (!o.res || (hw_has_a(o.op) && !o.a) || ((hw_has_b(o.op) || o.op == ACEHIP_HW_ROTATE) && !o.b))
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: null operand");
    if (hw_uses_prime(o.op) && o.prime_gi >= T) return fail(ACEHIP_EINVAL, "prime index out of range");
    if ((o.op == ACEHIP_HW_MULC || o.op == ACEHIP_HW_ADDC) && (u64)(uintptr_t)o.b >= c->hp.primes[o.prime_gi].q)
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: scalar operand is not a residue of the prime");
    if (o.op == ACEHIP_HW_ROTATE && limbs_overlap(o.res, o.a, span))
      return fail(ACEHIP_EINVAL, "acehip_hw_batch: in-place rotation is not supported");
  }
  // fewer, longer runs where the dependencies allow it (ACEHIP_HW_STAGES=0: program order, as before round 5)
  static const bool stages_on = [] { const char* e = getenv("ACEHIP_HW_STAGES"); return !e || atoi(e) != 0; }();
  static thread_local std::vector<acehip_hw_op> staged;
  if (stages_on && n_rot && n_rot < n && hw_stage_order(c, ops, n, staged)) ops = staged.data();
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
    if (n_mul) {
      stat(ST_EW_MUL, n_mul, 3 * n_mul * span);
      acehip_stat_slots()[ST_EW_MUL].calls--;  // (a subset of "elementwise": the limb-ops that multiply)
    }
    if (n_rot) {
      stat(ST_ROTATE, n_rot, 2 * n_rot * span);
      acehip_stat_slots()[ST_ROTATE].calls--;  // one entry point call, counted under elementwise
    }
  }
  return ACEHIP_OK;
}

extern "C" {

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

}  // extern "C"
