// hw_batch.hip -- many per-limb Hw_modadd / Hw_modmul / Hw_rotate calls (poly_arith.c:14-56) in a handful of
// launches.  ACE-generated code spells every ciphertext operation as a host loop over the RNS limbs, one
// Hw_* call per limb and component; launched one by one that is ~600k kernels of 1.5 MB each per ResNet-20
// image and the GPU idles on launch latency.  The host side (api.cpp acehip_hw_batch) groups a list of such
// ops into dependency chains (ops that touch a common written limb, kept in program order) and hands up to
// HW_BATCH_MAX ops to one launch: blockIdx.y walks one chain segment in order, so a coefficient's whole
// history stays in one lane and the sequential semantics of the original call sequence are preserved.
#include <algorithm>

#include "device_arith.hpp"
#include "kernels.hpp"

namespace acehip {

// Every lane owns ACEHIP_HW_LANES (2 or 4) consecutive coefficients of all limbs of its segment: read-after-write between ops of a chain goes
// through the lane itself.  The two most recent results stay in registers (see the kernel): an operand that is one of them
// is not reloaded, and a result is not stored when the next op of the segment writes the same limb again
// (accumulation runs res += a_j * b_j keep the accumulator in registers; the last op of a run stores, so every
// later reader -- in this segment, another launch or the host -- finds the final value in memory), nor when the host
// analysis found that only the next op of the segment reads it (HW_OP_NOSTORE: temporaries of blocks the caller has freed).
// Measured on ResNet-20 (1 stream s/image | 4 streams images/s): 2 lanes 1.466 | 1.324, 4 lanes 1.460 | 1.316, with the
// per-prime constants staged in LDS (ACEHIP_HW_STAGE=1) 1.42 | 1.29-1.31: the defaults are the best throughput.
#ifndef ACEHIP_HW_LANES
#define ACEHIP_HW_LANES 2
#endif
constexpr u32 kHwLanes = ACEHIP_HW_LANES;   // coefficients per lane (one or two 16-byte accesses)
#ifndef ACEHIP_HW_STAGE
#define ACEHIP_HW_STAGE 0
#endif
constexpr u32 kMaxPrimes = ACEHIP_HW_STAGE ? 96 : 1;   // per-prime constants staged in LDS (0: read them from memory)

struct V4 {
  ulong2 lo, hi;
};
// cache-policy experiment (tools/kernel_ab.sh): ACEHIP_HW_NT & 1: operand loads non-temporal, & 2: result stores non-temporal.
// Measured (profiles/r04ag_kernel_ab_hw_nt.txt, kernel seconds per 24 images): 2.506 default, 2.589 / 2.520 / 2.597 with 1 / 2 / 3: off.
#ifndef ACEHIP_HW_NT
#define ACEHIP_HW_NT 0
#endif
typedef u64 hw_u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulong2 ld2(const u64* p) {
#if ACEHIP_HW_NT & 1
  const hw_u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const hw_u64x2_t*>(p));
  return ulong2{v.x, v.y};
#else
  return *reinterpret_cast<const ulong2*>(p);
#endif
}
__device__ __forceinline__ void st2(u64* p, const ulong2& v) {
#if ACEHIP_HW_NT & 2
  __builtin_nontemporal_store(hw_u64x2_t{v.x, v.y}, reinterpret_cast<hw_u64x2_t*>(p));
#else
  *reinterpret_cast<ulong2*>(p) = v;
#endif
}
__device__ __forceinline__ V4 ld4(const u64* p) {
  if (kHwLanes == 2) return V4{ld2(p), ulong2{0, 0}};
  return V4{ld2(p), ld2(p + 2)};
}
__device__ __forceinline__ void st4(u64* p, const V4& v) {
  st2(p, v.lo);
  if (kHwLanes == 4) st2(p + 2, v.hi);
}
template <typename F>
__device__ __forceinline__ V4 map2(const V4& a, const V4& b, F f) {
  V4 r;
  r.lo.x = f(a.lo.x, b.lo.x);
  r.lo.y = f(a.lo.y, b.lo.y);
  if (kHwLanes == 4) {
    r.hi.x = f(a.hi.x, b.hi.x);
    r.hi.y = f(a.hi.y, b.hi.y);
  } else {
    r.hi = ulong2{0, 0};
  }
  return r;
}

// Thread layout (round 5): a workgroup is R images x C coefficient chunks of 64 lanes (R * C waves, wave w = image w % R, chunk w / R).
// Operands that all images share -- weight plaintexts above all -- used to be shared through L2 only (one workgroup per image), which
// holds for one stream and not next to the kernels of other image streams (those loads redirected to 4 KiB: headline +3.2 %,
// profiles/r05aq_*); with the images of a batch as the waves of ONE workgroup their requests for a shared line meet in the CU's own
// cache.  One image per launch: R = 1, C = 4, the old layout.
template <int CAP>
__global__ __launch_bounds__(1024) void hw_batch_ew_kernel(DevCtx c, HwBatchArgsT<CAP> args, u32 R, u32 C, u32 lockstep) {
  __shared__ u64 s_q[kMaxPrimes], s_mu[kMaxPrimes];
  __shared__ u32 s_nb[kMaxPrimes];
  if (ACEHIP_HW_STAGE) {
    const u32 n_primes = c.L + c.K;
    if (threadIdx.x < n_primes && threadIdx.x < kMaxPrimes) {
      const DevPrime& P = c.primes[threadIdx.x];
      s_q[threadIdx.x] = P.q;
      s_mu[threadIdx.x] = P.barrett_mu;
      s_nb[threadIdx.x] = P.nbits;
    }
    __syncthreads();
  }
  // (replica = blockIdx.z, the slowest index.  Dealing the replicas of a tile side by side -- kernels.hpp rep_block(), which pays
  // for the key inner product and the BSGS kernel -- measured 3 % slower here: the operands the replicas share, weight plaintexts, are a
  // small part of this kernel's traffic)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const u32 wr = wave % R, wc = wave / R;
  const u32 i = ((blockIdx.x * C + wc) * 64 + lane) * kHwLanes;
  const u32 rl = blockIdx.z * R + wr;  // image of this wave within the launch
  const bool active = i < c.N && rl < c.nrep;
  if (!active && !lockstep) return;
  const u32 beg = args.seg_start[blockIdx.y], end = args.seg_start[blockIdx.y + 1];
  const u32 rep = c.rep0 + rl;  // replica of this wave: operands inside the replicated arena move with it
  // The two most recent results of the segment stay in registers, keyed by their limb (the list's own addresses): slot 0 the
  // last result, slot 1 the last result of ANOTHER limb before it.  A chain that alternates between a temporary and an
  // accumulator (t = x * c; acc = acc + t; ... -- polynomial evaluation, convolution taps) then reads neither from memory.
  // A result replaces its own limb's entry; otherwise slot 0 moves to slot 1.  (The host analysis replays exactly this rule to
  // decide which results have to reach memory: api_hw_batch.cpp RegCache.)
  const u64 *r0 = nullptr, *r1 = nullptr;
  V4 v0{{0, 0}, {0, 0}}, v1{{0, 0}, {0, 0}};
  // Round 5: one more entry, for an OPERAND -- the second operand most recently loaded from memory.  Generated code multiplies both
  // polynomials of a ciphertext by the same plaintext limb, and both accumulators of a key inner product by the same raised digit
  // limb (mul c0, mul c1 per limb); the host puts such sibling chains into one segment (api_hw_batch.cpp "siblings"), where the shared
  // operand is then loaded once.  An op that writes the cached limb drops the entry.  (All address compares are on kernel arguments:
  // wave-uniform.)
  const u64* rb = nullptr;
  V4 vbc{{0, 0}, {0, 0}};
#ifdef HW_EXP  // timing experiment (results are wrong): operands all images share (outside the arena) are read from the first 4 KiB of their limb
  auto at = [&](const u64* real_addr) { return ((u64)real_addr - c.rep_lo < c.rep_span) ? real_addr + i : real_addr + (i & 510u); };
#else
  auto at = [&](const u64* real_addr) { return real_addr + i; };
#endif
  auto fetch = [&](const u64* list_addr, const u64* real_addr) {
    return list_addr == r0 ? v0 : (list_addr == r1 ? v1 : (list_addr == rb ? vbc : ld4(at(real_addr))));
  };
  auto fetch_b = [&](const u64* list_addr, const u64* real_addr) {
    if (list_addr == r0) return v0;
    if (list_addr == r1) return v1;
    if (list_addr != rb) {
      vbc = ld4(at(real_addr));
      rb = list_addr;
    }
    return vbc;
  };
  for (u32 k = beg; k < end; ++k) {
    if (lockstep) {  // experiment: the images of the workgroup start every op together, so that their loads of a shared operand meet in the CU's cache
      __syncthreads();
      if (!active) continue;
    }
    HwBatchOp op = args.op[k];
    const u32 kind = op.kind & HW_OP_KIND_MASK;
    const u64 *const res0 = op.res, *const a0 = op.a, *const b0 = op.b;  // (registers are matched by the list's own addresses)
    op.res = reb(c, op.res, rep);
    op.a = reb(c, op.a, rep);
    if (kind == HW_OP_ADD || kind == HW_OP_SUB || kind == HW_OP_MUL || kind == HW_OP_MULADD) op.b = reb(c, op.b, rep);  // (else an immediate)
    const bool keep_in_regs = (op.kind & HW_OP_NOSTORE) || (k + 1 < end && args.op[k + 1].res == res0);
    V4 vr;
    if (kind == HW_OP_ZERO) {
      vr = V4{{0, 0}, {0, 0}};
    } else {
      const V4 va = fetch(a0, op.a);
      if (kind == HW_OP_COPY) {
        vr = va;
      } else {
        const bool staged = ACEHIP_HW_STAGE && op.gi < kMaxPrimes;  // larger prime sets read the rest from memory
        const u64 q = staged ? s_q[op.gi] : c.primes[op.gi].q, mu = staged ? s_mu[op.gi] : c.primes[op.gi].barrett_mu;
        const u32 nb = staged ? s_nb[op.gi] : c.primes[op.gi].nbits;
        V4 vb;
        if (kind == HW_OP_MULC || kind == HW_OP_ADDC) {  // the second operand is an immediate
          const u64 imm = (u64)(uintptr_t)op.b;
          vb = V4{{imm, imm}, {imm, imm}};
        } else {
          vb = fetch_b(b0, op.b);
        }
        switch (kind) {
          case HW_OP_ADD:
          case HW_OP_ADDC:
            vr = map2(va, vb, [q](u64 x, u64 y) { return add_mod(x, y, q); });
            break;
          case HW_OP_SUB:
            vr = map2(va, vb, [q](u64 x, u64 y) { return sub_mod(x, y, q); });
            break;
          case HW_OP_MULADD: {
            const V4 acc = fetch(res0, op.res);
            const V4 pr = map2(va, vb, [q, mu, nb](u64 x, u64 y) { return mul_mod(x, y, q, mu, nb); });
            vr = map2(acc, pr, [q](u64 x, u64 y) { return add_mod(x, y, q); });
            break;
          }
          default:  // HW_OP_MUL, HW_OP_MULC
            vr = map2(va, vb, [q, mu, nb](u64 x, u64 y) { return mul_mod(x, y, q, mu, nb); });
            break;
        }
      }
    }
    if (!keep_in_regs) st4(op.res + i, vr);
    if (res0 == rb) rb = nullptr;  // the cached operand has a new value
    if (res0 != r0) {  // another limb than the last result's: that one becomes the older entry
      r1 = r0;
      v1 = v0;
    }
    r0 = res0;
    v0 = vr;
  }
}

// Round 6: IM images of a batch in ONE LANE (ACEHIP_HW_IMAGES_PER_LANE = 2 .. 4).  The images of a batch run the same op list on their own
// replicas of the arena; an operand OUTSIDE the arena -- a weight plaintext above all -- is the same limb for all of them.  One workgroup per
// image leaves that sharing to L2 (which holds for one stream and not next to the kernels of other image streams: those loads redirected to
// 4 KiB are worth 3.2 % of the headline, profiles/r05aq_*); the images as waves of a workgroup paid more in barriers than they saved
// (profiles/r05at_*, r05aw_*).  Here a lane keeps its coefficients of IM images: a shared operand is loaded ONCE and multiplied into all of
// them, the op list is decoded once per IM images, nothing is synchronised.  Same interpreter, same register-cache rule per image
// (the host's RegCache replay does not change): same bits.
template <int CAP, int IM>
__global__ __launch_bounds__(256) void hw_batch_ew_im_kernel(DevCtx c, HwBatchArgsT<CAP> args) {
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * kHwLanes;
  if (i >= c.N) return;
  const u32 rl0 = blockIdx.z * IM;  // first image of this lane within the launch
  const u32 beg = args.seg_start[blockIdx.y], end = args.seg_start[blockIdx.y + 1];
  const u64 *r0 = nullptr, *r1 = nullptr, *rb = nullptr;
  V4 v0[IM], v1[IM], vbc[IM];
#pragma unroll
  for (int m = 0; m < IM; ++m) v0[m] = v1[m] = vbc[m] = V4{{0, 0}, {0, 0}};
  bool on[IM];
#pragma unroll
  for (int m = 0; m < IM; ++m) on[m] = rl0 + (u32)m < c.nrep;  // (the last group of a batch may be short: wave-uniform)
  auto in_arena = [&](const u64* p) { return (u64)p - c.rep_lo < c.rep_span; };
  // operand `real` (list address `key`) for every image: registers, or memory -- once if all images share it
  auto load_all = [&](const u64* real, V4 (&out)[IM]) {
    if (!in_arena(real)) {
      const V4 t = ld4(real + i);
#pragma unroll
      for (int m = 0; m < IM; ++m) out[m] = t;
    } else {
#pragma unroll
      for (int m = 0; m < IM; ++m)
        if (on[m]) out[m] = ld4((const u64*)((u64)real + (u64)(c.rep0 + rl0 + (u32)m) * c.rep_stride) + i);
    }
  };
  for (u32 k = beg; k < end; ++k) {
    const HwBatchOp op = args.op[k];
    const u32 kind = op.kind & HW_OP_KIND_MASK;
    const u64 *const res0 = op.res, *const a0 = op.a, *const b0 = op.b;
    const bool keep_in_regs = (op.kind & HW_OP_NOSTORE) || (k + 1 < end && args.op[k + 1].res == res0);
    V4 vr[IM];
    if (kind == HW_OP_ZERO) {
#pragma unroll
      for (int m = 0; m < IM; ++m) vr[m] = V4{{0, 0}, {0, 0}};
    } else {
      V4 va[IM];
      if (a0 == r0) {
#pragma unroll
        for (int m = 0; m < IM; ++m) va[m] = v0[m];
      } else if (a0 == r1) {
#pragma unroll
        for (int m = 0; m < IM; ++m) va[m] = v1[m];
      } else if (a0 == rb) {
#pragma unroll
        for (int m = 0; m < IM; ++m) va[m] = vbc[m];
      } else {
        load_all(op.a, va);
      }
      if (kind == HW_OP_COPY) {
#pragma unroll
        for (int m = 0; m < IM; ++m) vr[m] = va[m];
      } else {
        const u64 q = c.primes[op.gi].q, mu = c.primes[op.gi].barrett_mu;
        const u32 nb = c.primes[op.gi].nbits;
        V4 vb[IM];
        if (kind == HW_OP_MULC || kind == HW_OP_ADDC) {
          const u64 imm = (u64)(uintptr_t)op.b;
#pragma unroll
          for (int m = 0; m < IM; ++m) vb[m] = V4{{imm, imm}, {imm, imm}};
        } else if (b0 == r0) {
#pragma unroll
          for (int m = 0; m < IM; ++m) vb[m] = v0[m];
        } else if (b0 == r1) {
#pragma unroll
          for (int m = 0; m < IM; ++m) vb[m] = v1[m];
        } else {
          if (b0 != rb) {
            load_all(op.b, vbc);
            rb = b0;
          }
#pragma unroll
          for (int m = 0; m < IM; ++m) vb[m] = vbc[m];
        }
        V4 acc[IM];
        if (kind == HW_OP_MULADD) {
          if (res0 == r0) {
#pragma unroll
            for (int m = 0; m < IM; ++m) acc[m] = v0[m];
          } else if (res0 == r1) {
#pragma unroll
            for (int m = 0; m < IM; ++m) acc[m] = v1[m];
          } else if (res0 == rb) {
#pragma unroll
            for (int m = 0; m < IM; ++m) acc[m] = vbc[m];
          } else {
            load_all(op.res, acc);
          }
        }
#pragma unroll
        for (int m = 0; m < IM; ++m) {
          switch (kind) {
            case HW_OP_ADD:
            case HW_OP_ADDC:
              vr[m] = map2(va[m], vb[m], [q](u64 x, u64 y) { return add_mod(x, y, q); });
              break;
            case HW_OP_SUB:
              vr[m] = map2(va[m], vb[m], [q](u64 x, u64 y) { return sub_mod(x, y, q); });
              break;
            case HW_OP_MULADD: {
              const V4 pr = map2(va[m], vb[m], [q, mu, nb](u64 x, u64 y) { return mul_mod(x, y, q, mu, nb); });
              vr[m] = map2(acc[m], pr, [q](u64 x, u64 y) { return add_mod(x, y, q); });
              break;
            }
            default:  // HW_OP_MUL, HW_OP_MULC
              vr[m] = map2(va[m], vb[m], [q, mu, nb](u64 x, u64 y) { return mul_mod(x, y, q, mu, nb); });
              break;
          }
        }
      }
    }
    if (!keep_in_regs) {
      if (!in_arena(op.res)) {  // (a result outside the arena is one limb for all images: they computed the same value)
        st4(op.res + i, vr[0]);
      } else {
#pragma unroll
        for (int m = 0; m < IM; ++m)
          if (on[m]) st4((u64*)((u64)op.res + (u64)(c.rep0 + rl0 + (u32)m) * c.rep_stride) + i, vr[m]);
      }
    }
    if (res0 == rb) rb = nullptr;
    if (res0 != r0) {
      r1 = r0;
#pragma unroll
      for (int m = 0; m < IM; ++m) v1[m] = v0[m];
    }
    r0 = res0;
#pragma unroll
    for (int m = 0; m < IM; ++m) v0[m] = vr[m];
  }
}

// independent gathers r[j] = a[perm[j]] (the host guarantees no result aliases any source of the launch)
// op.gi != 0: the table op.b is the automorphism X -> X^k of this context (k = op.gi), whose index map in the NTT (bit-reversed)
// order is perm[i] = rev(((2 rev(i) + 1) k mod 2N) >> 1) (host_params.cpp automorphism_order_ntt): computed here, 4 bytes per
// coefficient less to load.  op.gi == 0: a caller-supplied permutation, loaded.
template <int CAP>
__global__ __launch_bounds__(256) void hw_batch_rotate_kernel(DevCtx c, HwBatchArgsT<CAP> args) {
  const u32 N = c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= N) return;
  HwBatchOp op = args.op[blockIdx.y];
  op.res = reb(c, op.res, c.rep0 + blockIdx.z);
  op.a = reb(c, op.a, c.rep0 + blockIdx.z);
  uint2 p;
  if (op.gi != 0) {  // uniform for the workgroup
    const u32 sh = __builtin_clz(N) + 1;  // 32 - log2(N)
    const u32 j0 = __brev(i) >> sh, j1 = __brev(i + 1) >> sh;
    p.x = __brev((((2 * j0 + 1) * op.gi) & (2 * N - 1)) >> 1) >> sh;
    p.y = __brev((((2 * j1 + 1) * op.gi) & (2 * N - 1)) >> 1) >> sh;
  } else {
    p = *reinterpret_cast<const uint2*>(reinterpret_cast<const u32*>(op.b) + i);
  }
  ulong2 v;
  v.x = op.a[p.x];
  v.y = op.a[p.y];
  *reinterpret_cast<ulong2*>(op.res + i) = v;
}

// the same gathers for IM images of a batch in one lane: the index pair is computed (or loaded) once, the 2 * IM gathered loads are in
// flight together (ACEHIP_HW_IMAGES_PER_LANE, see hw_batch_ew_im_kernel)
template <int CAP, int IM>
__global__ __launch_bounds__(256) void hw_batch_rotate_im_kernel(DevCtx c, HwBatchArgsT<CAP> args) {
  const u32 N = c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= N) return;
  const HwBatchOp op = args.op[blockIdx.y];
  uint2 p;
  if (op.gi != 0) {  // uniform for the workgroup
    const u32 sh = __builtin_clz(N) + 1;  // 32 - log2(N)
    const u32 j0 = __brev(i) >> sh, j1 = __brev(i + 1) >> sh;
    p.x = __brev((((2 * j0 + 1) * op.gi) & (2 * N - 1)) >> 1) >> sh;
    p.y = __brev((((2 * j1 + 1) * op.gi) & (2 * N - 1)) >> 1) >> sh;
  } else {
    p = *reinterpret_cast<const uint2*>(reinterpret_cast<const u32*>(op.b) + i);
  }
  const u32 rl0 = blockIdx.z * IM;
  ulong2 v[IM];
#pragma unroll
  for (int m = 0; m < IM; ++m)
    if (rl0 + (u32)m < c.nrep) {
      const u64* a = reb(c, op.a, c.rep0 + rl0 + (u32)m);
      v[m].x = a[p.x];
      v[m].y = a[p.y];
    }
#pragma unroll
  for (int m = 0; m < IM; ++m)
    if (rl0 + (u32)m < c.nrep) *reinterpret_cast<ulong2*>(reb(c, op.res, c.rep0 + rl0 + (u32)m) + i) = v[m];
}

template <int CAP>
static HwBatchArgsT<CAP> shrink(const HwBatchArgs& a, u32 n_ops, u32 n_seg) {
  HwBatchArgsT<CAP> r;
  for (u32 i = 0; i < n_ops; ++i) r.op[i] = a.op[i];
  for (u32 i = 0; i <= n_seg; ++i) r.seg_start[i] = a.seg_start[i];
  return r;
}

void launch_hw_batch_ew(const DevCtx& c, const HwBatchArgs& args, u32 n_seg, hipStream_t s) {
  ACEHIP_ABLATE(ABL_EW);
  if (n_seg == 0) return;
  // images per workgroup: all of a batch up to 12 (ACEHIP_HW_REPS_WG=1: one, the layout that shares through L2); chunks so that a
  // workgroup has at least four waves
  static const u32 reps_cap = [] { const char* e = getenv("ACEHIP_HW_REPS_WG"); return e && atoi(e) > 0 ? (u32)atoi(e) : 1u; }();
  static const u32 lockstep = [] { const char* e = getenv("ACEHIP_HW_REPS_SYNC"); return e && atoi(e) > 0 ? 1u : 0u; }();
  // ACEHIP_HW_IMAGES_PER_LANE = 2 .. 4 (default 3; 0 / 1: off): the images-per-lane form (hw_batch_ew_im_kernel) for launches that cover
  // several images.  Same-box A/B of the headline (profiles/r06o_ab_bench_hw_images_per_lane.txt): 2.823 images/s with one image per
  // workgroup, 2.931 / 2.942 / 2.920 with 2 / 3 / 4 images per lane; bit-identical (the batch digests of tests/test_gpu_gen_parity.py)
  static const u32 im = [] { const char* e = getenv("ACEHIP_HW_IMAGES_PER_LANE"); const u32 v = e ? (u32)atoi(e) : 3u; return v >= 2 && v <= 4 ? v : 0u; }();
  if (im && c.nrep >= 2 && reps_cap == 1) {
    const u32 n_ops = args.seg_start[n_seg];
    const u32 per_wg = 256 * kHwLanes;
#define ACEHIP_HW_IM_LAUNCH(IMV)                                                                                                          \
  do {                                                                                                                                    \
    dim3 grid((c.N + per_wg - 1) / per_wg, n_seg, (c.nrep + IMV - 1) / IMV), block(256);                                                  \
    if (n_ops <= 16) hipLaunchKernelGGL((hw_batch_ew_im_kernel<16, IMV>), grid, block, 0, s, c, shrink<16>(args, n_ops, n_seg));          \
    else if (n_ops <= 48) hipLaunchKernelGGL((hw_batch_ew_im_kernel<48, IMV>), grid, block, 0, s, c, shrink<48>(args, n_ops, n_seg));     \
    else hipLaunchKernelGGL((hw_batch_ew_im_kernel<HW_BATCH_MAX, IMV>), grid, block, 0, s, c, args);                                      \
  } while (0)
    if (im == 2) ACEHIP_HW_IM_LAUNCH(2);
    else if (im == 3) ACEHIP_HW_IM_LAUNCH(3);
    else ACEHIP_HW_IM_LAUNCH(4);
#undef ACEHIP_HW_IM_LAUNCH
    return;
  }
  const u32 R = std::min(std::min(c.nrep, reps_cap), 16u), C = R >= 4 ? 1u : (R == 3 ? 1u : (R == 2 ? 2u : 4u));
  const u32 per_wg = 64 * kHwLanes * C;  // coefficients of a limb per workgroup
  dim3 grid((c.N + per_wg - 1) / per_wg, n_seg, (c.nrep + R - 1) / R), block(64 * R * C);
  const u32 n_ops = args.seg_start[n_seg];
  if (n_ops <= 16) hipLaunchKernelGGL(hw_batch_ew_kernel<16>, grid, block, 0, s, c, shrink<16>(args, n_ops, n_seg), R, C, lockstep);
  else if (n_ops <= 48) hipLaunchKernelGGL(hw_batch_ew_kernel<48>, grid, block, 0, s, c, shrink<48>(args, n_ops, n_seg), R, C, lockstep);
  else hipLaunchKernelGGL(hw_batch_ew_kernel<HW_BATCH_MAX>, grid, block, 0, s, c, args, R, C, lockstep);
}

void launch_hw_batch_rotate(const DevCtx& c, const HwBatchArgs& args, u32 n_ops, hipStream_t s) {
  ACEHIP_ABLATE(ABL_ROTATE);
  if (n_ops == 0) return;
  static const u32 im = [] { const char* e = getenv("ACEHIP_HW_ROTATE_IMAGES_PER_LANE"); const u32 v = e ? (u32)atoi(e) : 0u; return v == 2 || v == 3 ? v : 0u; }();
  if (im && c.nrep >= 2) {
    dim3 gim((c.N / 2 + 255) / 256, n_ops, (c.nrep + im - 1) / im), block(256);
    if (im == 2) {
      if (n_ops <= 16) hipLaunchKernelGGL((hw_batch_rotate_im_kernel<16, 2>), gim, block, 0, s, c, shrink<16>(args, n_ops, 0));
      else if (n_ops <= 48) hipLaunchKernelGGL((hw_batch_rotate_im_kernel<48, 2>), gim, block, 0, s, c, shrink<48>(args, n_ops, 0));
      else hipLaunchKernelGGL((hw_batch_rotate_im_kernel<HW_BATCH_MAX, 2>), gim, block, 0, s, c, args);
    } else {
      if (n_ops <= 16) hipLaunchKernelGGL((hw_batch_rotate_im_kernel<16, 3>), gim, block, 0, s, c, shrink<16>(args, n_ops, 0));
      else if (n_ops <= 48) hipLaunchKernelGGL((hw_batch_rotate_im_kernel<48, 3>), gim, block, 0, s, c, shrink<48>(args, n_ops, 0));
      else hipLaunchKernelGGL((hw_batch_rotate_im_kernel<HW_BATCH_MAX, 3>), gim, block, 0, s, c, args);
    }
    return;
  }
  dim3 grid((c.N / 2 + 255) / 256, n_ops, c.nrep), block(256);
  if (n_ops <= 16) hipLaunchKernelGGL(hw_batch_rotate_kernel<16>, grid, block, 0, s, c, shrink<16>(args, n_ops, 0));
  else if (n_ops <= 48) hipLaunchKernelGGL(hw_batch_rotate_kernel<48>, grid, block, 0, s, c, shrink<48>(args, n_ops, 0));
  else hipLaunchKernelGGL(hw_batch_rotate_kernel<HW_BATCH_MAX>, grid, block, 0, s, c, args);
}

}  // namespace acehip
