// hw_batch.hip -- many per-limb Hw_modadd / Hw_modmul / Hw_rotate calls (poly_arith.c:14-56) in a handful of
// launches.  ACE-generated code spells every ciphertext operation as a host loop over the RNS limbs, one
// Hw_* call per limb and component; launched one by one that is ~600k kernels of 1.5 MB each per ResNet-20
// image and the GPU idles on launch latency.  The host side (api.cpp acehip_hw_batch) groups a list of such
// ops into dependency chains (ops that touch a common written limb, kept in program order) and hands up to
// HW_BATCH_MAX ops to one launch: blockIdx.y walks one chain segment in order, so a coefficient's whole
// history stays in one lane and the sequential semantics of the original call sequence are preserved.
#include "device_arith.hpp"
#include "kernels.hpp"

namespace acehip {

// Every lane owns coefficients (i, i+1) of all limbs of its segment: read-after-write between ops of a chain goes
// through the lane itself.  The previous result stays in registers: an operand that is the previous op's result
// limb is not reloaded, and a result is not stored when the next op of the segment writes the same limb again
// (accumulation runs res += a_j * b_j keep the accumulator in registers; the last op of a run always stores, so
// every later reader -- in this segment, another launch or the host -- finds the final value in memory).
// CAP: capacity of the argument table (a launch with few ops ships a small kernel-argument block)
template <int CAP>
__global__ __launch_bounds__(256) void hw_batch_ew_kernel(DevCtx c, HwBatchArgsT<CAP> args) {
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u32 beg = args.seg_start[blockIdx.y], end = args.seg_start[blockIdx.y + 1];
  const u64* prev_res = nullptr;
  ulong2 vprev{0, 0};
  for (u32 k = beg; k < end; ++k) {
    const HwBatchOp op = args.op[k];
    const bool keep_in_regs = k + 1 < end && args.op[k + 1].res == op.res;
    ulong2 vr;
    if (op.kind == HW_OP_ZERO) {
      vr.x = 0;
      vr.y = 0;
    } else {
      const ulong2 va = op.a == prev_res ? vprev : *reinterpret_cast<const ulong2*>(op.a + i);
      if (op.kind == HW_OP_COPY) {
        vr = va;
      } else {
        const DevPrime P = c.primes[op.gi];
        ulong2 vb;
        if (op.kind == HW_OP_MULC || op.kind == HW_OP_ADDC) {  // the second operand is an immediate
          vb.x = vb.y = (u64)(uintptr_t)op.b;
        } else {
          vb = op.b == prev_res ? vprev : *reinterpret_cast<const ulong2*>(op.b + i);
        }
        switch (op.kind) {
          case HW_OP_ADD:
          case HW_OP_ADDC:
            vr.x = add_mod(va.x, vb.x, P.q);
            vr.y = add_mod(va.y, vb.y, P.q);
            break;
          case HW_OP_SUB:
            vr.x = sub_mod(va.x, vb.x, P.q);
            vr.y = sub_mod(va.y, vb.y, P.q);
            break;
          case HW_OP_MULADD: {
            const ulong2 acc = op.res == prev_res ? vprev : *reinterpret_cast<const ulong2*>(op.res + i);
            vr.x = add_mod(acc.x, mul_mod(va.x, vb.x, P), P.q);
            vr.y = add_mod(acc.y, mul_mod(va.y, vb.y, P), P.q);
            break;
          }
          default:  // HW_OP_MUL, HW_OP_MULC
            vr.x = mul_mod(va.x, vb.x, P);
            vr.y = mul_mod(va.y, vb.y, P);
            break;
        }
      }
    }
    if (!keep_in_regs) *reinterpret_cast<ulong2*>(op.res + i) = vr;
    prev_res = op.res;
    vprev = vr;
  }
}

// independent gathers r[j] = a[perm[j]] (the host guarantees no result aliases any source of the launch)
template <int CAP>
__global__ __launch_bounds__(256) void hw_batch_rotate_kernel(u32 N, HwBatchArgsT<CAP> args) {
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= N) return;
  const HwBatchOp op = args.op[blockIdx.y];
  const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const u32*>(op.b) + i);
  ulong2 v;
  v.x = op.a[p.x];
  v.y = op.a[p.y];
  *reinterpret_cast<ulong2*>(op.res + i) = v;
}

template <int CAP>
static HwBatchArgsT<CAP> shrink(const HwBatchArgs& a, u32 n_ops, u32 n_seg) {
  HwBatchArgsT<CAP> r;
  for (u32 i = 0; i < n_ops; ++i) r.op[i] = a.op[i];
  for (u32 i = 0; i <= n_seg; ++i) r.seg_start[i] = a.seg_start[i];
  return r;
}

void launch_hw_batch_ew(const DevCtx& c, const HwBatchArgs& args, u32 n_seg, hipStream_t s) {
  if (n_seg == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_seg), block(256);
  const u32 n_ops = args.seg_start[n_seg];
  if (n_ops <= 16) hipLaunchKernelGGL(hw_batch_ew_kernel<16>, grid, block, 0, s, c, shrink<16>(args, n_ops, n_seg));
  else if (n_ops <= 48) hipLaunchKernelGGL(hw_batch_ew_kernel<48>, grid, block, 0, s, c, shrink<48>(args, n_ops, n_seg));
  else hipLaunchKernelGGL(hw_batch_ew_kernel<HW_BATCH_MAX>, grid, block, 0, s, c, args);
}

void launch_hw_batch_rotate(const DevCtx& c, const HwBatchArgs& args, u32 n_ops, hipStream_t s) {
  if (n_ops == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_ops), block(256);
  if (n_ops <= 16) hipLaunchKernelGGL(hw_batch_rotate_kernel<16>, grid, block, 0, s, c.N, shrink<16>(args, n_ops, 0));
  else if (n_ops <= 48) hipLaunchKernelGGL(hw_batch_rotate_kernel<48>, grid, block, 0, s, c.N, shrink<48>(args, n_ops, 0));
  else hipLaunchKernelGGL(hw_batch_rotate_kernel<HW_BATCH_MAX>, grid, block, 0, s, c.N, args);
}

}  // namespace acehip
