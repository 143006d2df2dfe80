// shard.hip -- kernels of limb-sharded execution: every GPU of a node holds the limbs gi with
// gi % world == rank of every polynomial, PACKED (its y-th owned limb at word offset y*N), and only the base
// conversions exchange data.  These kernels are the packed-list counterparts of the position-indexed kernels of
// kernels.hip / keyswitch.hip: limb y belongs to prime gi[y].
#include "kernels.hpp"

namespace acehip {

__global__ __launch_bounds__(256) void packed_rescale_spread_kernel(DevCtx c, u64* __restrict__ t, size_t t_stride,
                                                                    const u64* __restrict__ last, size_t last_stride,
                                                                    const u32* __restrict__ gi, u32 gi_last,
                                                                    const u64* __restrict__ c1, const u64* __restrict__ c1p) {
  const u32 y = blockIdx.y;
  const u64 q = c.primes[gi[y]].q, ql = c.primes[gi_last].q, w = c1[y], wp = c1p[y];
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 v = *reinterpret_cast<const ulong2*>(last + blockIdx.z * last_stride + i);
  v.x = mul_shoup(switch_modulus(v.x, ql, q), w, wp, q);
  v.y = mul_shoup(switch_modulus(v.y, ql, q), w, wp, q);
  *reinterpret_cast<ulong2*>(t + blockIdx.z * t_stride + (size_t)y * c.N + i) = v;
}
void launch_packed_rescale_spread(const DevCtx& c, u64* t, size_t t_stride, const u64* last, size_t last_stride, const u32* gi,
                                  u32 gi_last, const u64* c1, const u64* c1p, u32 n_limbs, u32 n_polys, hipStream_t s) {
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, n_polys), block(256);
  hipLaunchKernelGGL(packed_rescale_spread_kernel, grid, block, 0, s, c, t, t_stride, last, last_stride, gi, gi_last, c1, c1p);
}

// MODE 0: out = shoup(x, w) + t (Rescale tail polynomial.c:1145-1158); MODE 1: out = shoup(x - t, w) (ModDown tail :956-965)
template <int MODE>
__global__ __launch_bounds__(256) void packed_tail_kernel(DevCtx c, u64* __restrict__ out0, u64* __restrict__ out1,
                                                          const u64* __restrict__ x0, const u64* __restrict__ x1,
                                                          const u64* __restrict__ t, size_t t_stride, const u32* __restrict__ gi,
                                                          const u64* __restrict__ wt, const u64* __restrict__ wtp) {
  const u32 y = blockIdx.y;
  const u64 q = c.primes[gi[y]].q, w = wt[y], wp = wtp[y];
  const size_t base = (size_t)y * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u64* x = blockIdx.z ? x1 : x0;
  u64* out = blockIdx.z ? out1 : out0;
  const ulong2 vx = *reinterpret_cast<const ulong2*>(x + base + i);
  const ulong2 vt = *reinterpret_cast<const ulong2*>(t + blockIdx.z * t_stride + base + i);
  ulong2 vo;
  if (MODE == 0) {
    vo.x = add_mod(mul_shoup(vx.x, w, wp, q), vt.x, q);
    vo.y = add_mod(mul_shoup(vx.y, w, wp, q), vt.y, q);
  } else {
    vo.x = mul_shoup(sub_mod(vx.x, vt.x, q), w, wp, q);
    vo.y = mul_shoup(sub_mod(vx.y, vt.y, q), w, wp, q);
  }
  *reinterpret_cast<ulong2*>(out + base + i) = vo;
}
void launch_packed_rescale_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                                const u32* gi, const u64* inv, const u64* invp, u32 n_limbs, u32 n_polys, hipStream_t s) {
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, n_polys), block(256);
  hipLaunchKernelGGL((packed_tail_kernel<0>), grid, block, 0, s, c, out0, out1, x0, x1, t, t_stride, gi, inv, invp);
}
void launch_packed_moddown_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                                const u32* gi, const u64* w, const u64* wp, u32 n_limbs, u32 n_polys, hipStream_t s) {
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, n_polys), block(256);
  hipLaunchKernelGGL((packed_tail_kernel<1>), grid, block, 0, s, c, out0, out1, x0, x1, t, t_stride, gi, w, wp);
}

// key inner product over the owned limbs, fused over digits (Multiply_add polynomial.c:148-183 for every part)
__global__ __launch_bounds__(256) void packed_key_mac_kernel(DevCtx c, u64* __restrict__ acc0, u64* __restrict__ acc1,
                                                             const u64* __restrict__ key, size_t key_stride,
                                                             PackedPtrs src, const u32* __restrict__ gi, u32 nd, u32 n_limbs) {
  const u32 y = blockIdx.y;
  const DevPrime P = c.primes[gi[y]];
  const size_t base = (size_t)y * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 r0{0, 0}, r1{0, 0};
  for (u32 d = 0; d < nd; ++d) {
    const ulong2 e = *reinterpret_cast<const ulong2*>(src.p[d * n_limbs + y] + i);
    const ulong2 k0 = *reinterpret_cast<const ulong2*>(key + (size_t)(2 * d) * key_stride + base + i);
    const ulong2 k1 = *reinterpret_cast<const ulong2*>(key + (size_t)(2 * d + 1) * key_stride + base + i);
    r0.x = add_mod(r0.x, mul_mod(k0.x, e.x, P), P.q);
    r0.y = add_mod(r0.y, mul_mod(k0.y, e.y, P), P.q);
    r1.x = add_mod(r1.x, mul_mod(k1.x, e.x, P), P.q);
    r1.y = add_mod(r1.y, mul_mod(k1.y, e.y, P), P.q);
  }
  *reinterpret_cast<ulong2*>(acc0 + base + i) = r0;
  *reinterpret_cast<ulong2*>(acc1 + base + i) = r1;
}
void launch_packed_key_mac(const DevCtx& c, u64* acc0, u64* acc1, const u64* key, size_t key_stride, const PackedPtrs& src,
                           const u32* gi, u32 nd, u32 n_limbs, hipStream_t s) {
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs), block(256);
  hipLaunchKernelGGL(packed_key_mac_kernel, grid, block, 0, s, c, acc0, acc1, key, key_stride, src, gi, nd, n_limbs);
}

__global__ __launch_bounds__(256) void packed_gather_kernel(DevCtx c, u64* __restrict__ dst, PackedPtrs src_tab) {
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  *reinterpret_cast<ulong2*>(dst + (size_t)blockIdx.y * c.N + i) = *reinterpret_cast<const ulong2*>(src_tab.p[blockIdx.y] + i);
}
void launch_packed_gather(const DevCtx& c, u64* dst, const PackedPtrs& src_tab, u32 n_limbs, hipStream_t s) {
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs), block(256);
  hipLaunchKernelGGL(packed_gather_kernel, grid, block, 0, s, c, dst, src_tab);
}

}  // namespace acehip
