// embed.hip -- canonical-embedding half of CKKS encode on the device, bit-identical to the reference's
// host FP64 code: Embedding_inv (rtlib/ant/src/util/ntt.c:713-753, special inverse FFT over the 5^i orbit,
// DIF butterflies from stride n/2 down to 1, bit reversal, divide by n) followed by the scaling / rounding
// loop of Encode_impl (rtlib/ant/src/util/ckks_encoder.c:240-268: x*Delta + 0.5 -> llround).
// IEEE-754 binary64 add/sub/mul/div are correctly rounded on gfx950 exactly as on the host, so the only
// things that could change a bit are the operation order (kept) and FMA contraction (disabled below: every
// product is rounded before it is added, as in the reference build which targets plain SSE2 doubles).
// The twiddle table (cos/sin from the host libm) and the 5^i table are uploaded, not recomputed.
#pragma clang fp contract(off)
#include "kernels.hpp"

namespace acehip {

struct cd {
  double x, y;
};
__device__ __forceinline__ cd c_add(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd c_sub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd c_mul(cd a, cd b) {
  const double rr = a.x * b.x, ii = a.y * b.y, ri = a.x * b.y, ir = a.y * b.x;
  return cd{rr - ii, ri + ir};
}

// kind: 0 = float real, 1 = double real, 2 = complex double (re, im interleaved); zero beyond len
__device__ __forceinline__ cd load_value(const void* __restrict__ vals, int kind, size_t len, size_t i) {
  if (i >= len) return cd{0.0, 0.0};
  if (kind == 0) return cd{(double)reinterpret_cast<const float*>(vals)[i], 0.0};
  if (kind == 1) return cd{reinterpret_cast<const double*>(vals)[i], 0.0};
  const double* p = reinterpret_cast<const double*>(vals) + 2 * i;
  return cd{p[0], p[1]};
}

// twiddle of butterfly i at stage logm: rou[(idx_mod - rot_group[i] % idx_mod) * gap], idx_mod = 2^(logm+2), as the
// reference looks it up (ntt.c:728-736); the host lays the values out per stage, tws[2^(logm-1) - 1 + i], so that
// the lanes of a wave read consecutive entries instead of chasing two tables
__device__ __forceinline__ cd stage_twiddle(const cd* __restrict__ tws, const u32* __restrict__, u32, u32 logm, u32 i) {
  return tws[(1u << (logm - 1)) - 1 + i];
}

constexpr u32 EMB_LOW = 8;    // stages 8..1 run on 256 contiguous values per workgroup
constexpr u32 EMB_COLS = 4;   // columns per workgroup in the strided pass

// Stages logn .. EMB_LOW+1 on a tile of R = n/256 rows (stride 256) x EMB_COLS columns held in LDS.
// Reads the message (zero padded), writes the partially transformed complex vector to work[n].
// blockIdx.y = message of a batch (EmbBatch: same kind / length / slots; work and msg of message b are b*work_stride /
// b*msg_stride further)
__global__ __launch_bounds__(256) void embed_inv_high_kernel(cd* __restrict__ work, EmbBatch batch, size_t work_stride, int kind,
                                                             size_t len, u32 logn, u32 log2m, const cd* __restrict__ rou,
                                                             const u32* __restrict__ rot_group) {
  extern __shared__ cd tile[];  // [R][EMB_COLS]
  const void* __restrict__ vals = batch.vals[blockIdx.y];
  work += blockIdx.y * work_stride;
  const u32 R = 1u << (logn - EMB_LOW);
  const u32 c0 = blockIdx.x * EMB_COLS;
  const u32 n_elem = R * EMB_COLS;
  for (u32 t = threadIdx.x; t < n_elem; t += blockDim.x) {
    const u32 r = t / EMB_COLS, cc = t % EMB_COLS;
    tile[t] = load_value(vals, kind, len, ((size_t)r << EMB_LOW) + c0 + cc);
  }
  __syncthreads();
  for (u32 logm = logn; logm > EMB_LOW; --logm) {
    const u32 rs = 1u << (logm - 1 - EMB_LOW);  // row distance of the pair
    for (u32 t = threadIdx.x; t < n_elem / 2; t += blockDim.x) {
      const u32 cc = t % EMB_COLS, h = t / EMB_COLS;          // h-th butterfly row pair
      const u32 r = ((h & ~(rs - 1)) << 1) | (h & (rs - 1));  // row with bit rs clear
      const u32 i = ((r & (rs - 1)) << EMB_LOW) + c0 + cc;    // index inside the 2^logm block, < 2^(logm-1)
      const cd e = tile[r * EMB_COLS + cc], o = tile[(r + rs) * EMB_COLS + cc];
      tile[r * EMB_COLS + cc] = c_add(e, o);
      tile[(r + rs) * EMB_COLS + cc] = c_mul(c_sub(e, o), stage_twiddle(rou, rot_group, log2m, logm, i));
    }
    __syncthreads();
  }
  for (u32 t = threadIdx.x; t < n_elem; t += blockDim.x) {
    const u32 r = t / EMB_COLS, cc = t % EMB_COLS;
    work[((size_t)r << EMB_LOW) + c0 + cc] = tile[t];
  }
}

// Stages min(logn, EMB_LOW) .. 1 on one contiguous block, then bit reversal, /n, *Delta + 0.5, llround and
// the scatter into the coefficient vector: msg[i*gap] = Re, msg[(i+slots)*gap] = Im (ckks_encoder.c:246-268).
// FROM_INPUT: logn <= EMB_LOW, read the message directly (no strided pass ran).
template <bool FROM_INPUT>
__global__ __launch_bounds__(128) void embed_inv_low_kernel(int64_t* __restrict__ msg, size_t msg_stride, const cd* __restrict__ work,
                                                            size_t work_stride, EmbBatch batch, int kind, size_t len, u32 logn,
                                                            u32 log2m, const cd* __restrict__ rou,
                                                            const u32* __restrict__ rot_group, double sf, u32 coef_gap,
                                                            int* __restrict__ err_flag, double round_add) {
  __shared__ cd blk[1u << EMB_LOW];
  const void* __restrict__ vals = batch.vals[blockIdx.y];
  msg += blockIdx.y * msg_stride;
  work += blockIdx.y * work_stride;
  const u32 lb = logn < EMB_LOW ? logn : EMB_LOW;
  const u32 bsz = 1u << lb;
  const size_t base = (size_t)blockIdx.x << lb;
  for (u32 t = threadIdx.x; t < bsz; t += blockDim.x)
    blk[t] = FROM_INPUT ? load_value(vals, kind, len, base + t) : work[base + t];
  __syncthreads();
  for (u32 logm = lb; logm > 0; --logm) {
    const u32 half = 1u << (logm - 1);
    for (u32 h = threadIdx.x; h < bsz / 2; h += blockDim.x) {
      const u32 i = h & (half - 1);
      const u32 e_idx = ((h & ~(half - 1)) << 1) | i;
      const cd e = blk[e_idx], o = blk[e_idx + half];
      blk[e_idx] = c_add(e, o);
      blk[e_idx + half] = c_mul(c_sub(e, o), stage_twiddle(rou, rot_group, log2m, logm, i));
    }
    __syncthreads();
  }
  // x / n with n a power of two == x * (1/n) bit for bit (one rounding of the same real number)
  const double inv_n = 1.0 / (double)(1u << logn);
  const u32 slots = 1u << logn;
  for (u32 t = threadIdx.x; t < bsz; t += blockDim.x) {
    const u32 p = (u32)base + t;
    const u32 i = logn ? (__brev(p) >> (32 - logn)) : 0;
    const cd v = blk[t];
    // round_add: 0.5 for Encode_impl (ckks_encoder.c:247-250: x*Delta + 0.5, then llround), 0 for Encode_impl_with_scale (:357-360)
    const double re = (v.x * inv_n) * sf + round_add, im = (v.y * inv_n) * sf + round_add;
    if (!(re <= 9.2e18 && re >= -9.2e18 && im <= 9.2e18 && im >= -9.2e18)) atomicOr(err_flag, 1);
    msg[(size_t)i * coef_gap] = llround(re);
    msg[(size_t)(i + slots) * coef_gap] = llround(im);
  }
}

// msg[b][N] (device) <- rounded, scaled inverse embedding of batch.vals[b] (len values each, zero padded to `slots`), b < n_batch;
// work must hold n_batch * slots complex values
void launch_embed_inv_batch(int64_t* msg, cd* work, const EmbBatch& batch, u32 n_batch, int kind, size_t len, u32 slots, u32 N,
                            const cd* rou, const u32* rot_group, double sf, int* err_flag, hipStream_t s, double round_add) {
  ACEHIP_ABLATE(ABL_EMBED);
  u32 logn = 0, log2m = 1;
  while ((1u << logn) < slots) ++logn;
  while ((1u << log2m) < 2 * N) ++log2m;
  const u32 coef_gap = N / (2 * slots);
  if (coef_gap > 1) (void)hipMemsetAsync(msg, 0, (size_t)n_batch * N * sizeof(int64_t), s);
  if (logn > EMB_LOW) {
    const u32 R = 1u << (logn - EMB_LOW);
    hipLaunchKernelGGL(embed_inv_high_kernel, dim3((1u << EMB_LOW) / EMB_COLS, n_batch), dim3(256), (size_t)R * EMB_COLS * sizeof(cd),
                       s, work, batch, (size_t)slots, kind, len, logn, log2m, rou, rot_group);
    hipLaunchKernelGGL(embed_inv_low_kernel<false>, dim3(slots >> EMB_LOW, n_batch), dim3(128), 0, s, msg, (size_t)N, work,
                       (size_t)slots, batch, kind, len, logn, log2m, rou, rot_group, sf, coef_gap, err_flag, round_add);
  } else {
    hipLaunchKernelGGL(embed_inv_low_kernel<true>, dim3(1, n_batch), dim3(128), 0, s, msg, (size_t)N, work, (size_t)slots, batch, kind,
                       len, logn, log2m, rou, rot_group, sf, coef_gap, err_flag, round_add);
  }
}
void launch_embed_inv(int64_t* msg, cd* work, const void* vals, int kind, size_t len, u32 slots, u32 N, const cd* rou,
                      const u32* rot_group, double sf, int* err_flag, hipStream_t s, double round_add) {
  EmbBatch b{};
  b.vals[0] = vals;
  launch_embed_inv_batch(msg, work, b, 1, kind, len, slots, N, rou, rot_group, sf, err_flag, s, round_add);
}

}  // namespace acehip
