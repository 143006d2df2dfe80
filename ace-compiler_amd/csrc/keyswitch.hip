// keyswitch.hip -- batched kernels of the hybrid key-switch: base conversion over all digits in one
// launch, the key inner product fused over digits, and the two-polynomial ModDown tail.
// Reference algorithm: Decompose_modup polynomial.c:1241-1335, Multiply_add :148-183,
// Reduce_rns_base :928-967 (generated Rotate()/Relinearize(), resnet20_cifar10_pre.onnx.inc:6972-7146).
#include <algorithm>

#include "kernels.hpp"

namespace acehip {

constexpr int kGroup = 8;       // output limbs per workgroup row: a source value is loaded once for all of them
constexpr int kConvMaxIn = 64;  // source limbs of one conversion (digit size alpha, or K): checked at context creation

// out[pos_j][n] = ( sum_i y_i[n] * hat[i][j] ) mod t_j,  y_i = in[src_pos0+i][n] (* scale_i mod q_i): canonical residues in,
// canonical residues out, the sum is exact (Reduce_rns_base polynomial.c:928-967 accumulates in 128 bits and reduces once).
//
// The multiply-accumulate is the whole cost of this kernel (n_in products per output), so it is arranged for
// v_mad_u64_u32 and nothing else: both factors are below 2^(2h) (h = DevCtx::split_bits = 30 for primes below 2^60) and
// are split into halves of h bits; the four partial products of a term are below 2^(2h), so 2^(64-2h) terms (16 for
// h = 30) accumulate in four plain 64-bit sums without any carry -- four multiply-adds per term, whose 64-bit addend is
// free -- and the sums are combined into the 128-bit total once per chunk of terms.  The constants hat[i][j] of the row are
// staged in LDS, already split (every lane reads the same address: a broadcast).
__global__ __launch_bounds__(256) void base_conv_batch_kernel(DevCtx c, u64* __restrict__ out, size_t out_stride,
                                                              const u64* __restrict__ in, size_t in_stride,
                                                              const ConvDesc* __restrict__ descs, u32 desc_step, PtrTab8 outz) {
  __shared__ uint2 s_hat[kConvMaxIn * kGroup];  // this row's constants, already split: {low half, high half}
  const RepZ rz = rep_of_z(c);  // blockIdx.z = problem + n_problems * replica
  const ConvDesc d = descs[rz.z * desc_step];
  const u32 j0 = blockIdx.y * kGroup;
  if (j0 >= d.n_out) return;  // uniform for the workgroup
  const u32 h = c.split_bits, mask = (1u << h) - 1u;
  for (u32 t = threadIdx.x; t < d.n_in * kGroup; t += 256) {
    const u32 i = t / kGroup, j = min(j0 + t % kGroup, d.n_out - 1);  // rows past n_out repeat the last one (computed, not stored)
    const u64 b = d.hat[(size_t)i * d.hat_ld + (d.col ? d.col[j] : j)];
    s_hat[t] = uint2{(u32)b & mask, (u32)(b >> h)};
  }
  __syncthreads();
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  if (n >= c.N) return;
  const u64* src = reb(c, in, rz.rep) + rz.z * in_stride + (size_t)d.src_pos0 * c.N + n;
  u64* dst = reb(c, outz.p[0] ? outz.p[rz.z] : out + rz.z * out_stride, rz.rep) + n;
  const u32 chunk = h <= 30 ? 16u : (h == 31 ? 4u : 1u);  // terms whose partial products fit 64-bit sums
  unsigned __int128 tot[kGroup];
  u64 s00[kGroup], s01[kGroup], s10[kGroup], s11[kGroup];
#pragma unroll
  for (int g = 0; g < kGroup; ++g) {
    tot[g] = 0;
    s00[g] = s01[g] = s10[g] = s11[g] = 0;
  }
  for (u32 i0 = 0; i0 < d.n_in; i0 += chunk) {
    const u32 i1 = min(i0 + chunk, d.n_in);
#pragma unroll 2
    for (u32 i = i0; i < i1; ++i) {
      u64 v = src[(size_t)i * c.N];
      if (d.scale) v = mul_shoup(v, d.scale[i], d.scale_prec[i], c.primes[d.src_gi[i]].q);
      const u32 a0 = (u32)v & mask, a1 = (u32)(v >> h);
#pragma unroll
      for (int g = 0; g < kGroup; ++g) {
        const uint2 b = s_hat[i * kGroup + g];  // same address in every lane: an LDS broadcast
        s00[g] += (u64)a0 * b.x;
        s01[g] += (u64)a0 * b.y;
        s10[g] += (u64)a1 * b.x;
        s11[g] += (u64)a1 * b.y;
      }
    }
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      tot[g] += (unsigned __int128)s00[g] + (((unsigned __int128)s01[g] + s10[g]) << h) + ((unsigned __int128)s11[g] << (2 * h));
      s00[g] = s01[g] = s10[g] = s11[g] = 0;
    }
  }
#pragma unroll
  for (int g = 0; g < kGroup; ++g) {
    const u32 j = j0 + g;
    if (j < d.n_out && owns(c, d.out_gi[j])) {  // (limb-sharded: the other ranks' outputs are not stored)
      const DevPrime& P = c.primes[d.out_gi[j]];
      dst[(size_t)d.out_pos[j] * c.N] = reduce128(U128{(u64)tot[g], (u64)(tot[g] >> 64)}, P.q, P.prec128_lo, P.prec128_hi);
    }
  }
}

// The usual case -- at most 16 sources, primes below 2^60 -- as its own kernel: every source value of the lane is requested
// before anything else waits (the loads overlap the descriptor / constant chain), one chunk of terms needs no 128-bit
// running total (<= 128 VGPRs: 4 workgroups per CU), and the per-output reduction constants come from LDS.
__global__ __launch_bounds__(256, 4) void base_conv_batch16_kernel(DevCtx c, u64* __restrict__ out, size_t out_stride,
                                                                   const u64* __restrict__ in, size_t in_stride,
                                                                   const ConvDesc* __restrict__ descs, u32 desc_step, PtrTab8 outz) {
  constexpr u32 kIn = 16;
  __shared__ uint2 s_hat[kIn * kGroup];
  __shared__ u64 s_q[kGroup], s_ml[kGroup], s_mh[kGroup];
  __shared__ u32 s_pos[kGroup], s_own[kGroup];
  const RepZ rz = rep_of_z(c);
  const ConvDesc d = descs[rz.z * desc_step];
  const u32 j0 = blockIdx.y * kGroup;
  if (j0 >= d.n_out) return;  // uniform for the workgroup
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  const bool active = n < c.N;
  const u32 h = c.split_bits, mask = (1u << h) - 1u;
  const u64* src = reb(c, in, rz.rep) + rz.z * in_stride + (size_t)d.src_pos0 * c.N + (active ? n : 0);
  u64 y[kIn];
#pragma unroll
  for (u32 i = 0; i < kIn; ++i) y[i] = i < d.n_in ? src[(size_t)i * c.N] : 0;
  for (u32 t = threadIdx.x; t < d.n_in * kGroup; t += 256) {
    const u32 i = t / kGroup, j = min(j0 + t % kGroup, d.n_out - 1);
    const u64 b = d.hat[(size_t)i * d.hat_ld + (d.col ? d.col[j] : j)];
    s_hat[t] = uint2{(u32)b & mask, (u32)(b >> h)};
  }
  if (threadIdx.x < kGroup) {
    const u32 j = min(j0 + threadIdx.x, d.n_out - 1);
    const DevPrime& P = c.primes[d.out_gi[j]];
    s_q[threadIdx.x] = P.q;
    s_ml[threadIdx.x] = P.prec128_lo;
    s_mh[threadIdx.x] = P.prec128_hi;
    s_pos[threadIdx.x] = d.out_pos[j];
    s_own[threadIdx.x] = owns(c, d.out_gi[j]);  // (limb-sharded: the other ranks' outputs are not stored)
  }
  __syncthreads();
  if (!active) return;
  u64 s00[kGroup], s01[kGroup], s10[kGroup], s11[kGroup];
#pragma unroll
  for (int g = 0; g < kGroup; ++g) s00[g] = s01[g] = s10[g] = s11[g] = 0;
#pragma unroll
  for (u32 i = 0; i < kIn; ++i) {
    if (i >= d.n_in) break;  // uniform
    u64 v = y[i];
    if (d.scale) v = mul_shoup(v, d.scale[i], d.scale_prec[i], c.primes[d.src_gi[i]].q);
    const u32 a0 = (u32)v & mask, a1 = (u32)(v >> h);
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
      const uint2 b = s_hat[i * kGroup + g];
      s00[g] += (u64)a0 * b.x;
      s01[g] += (u64)a0 * b.y;
      s10[g] += (u64)a1 * b.x;
      s11[g] += (u64)a1 * b.y;
    }
  }
  u64* dst = reb(c, outz.p[0] ? outz.p[rz.z] : out + rz.z * out_stride, rz.rep) + n;
#pragma unroll
  for (int g = 0; g < kGroup; ++g) {
    if (j0 + g < d.n_out && s_own[g]) {
      const unsigned __int128 tot =
          (unsigned __int128)s00[g] + (((unsigned __int128)s01[g] + s10[g]) << h) + ((unsigned __int128)s11[g] << (2 * h));
      dst[(size_t)s_pos[g] * c.N] = reduce128(U128{(u64)tot, (u64)(tot >> 64)}, s_q[g], s_ml[g], s_mh[g]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same conversion on the matrix cores.  out[j][n] = (sum_i y_i[n] * hat[i][j]) mod t_j is a matrix product
// [coefficients x sources] * [sources x outputs] over the integers followed by one reduction per entry, and the multiply-adds
// are the cost of the kernels above (4 v_mad_u64_u32 per term and output).  v_mfma_i32_16x16x64_i8 does 16 384 byte products per
// instruction, so the product is taken apart into bytes:
//   y_i = sum_a u_{i,a} 2^(8a)  (a < 8: the eight BYTES of the stored residue, no extraction needed),
//   G_{(i,a),j} = hat[i][j] * 2^(8a) mod t_j = sum_b g_{(i,a),j,b} 2^(7b)  (b < 9 seven-bit digits, host, once per level),
//   sum_i y_i hat[i][j]  ==  sum_b 2^(7b) C_b[n][j]   (mod t_j),      C_b[n][j] = sum_{k=(i,a)} u_k[n] * g_{k,j,b}:
// nine int8 matrix products with K = 8 * n_in.  The instruction multiplies SIGNED bytes: the A operand is the residue's bytes
// with the top bit flipped (u - 128, one XOR per register) and the accumulator starts at 128 * sum_k g_{k,j,b} (ConvDesc::boff),
// so it ends at exactly C_b >= 0 (below 2^23 for n_in <= 16).  The digits are put together in 128 bits (below 2^79) and reduced
// once, like the sums of the kernels above: the result is the canonical residue of an integer congruent to the reference's
// sum (Reduce_rns_base polynomial.c:928-967), i.e. the same bits.
// Layout: lane l = (r = l & 15, g = l >> 4).  A: row r = coefficient n0 + r, the 16 bytes of source limbs 8s + 2g, 8s + 2g + 1 of
// k-step s.  B: column r = output 16*tile + r, the same 16 (limb, byte) pairs (element order inside a lane group is the same
// for A and B whatever the hardware's k numbering, and a sum over k does not depend on it).  D: column r, rows 4g + reg.
// One wave keeps the B fragments of its output tile in registers and walks 256 coefficients in blocks of 16.
// ------------------------------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));
// (v1:v0) mod q for v below 2^80 (v1 < 2^16) and a prime above 2^32, i.e. m = floor(2^128/q) >> 64 below 2^32: reduce128()'s
// quotient (device_arith.hpp: three partial products, at most 4 below the true one) without the branches and the zero terms
__device__ __forceinline__ u64 reduce80(u64 v0, u32 v1, u64 q, u64 ml, u32 m) {
  const u64 A = (u64)v1 * m;                                      // v1 * mh (exact, < 2^48)
  const u64 Bq = ((u64)v1 * (u32)(ml >> 32) + (((u64)v1 * (u32)ml) >> 32)) >> 32;  // hi64(v1 * ml)
  const u64 t1 = (u64)(u32)v0 * m;
  const u64 t2 = (u64)(u32)(v0 >> 32) * m + (t1 >> 32);           // hi64(v0 * mh) = t2 >> 32
  const u64 qhat = A + Bq + (t2 >> 32);
  u64 r = v0 - qhat * q;
  const u64 q4 = 4 * q, q2 = 2 * q;
  r = r >= q4 ? r - q4 : r;
  r = r >= q2 ? r - q2 : r;
  return r >= q ? r - q : r;
}
constexpr u32 kMfmaWaveCoeffs = 256, kMfmaWgCoeffs = 4 * kMfmaWaveCoeffs;

#ifndef ACEHIP_CONV_MIN_WG
#define ACEHIP_CONV_MIN_WG 1  // workgroups per CU the register allocation must leave room for (experiments: tools/kernel_ab.sh)
#endif
#ifndef ACEHIP_BSGS_MIN_WG
#define ACEHIP_BSGS_MIN_WG 3  // measured (profiles/r04l_kernel_ab.txt): BSGS kernel time 0.931 -> 0.744 s per 24 images at 3 waves per SIMD (4: 0.844; conversion at 4: worse, spills)
#endif
template <int STEPS>
__global__ __launch_bounds__(256, ACEHIP_CONV_MIN_WG) void base_conv_mfma_kernel(DevCtx c, u64* __restrict__ out, size_t out_stride,
                                                             const u64* __restrict__ in, size_t in_stride,
                                                             const ConvDesc* __restrict__ descs, u32 desc_step, PtrTab8 outz) {
  constexpr int NB = (int)kConvMfmaDigits;
  __shared__ v4i_t sB[STEPS * NB * 64];  // the B fragments of this workgroup's output tile (its four waves share them)
  const RepZ rz = rep_of_z(c);  // blockIdx.z = problem + n_problems * replica
  const ConvDesc d = descs[rz.z * desc_step];
  const u32 tile = blockIdx.y;
  if (tile * 16 >= d.n_out) return;  // uniform for the workgroup
  {
    const v4i_t* bf = reinterpret_cast<const v4i_t*>(d.bfrag) + (size_t)tile * (STEPS * NB * 64);
    for (u32 t = threadIdx.x; t < STEPS * NB * 64; t += 256) sB[t] = bf[t];
  }
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 r = lane & 15u, g = lane >> 4;
  const u32 j = tile * 16 + r, jc = min(j, d.n_out - 1);  // this lane's output column (columns past n_out: zero constants, not stored)
  const u32 gi_out = d.out_gi[jc];
  const DevPrime& P = c.primes[gi_out];
  const u64 q = P.q, ml = P.prec128_lo;
  const u32 mh = (u32)P.prec128_hi;
  const bool store = j < d.n_out && owns(c, gi_out);  // (limb-sharded: the other ranks' outputs are not stored)
  // accumulator offsets of the column (128 * sum_k g_kb, see above), already weighted: the accumulators start at zero and the
  // digit sums below start from these two constants, which makes them the non-negative C_b sums
  u64 off_lo = 0, off_hi = 0;
#pragma unroll
  for (int b = 0; b < 5; ++b) off_lo += (u64)d.boff[(size_t)j * NB + b] << (7 * b);
#pragma unroll
  for (int b = 5; b < NB; ++b) off_hi += (u64)d.boff[(size_t)j * NB + b] << (7 * (b - 5));
  // this lane's source limbs of every k-step (past n_in: any valid limb, its constants are zero)
  u32 li[STEPS][2];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    li[s][0] = min((u32)(8 * s) + 2 * g, d.n_in - 1);
    li[s][1] = min((u32)(8 * s) + 2 * g + 1, d.n_in - 1);
  }
  const u64* src = reb(c, in, rz.rep) + rz.z * in_stride + (size_t)d.src_pos0 * c.N;
  u64* dst = reb(c, outz.p[0] ? outz.p[rz.z] : out + rz.z * out_stride, rz.rep) + (size_t)d.out_pos[jc] * c.N;
  typedef const __attribute__((address_space(1))) u64* gcptr_t;
  typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) u64x2_t* gptr2_t;
  const gcptr_t gsrc = (gcptr_t)(uintptr_t)src;
  const u32 n_base = (blockIdx.x * 4 + wave) * kMfmaWaveCoeffs;
  int w0 = 1, w7 = 1 << 7, w14 = 1 << 14, w21 = 1 << 21, w28 = 1 << 28;
  asm volatile("" : "+s"(w0), "+s"(w7), "+s"(w14), "+s"(w21), "+s"(w28));  // opaque: each digit term stays ONE v_mad_i64_i32
  // loads of a block are only requested here; scaling and the sign flip happen when the block is consumed (finish_a), so that
  // the next block's loads stay in flight while this block is multiplied and reduced
  auto load_raw = [&](u32 n0, u64 (&raw)[STEPS][2]) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // (global address space spelled out: as FLAT loads they would count on lgkmcnt too and every wait for an LDS fragment
      // would wait for the prefetch of the next block)
      raw[s][0] = gsrc[(size_t)li[s][0] * c.N + n0 + r];
      raw[s][1] = gsrc[(size_t)li[s][1] * c.N + n0 + r];
    }
  };
  auto finish_a = [&](const u64 (&raw)[STEPS][2], v4i_t (&a)[STEPS]) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      u64 v0 = raw[s][0], v1 = raw[s][1];
      if (d.scale) {  // pre-factor applied on load where the inverse NTT did not fold it in (N != 2^16)
        v0 = mul_shoup(v0, d.scale[li[s][0]], d.scale_prec[li[s][0]], c.primes[d.src_gi[li[s][0]]].q);
        v1 = mul_shoup(v1, d.scale[li[s][1]], d.scale_prec[li[s][1]], c.primes[d.src_gi[li[s][1]]].q);
      }
      a[s] = v4i_t{(int)((u32)v0 ^ 0x80808080u), (int)((u32)(v0 >> 32) ^ 0x80808080u), (int)((u32)v1 ^ 0x80808080u),
                   (int)((u32)(v1 >> 32) ^ 0x80808080u)};
    }
  };
  __syncthreads();
  constexpr u32 kBlocks = kMfmaWaveCoeffs / 16;
  u64 raw[2][STEPS][2];  // two blocks ahead: block rb + 2 is requested while block rb is multiplied and reduced
  load_raw(n_base, raw[0]);
  load_raw(n_base + 16, raw[1]);
#pragma unroll 2
  for (u32 rb = 0; rb < kBlocks; ++rb) {
    const u32 n0 = n_base + rb * 16;
    v4i_t a[STEPS];
    finish_a(raw[rb & 1], a);
    if (rb + 2 < kBlocks) load_raw(n0 + 32, raw[rb & 1]);
    v4i_t acc[NB];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // the nine fragments of a step are requested together (one LDS latency per step, not one per multiply)
      v4i_t Bs[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) Bs[b] = sB[(s * NB + b) * 64 + lane];
      static_assert(NB == 9, "the operand list below names nine fragments");
      asm volatile("s_waitcnt lgkmcnt(0)"  // (all nine live here: otherwise the scheduler requests them in pairs to save registers)
                   : "+v"(Bs[0]), "+v"(Bs[1]), "+v"(Bs[2]), "+v"(Bs[3]), "+v"(Bs[4]), "+v"(Bs[5]), "+v"(Bs[6]), "+v"(Bs[7]), "+v"(Bs[8]));
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], Bs[b], s == 0 ? v4i_t{0, 0, 0, 0} : acc[b], 0, 0, 0);
    }
    u64 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // coefficient n0 + 4g + t
      int64_t lo = (int64_t)off_lo;
      lo += (int64_t)acc[0][t] * w0;
      lo += (int64_t)acc[1][t] * w7;
      lo += (int64_t)acc[2][t] * w14;
      lo += (int64_t)acc[3][t] * w21;
      lo += (int64_t)acc[4][t] * w28;  // in [0, 2^52)
      int64_t hs = (int64_t)off_hi;
      hs += (int64_t)acc[5][t] * w0;
      hs += (int64_t)acc[6][t] * w7;
      hs += (int64_t)acc[7][t] * w14;
      hs += (int64_t)acc[8][t] * w21;  // in [0, 2^45), weight 2^35
      const u64 hi = (u64)hs;
      const u64 v0 = (u64)lo + (hi << 35);
      const u64 v1 = (hi >> 29) + (v0 < (u64)lo ? 1u : 0u);  // < 2^16
      o[t] = reduce80(v0, (u32)v1, q, ml, mh);
    }
    if (store) {
      *(gptr2_t)(uintptr_t)(dst + n0 + 4 * g) = u64x2_t{o[0], o[1]};
      *(gptr2_t)(uintptr_t)(dst + n0 + 4 * g + 2) = u64x2_t{o[2], o[3]};
    }
  }
}

void launch_base_conv_batch(const DevCtx& c, u64* out, size_t out_stride, const u64* in, size_t in_stride,
                            const ConvDesc* descs, u32 desc_step, u32 n_problems, u32 max_n_out, hipStream_t s, u32 max_n_in,
                            const PtrTab8& outz, u32 mfma_steps) {
  ACEHIP_ABLATE(ABL_CONV);
  if (n_problems == 0 || max_n_out == 0) return;
  if (mfma_steps != 0 && c.N % kMfmaWgCoeffs == 0) {
    dim3 grid(c.N / kMfmaWgCoeffs, (max_n_out + 15) / 16, n_problems * c.nrep), block(256);
    if (mfma_steps == 1) hipLaunchKernelGGL((base_conv_mfma_kernel<1>), grid, block, 0, s, c, out, out_stride, in, in_stride, descs, desc_step, outz);
    else                 hipLaunchKernelGGL((base_conv_mfma_kernel<2>), grid, block, 0, s, c, out, out_stride, in, in_stride, descs, desc_step, outz);
    return;
  }
  dim3 grid((c.N + 255) / 256, (max_n_out + kGroup - 1) / kGroup, n_problems * c.nrep), block(256);
  if (max_n_in != 0 && max_n_in <= 16 && c.split_bits <= 30)
    hipLaunchKernelGGL(base_conv_batch16_kernel, grid, block, 0, s, c, out, out_stride, in, in_stride, descs, desc_step, outz);
  else
    hipLaunchKernelGGL(base_conv_batch_kernel, grid, block, 0, s, c, out, out_stride, in, in_stride, descs, desc_step, outz);
}

// acc{0,1}[pos][n] = sum_d key{0,1}[d][gi][n] * e_d[pos][n];  key layout [nd][2][L+K][N]
// add0 (may be null): acc0[pos] += add0[pos] * w.w[pos] on the q-limbs -- the "+ P*c0" of Fast_rotate_ext (ckks_evaluator.c:539-575)
__global__ __launch_bounds__(256) void key_mac_fused_kernel(DevCtx c, u64* acc0, u64* acc1, const u64* key, const u64* ext,
                                                            size_t ext_stride, const u64* in, u32 level, u32 nd, u32 alpha,
                                                            const u64* add0, LimbConsts w) {
  const RepBlk rb = rep_block(c, (c.N / 2 + 255) / 256, level + c.K);  // replicas of a tile side by side: the key is shared
  const u32 pos = rb.y;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const u32 rep = rb.rep;
  acc0 = reb(c, acc0, rep);
  acc1 = reb(c, acc1, rep);
  key = reb(c, key, rep);
  ext = reb(c, ext, rep);
  in = reb(c, in, rep);
  add0 = reb(c, add0, rep);
  const DevPrime P = c.primes[gi];
  const size_t T = c.L + c.K;
  const size_t pb = (size_t)pos * c.N, kb = (size_t)gi * c.N;
  const u32 i = (rb.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  // digit whose ModUp passes this limb through (read from `in` when given; otherwise ext holds it)
  const u32 own = (in != nullptr && pos < level) ? pos / alpha : 0xffffffffu;
  ulong2 r0{0, 0}, r1{0, 0};
  for (u32 d = 0; d < nd; ++d) {
    const u64* e_src = (d == own) ? in + pb : ext + d * ext_stride + pb;
    const ulong2 e = *reinterpret_cast<const ulong2*>(e_src + i);
    const u64* k0p = key + ((size_t)d * 2) * T * c.N + kb;
    const ulong2 k0 = *reinterpret_cast<const ulong2*>(k0p + i);
    const ulong2 k1 = *reinterpret_cast<const ulong2*>(k0p + T * c.N + i);
    r0.x = add_mod(r0.x, mul_mod(k0.x, e.x, P), P.q);
    r0.y = add_mod(r0.y, mul_mod(k0.y, e.y, P), P.q);
    r1.x = add_mod(r1.x, mul_mod(k1.x, e.x, P), P.q);
    r1.y = add_mod(r1.y, mul_mod(k1.y, e.y, P), P.q);
  }
  if (add0 != nullptr && pos < level) {  // uniform for the workgroup
    const ulong2 a = *reinterpret_cast<const ulong2*>(add0 + pb + i);
    const u64 wl = w.w[pos];
    r0.x = add_mod(r0.x, mul_mod(a.x, wl, P), P.q);
    r0.y = add_mod(r0.y, mul_mod(a.y, wl, P), P.q);
  }
  *reinterpret_cast<ulong2*>(acc0 + pb + i) = r0;
  *reinterpret_cast<ulong2*>(acc1 + pb + i) = r1;
}

// The same products for a launch that covers several replicas (image batches): the key is the same for every image, so one
// workgroup keeps its 512 coefficients of the key limbs of all digits in registers and walks the replicas -- the key stream is read
// once per launch instead of once per replica (measured through L2 it arrived 6 times at 12 images per launch: 378 of the 1 190 MB
// a call moved, profiles/r04h_traffic.json).  ND = digits held (nd <= ND).
template <int ND>
__global__ __launch_bounds__(256) void key_mac_fused_reps_kernel(DevCtx c, u64* acc0, u64* acc1, const u64* key, const u64* ext,
                                                                 size_t ext_stride, const u64* in, u32 level, u32 nd, u32 alpha,
                                                                 const u64* add0, LimbConsts w) {
  const u32 X = (c.N / 2 + 255) / 256;
  const u32 pos = __builtin_amdgcn_readfirstlane(blockIdx.x / X), x = blockIdx.x % X;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const DevPrime P = c.primes[gi];
  const size_t T = c.L + c.K;
  const size_t pb = (size_t)pos * c.N, kb = (size_t)gi * c.N;
  const u32 i = (x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u32 own = (in != nullptr && pos < level) ? pos / alpha : 0xffffffffu;
  ulong2 k0[ND], k1[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const u32 dd = (u32)d < nd ? (u32)d : 0;  // (digits past nd: reload digit 0, never used)
    const u64* k0p = key + ((size_t)dd * 2) * T * c.N + kb;  // (keys lie outside the arena: the same for every replica)
    k0[d] = *reinterpret_cast<const ulong2*>(k0p + i);
    k1[d] = *reinterpret_cast<const ulong2*>(k0p + T * c.N + i);
  }
  const u64 wl = (add0 != nullptr && pos < level) ? w.w[pos] : 0;
  for (u32 r = 0; r < c.nrep; ++r) {
    const u32 rep = c.rep0 + r;
    const u64* ext_r = reb(c, ext, rep);
    const u64* in_r = reb(c, in, rep);
    // exact 128-bit sums (at most four products below 2^122), one reduction per output
    U128 s0x{0, 0}, s0y{0, 0}, s1x{0, 0}, s1y{0, 0};
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      if ((u32)d < nd) {
        const u64* e_src = ((u32)d == own) ? in_r + pb : ext_r + (size_t)d * ext_stride + pb;
        const ulong2 e = *reinterpret_cast<const ulong2*>(e_src + i);
        mac128(s0x, k0[d].x, e.x);
        mac128(s0y, k0[d].y, e.y);
        mac128(s1x, k1[d].x, e.x);
        mac128(s1y, k1[d].y, e.y);
      }
    }
    ulong2 r0{reduce128(s0x, P.q, P.prec128_lo, P.prec128_hi), reduce128(s0y, P.q, P.prec128_lo, P.prec128_hi)};
    ulong2 r1{reduce128(s1x, P.q, P.prec128_lo, P.prec128_hi), reduce128(s1y, P.q, P.prec128_lo, P.prec128_hi)};
    if (add0 != nullptr && pos < level) {
      const ulong2 a = *reinterpret_cast<const ulong2*>(reb(c, add0, rep) + pb + i);
      r0.x = add_mod(r0.x, mul_mod(a.x, wl, P), P.q);
      r0.y = add_mod(r0.y, mul_mod(a.y, wl, P), P.q);
    }
    *reinterpret_cast<ulong2*>(reb(c, acc0, rep) + pb + i) = r0;
    *reinterpret_cast<ulong2*>(reb(c, acc1, rep) + pb + i) = r1;
  }
}

void launch_key_mac_fused(const DevCtx& c, u64* acc0, u64* acc1, const u64* key, const u64* ext, size_t ext_stride,
                          const u64* in, u32 level, u32 nd, u32 alpha, hipStream_t s, const u64* add0, const LimbConsts* w) {
  ACEHIP_ABLATE(ABL_KEYMAC);
  const u32 X = (c.N / 2 + 255) / 256;
  // several replicas and a key outside the arena (the usual case: keys are shared by all images): the replica-walking form
  static const bool reps_on = [] { const char* e = getenv("ACEHIP_KEYMAC_REPS"); return !e || atoi(e) != 0; }();
  const bool key_shared = !((u64)key - c.rep_lo < c.rep_span);
  if (reps_on && c.nrep > 1 && key_shared && nd <= 4) {
    dim3 grid(X * (level + c.K)), block(256);
    const LimbConsts lw = w ? *w : LimbConsts{};
    if (nd <= 2) hipLaunchKernelGGL(key_mac_fused_reps_kernel<2>, grid, block, 0, s, c, acc0, acc1, key, ext, ext_stride, in, level, nd, alpha, add0, lw);
    else         hipLaunchKernelGGL(key_mac_fused_reps_kernel<4>, grid, block, 0, s, c, acc0, acc1, key, ext, ext_stride, in, level, nd, alpha, add0, lw);
    return;
  }
  dim3 grid(X * (level + c.K) * c.nrep), block(256);  // 1-D: rep_block() maps it
  hipLaunchKernelGGL(key_mac_fused_kernel, grid, block, 0, s, c, acc0, acc1, key, ext, ext_stride, in, level, nd, alpha, add0,
                     w ? *w : LimbConsts{});
}

// Several key inner products over the SAME raised digits (the hoisted rotations of Rotate_iteration ckks_bootstrap_context.c:1276-1290:
// one Switch_key_precompute, then Fast_rotate_ext with one key per rotation): rotation j gets acc{0,1}_j = sum_d key_j{0,1}[d] (*) ext[d]
// (+ add0 * w on the q-limbs of acc0_j).  One key inner product per launch reads the digits -- beta (l+K) limbs per image, hundreds of MB per
// batch, far beyond any cache -- once per ROTATION; here a lane keeps its coefficient of every digit of up to 12 images in registers
// and walks the rotations: the digits are read once per launch, every key part once per launch (for all images), only the sums are
// per rotation and image.  One coefficient per lane (8-byte accesses, 512 contiguous bytes per wave instruction) so that 12 images x ND
// digits fit the register file.  Same arithmetic as key_mac_fused_kernel: canonical residues of the same sums.
constexpr u32 kKmMultiReps = 12;
struct KeyMultiArgs {
  u64* acc0[KEY_MULTI_MAX];
  u64* acc1[KEY_MULTI_MAX];
  const u64* key[KEY_MULTI_MAX];  // key set of rotation j: [nd][2][L+K][N]
  u32 n;
};
template <int ND>
__global__ __launch_bounds__(256) void key_mac_multi_kernel(DevCtx c, KeyMultiArgs a, const u64* ext, size_t ext_stride, u32 level, u32 nd,
                                                            const u64* add0, LimbConsts w, u32 rep_first, u32 rep_count) {
  const u32 X = (c.N + 255) / 256;
  const u32 pos = __builtin_amdgcn_readfirstlane(blockIdx.x / X), x = blockIdx.x % X;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const DevPrime P = c.primes[gi];
  const size_t T = c.L + c.K;
  const size_t pb = (size_t)pos * c.N, kb = (size_t)gi * c.N;
  const u32 i = x * 256 + threadIdx.x;
  if (i >= c.N) return;
  const bool with_add = add0 != nullptr && pos < level;  // uniform for the workgroup
  const u64 wl = with_add ? w.w[pos] : 0;
  u64 e[kKmMultiReps][ND], pc[kKmMultiReps];
#pragma unroll
  for (u32 r = 0; r < kKmMultiReps; ++r) {
    const u32 rr = r < rep_count ? r : 0;  // (images past the group: reload the first, never stored)
    const u64* ext_r = reb(c, ext, c.rep0 + rep_first + rr);
#pragma unroll
    for (int d = 0; d < ND; ++d) e[r][d] = (u32)d < nd ? ext_r[(size_t)d * ext_stride + pb + i] : 0;
    pc[r] = with_add ? reb(c, add0, c.rep0 + rep_first + rr)[pb + i] : 0;
  }
  if (with_add) {
#pragma unroll
    for (u32 r = 0; r < kKmMultiReps; ++r) pc[r] = mul_mod(pc[r], wl, P);
  }
  for (u32 j = 0; j < a.n; ++j) {
    u64 k0[ND], k1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const u32 dd = (u32)d < nd ? (u32)d : 0;
      const u64* kp = a.key[j] + ((size_t)dd * 2) * T * c.N + kb;  // (keys lie outside the arena: the same for every image)
      k0[d] = kp[i];
      k1[d] = kp[T * c.N + i];
    }
    u64* o0 = a.acc0[j];
    u64* o1 = a.acc1[j];
#pragma unroll
    for (u32 r = 0; r < kKmMultiReps; ++r) {
      if (r < rep_count) {  // uniform
        // exact 128-bit sums of the ND products (below 2^124), ONE reduction per output: the kernel is bound by instruction issue, and a
        // reduction per product costs three times the multiply-adds of the product (digits past nd are zeros)
        U128 s0{0, 0}, s1{0, 0};
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          mac128(s0, k0[d], e[r][d]);
          mac128(s1, k1[d], e[r][d]);
        }
        const u64 r0 = add_mod(reduce128(s0, P.q, P.prec128_lo, P.prec128_hi), pc[r], P.q);
        const u64 r1 = reduce128(s1, P.q, P.prec128_lo, P.prec128_hi);
        reb(c, o0, c.rep0 + rep_first + r)[pb + i] = r0;
        reb(c, o1, c.rep0 + rep_first + r)[pb + i] = r1;
      }
    }
  }
}

void launch_key_mac_multi(const DevCtx& c, u64* const* acc0, u64* const* acc1, const u64* const* keys, u32 n_keys, const u64* ext,
                          size_t ext_stride, u32 level, u32 nd, hipStream_t s, const u64* add0, const LimbConsts* w) {
  ACEHIP_ABLATE(ABL_KEYMAC);
  KeyMultiArgs a;
  a.n = n_keys;
  for (u32 j = 0; j < KEY_MULTI_MAX; ++j) {
    a.acc0[j] = j < n_keys ? acc0[j] : nullptr;
    a.acc1[j] = j < n_keys ? acc1[j] : nullptr;
    a.key[j] = j < n_keys ? keys[j] : nullptr;
  }
  const LimbConsts lw = w ? *w : LimbConsts{};
  dim3 grid(((c.N + 255) / 256) * (level + c.K)), block(256);
  for (u32 r0 = 0; r0 < c.nrep; r0 += kKmMultiReps) {  // the images of the launch in groups the registers hold
    const u32 cnt = std::min(kKmMultiReps, c.nrep - r0);
    if (nd <= 1)      hipLaunchKernelGGL(key_mac_multi_kernel<1>, grid, block, 0, s, c, a, ext, ext_stride, level, nd, add0, lw, r0, cnt);
    else if (nd == 2) hipLaunchKernelGGL(key_mac_multi_kernel<2>, grid, block, 0, s, c, a, ext, ext_stride, level, nd, add0, lw, r0, cnt);
    else if (nd == 3) hipLaunchKernelGGL(key_mac_multi_kernel<3>, grid, block, 0, s, c, a, ext, ext_stride, level, nd, add0, lw, r0, cnt);
    else              hipLaunchKernelGGL(key_mac_multi_kernel<4>, grid, block, 0, s, c, a, ext, ext_stride, level, nd, add0, lw, r0, cnt);
  }
}

// ------------------------------------------------------------------------------------------------
// Baby-step giant-step inner products of a homomorphic linear transform (Rotate_iteration
// ckks_bootstrap_context.c:1326-1341, the Mul_plaintext / Add loops over one giant step):
//   out_i = sum_{j < g} rot_j (*) pt_{i,j}      for every i < b, both polynomials of the PQ-extended ciphertexts
// One lane keeps its two coefficients of all g pre-rotated ciphertexts (c0 and c1) in registers and streams the b*g
// plaintext diagonals past them once: per limb 2g + b*g + 2b limb transfers instead of the 3*b*g + 2b of one
// multiply-accumulate chain per output (the rotated ciphertexts -- 2g*(l+K) limbs, hundreds of MiB -- are otherwise
// re-read for every i).  Products are accumulated exactly in 128 bits and reduced once per output (g <= 16 products
// below 2^122 each), so the result is the canonical residue of the exact sum whatever the order.
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256, ACEHIP_BSGS_MIN_WG) void bsgs_inner_kernel(DevCtx c, BsgsArgs a, u32 level) {
  const RepBlk rb = rep_block(c, (c.N / 2 + 255) / 256, level + c.K);  // replicas of a tile side by side: the diagonals are shared
  const u32 pos = rb.y;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const u32 rep = rb.rep;
  const DevPrime& P = c.primes[gi];
  const u64 q = P.q, ml = P.prec128_lo, mh = P.prec128_hi;
  const size_t ct_off = (size_t)pos * c.N;  // PQ-extended ciphertext: p-limbs follow the q-limbs
  // plaintext: q-limb `pos`, or p-limb (pos - level) behind its own (possibly higher) number of q-limbs
  const size_t pt_off = pos < level ? ct_off : (size_t)(a.pt_q_alloc + (pos - level)) * c.N;
  const u32 i = (rb.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 r0[G], r1[G];
  const u32 sh = __builtin_clz(c.N) + 1;  // 32 - log2(N)
  const u32 b0 = __brev(i) >> sh, b1 = __brev(i + 1) >> sh;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (j < (int)a.g) {
      const u32 k = a.in_auto[j];
      const u64 *in0 = reb(c, a.in0[j], rep), *in1 = reb(c, a.in1[j], rep);
      if (k == 0) {
        r0[j] = *reinterpret_cast<const ulong2*>(in0 + ct_off + i);
        r1[j] = *reinterpret_cast<const ulong2*>(in1 + ct_off + i);
      } else {  // rotated input: in[perm_k(i)], perm_k(i) = rev(((2 rev(i) + 1) k mod 2N) / 2)  (automorphism_order_ntt)
        const u32 px = __brev((((2 * b0 + 1) * k) & (2 * c.N - 1)) >> 1) >> sh;
        const u32 py = __brev((((2 * b1 + 1) * k) & (2 * c.N - 1)) >> 1) >> sh;
        r0[j] = ulong2{in0[ct_off + px], in0[ct_off + py]};
        r1[j] = ulong2{in1[ct_off + px], in1[ct_off + py]};
      }
    }
  }
  for (u32 bi = 0; bi < a.b; ++bi) {
    U128 s0x{0, 0}, s0y{0, 0}, s1x{0, 0}, s1y{0, 0};
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const u64* pt = j < (int)a.g ? reb(c, a.pt[bi * a.g + j], rep) : nullptr;
      if (pt != nullptr) {  // a missing diagonal (giant + j == num_rot) contributes nothing
#ifdef BSGS_EXP  // timing experiment (results are wrong): every diagonal load hits the first 4 KiB of its limb -- what the diagonal stream costs
        const ulong2 p = *reinterpret_cast<const ulong2*>(pt + pt_off + (i & 510u));
#else
        const ulong2 p = *reinterpret_cast<const ulong2*>(pt + pt_off + i);
#endif
        mac128(s0x, r0[j].x, p.x);
        mac128(s0y, r0[j].y, p.y);
        mac128(s1x, r1[j].x, p.x);
        mac128(s1y, r1[j].y, p.y);
      }
    }
    ulong2 o0, o1;
    o0.x = reduce128(s0x, q, ml, mh);
    o0.y = reduce128(s0y, q, ml, mh);
    o1.x = reduce128(s1x, q, ml, mh);
    o1.y = reduce128(s1y, q, ml, mh);
    *reinterpret_cast<ulong2*>(reb(c, a.out0[bi], rep) + ct_off + i) = o0;
    *reinterpret_cast<ulong2*>(reb(c, a.out1[bi], rep) + ct_off + i) = o1;
  }
}

// The same inner products with ONE coefficient per lane: half the registers per lane (G = 16: 64 instead of 128 for the resident
// inputs), twice the lanes.  The two-coefficient form of G = 16 needs 168 VGPRs at three waves per SIMD and spills 16 of them into
// scratch inside the diagonal loop (ISA metadata of round 4: vgpr_spill_count 16, private_segment_fixed_size 20); this form has no
// spills at ACEHIP_BSGS1_MIN_WG waves.  Which one runs for g > 8: ACEHIP_BSGS16_CPL (tools/kernel_ab.sh decides).
#ifndef ACEHIP_BSGS1_MIN_WG
#define ACEHIP_BSGS1_MIN_WG 4
#endif
#ifndef ACEHIP_BSGS16_CPL
#define ACEHIP_BSGS16_CPL 2
#endif
template <int G>
__global__ __launch_bounds__(256, ACEHIP_BSGS1_MIN_WG) void bsgs_inner1_kernel(DevCtx c, BsgsArgs a, u32 level) {
  const RepBlk rb = rep_block(c, (c.N + 255) / 256, level + c.K);
  const u32 pos = rb.y;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const u32 rep = rb.rep;
  const DevPrime& P = c.primes[gi];
  const u64 q = P.q, ml = P.prec128_lo, mh = P.prec128_hi;
  const size_t ct_off = (size_t)pos * c.N;
  const size_t pt_off = pos < level ? ct_off : (size_t)(a.pt_q_alloc + (pos - level)) * c.N;
  const u32 i = rb.x * 256 + threadIdx.x;
  if (i >= c.N) return;
  u64 r0[G], r1[G];
  const u32 sh = __builtin_clz(c.N) + 1;  // 32 - log2(N)
  const u32 b0 = __brev(i) >> sh;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (j < (int)a.g) {
      const u32 k = a.in_auto[j];
      const u64 *in0 = reb(c, a.in0[j], rep), *in1 = reb(c, a.in1[j], rep);
      const u32 px = k == 0 ? i : __brev((((2 * b0 + 1) * k) & (2 * c.N - 1)) >> 1) >> sh;
      r0[j] = in0[ct_off + px];
      r1[j] = in1[ct_off + px];
    }
  }
  for (u32 bi = 0; bi < a.b; ++bi) {
    U128 s0{0, 0}, s1{0, 0};
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const u64* pt = j < (int)a.g ? reb(c, a.pt[bi * a.g + j], rep) : nullptr;
      if (pt != nullptr) {
        const u64 p = pt[pt_off + i];
        mac128(s0, r0[j], p);
        mac128(s1, r1[j], p);
      }
    }
    reb(c, a.out0[bi], rep)[ct_off + i] = reduce128(s0, q, ml, mh);
    reb(c, a.out1[bi], rep)[ct_off + i] = reduce128(s1, q, ml, mh);
  }
}

// The images of a batch as the WAVES of one workgroup (round 5).  The diagonals are the same for every image; the forms above leave their
// sharing to L2 (workgroups of the same tile side by side), which holds for one stream and not next to the kernels of other image streams
// (every diagonal load redirected to 4 KiB: headline +1.2 %, profiles/r05al_*).  Here wave w of a workgroup is image rep_first + w, a lane
// one coefficient: the g diagonals of an output are staged ONCE per workgroup in LDS (64 coefficients each, double-buffered: the next
// output's are requested before this one's are multiplied) and read by all waves.  Same sums, same single reduction: same bits.
template <int G>
__global__ __launch_bounds__(1024) void bsgs_inner_reps_kernel(DevCtx c, BsgsArgs a, u32 level, u32 rep_first) {
  __shared__ u64 sd[2][G][64];
  const u32 X = c.N / 64;
  const u32 pos = __builtin_amdgcn_readfirstlane(blockIdx.x / X), x = blockIdx.x % X;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;  // (uniform for the workgroup)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const u32 rep = c.rep0 + rep_first + wave;
  const DevPrime& P = c.primes[gi];
  const u64 q = P.q, ml = P.prec128_lo, mh = P.prec128_hi;
  const size_t ct_off = (size_t)pos * c.N;
  const size_t pt_off = pos < level ? ct_off : (size_t)(a.pt_q_alloc + (pos - level)) * c.N;
  const u32 i = x * 64 + lane;
  u64 r0[G], r1[G];
  const u32 sh = __builtin_clz(c.N) + 1;  // 32 - log2(N)
  const u32 b0 = __brev(i) >> sh;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (j < (int)a.g) {
      const u32 k = a.in_auto[j];
      const u64 *in0 = reb(c, a.in0[j], rep), *in1 = reb(c, a.in1[j], rep);
      const u32 px = k == 0 ? i : __brev((((2 * b0 + 1) * k) & (2 * c.N - 1)) >> 1) >> sh;
      r0[j] = in0[ct_off + px];
      r1[j] = in1[ct_off + px];
    } else {
      r0[j] = r1[j] = 0;
    }
  }
  auto stage = [&](u32 bi, u32 buf) {  // the g diagonals of output bi, this workgroup's 64 coefficients (a missing one: zeros)
    for (u32 t = threadIdx.x; t < (u32)G * 64u; t += blockDim.x) {
      const u32 j = t >> 6, l = t & 63u;
      const u64* pt = j < a.g ? a.pt[bi * a.g + j] : nullptr;
      sd[buf][j][l] = pt != nullptr ? pt[pt_off + x * 64 + l] : 0;
    }
  };
  stage(0, 0);
  __syncthreads();
  for (u32 bi = 0; bi < a.b; ++bi) {
    if (bi + 1 < a.b) stage(bi + 1, (bi + 1) & 1u);
    const u32 buf = bi & 1u;
    U128 s0{0, 0}, s1{0, 0};
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const u64 p = sd[buf][j][lane];
      mac128(s0, r0[j], p);
      mac128(s1, r1[j], p);
    }
    reb(c, a.out0[bi], rep)[ct_off + i] = reduce128(s0, q, ml, mh);
    reb(c, a.out1[bi], rep)[ct_off + i] = reduce128(s1, q, ml, mh);
    __syncthreads();  // everybody has read this output's diagonals (the buffer is rewritten two outputs on) and the next one's are in place
  }
}

void launch_bsgs_inner(const DevCtx& c, const BsgsArgs& a, u32 level, hipStream_t s) {
  ACEHIP_ABLATE(ABL_BSGS);
  // several images per launch and every diagonal outside the replicated arena (the usual case: precomputed plaintexts are shared by all
  // images): the images-as-waves form.  ACEHIP_BSGS_REPS=0: the forms that share through L2.
  static const bool reps_on = [] { const char* e = getenv("ACEHIP_BSGS_REPS"); return !e || atoi(e) != 0; }();
  if (reps_on && c.nrep >= 4 && c.N % 64 == 0 && a.g > 8) {
    bool shared = true;
    for (u32 t = 0; t < a.g * a.b && shared; ++t) shared = !((u64)a.pt[t] - c.rep_lo < c.rep_span);
    if (shared) {
      dim3 grid((c.N / 64) * (level + c.K));
      // ACEHIP_BSGS_WG_IMAGES: images (= waves) per workgroup, default 16 = the whole batch in one workgroup.  A 12-wave workgroup needs
      // three wave slots with 128 registers on EVERY SIMD of one CU at once: next to the kernels of other image streams it waits for a CU
      // to drain (8.3 x its standalone time in the three-stream trace of round 5); smaller groups fit sooner and stage the diagonals
      // once per group.  Measured: profiles/r06g_*.
      static const u32 wg_images = [] { const char* e = getenv("ACEHIP_BSGS_WG_IMAGES"); const u32 v = e ? (u32)atoi(e) : 16u; return v >= 1 && v <= 16 ? v : 16u; }();
      for (u32 r0 = 0; r0 < c.nrep; r0 += wg_images) {
        const u32 cnt = std::min(wg_images, c.nrep - r0);
        hipLaunchKernelGGL((bsgs_inner_reps_kernel<16>), grid, dim3(64 * cnt), 0, s, c, a, level, r0);
      }
      return;
    }
  }
  dim3 grid(((c.N / 2 + 255) / 256) * (level + c.K) * c.nrep), block(256);  // 1-D: rep_block() maps it
  if (a.g <= 4)      hipLaunchKernelGGL((bsgs_inner_kernel<4>), grid, block, 0, s, c, a, level);
  else if (a.g <= 8) hipLaunchKernelGGL((bsgs_inner_kernel<8>), grid, block, 0, s, c, a, level);
  else if (ACEHIP_BSGS16_CPL == 1) {
    dim3 grid1(((c.N + 255) / 256) * (level + c.K) * c.nrep);
    hipLaunchKernelGGL((bsgs_inner1_kernel<16>), grid1, block, 0, s, c, a, level);
  } else             hipLaunchKernelGGL((bsgs_inner_kernel<16>), grid, block, 0, s, c, a, level);
}

__global__ __launch_bounds__(256) void moddown_tail2_kernel(DevCtx c, u64* __restrict__ out0, u64* __restrict__ out1,
                                                            const u64* __restrict__ x0, const u64* __restrict__ x1,
                                                            const u64* __restrict__ t0, const u64* __restrict__ t1,
                                                            const u64* __restrict__ pinv,
                                                            const u64* __restrict__ pinv_prec) {
  const u32 l = blockIdx.y;
  if (!owns(c, l)) return;
  const RepZ rz = rep_of_z(c);
  const u64 q = c.primes[l].q, w = pinv[l], wp = pinv_prec[l];
  const size_t base = (size_t)l * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u64* x = reb(c, rz.z ? x1 : x0, rz.rep);
  const u64* t = reb(c, rz.z ? t1 : t0, rz.rep);
  u64* o = reb(c, rz.z ? out1 : out0, rz.rep);
  const ulong2 vx = *reinterpret_cast<const ulong2*>(x + base + i);
  ulong2 vt = *reinterpret_cast<const ulong2*>(t + base + i);
  vt.x = mul_shoup(sub_mod(vx.x, vt.x, q), w, wp, q);
  vt.y = mul_shoup(sub_mod(vx.y, vt.y, q), w, wp, q);
  *reinterpret_cast<ulong2*>(o + base + i) = vt;
}

void launch_moddown_tail2(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t0,
                          const u64* t1, const u64* pinv, const u64* pinv_prec, u32 level, hipStream_t s, u32 n_polys) {
  ACEHIP_ABLATE(ABL_OTHER);
  dim3 grid((c.N / 2 + 255) / 256, level, n_polys * c.nrep), block(256);
  hipLaunchKernelGGL(moddown_tail2_kernel, grid, block, 0, s, c, out0, out1, x0, x1, t0, t1, pinv, pinv_prec);
}

}  // namespace acehip
