// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): negacyclic NTT/iNTT, limb-wise modular
// ops, RNS base conversion, ModDown/Rescale tails, key inner product.
//
// Everything here is 64-bit integer modular arithmetic (no MFMA).  The kernels are HBM-bandwidth
// bound by design: limb-major [limb][N] layout, 16-byte per-lane accesses, LDS-staged butterflies.
// Reference algorithms: fhe-cmplr/rtlib/ant/src/util/{ntt.c,polynomial.c}, src/poly/poly_arith.c.
#include <cstdlib>
#include "kernels.hpp"

namespace acehip {

#ifdef ACEHIP_ABLATION
unsigned ablate_mask() {
  static const unsigned m = [] {
    const char* e = getenv("ACEHIP_ABLATE");
    return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
  }();
  return m;
}
#endif


// ------------------------------------------------------------------------------------------------
// NTT passes.
//
// A pass performs `r` consecutive radix-2 stages [s0, s0+r) of the length-N transform on a tile held
// in LDS.  At stage s (m = 2^s butterfly groups, half-distance t = N / 2^(s+1)) the butterfly pairs
// differ in bit (logN-1-s) of the coefficient index j, and the twiddle is W[m + (j >> (logN-s))]
// (reference ntt.c:206-236: omega = rous[i + m], i = block index).  So a pass touches index bits
// [logN-s0-r, logN-s0): "rows"; the bits above are the outer block `o`, the bits below the column.
//   STRIDED tile : R = 2^r rows x C adjacent columns (C <= S = 2^(logN-s0-r)), LDS index row*C + col
//   CONTIG  tile : S = 1; C consecutive outer blocks of R contiguous elements, LDS index col*R + row
// Forward (Cooley-Tukey, natural -> bit-reversed): stages ascending.  Inverse (Gentleman-Sande,
// bit-reversed -> natural): stages descending, N^-1 folded into the last stage (s = 0).  All values
// stay canonical in [0,q), hence bit-identical to the reference whatever the stage grouping.
// ------------------------------------------------------------------------------------------------
template <bool CONTIG, bool INVERSE>
__global__ __launch_bounds__(256) void ntt_pass_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride, u32 level, u32 pos0, u32 pos_off,
                                                       u32 s0, u32 r, u32 log_c, u32 skip_alpha) {
  extern __shared__ u64 tile[];
  u32 pos;
  const RepZ rz = rep_of_z(c);  // blockIdx.z = polynomial + n_polys * replica
  if (!ntt_limb_pos(pos, pos0, level, c.K, skip_alpha, blockIdx.y, rz.z)) return;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const DevPrime P = c.primes[gi];
  const u64 q = P.q;
  u64* x = reb(c, poly, rz.rep) + rz.z * poly_stride + (size_t)(pos - pos_off) * c.N;
  const ulong2* W = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)gi * c.N;
  const u32 R = 1u << r, C = 1u << log_c;
  const u32 log_s = c.logN - s0 - r;
  const u32 chunk = blockIdx.x, tid = threadIdx.x;
  const u32 tile_elems = R * C;
  const u32 o_strided = (chunk << log_c) >> log_s;  // C <= S: one outer block per tile
  const u32 ci0 = (chunk << log_c) & ((1u << log_s) - 1);

  for (u32 e = tid; e < tile_elems; e += 256) {
    size_t j;
    if (CONTIG) {
      j = (size_t)chunk * tile_elems + e;
    } else {
      u32 row = e >> log_c, cc = e & (C - 1);
      j = ((size_t)o_strided << (r + log_s)) | ((size_t)row << log_s) | (ci0 + cc);
    }
    tile[e] = x[j];
  }
  __syncthreads();

  const u32 n_bfly = tile_elems >> 1;
  for (u32 step = 0; step < r; ++step) {
    const u32 ss = INVERSE ? (r - 1 - step) : step;  // stage offset inside the pass
    const u32 s = s0 + ss;
    const u32 log_half = r - 1 - ss, half = 1u << log_half;
    for (u32 b = tid; b < n_bfly; b += 256) {
      u32 col, pr, o;
      if (CONTIG) {
        pr = b & ((R >> 1) - 1);
        col = b >> (r - 1);
        o = chunk * C + col;
      } else {
        col = b & (C - 1);
        pr = b >> log_c;
        o = o_strided;
      }
      const u32 grp = pr >> log_half, k = pr & (half - 1);
      const u32 row_lo = (grp << (log_half + 1)) | k;
      u32 i_lo, i_hi;
      if (CONTIG) {
        i_lo = col * R + row_lo;
        i_hi = i_lo + half;
      } else {
        i_lo = row_lo * C + col;
        i_hi = i_lo + half * C;
      }
      const u32 tw = (1u << s) + (o << ss) + grp;
      const u64 u = tile[i_lo], v = tile[i_hi];
      const ulong2 w = W[tw];
      if (!INVERSE) {
        const u64 wv = mul_shoup(v, w.x, w.y, q);
        tile[i_lo] = add_mod(u, wv, q);
        tile[i_hi] = sub_mod(u, wv, q);
      } else if (s != 0) {
        tile[i_lo] = add_mod(u, v, q);
        tile[i_hi] = mul_shoup(sub_mod(u, v, q), w.x, w.y, q);
      } else {  // last inverse stage: fold N^-1 (reference folds it into its first stage, ntt.c:282-317)
        tile[i_lo] = mul_shoup(add_mod(u, v, q), P.n_inv, P.n_inv_prec, q);
        tile[i_hi] = mul_shoup(sub_mod(u, v, q), P.inv_w1_ninv, P.inv_w1_ninv_prec, q);
      }
    }
    __syncthreads();
  }

  for (u32 e = tid; e < tile_elems; e += 256) {
    size_t j;
    if (CONTIG) {
      j = (size_t)chunk * tile_elems + e;
    } else {
      u32 row = e >> log_c, cc = e & (C - 1);
      j = ((size_t)o_strided << (r + log_s)) | ((size_t)row << log_s) | (ci0 + cc);
    }
    x[j] = tile[e];
  }
}

template <bool CONTIG, bool INVERSE>
static void launch_pass(const DevCtx& c, u64* poly, size_t poly_stride, u32 n_polys, u32 level, u32 pos0, u32 pos_off, u32 n_limbs, u32 s0, u32 r,
                        u32 log_c, hipStream_t s, u32 skip_alpha) {
  ACEHIP_ABLATE(ABL_NTT);
  const u32 tiles = c.N >> (r + log_c);
  dim3 grid(tiles, n_limbs, n_polys * c.nrep), block(256);
  size_t lds = sizeof(u64) << (r + log_c);
  hipLaunchKernelGGL((ntt_pass_kernel<CONTIG, INVERSE>), grid, block, lds, s, c, poly, poly_stride, level, pos0, pos_off, s0, r, log_c, skip_alpha);
}

void launch_ntt(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                u32 n_polys, size_t poly_stride, u32 skip_alpha) {
  if (n_limbs == 0) return;
  const u32 logN = c.logN;
  if (logN != 16) ntt_count((u64)n_limbs * n_polys * c.nrep);  // (N = 2^16: counted by launch_ntt_fused)
  if (logN <= 12) {  // whole limb in LDS (<= 32 KiB): one pass
    if (!inverse) launch_pass<true, false>(c, poly, poly_stride, n_polys, level, pos0, pos_off, n_limbs, 0, logN, 0, s, skip_alpha);
    else          launch_pass<true, true>(c, poly, poly_stride, n_polys, level, pos0, pos_off, n_limbs, 0, logN, 0, s, skip_alpha);
    return;
  }
  if (logN == 16) {  // two register-tiled passes of 8 stages (ntt_fast.hip)
    launch_ntt_fast(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha);
    return;
  }
  // leading logN-8 stages: LDS radix-2 pass on column tiles; trailing 8 stages: register-tiled contig pass
  const u32 r1 = logN - 8, log_c = 4;
  if (!inverse) {
    launch_pass<false, false>(c, poly, poly_stride, n_polys, level, pos0, pos_off, n_limbs, 0, r1, log_c, s, skip_alpha);
    launch_ntt_contig8(c, poly, level, pos0, n_limbs, false, s, pos_off, n_polys, poly_stride, skip_alpha);
  } else {
    launch_ntt_contig8(c, poly, level, pos0, n_limbs, true, s, pos_off, n_polys, poly_stride, skip_alpha);
    launch_pass<false, true>(c, poly, poly_stride, n_polys, level, pos0, pos_off, n_limbs, 0, r1, log_c, s, skip_alpha);
  }
}

// ------------------------------------------------------------------------------------------------
// limb-wise elementwise ops (Hw_modadd / Hw_modmul poly_arith.c:14-39, Multiply_add polynomial.c:148)
// grid: (ceil(N/512), n_limbs); each lane handles 2 coefficients (16-byte accesses)
// ------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(DevCtx c, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 pos_off) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  r = reb(c, r, c.rep0 + blockIdx.z);
  a = reb(c, a, c.rep0 + blockIdx.z);
  b = reb(c, b, c.rep0 + blockIdx.z);
  const DevPrime P = c.primes[limb_prime(pos, level, c.L)];
  const size_t base = (size_t)(pos - pos_off) * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const ulong2 va = *reinterpret_cast<const ulong2*>(a + base + i);
  const ulong2 vb = *reinterpret_cast<const ulong2*>(b + base + i);
  ulong2 vr;
  if (OP == 0) {
    vr.x = add_mod(va.x, vb.x, P.q);
    vr.y = add_mod(va.y, vb.y, P.q);
  } else if (OP == 1) {
    vr.x = sub_mod(va.x, vb.x, P.q);
    vr.y = sub_mod(va.y, vb.y, P.q);
  } else if (OP == 2) {
    vr.x = mul_mod(va.x, vb.x, P);
    vr.y = mul_mod(va.y, vb.y, P);
  } else {
    const ulong2 acc = *reinterpret_cast<const ulong2*>(r + base + i);
    vr.x = add_mod(acc.x, mul_mod(va.x, vb.x, P), P.q);
    vr.y = add_mod(acc.y, mul_mod(va.y, vb.y, P), P.q);
  }
  *reinterpret_cast<ulong2*>(r + base + i) = vr;
}

void launch_ew(const DevCtx& c, EwOp op, u64* r, const u64* a, const u64* b, u32 level, u32 pos0, u32 n_limbs,
               hipStream_t s, u32 pos_off) {
  ACEHIP_ABLATE(ABL_EW);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, c.nrep), block(256);
  switch (op) {
    case EwOp::Add: hipLaunchKernelGGL(ew_kernel<0>, grid, block, 0, s, c, r, a, b, level, pos0, pos_off); break;
    case EwOp::Sub: hipLaunchKernelGGL(ew_kernel<1>, grid, block, 0, s, c, r, a, b, level, pos0, pos_off); break;
    case EwOp::Mul: hipLaunchKernelGGL(ew_kernel<2>, grid, block, 0, s, c, r, a, b, level, pos0, pos_off); break;
    case EwOp::MulAdd: hipLaunchKernelGGL(ew_kernel<3>, grid, block, 0, s, c, r, a, b, level, pos0, pos_off); break;
  }
}

// Hw_rotate (poly_arith.c:41-56) with an NTT-domain table: pure gather r[j] = a[perm[j]]
__global__ __launch_bounds__(256) void rotate_kernel(DevCtx c, u64* r, const u64* a, const u32* __restrict__ perm, u32 level, u32 pos0) {
  const u32 N = c.N;
  if (!owns(c, limb_prime(pos0 + blockIdx.y, level, c.L))) return;
  r = reb(c, r, c.rep0 + blockIdx.z);
  a = reb(c, a, c.rep0 + blockIdx.z);
  const size_t base = (size_t)(pos0 + blockIdx.y) * N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= N) return;
  const uint2 p = *reinterpret_cast<const uint2*>(perm + i);
  ulong2 v;
  v.x = a[base + p.x];
  v.y = a[base + p.y];
  *reinterpret_cast<ulong2*>(r + base + i) = v;
}

void launch_rotate(const DevCtx& c, u64* r, const u64* a, const u32* perm, u32 level, u32 pos0, u32 n_limbs, hipStream_t s) {
  ACEHIP_ABLATE(ABL_ROTATE);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(rotate_kernel, grid, block, 0, s, c, r, a, perm, level, pos0);
}

// r_z[pos][j] = acc_z[pos][j] + a_z[pos][perm_k(j)] for one or two polynomials (blockIdx.z): the accumulation of a rotated
// ciphertext (Rotate_iteration's outer sums, ckks_bootstrap_context.c:1343-1377: Automorphism_transform, then Add_poly)
// in one pass; the index map of the automorphism X -> X^k is computed (automorphism_order_ntt), not loaded
__global__ __launch_bounds__(256) void rotate_add2_kernel(DevCtx c, u64* __restrict__ r0, u64* __restrict__ r1,
                                                          const u64* acc0, const u64* acc1, const u64* __restrict__ a0,
                                                          const u64* __restrict__ a1, u32 auto_k, u32 level, u32 pos0) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  const RepZ rz = rep_of_z(c);  // blockIdx.z = polynomial + (1 or 2) * replica
  const u64 q = c.primes[limb_prime(pos, level, c.L)].q;
  const size_t base = (size_t)pos * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u32 sh = __builtin_clz(c.N) + 1;  // 32 - log2(N)
  const u32 b0 = __brev(i) >> sh, b1 = __brev(i + 1) >> sh;
  const u32 px = __brev((((2 * b0 + 1) * auto_k) & (2 * c.N - 1)) >> 1) >> sh;
  const u32 py = __brev((((2 * b1 + 1) * auto_k) & (2 * c.N - 1)) >> 1) >> sh;
  const u64* a = reb(c, rz.z ? a1 : a0, rz.rep);
  const u64* acc = reb(c, rz.z ? acc1 : acc0, rz.rep);
  u64* r = reb(c, rz.z ? r1 : r0, rz.rep);
  ulong2 v = *reinterpret_cast<const ulong2*>(acc + base + i);
  v.x = add_mod(v.x, a[base + px], q);
  v.y = add_mod(v.y, a[base + py], q);
  *reinterpret_cast<ulong2*>(r + base + i) = v;
}

void launch_rotate_add2(const DevCtx& c, u64* r0, u64* r1, const u64* acc0, const u64* acc1, const u64* a0, const u64* a1, u32 auto_k,
                        u32 level, u32 pos0, u32 n_limbs, hipStream_t s) {
  ACEHIP_ABLATE(ABL_ROTATE);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, (r1 ? 2 : 1) * c.nrep), block(256);
  hipLaunchKernelGGL(rotate_add2_kernel, grid, block, 0, s, c, r0, r1, acc0, acc1, a0, a1, auto_k, level, pos0);
}

// r[l][n] = a[l][n] * w[l] mod prime(gi[l])   (Shoup; per-limb constants in HBM)
__global__ __launch_bounds__(256) void mul_const_kernel(DevCtx c, u64* r, const u64* a, const u64* __restrict__ w,
                                                        const u64* __restrict__ wp, const u32* __restrict__ gi) {
  const u32 l = blockIdx.y;
  if (!owns(c, gi[l])) return;
  r = reb(c, r, c.rep0 + blockIdx.z);
  a = reb(c, a, c.rep0 + blockIdx.z);
  const u64 q = c.primes[gi[l]].q, wl = w[l], wpl = wp[l];
  const size_t base = (size_t)l * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 v = *reinterpret_cast<const ulong2*>(a + base + i);
  v.x = mul_shoup(v.x, wl, wpl, q);
  v.y = mul_shoup(v.y, wl, wpl, q);
  *reinterpret_cast<ulong2*>(r + base + i) = v;
}

void launch_mul_const(const DevCtx& c, u64* r, const u64* a, const u64* w, const u64* wp, const u32* gi, u32 n_limbs,
                      hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(mul_const_kernel, grid, block, 0, s, c, r, a, w, wp, gi);
}

// ------------------------------------------------------------------------------------------------
// RNS base conversion (Fast_base_conv polynomial.c:755-807, Decompose_modup :1297-1320):
//   out[pos_j][n] = ( sum_i in[i][n] * hat[i][j] ) mod t_j,  exact 128-bit sum then one reduction.
// One lane per coefficient; the n_in source residues stay in registers and are reused for the
// JG output limbs of this workgroup (blockIdx.y selects the group).
// ------------------------------------------------------------------------------------------------
constexpr int kConvChunk = 16;  // source limbs held in registers at a time (alpha, K <= 12 for the reference's sets)
constexpr int kConvGroup = 4;

// n_in <= 64: 64 products below 2^122 cannot overflow the 128-bit sums.  Sources are taken in chunks of 16, the output
// sums of the group stay in registers across chunks (the usual case is a single chunk: the loop runs once).
__global__ __launch_bounds__(256) void base_conv_kernel(DevCtx c, u64* out, const u64* in, const u64* __restrict__ hat,
                                                        const u32* __restrict__ out_gi, const u32* __restrict__ out_pos, u32 n_in,
                                                        u32 n_out, u32 hat_ld) {
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  if (n >= c.N) return;
  out = reb(c, out, c.rep0 + blockIdx.z);
  in = reb(c, in, c.rep0 + blockIdx.z);
  const u32 j0 = blockIdx.y * kConvGroup;
  U128 acc[kConvGroup];
#pragma unroll
  for (int g = 0; g < kConvGroup; ++g) acc[g] = U128{0, 0};
  for (u32 i0 = 0; i0 < n_in; i0 += kConvChunk) {
    u64 y[kConvChunk];
#pragma unroll
    for (int i = 0; i < kConvChunk; ++i) y[i] = i0 + i < n_in ? in[(size_t)(i0 + i) * c.N + n] : 0;
#pragma unroll
    for (int g = 0; g < kConvGroup; ++g) {
      const u32 j = j0 + g;
      if (j < n_out) {
#pragma unroll
        for (int i = 0; i < kConvChunk; ++i)
          if (i0 + i < n_in) mac128(acc[g], y[i], hat[(size_t)(i0 + i) * hat_ld + j]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < kConvGroup; ++g) {
    const u32 j = j0 + g;
    if (j < n_out && owns(c, out_gi[j])) {
      const DevPrime P = c.primes[out_gi[j]];
      out[(size_t)out_pos[j] * c.N + n] = reduce128(acc[g], P.q, P.prec128_lo, P.prec128_hi);
    }
  }
}

void launch_base_conv(const DevCtx& c, u64* out, const u64* in, const u64* hat, const u32* out_gi, const u32* out_pos,
                      u32 n_in, u32 n_out, u32 hat_ld, hipStream_t s) {
  ACEHIP_ABLATE(ABL_CONV);
  if (n_out == 0) return;
  dim3 grid((c.N + 255) / 256, (n_out + kConvGroup - 1) / kConvGroup, c.nrep), block(256);
  hipLaunchKernelGGL(base_conv_kernel, grid, block, 0, s, c, out, in, hat, out_gi, out_pos, n_in, n_out, hat_ld);
}

// ModDown tail (Reduce_rns_base polynomial.c:956-965): out = (x - out) * P^-1 mod q_i
__global__ __launch_bounds__(256) void moddown_tail_kernel(DevCtx c, u64* out, const u64* x, const u64* __restrict__ pinv,
                                                           const u64* __restrict__ pinv_prec) {
  const u32 l = blockIdx.y;
  if (!owns(c, l)) return;
  out = reb(c, out, c.rep0 + blockIdx.z);
  x = reb(c, x, c.rep0 + blockIdx.z);
  const u64 q = c.primes[l].q, w = pinv[l], wp = pinv_prec[l];
  const size_t base = (size_t)l * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const ulong2 vx = *reinterpret_cast<const ulong2*>(x + base + i);
  ulong2 vo = *reinterpret_cast<const ulong2*>(out + base + i);
  vo.x = mul_shoup(sub_mod(vx.x, vo.x, q), w, wp, q);
  vo.y = mul_shoup(sub_mod(vx.y, vo.y, q), w, wp, q);
  *reinterpret_cast<ulong2*>(out + base + i) = vo;
}

void launch_moddown_tail(const DevCtx& c, u64* out, const u64* x, const u64* pinv, const u64* pinv_prec, u32 level,
                         hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  dim3 grid((c.N / 2 + 255) / 256, level, c.nrep), block(256);
  hipLaunchKernelGGL(moddown_tail_kernel, grid, block, 0, s, c, out, x, pinv, pinv_prec);
}

// Rescale (Rescale_poly polynomial.c:1132-1144): spread the iNTT'd last limb to every remaining limb.
// blockIdx.z = polynomial (c0 / c1 of a ciphertext are rescaled together)
__global__ __launch_bounds__(256) void rescale_spread_kernel(DevCtx c, u64* t, size_t t_stride, const u64* last, size_t last_stride,
                                                             const u64* __restrict__ c1, const u64* __restrict__ c1p,
                                                             u32 level) {
  const u32 l = blockIdx.y;
  if (!owns(c, l)) return;
  const RepZ rz = rep_of_z(c);  // blockIdx.z = polynomial + n_polys * replica
  const u64 q = c.primes[l].q, ql = c.primes[level - 1].q, w = c1[l], wp = c1p[l];
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 v = *reinterpret_cast<const ulong2*>(reb(c, last, rz.rep) + rz.z * last_stride + i);
  v.x = mul_shoup(switch_modulus(v.x, ql, q), w, wp, q);
  v.y = mul_shoup(switch_modulus(v.y, ql, q), w, wp, q);
  *reinterpret_cast<ulong2*>(reb(c, t, rz.rep) + rz.z * t_stride + (size_t)l * c.N + i) = v;
}

void launch_rescale_spread(const DevCtx& c, u64* t, size_t t_stride, const u64* last, size_t last_stride, const u64* c1,
                           const u64* c1p, u32 level, u32 n_polys, hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  dim3 grid((c.N / 2 + 255) / 256, level - 1, n_polys * c.nrep), block(256);
  hipLaunchKernelGGL(rescale_spread_kernel, grid, block, 0, s, c, t, t_stride, last, last_stride, c1, c1p, level);
}

// Rescale tail (polynomial.c:1145-1158): out = x * q_l^-1 + t
__global__ __launch_bounds__(256) void rescale_tail_kernel(DevCtx c, u64* __restrict__ out0, u64* __restrict__ out1,
                                                           const u64* __restrict__ x0, const u64* __restrict__ x1,
                                                           const u64* t, size_t t_stride,
                                                           const u64* __restrict__ inv, const u64* __restrict__ invp) {
  const u32 l = blockIdx.y;
  if (!owns(c, l)) return;
  const RepZ rz = rep_of_z(c);
  const u64 q = c.primes[l].q, w = inv[l], wp = invp[l];
  const size_t base = (size_t)l * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const u64* x = reb(c, rz.z ? x1 : x0, rz.rep);
  u64* out = reb(c, rz.z ? out1 : out0, rz.rep);
  const ulong2 vx = *reinterpret_cast<const ulong2*>(x + base + i);
  const ulong2 vt = *reinterpret_cast<const ulong2*>(reb(c, t, rz.rep) + rz.z * t_stride + base + i);
  ulong2 vo;
  vo.x = add_mod(mul_shoup(vx.x, w, wp, q), vt.x, q);
  vo.y = add_mod(mul_shoup(vx.y, w, wp, q), vt.y, q);
  *reinterpret_cast<ulong2*>(out + base + i) = vo;
}

void launch_rescale_tail(const DevCtx& c, u64* out0, u64* out1, const u64* x0, const u64* x1, const u64* t, size_t t_stride,
                         const u64* inv, const u64* invp, u32 level, u32 n_polys, hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  dim3 grid((c.N / 2 + 255) / 256, level - 1, n_polys * c.nrep), block(256);
  hipLaunchKernelGGL(rescale_tail_kernel, grid, block, 0, s, c, out0, out1, x0, x1, t, t_stride, inv, invp);
}

// key inner product for one digit (generated code inc:7011-7036 == Multiply_add polynomial.c:148-183)
template <bool ACC>
__global__ __launch_bounds__(256) void key_mac_kernel(DevCtx c, u64* acc0, u64* acc1, const u64* key0, const u64* key1, const u64* ext,
                                                      u32 level) {
  const u32 pos = blockIdx.y;
  const u32 gi = limb_prime(pos, level, c.L);
  if (!owns(c, gi)) return;
  const u32 rep = c.rep0 + blockIdx.z;
  acc0 = reb(c, acc0, rep);
  acc1 = reb(c, acc1, rep);
  key0 = reb(c, key0, rep);
  key1 = reb(c, key1, rep);
  ext = reb(c, ext, rep);
  const DevPrime P = c.primes[gi];
  const size_t pb = (size_t)pos * c.N, kb = (size_t)gi * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  const ulong2 e = *reinterpret_cast<const ulong2*>(ext + pb + i);
  const ulong2 k0 = *reinterpret_cast<const ulong2*>(key0 + kb + i);
  const ulong2 k1 = *reinterpret_cast<const ulong2*>(key1 + kb + i);
  ulong2 r0, r1;
  r0.x = mul_mod(k0.x, e.x, P);
  r0.y = mul_mod(k0.y, e.y, P);
  r1.x = mul_mod(k1.x, e.x, P);
  r1.y = mul_mod(k1.y, e.y, P);
  if (ACC) {
    const ulong2 a0 = *reinterpret_cast<const ulong2*>(acc0 + pb + i);
    const ulong2 a1 = *reinterpret_cast<const ulong2*>(acc1 + pb + i);
    r0.x = add_mod(r0.x, a0.x, P.q);
    r0.y = add_mod(r0.y, a0.y, P.q);
    r1.x = add_mod(r1.x, a1.x, P.q);
    r1.y = add_mod(r1.y, a1.y, P.q);
  }
  *reinterpret_cast<ulong2*>(acc0 + pb + i) = r0;
  *reinterpret_cast<ulong2*>(acc1 + pb + i) = r1;
}

void launch_key_mac(const DevCtx& c, u64* acc0, u64* acc1, const u64* key0, const u64* key1, const u64* ext, u32 level,
                    bool accumulate, hipStream_t s) {
  ACEHIP_ABLATE(ABL_KEYMAC);
  dim3 grid((c.N / 2 + 255) / 256, level + c.K, c.nrep), block(256);
  if (accumulate) hipLaunchKernelGGL(key_mac_kernel<true>, grid, block, 0, s, c, acc0, acc1, key0, key1, ext, level);
  else            hipLaunchKernelGGL(key_mac_kernel<false>, grid, block, 0, s, c, acc0, acc1, key0, key1, ext, level);
}

}  // namespace acehip
