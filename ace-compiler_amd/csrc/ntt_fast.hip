// ntt_fast.hip -- register-tiled negacyclic NTT passes for gfx950.
//
// One pass = 8 radix-2 stages = two rounds of radix-16 held in registers (16 coefficients per lane,
// 4 stages per round) with one LDS transpose in between; 256 lanes x 16 = 4096 coefficients (32 KiB)
// per workgroup.  N = 2^16 is exactly two passes:
//   STRIDED pass: stages [0,8)          rows = index bits [logN-8, logN), 16 adjacent columns/tile
//   CONTIG  pass: stages [logN-8,logN)  16 consecutive 256-coefficient blocks per tile
// The kernels are bound by VALU issue: on gfx950 every VOP3 / 64-bit instruction (v_mad_u64_u32, v_lshl_add_u64, v_mul_lo_u32,
// v_cmp_*_u64 ...) costs the same ~4 cycles per wave and a plain VOP1/VOP2 ~2 (tools/ubench_ops.hip), so the butterflies are
// built to need as few INSTRUCTIONS as possible, not only as few multiplies:
//   * twiddle products use a "sloppy" Shoup reduction: the quotient estimate takes three of the four partial products of
//     a * companion and every partial product is a v_mad_u64_u32 whose 64-bit accumulate is free -- 9 multiply-adds per product;
//   * SMALL primes (81q < 2^63: every q-limb of the 50..56-bit scaling primes): values stay below 2^63 and the companion is
//     used as floor(w*2^63/q), so the two cross products of the quotient add without carry in ONE accumulation chain (3
//     instructions + a move); the product comes out in [0,5q).  Forward transforms never reduce the butterfly sums: the low
//     product chain starts from x, so X' = x + m costs no instruction of its own, Y' = (2x + 4q) - X' one shift-add and one
//     64-bit subtraction (15 instructions per butterfly); values grow by 5q per stage, stay below 81q over the 16 stages and
//     are reduced once, at the very end (quotient from floor(2^64/q));
//   * larger primes (q0, the P primes, < 2^61) use a 64-bit companion (shoup4, result in [0,4q)) and keep values in [0,8q) with
//     one conditional subtraction of 4q per forward butterfly;
//   * inverse transforms keep values in [0,5q) (SMALL) or [0,4q) with one conditional subtraction per butterfly.
// The last pass writes canonical residues in [0,q), so results are bit-identical to the reference (ntt.c:190-353),
// which keeps every intermediate canonical.  Twiddles come from the interleaved {w, floor(w*2^64/q)} tables (16-byte
// loads); workgroups that share twiddles are placed on one XCD (kernels.hpp ntt_block).
#include <cstdlib>

#include "device_arith.hpp"
#include "kernels.hpp"
#include "ntt_fp.hpp"

#ifndef ACEHIP_NTT_MIN_WG
#define ACEHIP_NTT_MIN_WG 4  // workgroups per CU the wide passes' register allocation leaves room for (experiments: tools/kernel_ab.sh)
#endif

namespace acehip {

struct Tw {
  u64 w, p;
};

#ifndef NTT_EXP
#define NTT_EXP 0  // timing experiments only (results are wrong): 1 no butterflies, 2 no data loads/stores, 4 no twiddle loads,
                   // 8 no strided pass at all (what a one-pass transform would save), 16 the forward contiguous pass does not load its
                   // input (what reading the first pass' output out of the XCD's L2 could save at most), 32 every per-lane twiddle
                   // load hits the same few KiB (what the twiddle stream costs beyond its instructions), 64 the same for the key parts of the
                   // key inner product formed inside a pass (kmac_tile)
#endif
// SMALL: the companion is used as floor(w*2^63/q) = floor(w*2^64/q) >> 1 (see shoup5_add)
template <bool SMALL>
__device__ __forceinline__ Tw ldtw(const ulong2* __restrict__ t, u32 idx) {
#if NTT_EXP & 4
  return Tw{(u64)idx * 0x9E3779B97F4A7C15ull, ((u64)idx * 0xC2B2AE3D27D4EB4Full + threadIdx.x) >> 1};
#else
#if NTT_EXP & 32
  const ulong2 v = t[idx & 255u];  // timing experiment (results are wrong): the twiddle stream always hits the same 4 KiB
#else
  const ulong2 v = t[idx];
#endif
  return Tw{v.x, SMALL ? v.y >> 1 : v.y};
#endif
}

// a*w mod q in [0,2q) for any 64-bit a (exact Shoup quotient: at most 1 below the true quotient)
__device__ __forceinline__ u64 shoup_lazy(u64 a, Tw t, u64 q) { return a * t.w - mulhi64(a, t.p) * q; }

#define ACEHIP_PIN(x) asm("" : "+v"(x))
// {r.lo, r.hi + c.lo}: the low product chain r and the cross-term chain c of a 64-bit product joined with ONE 32-bit add
// (inline asm keeps the compiler from rewriting it as a 64-bit add of a shifted value: two moves and a v_lshl_add_u64)
__device__ __forceinline__ u64 join_hi_add(u64 r, u64 c) {
  u32 rhi;
  asm("v_add_u32 %0, %1, %2" : "=v"(rhi) : "v"((u32)(r >> 32)), "v"((u32)c));
  return ((u64)rhi << 32) | (u32)r;
}

// addend + (a*w - h*q) mod 2^64 with a*w - h*q in [0,4q), for any 64-bit a: quotient estimate
// h = a1*p1 + hi32(a0*p1) + hi32(a1*p0), which is floor(a*wp/2^64) less at most 2 (the dropped a0*p0 and the two truncated
// cross terms), itself at most 1 below floor(a*w/q).
// Every partial product is written as a full 32x32->64 product and pinned by an empty asm, so that the compiler emits
// v_mad_u64_u32 (with its free 64-bit accumulate) for it instead of narrowing to v_mul_lo_u32 / v_mul_hi_u32: 9
// v_mad_u64_u32 per product and no separate 64-bit subtraction (the result is a*w + h*(2^64-q)).
__device__ __forceinline__ u64 shoup4_add(u64 a, Tw t, u64 q, u64 addend) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)t.p, p1 = (u32)(t.p >> 32);
  const u32 w0 = (u32)t.w, w1 = (u32)(t.w >> 32);
  const u64 nq = 0 - q;
  const u32 n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 U = (u64)a0 * p1;
  ACEHIP_PIN(U);
  u64 V = (u64)a1 * p0;
  ACEHIP_PIN(V);
  const u64 h = (u64)a1 * p1 + (U >> 32) + (V >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 c = (u64)a0 * w1;
  c += (u64)a1 * w0;
  c += (u64)h0 * n1;
  c += (u64)h1 * n0;
  ACEHIP_PIN(c);
  u64 r = (u64)a0 * w0 + addend;
  r += (u64)h0 * n0;
  ACEHIP_PIN(r);
  return join_hi_add(r, c);
}

// SMALL primes: addend + (a*w - 2h*q) mod 2^64 for a < 2^63 and t.p = floor(w*2^63/q) < 2^63 (ldtw<true>), nq2 = -2q.
// The cross products a0*p1 + a1*p0 cannot carry out of 64 bits (a1, p1 < 2^31), so the quotient estimate is one chain:
// h = a1*p1 + hi32(a0*p1 + a1*p0) >= floor(a*p/2^64) - 1, 2h > a*w/q - 5, and the product a*w - 2h*q lies in [0,5q).
__device__ __forceinline__ u64 shoup5_add(u64 a, Tw t, u64 nq2, u64 addend) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)t.p, p1 = (u32)(t.p >> 32);
  const u32 w0 = (u32)t.w, w1 = (u32)(t.w >> 32), m0 = (u32)nq2, m1 = (u32)(nq2 >> 32);
  u64 M = (u64)a0 * p1;
  M += (u64)a1 * p0;
  ACEHIP_PIN(M);
  const u64 h = (u64)a1 * p1 + (M >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 c = (u64)a0 * w1;
  c += (u64)a1 * w0;
  c += (u64)h0 * m1;
  c += (u64)h1 * m0;
  ACEHIP_PIN(c);
  u64 r = (u64)a0 * w0 + addend;
  r += (u64)h0 * m0;
  ACEHIP_PIN(r);
  return join_hi_add(r, c);
}

// SMALL primes: 16 forward stages without any reduction grow a canonical input to q + 16*5q = 81q, which must stay below
// 2^63 (shoup5_add); the inverse keeps values below 10q
constexpr u64 kSmallPrimeMax = (~0ull >> 1) / 81;

// per-prime constants of the butterflies: SMALL: lim = 5q (inverse range), nq = -2q; otherwise lim = 4q, nq unused
struct BfK {
  u64 q, q4, lim, nq;
};
// (pinned to SGPRs through readfirstlane: otherwise the compiler re-derives 5q inside every butterfly with a multiply-add
// and an add)
__device__ __forceinline__ u64 uniform64(u64 v) {
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
template <bool SMALL>
__device__ __forceinline__ BfK bf_consts(u64 q) {
  return BfK{q, uniform64(4 * q), uniform64(SMALL ? 5 * q : 4 * q), uniform64(SMALL ? 0 - 2 * q : 0 - q)};
}

// forward (Cooley-Tukey) lazy butterfly.  SMALL: X,Y < B -> X,Y < B + 5q, no reduction.
// !SMALL: X,Y in [0,8q) -> [0,8q) (q < 2^61)
template <bool SMALL>
__device__ __forceinline__ void bf_fwd(u64& X, u64& Y, Tw t, const BfK& k) {
#if NTT_EXP & 1
  X ^= t.w; Y += t.p;
  return;
#endif
  if (SMALL) {
    const u64 x = X;
    const u64 nx = shoup5_add(Y, t, k.nq, x);  // x + m: the low product chain starts from x
    X = nx;
    Y = (x << 1) + k.lim - nx;                 // x + 5q - m  (m < 5q), as (2x + 5q) - (x + m): one shift-add, one subtraction
  } else {
    const u64 x = X >= k.q4 ? X - k.q4 : X;
    const u64 nx = shoup4_add(Y, t, k.q, x);   // x + m, m < 4q
    X = nx;
    Y = (x << 1) + k.q4 - nx;
  }
}
// inverse (Gentleman-Sande) lazy butterfly: X,Y in [0,lim) -> [0,lim), lim = 5q (SMALL) or 4q
template <bool SMALL>
__device__ __forceinline__ void bf_inv(u64& X, u64& Y, Tw t, const BfK& k) {
  const u64 s = X + Y;
  const u64 d = X + k.lim - Y;
  X = s >= k.lim ? s - k.lim : s;
  Y = SMALL ? shoup5_add(d, t, k.nq, 0) : shoup4_add(d, t, k.q, 0);
}
// canonical residue of a forward result: v < 81q (SMALL; mu = floor(2^64/q)) or v < 8q
template <bool SMALL>
__device__ __forceinline__ u64 canon_fwd(u64 v, u64 q, u64 mu) {
  if (SMALL) {
    const u64 r = v - mulhi64(v, mu) * q;  // quotient at most 1 low: [0,2q)
    return r >= q ? r - q : r;
  }
  const u64 q4 = 4 * q, q2 = 2 * q;
  v = v >= q4 ? v - q4 : v;
  v = v >= q2 ? v - q2 : v;
  return v >= q ? v - q : v;
}

// 4 forward stages on 16 registers; stage u pairs (k, k + (8>>u)), twiddle T_u[k / (16>>u)]
template <bool SMALL>
__device__ __forceinline__ void radix16_fwd(u64 (&x)[16], const Tw& t0, const Tw (&t1)[2], const Tw (&t2)[4],
                                            const Tw (&t3)[8], const BfK& k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) bf_fwd<SMALL>(x[i], x[i + 8], t0, k);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) bf_fwd<SMALL>(x[8 * g + i], x[8 * g + i + 4], t1[g], k);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 2; ++i) bf_fwd<SMALL>(x[4 * g + i], x[4 * g + i + 2], t2[g], k);
#pragma unroll
  for (int g = 0; g < 8; ++g) bf_fwd<SMALL>(x[2 * g], x[2 * g + 1], t3[g], k);
}

// 4 inverse stages (u = 3..1); stage u = 0 is handled by the caller (it may carry the N^-1 fold)
template <bool SMALL>
__device__ __forceinline__ void radix16_inv_321(u64 (&x)[16], const Tw (&t1)[2], const Tw (&t2)[4], const Tw (&t3)[8],
                                                const BfK& k) {
#pragma unroll
  for (int g = 0; g < 8; ++g) bf_inv<SMALL>(x[2 * g], x[2 * g + 1], t3[g], k);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 2; ++i) bf_inv<SMALL>(x[4 * g + i], x[4 * g + i + 2], t2[g], k);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) bf_inv<SMALL>(x[8 * g + i], x[8 * g + i + 4], t1[g], k);
}
template <bool SMALL>
__device__ __forceinline__ void radix16_inv_0(u64 (&x)[16], const Tw& t0, const BfK& k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) bf_inv<SMALL>(x[i], x[i + 8], t0, k);
}

// twiddles of the 16-group with index `prefix` at stage `sbase`: T_u[i] = TW[2^(sbase+u) + (prefix<<u) + i]
template <bool SMALL>
__device__ __forceinline__ void load_tw(const ulong2* __restrict__ TW, u32 sbase, u32 prefix, Tw& t0, Tw (&t1)[2],
                                        Tw (&t2)[4], Tw (&t3)[8]) {
  t0 = ldtw<SMALL>(TW, (1u << sbase) + prefix);
#pragma unroll
  for (int i = 0; i < 2; ++i) t1[i] = ldtw<SMALL>(TW, (2u << sbase) + (prefix << 1) + i);
#pragma unroll
  for (int i = 0; i < 4; ++i) t2[i] = ldtw<SMALL>(TW, (4u << sbase) + (prefix << 2) + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) t3[i] = ldtw<SMALL>(TW, (8u << sbase) + (prefix << 3) + i);
}

// Companion-only twiddles (TW8: 8 bytes per entry instead of 16).  p = floor(w*2^64/q) determines w: p*q = w*2^64 - r with
// 0 < r < q, so w = mulhi64(p, q) + 1 exactly.  In the small launches of the workload the passes wait for memory (VALU issue in
// 30 % of the cycles, profiles/r02z) and the 16-byte twiddle stream of the last four stages is as large as the data of the
// pass; loading p alone halves it for 4 multiply-adds per twiddle.  The twiddle is derived right before its butterflies, so
// only the companions (30 registers instead of 60) stay live.  Used when few polynomials share a limb's twiddles.
struct Tp15 {
  u64 p0, p1[2], p2[4], p3[8];
};
__device__ __forceinline__ void load_tp(const u64* __restrict__ TP, u32 sbase, u32 prefix, Tp15& t) {
#if NTT_EXP & 32
#define ACEHIP_TPIDX(i) ((i) & 255u)
#else
#define ACEHIP_TPIDX(i) (i)
#endif
  t.p0 = TP[ACEHIP_TPIDX((1u << sbase) + prefix)];
#pragma unroll
  for (int i = 0; i < 2; ++i) t.p1[i] = TP[ACEHIP_TPIDX((2u << sbase) + (prefix << 1) + i)];
#pragma unroll
  for (int i = 0; i < 4; ++i) t.p2[i] = TP[ACEHIP_TPIDX((4u << sbase) + (prefix << 2) + i)];
#pragma unroll
  for (int i = 0; i < 8; ++i) t.p3[i] = TP[ACEHIP_TPIDX((8u << sbase) + (prefix << 3) + i)];
}
template <bool SMALL>
__device__ __forceinline__ Tw tw_of(u64 p, u64 q) {
  return Tw{mulhi64(p, q) + 1, SMALL ? p >> 1 : p};
}
template <bool SMALL>
__device__ __forceinline__ void radix16_fwd_p(u64 (&x)[16], const Tp15& t, const BfK& k) {
  {
    const Tw w = tw_of<SMALL>(t.p0, k.q);
#pragma unroll
    for (int i = 0; i < 8; ++i) bf_fwd<SMALL>(x[i], x[i + 8], w, k);
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const Tw w = tw_of<SMALL>(t.p1[g], k.q);
#pragma unroll
    for (int i = 0; i < 4; ++i) bf_fwd<SMALL>(x[8 * g + i], x[8 * g + i + 4], w, k);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const Tw w = tw_of<SMALL>(t.p2[g], k.q);
#pragma unroll
    for (int i = 0; i < 2; ++i) bf_fwd<SMALL>(x[4 * g + i], x[4 * g + i + 2], w, k);
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) bf_fwd<SMALL>(x[2 * g], x[2 * g + 1], tw_of<SMALL>(t.p3[g], k.q), k);
}
template <bool SMALL>
__device__ __forceinline__ void radix16_inv_p(u64 (&x)[16], const Tp15& t, const BfK& k) {  // stages 3..0
#pragma unroll
  for (int g = 0; g < 8; ++g) bf_inv<SMALL>(x[2 * g], x[2 * g + 1], tw_of<SMALL>(t.p3[g], k.q), k);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const Tw w = tw_of<SMALL>(t.p2[g], k.q);
#pragma unroll
    for (int i = 0; i < 2; ++i) bf_inv<SMALL>(x[4 * g + i], x[4 * g + i + 2], w, k);
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const Tw w = tw_of<SMALL>(t.p1[g], k.q);
#pragma unroll
    for (int i = 0; i < 4; ++i) bf_inv<SMALL>(x[8 * g + i], x[8 * g + i + 4], w, k);
  }
  {
    const Tw w = tw_of<SMALL>(t.p0, k.q);
#pragma unroll
    for (int i = 0; i < 8; ++i) bf_inv<SMALL>(x[i], x[i + 8], w, k);
  }
}

// The 15 twiddles of stages 0..3 (TW[1..15]) are the same for every lane: read through the constant address space they
// become scalar loads, live in SGPRs (60 of them instead of 60 VGPRs) and enter the multiply-adds as their one scalar operand.
// (The tables are written once at context creation, never by a kernel.)
typedef const __attribute__((address_space(4))) u64* ctw_ptr;
template <bool SMALL>
__device__ __forceinline__ void load_tw_uniform(const ulong2* __restrict__ TW, Tw& t0, Tw (&t1)[2], Tw (&t2)[4], Tw (&t3)[8]) {
#if NTT_EXP & 4
  load_tw<SMALL>(TW, 0, 0, t0, t1, t2, t3);
#else
  ctw_ptr T = (ctw_ptr)(reinterpret_cast<const u64*>(TW));
  auto ld = [&](u32 i) {
    const u64 w = T[2 * i], p = T[2 * i + 1];
    return Tw{w, SMALL ? p >> 1 : p};
  };
  t0 = ld(1);
#pragma unroll
  for (int i = 0; i < 2; ++i) t1[i] = ld(2 + i);
#pragma unroll
  for (int i = 0; i < 4; ++i) t2[i] = ld(4 + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) t3[i] = ld(8 + i);
#endif
}

// One limb seen through a buffer descriptor: addresses are descriptor (SGPRs) + one 32-bit per-lane byte offset + a scalar
// byte offset, so the 16 strided accesses of a lane share ONE address VGPR instead of 16 64-bit pairs (the difference
// between 4 and 2 waves per SIMD for the forward strided pass).  `base` must be wave-uniform.
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
struct LimbBuf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ LimbBuf limb_buf(const u64* base, u32 bytes) {
  const u64 a = reinterpret_cast<u64>(base);
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)a), hi = __builtin_amdgcn_readfirstlane((u32)(a >> 32));
  void* p = reinterpret_cast<void*>(((u64)hi << 32) | lo);
  return LimbBuf{__builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000)};
}
// cache-policy experiments (tools): NTT_NT & 1: polynomial data loads non-temporal, & 2: stores non-temporal
#ifndef NTT_NT
#define NTT_NT 3  // measured: roofline batch 0.573 -> 0.556 ms, headline unchanged
#endif
#define NTT_NT_LD ((NTT_NT & 1) ? 2 : 0)
#define NTT_NT_ST ((NTT_NT & 2) ? 2 : 0)
typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T ntld(const T* p) {
#if NTT_NT & 1
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <class T> __device__ __forceinline__ void ntst(T* p, T v) {
#if NTT_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ u64 bld(const LimbBuf& b, u32 voff, u32 soff) {
#if NTT_EXP & 2
  return (u64)voff * 0x9E3779B97F4A7C15ull + soff;
#endif
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, NTT_NT_LD);
  return ((u64)v.y << 32) | v.x;
}
__device__ __forceinline__ void bst(const LimbBuf& b, u32 voff, u32 soff, u64 v) {
#if NTT_EXP & 2
  if (v != 0x123456789ull) return;
#endif
  u32x2_t w;
  w.x = (u32)v;
  w.y = (u32)(v >> 32);
  __builtin_amdgcn_raw_buffer_store_b64(w, b.r, voff, soff, NTT_NT_ST);
}

constexpr u32 kRowPitch = 17;        // strided tile: 256 rows x 16 cols, row pitch 17 words
constexpr u32 kBlkPitch = 272;       // contig tile: 16 blocks x (256 + 16 pad) words
__device__ __forceinline__ u32 cpad(u32 rho) { return rho + (rho >> 4); }

// ------------------------------------------------------------------------------------------------
// STRIDED pass (stages 0..7): tile = rows rho in [0,256) (coefficient index bits [logN-8,logN)),
// columns col = chunk*16 + cc.  Round A lanes (g = tid>>4, cc = tid&15) hold rows 16k+g;
// round B lanes (h = tid>>4, cc) hold rows 16h+g'.
// ------------------------------------------------------------------------------------------------
struct StridedArgs {
  LimbBuf buf;
  const ulong2* __restrict__ TW;
  const double* __restrict__ TWD;  // the limb's twiddles as doubles (FP class, ntt_fp.hpp)
  u64* lds;
  u32 cc, hg, col;
  u64 q;
  static constexpr u32 log_s = 8;
};

// forward: rows 16k+hg are read in place unless FROM_MSG filled x[]; the result (lazy: < 41q SMALL, < 8q otherwise) is
// stored in place.  The two prime classes are separate code regions from the first load to the last store (one
// scalar branch at the top): sharing the loads lets the compiler hoist both paths' twiddles above the branch, which
// costs half of the occupancy.
// SRC: where x[] comes from: SRC_MEM the limb itself (in place), SRC_MSG the signed message f.msg reduced mod the limb's prime
// (encode), SRC_CONV4/8/12 the fast base conversion of up to 4/8/12 coefficient-domain source limbs (ModUp / ModDown: the
// converted limbs are never written in coefficient form; 16 sources would spill registers: those take the separate kernel)
enum : int { SRC_MEM = 0, SRC_MSG = 1, SRC_CONV4 = 4, SRC_CONV8 = 8, SRC_CONV12 = 12 };
// canonical input of the forward strided pass, rows 16k+hg of the workgroup's tile (SRC: see above)
template <int SRC>
__device__ __forceinline__ void strided_fwd_source(const DevCtx& c, const StridedArgs& a, const DevPrime& P, const NttFuse& f, u32 pos,
                                                   u32 row, u32 z, u32 rep, u32 n_bytes, u32 split_bits, u64 (&x)[16]) {
  constexpr bool FROM_MSG = SRC == SRC_MSG;
  const u64 q = a.q;
  if (FROM_MSG) {  // the signed message, reduced mod this limb's prime (and scaled): Encode_impl ckks_encoder.c:262-285
    const u64 sc = f.msg_scale ? f.msg_scale[pos] : 0;
    const LimbBuf mbuf = limb_buf(reinterpret_cast<const u64*>(reb(c, f.msg, rep) + z * f.msg_stride), n_bytes);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t v = (int64_t)bld(mbuf, (a.hg << 11) + a.col * 8, (u32)k << 15);
      const u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
      u64 r = mag;
      if (__any(mag >= q))  // encoded weights are far below the primes: the wide reduction is the rare path, skip it per wave
        r = mag < q ? mag : reduce128(U128{mag, 0}, q, P.prec128_lo, P.prec128_hi);
      if (v < 0 && r != 0) r = q - r;
      x[k] = f.msg_scale ? mul_mod(r, sc, P) : r;
    }
  } else if (SRC >= SRC_CONV4) {
    // x[n] = ( sum_i y_i[n] * hat[i][row] ) mod q: exact 128-bit sum, one reduction (Reduce_rns_base polynomial.c:928-967,
    // Decompose_modup :1302-1320); the pre-factors (Q_d/q_i)^-1 were folded into the inverse NTT that produced y.  Same
    // arithmetic as base_conv_batch16_kernel (keyswitch.hip): halves of split_bits <= 30 bits, four carry-free
    // multiply-add chains per term; the row's constants are wave-uniform (scalar registers).
    constexpr int NI = SRC >= SRC_CONV4 ? SRC : 1;  // (the branch is dead for the other sources)
    const ConvDesc& d = f.conv[z * f.conv_step];
    const u32 n_in = d.n_in, jcol = d.col ? d.col[row] : row;
    if (d.out_pos[row] != pos) __builtin_trap();  // the launch must enumerate the limbs like the descriptor does
    const u32 h = split_bits, mask = (1u << h) - 1u;
    const LimbBuf sbuf = limb_buf(reb(c, f.conv_src, rep) + z * f.conv_src_stride, (d.src_pos0 + n_in) * n_bytes);
    u32 b0[NI], b1[NI], soff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const u32 ii = min((u32)i, n_in - 1);  // rows past n_in: reload the last source, multiply by zero
      const u64 b = (u32)i < n_in ? d.hat[(size_t)ii * d.hat_ld + jcol] : 0;
      b0[i] = (u32)b & mask;
      b1[i] = (u32)(b >> h);
      soff[i] = (d.src_pos0 + ii) * n_bytes;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      u64 s00 = 0, s01 = 0, s10 = 0, s11 = 0;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const u64 v = bld(sbuf, (a.hg << 11) + a.col * 8, ((u32)k << 15) + soff[i]);
        const u32 a0 = (u32)v & mask, a1 = (u32)(v >> h);
        s00 += (u64)a0 * b0[i];
        s01 += (u64)a0 * b1[i];
        s10 += (u64)a1 * b0[i];
        s11 += (u64)a1 * b1[i];
      }
      const unsigned __int128 tot =
          (unsigned __int128)s00 + (((unsigned __int128)s01 + s10) << h) + ((unsigned __int128)s11 << (2 * h));
      x[k] = reduce128(U128{(u64)tot, (u64)(tot >> 64)}, q, P.prec128_lo, P.prec128_hi);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = bld(a.buf, (a.hg << 11) + a.col * 8, (u32)k << 15);  // row 16k+hg, row pitch 2 KiB
  }
}

template <bool SMALL, int SRC>
__device__ __forceinline__ void strided_fwd_body(const DevCtx& c, const StridedArgs& a, const DevPrime& P, const NttFuse& f, u32 pos,
                                                 u32 row, u32 z, u32 rep, u32 n_bytes, u32 split_bits) {
  const u64 q = a.q;
  const BfK bk = bf_consts<SMALL>(q);
  u64 x[16];
  Tw t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");  // keeps this path's loads below the class branch (no hoisting / merging across paths)
  strided_fwd_source<SRC>(c, a, P, f, pos, row, z, rep, n_bytes, split_bits, x);
  load_tw_uniform<SMALL>(a.TW, t0, t1, t2, t3);  // round A: stages 0..3 (uniform twiddles TW[1..15])
  radix16_fwd<SMALL>(x, t0, t1, t2, t3, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) a.lds[(16 * k + a.hg) * kRowPitch + a.cc] = x[k];
  load_tw<SMALL>(a.TW, 4, a.hg, t0, t1, t2, t3);
  __syncthreads();
  // round B: stages 4..7
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = a.lds[(16 * a.hg + k) * kRowPitch + a.cc];
  radix16_fwd<SMALL>(x, t0, t1, t2, t3, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) bst(a.buf, (a.hg << 15) + a.col * 8, (u32)k << 11, x[k]);  // row 16hg+k
  asm volatile("" ::: "memory");  // and its stores above the join
}

// inverse: round B first (stages 7..4 on rows 16h+g'; input lazy [0,lim) from the contiguous pass), then round A (stages
// 3..1 and stage 0 with N^-1 -- or the caller's scale -- folded in); canonical (or centred) output
template <bool SMALL>
__device__ __forceinline__ void strided_inv_body(u64* __restrict__ X, const ulong2* __restrict__ TW, u64* lds, const DevPrime& P,
                                                 const NttFuse& f, u32 pos, u32 cc, u32 hg, u32 col) {
  constexpr u32 log_s = 8;
  const u64 q = P.q;
  const BfK bk = bf_consts<SMALL>(q);
  u64 x[16];
  Tw t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");  // keeps this path's loads below the class branch
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = ntld(&X[((size_t)(16 * hg + k) << log_s) + col]);
  load_tw<SMALL>(TW, 4, hg, t0, t1, t2, t3);
  radix16_inv_321<SMALL>(x, t1, t2, t3, bk);
  radix16_inv_0<SMALL>(x, t0, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[(16 * hg + k) * kRowPitch + cc] = x[k];
  load_tw_uniform<SMALL>(TW, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[(16 * k + hg) * kRowPitch + cc];
  radix16_inv_321<SMALL>(x, t1, t2, t3, bk);
  Tw tn{P.n_inv, P.n_inv_prec}, tw{P.inv_w1_ninv, P.inv_w1_ninv_prec};
  if (f.inv_scale) {
    const u64* sc = f.inv_scale + 4 * (size_t)pos;
    tn = Tw{sc[0], sc[1]};
    tw = Tw{sc[2], sc[3]};
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u64 s = x[k] + x[k + 8];            // [0,2 lim)
    const u64 d = x[k] + bk.lim - x[k + 8];   // (0,2 lim)
    u64 a = shoup_lazy(s, tn, q), b = shoup_lazy(d, tw, q);  // exact quotient: [0,2q)
    x[k] = a >= q ? a - q : a;
    x[k + 8] = b >= q ? b - q : b;
  }
  if (f.center_out) {
    const u64 half = q >> 1;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = x[k] > half ? x[k] - q : x[k];  // two's complement of the negative lift
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) ntst(&X[((size_t)(16 * k + hg) << log_s) + col], x[k]);
  asm volatile("" ::: "memory");
}

// ---- FP class (ntt_fp.hpp): the same passes with FP64 butterflies.  Layouts, LDS traffic and twiddle indices are those of the
// integer bodies; x[] holds integer-valued doubles, the LDS tile and the intermediate between the passes their bit patterns.
template <int SRC>
__device__ __forceinline__ void strided_fwd_body_fp(const DevCtx& c, const StridedArgs& a, const DevPrime& P, const NttFuse& f, u32 pos,
                                                    u32 row, u32 z, u32 rep, u32 n_bytes, u32 split_bits) {
  const FpK k = fp_consts(a.q);
  double x[16], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");  // keeps this path's loads below the class branch
  {
    u64 xi[16];
    strided_fwd_source<SRC>(c, a, P, f, pos, row, z, rep, n_bytes, split_bits, xi);  // canonical residues
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fp_from_u64(xi[i]);
  }
  fp_load_tw_uniform(a.TWD, t0, t1, t2, t3);  // round A: stages 0..3
  fp_radix16_fwd(x, t0, t1, t2, t3, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) a.lds[(16 * i + a.hg) * kRowPitch + a.cc] = fp_bits(x[i]);
  fp_load_tw(a.TWD, 4, a.hg, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(a.lds[(16 * a.hg + i) * kRowPitch + a.cc]);
  fp_radix16_fwd(x, t0, t1, t2, t3, k);  // round B: stages 4..7
#pragma unroll
  for (int i = 0; i < 16; ++i) bst(a.buf, (a.hg << 15) + a.col * 8, (u32)i << 11, fp_bits(x[i]));  // |v| <= 0.51q, as doubles
  asm volatile("" ::: "memory");
}

// inverse: input = the contiguous FP pass' doubles (|v| <= 0.51q); canonical (or centred) u64 output
__device__ __forceinline__ void strided_inv_body_fp(u64* __restrict__ X, const double* __restrict__ TWD, u64* lds, const DevPrime& P,
                                                    const NttFuse& f, u32 pos, u32 cc, u32 hg, u32 col) {
  constexpr u32 log_s = 8;
  const u64 q = P.q;
  const FpK k = fp_consts(q);
  double x[16], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(ntld(&X[((size_t)(16 * hg + i) << log_s) + col]));
  fp_load_tw(TWD, 4, hg, t0, t1, t2, t3);
  fp_radix16_inv_321(x, t1, t2, t3, k);
  fp_radix16_inv_0(x, t0, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[(16 * hg + i) * kRowPitch + cc] = fp_bits(x[i]);
  fp_load_tw_uniform(TWD, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(lds[(16 * i + hg) * kRowPitch + cc]);
  fp_radix16_inv_321(x, t1, t2, t3, k);  // |v| <= 2.53q
  // stage 0 with N^-1 (or the caller's scale) folded into its two multiplications, as in strided_inv_body
  u64 un = P.n_inv, uw = P.inv_w1_ninv;
  if (f.inv_scale) {
    const u64* sc = f.inv_scale + 4 * (size_t)pos;
    un = sc[0];
    uw = sc[2];
  }
  const double tn = fp_from_u64(un), tw = fp_from_u64(uw);
  const double half = fp_from_u64(q >> 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double s = x[i] + x[i + 8], d = x[i] - x[i + 8];  // |.| <= 5.06q
    double a = fp_canon_f(fp_mulmod(s, tn, k), k), b = fp_canon_f(fp_mulmod(d, tw, k), k);
    if (f.center_out) {  // the centred lift, as the two's complement the integer path stores
      a = a > half ? a - k.q : a;
      b = b > half ? b - k.q : b;
      x[i] = a;
      x[i + 8] = b;
    } else {
      x[i] = a;
      x[i + 8] = b;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    u64 v;
    if (f.center_out) v = (u64)(int64_t)x[i];  // |x| < 2^50: exact
    else              v = fp_to_u64(x[i]);
    ntst(&X[((size_t)(16 * i + hg) << log_s) + col], v);
  }
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// PIPELINED forms of the strided pass (ACEHIP_NTT_PIPE = T in {2, 4}; round 6).  A workgroup walks T adjacent tiles of ONE limb.
// Every twiddle of the strided pass depends on the lane only (stages 0..3 are uniform, stages 4..7 go by the row group hg), never on
// the tile, so both sets are fetched once per workgroup and stay in registers (SGPRs / VGPRs) over the walk; the 16 loads of tile
// t+1 are issued BEFORE the butterflies of tile t (a second set of 16 registers), so that a wave's memory phase lies under its own
// arithmetic instead of under the other workgroups' of the CU.  Same butterflies in the same order on the same values as the
// one-tile bodies above: bit-identical results (tests force both forms).  Costs 3 instead of 4 workgroups per CU in registers.
// ------------------------------------------------------------------------------------------------
// keeps the compiler from hoisting the twiddle re-derivation (tw_of) out of the tile walk: the derived twiddles are loop invariant, and
// kept resident they are the 60 registers the companion-only form is there to avoid
__device__ __forceinline__ void pin_tp(Tp15& t) {
  ACEHIP_PIN(t.p0);
#pragma unroll
  for (int i = 0; i < 2; ++i) ACEHIP_PIN(t.p1[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) ACEHIP_PIN(t.p2[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) ACEHIP_PIN(t.p3[i]);
}
#ifndef NTT_PIPE_PREFETCH
#define NTT_PIPE_PREFETCH 1  // 0 (experiment): the walk keeps the twiddles but loads every tile when its turn comes -- no second register set
#endif
// (integer classes: the per-lane twiddles of stages 4..7 stay resident in their companion-only form -- 30 registers instead of 60, the
//  twiddle re-derived where it is used, radix16_fwd_p -- or they would not fit beside round A's temporaries at four workgroups per CU)
template <bool SMALL, int T>
__device__ __forceinline__ void strided_fwd_pipe_body(const StridedArgs& a, const u64* __restrict__ TP) {
  const BfK bk = bf_consts<SMALL>(a.q);
  u64 x[16], xn[16];
  Tw u0, u1[2], u2[4], u3[8];
  Tp15 tp;
  asm volatile("" ::: "memory");
  const u32 vld = (a.hg << 11) + a.col * 8, vst = (a.hg << 15) + a.col * 8;
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = bld(a.buf, vld, (u32)k << 15);
  load_tw_uniform<SMALL>(a.TW, u0, u1, u2, u3);
  load_tp(TP, 4, a.hg, tp);
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    if (NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int k = 0; k < 16; ++k) xn[k] = bld(a.buf, vld + 128u * (u32)(t + 1), (u32)k << 15);
    }
    radix16_fwd<SMALL>(x, u0, u1, u2, u3, bk);
#pragma unroll
    for (int k = 0; k < 16; ++k) a.lds[(16 * k + a.hg) * kRowPitch + a.cc] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = a.lds[(16 * a.hg + k) * kRowPitch + a.cc];
    if (t + 1 < T) __syncthreads();  // the next tile's round A writes the LDS tile again
    pin_tp(tp);
    radix16_fwd_p<SMALL>(x, tp, bk);
#pragma unroll
    for (int k = 0; k < 16; ++k) bst(a.buf, vst + 128u * (u32)t, (u32)k << 11, x[k]);
    if (t + 1 < T) {
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = NTT_PIPE_PREFETCH ? xn[k] : bld(a.buf, vld + 128u * (u32)(t + 1), (u32)k << 15);
    }
  }
  asm volatile("" ::: "memory");
}
template <int T>
__device__ __forceinline__ void strided_fwd_pipe_body_fp(const StridedArgs& a) {
  const FpK k = fp_consts(a.q);
  u64 xn[16];
  double x[16], u0, u1[2], u2[4], u3[8], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");
  const u32 vld = (a.hg << 11) + a.col * 8, vst = (a.hg << 15) + a.col * 8;
#pragma unroll
  for (int i = 0; i < 16; ++i) xn[i] = bld(a.buf, vld, (u32)i << 15);
  fp_load_tw_uniform(a.TWD, u0, u1, u2, u3);
  fp_load_tw(a.TWD, 4, a.hg, t0, t1, t2, t3);
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fp_from_u64(xn[i]);  // canonical residues
    if (NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = bld(a.buf, vld + 128u * (u32)(t + 1), (u32)i << 15);
    }
    fp_radix16_fwd(x, u0, u1, u2, u3, k);
#pragma unroll
    for (int i = 0; i < 16; ++i) a.lds[(16 * i + a.hg) * kRowPitch + a.cc] = fp_bits(x[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(a.lds[(16 * a.hg + i) * kRowPitch + a.cc]);
    if (t + 1 < T) __syncthreads();
    fp_radix16_fwd(x, t0, t1, t2, t3, k);
#pragma unroll
    for (int i = 0; i < 16; ++i) bst(a.buf, vst + 128u * (u32)t, (u32)i << 11, fp_bits(x[i]));
    if (!NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = bld(a.buf, vld + 128u * (u32)(t + 1), (u32)i << 15);
    }
  }
  asm volatile("" ::: "memory");
}

// inverse, pipelined: tiles walked like the forward form; X = the limb, col = the first tile's column of this lane
template <bool SMALL, int T>
__device__ __forceinline__ void strided_inv_pipe_body(u64* __restrict__ X, const ulong2* __restrict__ TW, const u64* __restrict__ TP, u64* lds,
                                                      const DevPrime& P, const NttFuse& f, u32 pos, u32 cc, u32 hg, u32 col) {
  const u64 q = P.q;
  const BfK bk = bf_consts<SMALL>(q);
  u64 x[16], xn[16];
  Tw u0, u1[2], u2[4], u3[8];
  Tp15 tp;
  asm volatile("" ::: "memory");
  // (buffer addressing like the forward pass: one address register for the 16 rows of a lane instead of 16 pointer pairs -- which a loop
  //  over tiles would keep resident)
  const LimbBuf buf = limb_buf(X, 8u << 16);
  const u32 vld = (hg << 15) + col * 8, vst = (hg << 11) + col * 8;  // rows 16 hg + k (2 KiB apart) / rows 16 k + hg (32 KiB apart)
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = bld(buf, vld, (u32)k << 11);
  load_tp(TP, 4, hg, tp);
  load_tw_uniform<SMALL>(TW, u0, u1, u2, u3);
  Tw tn{P.n_inv, P.n_inv_prec}, tw{P.inv_w1_ninv, P.inv_w1_ninv_prec};
  if (f.inv_scale) {
    const u64* sc = f.inv_scale + 4 * (size_t)pos;
    tn = Tw{sc[0], sc[1]};
    tw = Tw{sc[2], sc[3]};
  }
  tn = Tw{uniform64(tn.w), uniform64(tn.p)};  // (the same for every lane: scalar registers over the walk)
  tw = Tw{uniform64(tw.w), uniform64(tw.p)};
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    if (NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int k = 0; k < 16; ++k) xn[k] = bld(buf, vld + 128u * (u32)(t + 1), (u32)k << 11);
    }
    pin_tp(tp);
    radix16_inv_p<SMALL>(x, tp, bk);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[(16 * hg + k) * kRowPitch + cc] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[(16 * k + hg) * kRowPitch + cc];
    if (t + 1 < T) __syncthreads();
    radix16_inv_321<SMALL>(x, u1, u2, u3, bk);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 s = x[k] + x[k + 8];
      const u64 d = x[k] + bk.lim - x[k + 8];
      u64 a = shoup_lazy(s, tn, q), b = shoup_lazy(d, tw, q);
      x[k] = a >= q ? a - q : a;
      x[k + 8] = b >= q ? b - q : b;
    }
    if (f.center_out) {
      const u64 half = q >> 1;
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = x[k] > half ? x[k] - q : x[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) bst(buf, vst + 128u * (u32)t, (u32)k << 15, x[k]);
    if (t + 1 < T) {
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = NTT_PIPE_PREFETCH ? xn[k] : bld(buf, vld + 128u * (u32)(t + 1), (u32)k << 11);
    }
  }
  asm volatile("" ::: "memory");
}
template <int T>
__device__ __forceinline__ void strided_inv_pipe_body_fp(u64* __restrict__ X, const double* __restrict__ TWD, u64* lds, const DevPrime& P,
                                                         const NttFuse& f, u32 pos, u32 cc, u32 hg, u32 col) {
  const u64 q = P.q;
  const FpK k = fp_consts(q);
  u64 xn[16];
  double x[16], u0, u1[2], u2[4], u3[8], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");
  const LimbBuf buf = limb_buf(X, 8u << 16);
  const u32 vld = (hg << 15) + col * 8, vst = (hg << 11) + col * 8;
#pragma unroll
  for (int i = 0; i < 16; ++i) xn[i] = bld(buf, vld, (u32)i << 11);
  fp_load_tw(TWD, 4, hg, t0, t1, t2, t3);
  fp_load_tw_uniform(TWD, u0, u1, u2, u3);
  u64 un = P.n_inv, uw = P.inv_w1_ninv;
  if (f.inv_scale) {
    const u64* sc = f.inv_scale + 4 * (size_t)pos;
    un = sc[0];
    uw = sc[2];
  }
  const double tn = fp_from_u64(un), tw = fp_from_u64(uw);
  const double half = fp_from_u64(q >> 1);
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(xn[i]);
    if (NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = bld(buf, vld + 128u * (u32)(t + 1), (u32)i << 11);
    }
    fp_radix16_inv_321(x, t1, t2, t3, k);
    fp_radix16_inv_0(x, t0, k);
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[(16 * hg + i) * kRowPitch + cc] = fp_bits(x[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(lds[(16 * i + hg) * kRowPitch + cc]);
    if (t + 1 < T) __syncthreads();
    fp_radix16_inv_321(x, u1, u2, u3, k);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double s = x[i] + x[i + 8], d = x[i] - x[i + 8];
      double a = fp_canon_f(fp_mulmod(s, tn, k), k), b = fp_canon_f(fp_mulmod(d, tw, k), k);
      if (f.center_out) {
        a = a > half ? a - k.q : a;
        b = b > half ? b - k.q : b;
      }
      x[i] = a;
      x[i + 8] = b;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      u64 v;
      if (f.center_out) v = (u64)(int64_t)x[i];
      else              v = fp_to_u64(x[i]);
      bst(buf, vst + 128u * (u32)t, (u32)i << 15, v);
    }
    if (!NTT_PIPE_PREFETCH && t + 1 < T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = bld(buf, vld + 128u * (u32)(t + 1), (u32)i << 11);
    }
  }
  asm volatile("" ::: "memory");
}

// (uniform64: a load issued after stores of the same kernel is a vector load even when its address is uniform)
// One resolved workgroup of a pass: tile of limb row y of polynomial z, the limb's position / prime
struct NttWg {
  u32 tile, y, z, pos, gi;
  u32 rep;  // replica (kernels.hpp DevCtx): the launch's polynomial index is z + n_polys * (rep - rep0)
};
// polynomial index of the launch -> polynomial of the call and replica
__device__ __forceinline__ void ntt_split_z(NttWg& w, const DevCtx& c, u32 n_polys) {
  const u32 r = c.nrep == 1 ? 0 : w.z / n_polys;
  w.z -= r * n_polys;
  w.rep = c.rep0 + r;
}
// limb row (y, z) of the launch -> position and prime; false: the row does not exist (a digit's own limbs, grid padding)
__device__ __forceinline__ bool ntt_resolve(NttWg& w, const DevCtx& c, const NttFuse& f, u32 level, u32 pos0, u32 skip_alpha) {
  u32 pos, gi;
  if (f.gi_tab != nullptr) {  // packed limb list (limb-sharded execution)
    pos = w.y;
    gi = f.gi_tab[pos];
  } else {
    if (!ntt_limb_pos(pos, pos0, level, c.K, skip_alpha, w.y, w.z)) return false;
    gi = limb_prime(pos, level, c.L);
  }
  w.pos = __builtin_amdgcn_readfirstlane(pos);  // wave-uniform: prime constants and base pointers live in SGPRs
  w.gi = __builtin_amdgcn_readfirstlane(gi);
  return owns(c, w.gi);  // limb-sharded execution: the other ranks' limbs are skipped
}

// STRIDED pass of one workgroup.  SRC (forward only): source of the first pass' input, see strided_fwd_body.
template <bool INVERSE, int SRC>
__device__ __forceinline__ void strided_pass(const DevCtx& c, u64* __restrict__ poly, size_t poly_stride, u32 pos_off, const NttFuse& f,
                                             u64* lds, const NttWg& w) {
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  u64* __restrict__ X = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + (size_t)(w.pos - pos_off) * c.N;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const u32 tid = threadIdx.x, cc = tid & 15, hg = tid >> 4;
  const u32 col = w.tile * 16 + cc;  // N = 2^16 only (launch_ntt_fused): constant row stride, addresses = one base + immediates
  const double* __restrict__ TWD = c.twd_fwd ? (INVERSE ? c.twd_inv : c.twd_fwd) + (size_t)w.gi * c.N : nullptr;
  const bool fp = TWD != nullptr && q < kFpPrimeMax;  // FP class: both passes of the transform decide alike (wave-uniform)
  if (!INVERSE) {
    const StridedArgs a{limb_buf(X, c.N * 8), TW, TWD, lds, cc, hg, col, q};
    if (fp)                       strided_fwd_body_fp<SRC>(c, a, P, f, w.pos, w.y, w.z, w.rep, c.N * 8, c.split_bits);
    else if (q <= kSmallPrimeMax) strided_fwd_body<true, SRC>(c, a, P, f, w.pos, w.y, w.z, w.rep, c.N * 8, c.split_bits);
    else                          strided_fwd_body<false, SRC>(c, a, P, f, w.pos, w.y, w.z, w.rep, c.N * 8, c.split_bits);
  } else {
    if (fp)                       strided_inv_body_fp(X, TWD, lds, P, f, w.pos, cc, hg, col);
    else if (q <= kSmallPrimeMax) strided_inv_body<true>(X, TW, lds, P, f, w.pos, cc, hg, col);
    else                          strided_inv_body<false>(X, TW, lds, P, f, w.pos, cc, hg, col);
  }
}

template <bool INVERSE, int SRC>
__global__ __launch_bounds__(256, ACEHIP_NTT_MIN_WG) void ntt8_strided_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride,
                                                           u32 level, u32 pos0, u32 pos_off, u32 skip_alpha, NttFuse f,
                                                           u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[256 * kRowPitch];
#if NTT_EXP & 8
  if (SRC != SRC_MSG) return;
#endif
  const NttBlk blk = ntt_block(c.logN - 12, n_limbs, n_polys * c.nrep);
  NttWg w{blk.tile, blk.y, blk.z, 0, 0, 0};
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  strided_pass<INVERSE, SRC>(c, poly, poly_stride, pos_off, f, lds, w);
}

// pipelined strided pass (in place, SRC_MEM): workgroup = tiles [T*tile, T*tile + T) of limb row y of polynomial z
#ifndef NTT_PIPE_T4
#define NTT_PIPE_T4 0  // 1: also compile the four-tile walk (experiments; measured slower than two tiles on every batch, profiles/r06a_*)
#endif
#ifndef NTT_PIPE_WG
#define NTT_PIPE_WG 4  // workgroups per CU the strided walk is compiled for (121 / 128 VGPRs with the companion-only resident twiddles)
#endif
#ifndef NTT_PIPE_CONTIG_WG
#define NTT_PIPE_CONTIG_WG 3  // ... and the contiguous walk (it needs the registers: its twiddles change with the tile)
#endif
#ifdef NTT_PIPE_FP_ONLY  // experiment: the pipelined kernels carry the FP class only (register allocation of that class alone)
#define NTT_PIPE_INT(x) (void)0  // (results of the integer classes are wrong: timing experiment)
#else
#define NTT_PIPE_INT(x) x
#endif
template <bool INVERSE, int T>
__global__ __launch_bounds__(256, NTT_PIPE_WG) void ntt8_strided_pipe_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride, u32 level, u32 pos0,
                                                                u32 pos_off, u32 skip_alpha, NttFuse f, u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[256 * kRowPitch];
  constexpr u32 logT = T == 2 ? 1 : (T == 4 ? 2 : 3);
  static_assert(T == 2 || T == 4 || T == 8, "tiles per workgroup");
  const NttBlk blk = ntt_block(c.logN - 12 - logT, n_limbs, n_polys * c.nrep);
  NttWg w{blk.tile * (u32)T, blk.y, blk.z, 0, 0, 0};
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  u64* __restrict__ X = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + (size_t)(w.pos - pos_off) * c.N;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const u64* __restrict__ TP = (INVERSE ? c.twp_inv : c.twp_fwd) + (size_t)w.gi * c.N;  // (the launcher checks that the tables exist)
  const u32 tid = threadIdx.x, cc = tid & 15, hg = tid >> 4;
  const u32 col = w.tile * 16 + cc;
  const double* __restrict__ TWD = c.twd_fwd ? (INVERSE ? c.twd_inv : c.twd_fwd) + (size_t)w.gi * c.N : nullptr;
  const bool fp = TWD != nullptr && q < kFpPrimeMax;
  if (!INVERSE) {
    const StridedArgs a{limb_buf(X, c.N * 8), TW, TWD, lds, cc, hg, col, q};
    if (fp)                       strided_fwd_pipe_body_fp<T>(a);
    else if (q <= kSmallPrimeMax) NTT_PIPE_INT((strided_fwd_pipe_body<true, T>(a, TP)));
    else                          NTT_PIPE_INT((strided_fwd_pipe_body<false, T>(a, TP)));
  } else {
    if (fp)                       strided_inv_pipe_body_fp<T>(X, TWD, lds, P, f, w.pos, cc, hg, col);
    else if (q <= kSmallPrimeMax) NTT_PIPE_INT((strided_inv_pipe_body<true, T>(X, TW, TP, lds, P, f, w.pos, cc, hg, col)));
    else                          NTT_PIPE_INT((strided_inv_pipe_body<false, T>(X, TW, TP, lds, P, f, w.pos, cc, hg, col)));
  }
}

// ------------------------------------------------------------------------------------------------
// CONTIG pass (stages logN-8 .. logN-1): tile = 16 consecutive blocks o = chunk*16 + b of 256
// contiguous coefficients, rho = index within the block.  Round A lanes (g = tid&15, b = tid>>4)
// hold rho = 16k+g (each load instruction reads whole 128-byte lines); round B lanes (h = tid&15, b)
// hold the 16 contiguous rho = 16h+g'.  The contiguous side goes through LDS so that global accesses
// stay 16 bytes per lane, 1 KiB contiguous per wave instruction.
// CANON_OUT (inverse only): write canonical values instead of lazy [0,lim) (needed when a generic
// pass follows instead of the strided fast pass).
// ------------------------------------------------------------------------------------------------
// ---- key inner product as the source of a contiguous pass (kernels.hpp Kmac).  One workgroup = tile `tile` of polynomial z at extended
// limb position xpos (prime gi): the value at coefficient tile*4096 + e is sum_d E_d[e] * key_d[z][gi][e] -- exact 128-bit sums of at most 8
// products below 2^122, reduced once (the canonical residue, whatever order generated code's Hw_modmul / Hw_modadd chains use).  The
// digit and key base pointers are kernel arguments indexed by the wave-uniform d: scalar loads.
struct KmacSrc {
  u32 xpos, gi, z, rep, tile;
  u64 q, ml, mh;
};
// generic form (any number of digits): one coefficient pair, digits one after the other
__device__ __forceinline__ ulong2 kmac_at(const DevCtx& c, const Kmac& km, const KmacSrc& k, u32 e) {
  U128 s0{0, 0}, s1{0, 0};
  const size_t eoff = (size_t)k.xpos * c.N + (size_t)k.tile * 4096 + e;
  const size_t koff = ((size_t)k.z * km.key_T + k.gi) * c.N + (size_t)k.tile * 4096 + e;
  const u32 own_d = (km.own != nullptr && k.xpos < km.level) ? k.xpos / km.alpha : 0xffffffffu;
  for (u32 d = 0; d < km.nd; ++d) {
    const u64* eb = d == own_d ? km.own : km.ext[d];
    const ulong2 ev = *reinterpret_cast<const ulong2*>(reb(c, eb, k.rep) + eoff);
    const ulong2 kv = *reinterpret_cast<const ulong2*>(reb(c, km.key[d], k.rep) + koff);
    mac128(s0, ev.x, kv.x);
    mac128(s1, ev.y, kv.y);
  }
  return ulong2{reduce128(s0, k.q, k.ml, k.mh), reduce128(s1, k.q, k.ml, k.mh)};
}
// The workgroup's 8 coefficient pairs per lane (e = 2*tid + 512*i) for ND digits known at compile time: the 2*ND*CH loads of CH pairs
// are issued together before any of them is used (the generic loop above waits for every pair of loads in turn: eight or more memory
// latencies one after the other, which is what the pass then costs next to the kernels of other image streams).  sink(i, value).
template <int ND>
struct KmacPtrs {
  const u64* e[ND];
  const u64* k[ND];
};
template <int ND>
__device__ __forceinline__ KmacPtrs<ND> kmac_ptrs(const DevCtx& c, const Kmac& km, const KmacSrc& k) {
  KmacPtrs<ND> p;
  const size_t eoff = (size_t)k.xpos * c.N + (size_t)k.tile * 4096;
  const size_t koff = ((size_t)k.z * km.key_T + k.gi) * c.N + (size_t)k.tile * 4096;
  const u32 own_d = (km.own != nullptr && k.xpos < km.level) ? k.xpos / km.alpha : 0xffffffffu;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    p.e[d] = reb(c, (u32)d == own_d ? km.own : km.ext[d], k.rep) + eoff;
    p.k[d] = reb(c, km.key[d], k.rep) + koff;
  }
  return p;
}
template <int ND, class Sink>
__device__ __forceinline__ void kmac_tile(const DevCtx& c, const Kmac& km, const KmacSrc& k, u32 tid, Sink sink) {
  constexpr int CH = ND == 1 ? 8 : (ND == 2 ? 4 : 2);  // 64 (48 for three digits) registers of loads in flight
  const KmacPtrs<ND> p = kmac_ptrs<ND>(c, km, k);
#pragma unroll
  for (int i0 = 0; i0 < 8; i0 += CH) {
    ulong2 ev[CH][ND], kv[CH][ND];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const u32 e = 2 * tid + 512 * (u32)(i0 + i);
        ev[i][d] = *reinterpret_cast<const ulong2*>(p.e[d] + e);
#if NTT_EXP & 64  // timing experiment (results are wrong): every key load of the limb hits the same 4 KiB -- what the key stream costs a pass
        kv[i][d] = *reinterpret_cast<const ulong2*>(p.k[d] - (size_t)k.tile * 4096 + (e & 510u));
#else
        kv[i][d] = *reinterpret_cast<const ulong2*>(p.k[d] + e);
#endif
      }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      U128 s0{0, 0}, s1{0, 0};
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        mac128(s0, ev[i][d].x, kv[i][d].x);
        mac128(s1, ev[i][d].y, kv[i][d].y);
      }
      sink(i0 + i, ulong2{reduce128(s0, k.q, k.ml, k.mh), reduce128(s1, k.q, k.ml, k.mh)});
    }
  }
}
template <class Sink>
__device__ __forceinline__ void kmac_tile_any(const DevCtx& c, const Kmac& km, const KmacSrc& k, u32 tid, Sink sink) {
  switch (km.nd) {  // (wave-uniform: a kernel argument)
    case 1: kmac_tile<1>(c, km, k, tid, sink); break;
    case 2: kmac_tile<2>(c, km, k, tid, sink); break;
    case 3: kmac_tile<3>(c, km, k, tid, sink); break;
    case 4: kmac_tile<4>(c, km, k, tid, sink); break;
    default:
      for (u32 i = 0; i < 8; ++i) sink((int)i, kmac_at(c, km, k, 2 * tid + 512 * i));
  }
}

// forward rounds: x[] holds rho = 16k+lo4 of block b on entry and the canonical values of the 16 contiguous
// rho = 16*lo4+k on return
// PRE (pipelined form): x[] already holds the tile's values as they lie in memory (loaded during the previous tile's butterflies);
// pf() issues the NEXT tile's loads.  It is called right after the tile's first twiddle loads: vmcnt counts in order, so a wait for
// twiddles issued behind the prefetch would wait for the prefetch as well.
struct NoPf {
  __device__ __forceinline__ void operator()() const {}
};
template <bool SMALL, bool TW8, bool PRE = false, class PF = NoPf>
__device__ __forceinline__ void contig_fwd_body(const u64* __restrict__ X, const ulong2* __restrict__ TW, const u64* __restrict__ TP,
                                                u64* lds, u32 s8, u32 o, u32 b, u32 lo4, u64 q, u64 mu, u64 (&x)[16], PF pf = PF{}) {
  const BfK bk = bf_consts<SMALL>(q);
  Tw t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");  // keeps this path's loads below the class branch
  if (!PRE) {
#pragma unroll
#if NTT_EXP & (2 | 16)
    for (int k = 0; k < 16; ++k) x[k] = (u64)(b * 256 + 16 * k + lo4) * 0x9E3779B97F4A7C15ull + o;
#else
    for (int k = 0; k < 16; ++k) x[k] = ntld(&X[b * 256 + 16 * k + lo4]);
#endif
  }
  load_tw<SMALL>(TW, s8, o, t0, t1, t2, t3);  // round A: stages s8..s8+3 on rho = 16k + g
  pf();
  radix16_fwd<SMALL>(x, t0, t1, t2, t3, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * kBlkPitch + 17 * k + lo4] = x[k];
  Tp15 tp;
  if (TW8) load_tp(TP, s8 + 4, 16 * o + lo4, tp);
  else     load_tw<SMALL>(TW, s8 + 4, 16 * o + lo4, t0, t1, t2, t3);
  __syncthreads();
  // round B: stages s8+4..s8+7 on rho = 16h + g'
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[b * kBlkPitch + 17 * lo4 + k];
  if (TW8) radix16_fwd_p<SMALL>(x, tp, bk);
  else     radix16_fwd<SMALL>(x, t0, t1, t2, t3, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = canon_fwd<SMALL>(x[k], q, mu);
}

// inverse rounds: the tile is read from S (coalesced 16-byte loads through LDS), round B first (stages s8+7..s8+4 on the 16
// contiguous rho = 16h + g'), then round A (stages s8+3..s8 on rho = 16k + g); lazy [0,lim) output unless CANON_OUT
// PRE (pipelined form): the tile's 8 coalesced 16-byte loads are already in pre[]; a barrier follows the last LDS read (the next
// tile of the walk writes the LDS tile again)
template <bool SMALL, bool CANON_OUT, bool TW8, bool KM = false, bool PRE = false, class PF = NoPf>
__device__ __forceinline__ void contig_inv_body(u64* __restrict__ X, const u64* __restrict__ S, const ulong2* __restrict__ TW,
                                                const u64* __restrict__ TP, u64* lds, u32 s8, u32 o, u32 b, u32 lo4, u64 q,
                                                const DevCtx* kc = nullptr, const Kmac* km = nullptr, const KmacSrc* ks = nullptr,
                                                const u64x2_t* pre = nullptr, PF pf = PF{}) {
  const BfK bk = bf_consts<SMALL>(q);
  const u32 tid = threadIdx.x;
  u64 x[16];
  Tw t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");  // keeps this path's loads below the class branch
  if (KM) {  // the tile is the key inner product (canonical residues)
    kmac_tile_any(*kc, *km, *ks, tid, [&](int i, ulong2 v) {
      const u32 e = 2 * tid + 512 * (u32)i, bb = e >> 8, rho = e & 255;
      lds[bb * kBlkPitch + cpad(rho)] = v.x;
      lds[bb * kBlkPitch + cpad(rho) + 1] = v.y;
    });
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // coalesced 16-byte loads
      const u32 e = 2 * tid + 512 * i, bb = e >> 8, rho = e & 255;
      const u64x2_t vv = PRE ? pre[i] : ntld(reinterpret_cast<const u64x2_t*>(S + e));
      lds[bb * kBlkPitch + cpad(rho)] = vv.x;
      lds[bb * kBlkPitch + cpad(rho) + 1] = vv.y;
    }
  }
  Tp15 tp;
  if (TW8) load_tp(TP, s8 + 4, 16 * o + lo4, tp);
  else     load_tw<SMALL>(TW, s8 + 4, 16 * o + lo4, t0, t1, t2, t3);
  pf();
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[b * kBlkPitch + 17 * lo4 + k];
  if (TW8) {
    radix16_inv_p<SMALL>(x, tp, bk);
  } else {
    radix16_inv_321<SMALL>(x, t1, t2, t3, bk);
    radix16_inv_0<SMALL>(x, t0, bk);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * kBlkPitch + 17 * lo4 + k] = x[k];
  load_tw<SMALL>(TW, s8, o, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[b * kBlkPitch + 17 * k + lo4];
  if (PRE) __syncthreads();
  radix16_inv_321<SMALL>(x, t1, t2, t3, bk);
  radix16_inv_0<SMALL>(x, t0, bk);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    u64 v = x[k];  // [0,lim)
    if (CANON_OUT) {
      if (SMALL) v = v >= 4 * q ? v - 4 * q : v;
      v = v >= 2 * q ? v - 2 * q : v;
      v = v >= q ? v - q : v;
    }
    ntst(&X[b * 256 + 16 * k + lo4], v);
  }
  asm volatile("" ::: "memory");
}

// ---- FP class, contiguous pass.  forward: input = the strided FP pass' doubles; x[] returns the canonical u64 residues of the 16
// contiguous rho = 16*lo4+k, like contig_fwd_body
template <bool PRE = false, class PF = NoPf>
__device__ __forceinline__ void contig_fwd_body_fp(const u64* __restrict__ X, const double* __restrict__ TWD, u64* lds, u32 s8, u32 o, u32 b,
                                                   u32 lo4, u64 q, u64 (&xo)[16], PF pf = PF{}) {
  const FpK k = fp_consts(q);
  double x[16], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(PRE ? xo[i] : ntld(&X[b * 256 + 16 * i + lo4]));
  fp_load_tw(TWD, s8, o, t0, t1, t2, t3);  // round A: stages s8..s8+3 on rho = 16k + g
  pf();
  fp_radix16_fwd(x, t0, t1, t2, t3, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[b * kBlkPitch + 17 * i + lo4] = fp_bits(x[i]);
  fp_load_tw(TWD, s8 + 4, 16 * o + lo4, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(lds[b * kBlkPitch + 17 * lo4 + i]);
  fp_radix16_fwd(x, t0, t1, t2, t3, k);  // round B; |v| <= 0.51q
#pragma unroll
  for (int i = 0; i < 16; ++i) xo[i] = fp_to_u64(x[i] < 0 ? x[i] + k.q : x[i]);
}

// inverse: canonical u64 input from S (coalesced through LDS, as contig_inv_body), output doubles |v| <= 0.51q for the strided FP pass
template <bool KM = false, bool PRE = false, class PF = NoPf>
__device__ __forceinline__ void contig_inv_body_fp(u64* __restrict__ X, const u64* __restrict__ S, const double* __restrict__ TWD, u64* lds,
                                                   u32 s8, u32 o, u32 b, u32 lo4, u64 q, const DevCtx* kc = nullptr, const Kmac* km = nullptr,
                                                   const KmacSrc* ks = nullptr, const u64x2_t* pre = nullptr, PF pf = PF{}) {
  const FpK k = fp_consts(q);
  const u32 tid = threadIdx.x;
  double x[16], t0, t1[2], t2[4], t3[8];
  asm volatile("" ::: "memory");
  if (KM) {
    kmac_tile_any(*kc, *km, *ks, tid, [&](int i, ulong2 v) {
      const u32 e = 2 * tid + 512 * (u32)i, bb = e >> 8, rho = e & 255;
      lds[bb * kBlkPitch + cpad(rho)] = v.x;
      lds[bb * kBlkPitch + cpad(rho) + 1] = v.y;
    });
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // coalesced 16-byte loads
      const u32 e = 2 * tid + 512 * i, bb = e >> 8, rho = e & 255;
      const u64x2_t vv = PRE ? pre[i] : ntld(reinterpret_cast<const u64x2_t*>(S + e));
      lds[bb * kBlkPitch + cpad(rho)] = vv.x;
      lds[bb * kBlkPitch + cpad(rho) + 1] = vv.y;
    }
  }
  fp_load_tw(TWD, s8 + 4, 16 * o + lo4, t0, t1, t2, t3);
  pf();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_red(fp_from_u64(lds[b * kBlkPitch + 17 * lo4 + i]), k);  // canonical -> |v| <= 0.51q (ntt_fp.hpp)
  fp_radix16_inv_321(x, t1, t2, t3, k);
  fp_radix16_inv_0(x, t0, k);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[b * kBlkPitch + 17 * lo4 + i] = fp_bits(x[i]);
  fp_load_tw(TWD, s8, o, t0, t1, t2, t3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_of_bits(lds[b * kBlkPitch + 17 * i + lo4]);
  if (PRE) __syncthreads();
  fp_radix16_inv_321(x, t1, t2, t3, k);
  fp_radix16_inv_0(x, t0, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) ntst(&X[b * 256 + 16 * i + lo4], fp_bits(x[i]));
  asm volatile("" ::: "memory");
}

// the end of a forward contiguous tile: x[] (canonical, the lane's 16 contiguous coefficients) goes through LDS into coalesced 16-byte
// stores, combined with the fused neighbour (FUSE 1 / 2: Rescale / ModDown tail, 3: the ModDown tail on the key inner product).
// X = the tile in the transformed polynomial, `tile` its index within the limb.
template <int FUSE>
__device__ __forceinline__ void contig_fwd_finish(const DevCtx& c, const NttFuse& f, u64* lds, const NttWg& w, u32 tile, const DevPrime& P, u64 q,
                                                  u64* __restrict__ X, const u64 (&x)[16]) {
  const u32 tid = threadIdx.x, lo4 = tid & 15, b = tid >> 4, pos = w.pos;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[b * kBlkPitch + 17 * lo4 + k] = x[k];
    __syncthreads();
    const size_t tail_off = (size_t)pos * c.N + (size_t)tile * 4096;  // q-limb `pos` of x_z / out_z
    const KmacSrc ks{f.km_pos0 + pos, w.gi, w.z, w.rep, tile, q, P.prec128_lo, P.prec128_hi};
    const u64* __restrict__ xin = (FUSE == 1 || FUSE == 2) ? reb(c, w.z ? f.x1 : f.x0, w.rep) + tail_off : nullptr;
    u64* __restrict__ dst = FUSE ? reb(c, w.z ? f.out1 : f.out0, w.rep) + tail_off : X;
    const u64 tw_w = FUSE ? f.w[pos] : 0, tw_p = FUSE ? f.wp[pos] : 0;
    if (FUSE == 3) {  // the ModDown tail on the key inner product itself
      kmac_tile_any(c, f.km, ks, tid, [&](int i, ulong2 xv) {
        const u32 e = 2 * tid + 512 * (u32)i, bb = e >> 8, rho = e & 255;
        u64x2_t vv;
        vv.x = mul_shoup(sub_mod(xv.x, lds[bb * kBlkPitch + cpad(rho)], q), tw_w, tw_p, q);
        vv.y = mul_shoup(sub_mod(xv.y, lds[bb * kBlkPitch + cpad(rho) + 1], q), tw_w, tw_p, q);
        ntst(reinterpret_cast<u64x2_t*>(dst + e), vv);
      });
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // coalesced 16-byte stores
      const u32 e = 2 * tid + 512 * i, bb = e >> 8, rho = e & 255;
      ulong2 v;
      v.x = lds[bb * kBlkPitch + cpad(rho)];
      v.y = lds[bb * kBlkPitch + cpad(rho) + 1];
      if (FUSE == 1) {
        const ulong2 xv = *reinterpret_cast<const ulong2*>(xin + e);
        v.x = add_mod(mul_shoup(xv.x, tw_w, tw_p, q), v.x, q);
        v.y = add_mod(mul_shoup(xv.y, tw_w, tw_p, q), v.y, q);
      } else if (FUSE == 2) {
        const ulong2 xv = *reinterpret_cast<const ulong2*>(xin + e);
        v.x = mul_shoup(sub_mod(xv.x, v.x, q), tw_w, tw_p, q);
        v.y = mul_shoup(sub_mod(xv.y, v.y, q), tw_w, tw_p, q);
      }
#if NTT_EXP & 2
      if (v.x != 0x123456789ull) continue;
#endif
      u64x2_t vv;
      vv.x = v.x;
      vv.y = v.y;
      ntst(reinterpret_cast<u64x2_t*>(dst + e), vv);
    }
}

// CONTIG pass of one workgroup.
// FUSE: inverse -> 1: read the input from f.src_z (out of place);  forward -> 1 / 2: combine the result with
// f.x_z and write it to f.out_z (Rescale / ModDown tail) instead of storing it in place
template <bool INVERSE, bool CANON_OUT, int FUSE, bool TW8, bool FP_OK>
__device__ __forceinline__ void contig_pass(const DevCtx& c, u64* __restrict__ poly, size_t poly_stride, u32 pos_off, const NttFuse& f,
                                            u64* lds, const NttWg& w) {
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  const u32 pos = w.pos;
  u64* __restrict__ X = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + (size_t)(pos - pos_off) * c.N + (size_t)w.tile * 4096;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const u64* __restrict__ TP = TW8 ? (INVERSE ? c.twp_inv : c.twp_fwd) + (size_t)w.gi * c.N : nullptr;
  const u32 s8 = c.logN - 8;
  const u32 tid = threadIdx.x, lo4 = tid & 15, b = tid >> 4;
  const u32 o = w.tile * 16 + b;

  // FP class (ntt_fp.hpp): only where the OTHER pass of the transform is the wide strided pass (launch_ntt_fused sets FP_OK), never
  // with CANON_OUT (the hybrid sizes, whose other stages are the generic integer kernel)
  const double* __restrict__ TWD = (FP_OK && c.twd_fwd) ? (INVERSE ? c.twd_inv : c.twd_fwd) + (size_t)w.gi * c.N : nullptr;
  const bool fp = TWD != nullptr && q < kFpPrimeMax;
  if (!INVERSE) {
    u64 x[16];
    if (fp)                       contig_fwd_body_fp(X, TWD, lds, s8, o, b, lo4, q, x);
    else if (q <= kSmallPrimeMax) contig_fwd_body<true, TW8>(X, TW, TP, lds, s8, o, b, lo4, q, P.prec128_hi, x);
    else                          contig_fwd_body<false, TW8>(X, TW, TP, lds, s8, o, b, lo4, q, P.prec128_hi, x);
    contig_fwd_finish<FUSE>(c, f, lds, w, w.tile, P, q, X, x);
  } else {
    if (FUSE == 2) {  // input = the key inner product
      const KmacSrc ks{f.km_pos0 + pos, w.gi, w.z, w.rep, w.tile, q, P.prec128_lo, P.prec128_hi};
      if (fp)                       contig_inv_body_fp<true>(X, nullptr, TWD, lds, s8, o, b, lo4, q, &c, &f.km, &ks);
      else if (q <= kSmallPrimeMax) contig_inv_body<true, CANON_OUT, TW8, true>(X, nullptr, TW, TP, lds, s8, o, b, lo4, q, &c, &f.km, &ks);
      else                          contig_inv_body<false, CANON_OUT, TW8, true>(X, nullptr, TW, TP, lds, s8, o, b, lo4, q, &c, &f.km, &ks);
      return;
    }
    const u64* __restrict__ S =
        FUSE ? reb(c, w.z ? f.src1 : f.src0, w.rep) + (size_t)(pos - pos_off) * c.N + (size_t)w.tile * 4096 : X;
    if (fp)                       contig_inv_body_fp(X, S, TWD, lds, s8, o, b, lo4, q);
    else if (q <= kSmallPrimeMax) contig_inv_body<true, CANON_OUT, TW8>(X, S, TW, TP, lds, s8, o, b, lo4, q);
    else                          contig_inv_body<false, CANON_OUT, TW8>(X, S, TW, TP, lds, s8, o, b, lo4, q);
  }
}

template <bool INVERSE, bool CANON_OUT, int FUSE, bool TW8 = false, bool FP_OK = false>
__global__ __launch_bounds__(256, ACEHIP_NTT_MIN_WG) void ntt8_contig_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride,
                                                          u32 level, u32 pos0, u32 pos_off, u32 skip_alpha, NttFuse f,
                                                          u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[16 * kBlkPitch];
  const NttBlk blk = ntt_block(c.logN - 12, n_limbs, n_polys * c.nrep);
  NttWg w{blk.tile, blk.y, blk.z, 0, 0, 0};
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  contig_pass<INVERSE, CANON_OUT, FUSE, TW8, FP_OK>(c, poly, poly_stride, pos_off, f, lds, w);
}

// PIPELINED contiguous pass (ACEHIP_NTT_PIPE = T; round 6): the workgroup walks tiles [T*tile, T*tile + T) of one limb; the global loads of
// tile t+1 are issued before the butterflies of tile t.  The twiddles of this pass differ per tile and are fetched at the head of each
// tile like in the one-tile form (they come from the XCD's L2: the polynomials of a launch share them).  FUSE as in contig_pass
// (forward: 0..3; inverse: 0 / 1; the key inner product as the inverse's source keeps the one-tile form).
template <bool INVERSE, int FUSE, bool TW8, int T>
__global__ __launch_bounds__(256, NTT_PIPE_CONTIG_WG) void ntt8_contig_pipe_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride, u32 level, u32 pos0,
                                                               u32 pos_off, u32 skip_alpha, NttFuse f, u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[16 * kBlkPitch];
  constexpr u32 logT = T == 2 ? 1 : (T == 4 ? 2 : 3);
  static_assert(T == 2 || T == 4 || T == 8, "tiles per workgroup");
  static_assert(!INVERSE || FUSE <= 1, "the key inner product as the source of the inverse pass keeps the one-tile kernel");
  const NttBlk blk = ntt_block(c.logN - 12 - logT, n_limbs, n_polys * c.nrep);
  NttWg w{blk.tile * (u32)T, blk.y, blk.z, 0, 0, 0};
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  const u32 pos = w.pos;
  u64* __restrict__ X = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + (size_t)(pos - pos_off) * c.N + (size_t)w.tile * 4096;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const u64* __restrict__ TP = TW8 ? (INVERSE ? c.twp_inv : c.twp_fwd) + (size_t)w.gi * c.N : nullptr;
  const u32 s8 = c.logN - 8;
  const u32 tid = threadIdx.x, lo4 = tid & 15, b = tid >> 4;
  const double* __restrict__ TWD = c.twd_fwd ? (INVERSE ? c.twd_inv : c.twd_fwd) + (size_t)w.gi * c.N : nullptr;
  const bool fp = TWD != nullptr && q < kFpPrimeMax;
  const bool small = q <= kSmallPrimeMax;
  if (!INVERSE) {
    u64 x[16], xn[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ntld(&X[b * 256 + 16 * k + lo4]);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      u64* __restrict__ Xt = X + (size_t)t * 4096;
      const u32 tile = w.tile + (u32)t, o = tile * 16 + b;
      auto pf = [&]() {
        if (t + 1 < T) {
#pragma unroll
          for (int k = 0; k < 16; ++k) xn[k] = ntld(&Xt[4096 + b * 256 + 16 * k + lo4]);
        }
      };
      if (fp)         contig_fwd_body_fp<true>(Xt, TWD, lds, s8, o, b, lo4, q, x, pf);
      else if (small) NTT_PIPE_INT((contig_fwd_body<true, TW8, true>(Xt, TW, TP, lds, s8, o, b, lo4, q, P.prec128_hi, x, pf)));
      else            NTT_PIPE_INT((contig_fwd_body<false, TW8, true>(Xt, TW, TP, lds, s8, o, b, lo4, q, P.prec128_hi, x, pf)));
      contig_fwd_finish<FUSE>(c, f, lds, w, tile, P, q, Xt, x);
      if (t + 1 < T) {
        __syncthreads();  // the next tile's round A writes the LDS tile again
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xn[k];
      }
    }
  } else {
    const u64* __restrict__ S =
        FUSE ? reb(c, w.z ? f.src1 : f.src0, w.rep) + (size_t)(pos - pos_off) * c.N + (size_t)w.tile * 4096 : X;
    u64x2_t v[8], vn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ntld(reinterpret_cast<const u64x2_t*>(S + 2 * tid + 512 * i));
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      u64* __restrict__ Xt = X + (size_t)t * 4096;
      const u32 o = (w.tile + (u32)t) * 16 + b;
      auto pf = [&]() {
        if (t + 1 < T) {
#pragma unroll
          for (int i = 0; i < 8; ++i) vn[i] = ntld(reinterpret_cast<const u64x2_t*>(S + (size_t)(t + 1) * 4096 + 2 * tid + 512 * i));
        }
      };
      if (fp)         contig_inv_body_fp<false, true>(Xt, nullptr, TWD, lds, s8, o, b, lo4, q, nullptr, nullptr, nullptr, v, pf);
      else if (small) NTT_PIPE_INT((contig_inv_body<true, false, TW8, false, true>(Xt, nullptr, TW, TP, lds, s8, o, b, lo4, q, nullptr, nullptr, nullptr, v, pf)));
      else            NTT_PIPE_INT((contig_inv_body<false, false, TW8, false, true>(Xt, nullptr, TW, TP, lds, s8, o, b, lo4, q, nullptr, nullptr, nullptr, v, pf)));
      if (t + 1 < T) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = vn[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// NARROW passes for small launches: 1024-coefficient tiles, 4 coefficients per lane, four radix-4 rounds per pass.
// Most transforms of the workload cover 1..30 limbs (a Rescale's last limb, the P-limbs of a ModDown, low levels): with
// 4096-coefficient tiles that is 16..500 workgroups of one wave per SIMD, each wave running 128 butterflies one after the
// other at a lone wave's issue rate (a v_mad_u64_u32 every 9 cycles, profiles/r02v): ~9.5 us per pass whatever the size.
// The same work cut into 4x as many workgroups of 16 butterflies per wave finishes in about half the time and fills 4x as
// many CUs.  Same butterflies, same twiddle tables, same lazy ranges as the wide passes: results are bit-identical.
//   STRIDED tile: 256 rows x 4 adjacent columns (a 32-byte segment per row); thread t = 4*g + c; rounds hold rows
//       g + 64k | 64a + j + 16k | 16a' + j' + 4k | 4g + k   (k = 0..3), exchanged through LDS (row pitch 5: conflict-free)
//   CONTIG tile: 4 blocks of 256 contiguous coefficients, ONE WAVE PER BLOCK (lane l): rounds hold rho =
//       l + 64k | 64(l>>4) + (l&15) + 16k | 16(l>>2) + (l&3) + 4k | 4l + k; the exchanges stay inside the wave (its own
//       LDS region, no workgroup barrier)
// Every round is the same radix-4 step on 4 registers: stage A pairs (0,2),(1,3) with twiddle tA, stage B pairs (0,1) with
// tB0 and (2,3) with tB1, where tA = TW[2^s + p], tB_i = TW[2^(s+1) + 2p + i] for the round's first stage s and the index p of
// the lane's butterfly group at that stage.
// ------------------------------------------------------------------------------------------------
constexpr u32 kNarrowPitch = 5;  // strided narrow tile: 256 rows x 4 columns, row pitch 5 words

struct Tw3 {
  Tw a, b0, b1;
};
// scalar (wave-uniform p: constant address space) or per-lane twiddles of one round
template <bool SMALL, bool UNIFORM>
__device__ __forceinline__ Tw3 load_tw3(const ulong2* __restrict__ TW, u32 s, u32 p) {
#if NTT_EXP & 4
  return Tw3{ldtw<SMALL>(TW, 1), ldtw<SMALL>(TW, 2), ldtw<SMALL>(TW, 3)};
#else
  const u32 ia = (1u << s) + p, ib = (2u << s) + 2 * p;
  if (UNIFORM) {
    ctw_ptr T = (ctw_ptr)(reinterpret_cast<const u64*>(TW));
    auto ld = [&](u32 i) {
      const u64 w = T[2 * i], pp = T[2 * i + 1];
      return Tw{w, SMALL ? pp >> 1 : pp};
    };
    return Tw3{ld(ia), ld(ib), ld(ib + 1)};
  }
  return Tw3{ldtw<SMALL>(TW, ia), ldtw<SMALL>(TW, ib), ldtw<SMALL>(TW, ib + 1)};
#endif
}
template <bool SMALL>
__device__ __forceinline__ void radix4_fwd(u64 (&x)[4], const Tw3& t, const BfK& k) {
  bf_fwd<SMALL>(x[0], x[2], t.a, k);
  bf_fwd<SMALL>(x[1], x[3], t.a, k);
  bf_fwd<SMALL>(x[0], x[1], t.b0, k);
  bf_fwd<SMALL>(x[2], x[3], t.b1, k);
}
// inverse: stage B first, then stage A (skipped when the caller folds it: the very last stage carries N^-1)
template <bool SMALL, bool WITH_A>
__device__ __forceinline__ void radix4_inv(u64 (&x)[4], const Tw3& t, const BfK& k) {
  bf_inv<SMALL>(x[0], x[1], t.b0, k);
  bf_inv<SMALL>(x[2], x[3], t.b1, k);
  if (WITH_A) {
    bf_inv<SMALL>(x[0], x[2], t.a, k);
    bf_inv<SMALL>(x[1], x[3], t.a, k);
  }
}

// ---- strided narrow pass (stages 0..7).  lds: 256 * kNarrowPitch words.
template <bool SMALL, bool INVERSE, int SRC>
__device__ __forceinline__ void strided4_body(u64* __restrict__ X, const ulong2* __restrict__ TW, u64* lds, const DevPrime& P,
                                              const NttFuse& f, const int64_t* __restrict__ M, u32 pos, u32 tile, u64 q, u32 n_words) {
  const BfK bk = bf_consts<SMALL>(q);
  const u32 t = threadIdx.x, c = t & 3, g = t >> 2;
  const u32 col = tile * 4 + c;
  // the round layouts: row of register k
  const u32 a2 = __builtin_amdgcn_readfirstlane(t >> 6), j2 = (t >> 2) & 15;  // R2: 64*a2 + j2 + 16k (a2 is wave-uniform)
  const u32 a3 = t >> 4, j3 = (t >> 2) & 3;                                    // R3: 16*a3 + j3 + 4k
  u64 x[4];
  asm volatile("" ::: "memory");
  // every round's twiddles are requested up front, next to the data: four dependent fetches would cost four memory latencies
  const Tw3 t1 = load_tw3<SMALL, true>(TW, 0, 0), t2 = load_tw3<SMALL, true>(TW, 2, a2), t3 = load_tw3<SMALL, false>(TW, 4, a3),
            t4 = load_tw3<SMALL, false>(TW, 6, g);
  if (!INVERSE) {
    if (SRC == SRC_MSG) {  // Encode_impl ckks_encoder.c:262-285, as in strided_fwd_body
      const u64 sc = f.msg_scale ? f.msg_scale[pos] : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t v = M[(size_t)(g + 64 * k) * 256 + col];
        const u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
        u64 r = mag;
        if (__any(mag >= q)) r = mag < q ? mag : reduce128(U128{mag, 0}, q, P.prec128_lo, P.prec128_hi);
        if (v < 0 && r != 0) r = q - r;
        x[k] = f.msg_scale ? mul_mod(r, sc, P) : r;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = ntld(&X[(size_t)(g + 64 * k) * 256 + col]);
    }
    radix4_fwd<SMALL>(x, t1, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(g + 64 * k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(64 * a2 + j2 + 16 * k) * kNarrowPitch + c];
    radix4_fwd<SMALL>(x, t2, bk);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(64 * a2 + j2 + 16 * k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(16 * a3 + j3 + 4 * k) * kNarrowPitch + c];
    radix4_fwd<SMALL>(x, t3, bk);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(16 * a3 + j3 + 4 * k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(4 * g + k) * kNarrowPitch + c];
    radix4_fwd<SMALL>(x, t4, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) ntst(&X[(size_t)(4 * g + k) * 256 + col], x[k]);  // lazy: < 41q SMALL, < 8q otherwise
  } else {
    // input lazy [0,lim) from the contiguous pass; stages 7,6 | 5,4 | 3,2 | 1 and stage 0 with N^-1 (or the caller's scale)
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = ntld(&X[(size_t)(4 * g + k) * 256 + col]);
    radix4_inv<SMALL, true>(x, t4, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(4 * g + k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(16 * a3 + j3 + 4 * k) * kNarrowPitch + c];
    radix4_inv<SMALL, true>(x, t3, bk);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(16 * a3 + j3 + 4 * k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(64 * a2 + j2 + 16 * k) * kNarrowPitch + c];
    radix4_inv<SMALL, true>(x, t2, bk);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) lds[(64 * a2 + j2 + 16 * k) * kNarrowPitch + c] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = lds[(g + 64 * k) * kNarrowPitch + c];
    radix4_inv<SMALL, false>(x, t1, bk);  // stage 1; stage 0 below
    Tw tn{P.n_inv, P.n_inv_prec}, tw{P.inv_w1_ninv, P.inv_w1_ninv_prec};
    if (f.inv_scale) {
      const u64* sc = f.inv_scale + 4 * (size_t)pos;
      tn = Tw{sc[0], sc[1]};
      tw = Tw{sc[2], sc[3]};
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u64 s = x[k] + x[k + 2];
      const u64 d = x[k] + bk.lim - x[k + 2];
      const u64 a = shoup_lazy(s, tn, q), b = shoup_lazy(d, tw, q);  // exact quotient: [0,2q)
      x[k] = a >= q ? a - q : a;
      x[k + 2] = b >= q ? b - q : b;
    }
    if (f.center_out) {
      const u64 half = q >> 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = x[k] > half ? x[k] - q : x[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ntst(&X[(size_t)(g + 64 * k) * 256 + col], x[k]);
  }
  asm volatile("" ::: "memory");
}

template <bool INVERSE, int SRC>
__global__ __launch_bounds__(256) void ntt4_strided_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride, u32 level, u32 pos0,
                                                        u32 pos_off, u32 skip_alpha, NttFuse f, u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[256 * kNarrowPitch];
  // the 4 tiles that share the 128-byte lines of a row (tile = 4*line + r) get block ids congruent mod 8: one XCD's L2
  const u32 b = blockIdx.x, xl = b & 7u, m = b >> 3;
  const u32 tile = 4 * (xl + 8 * ((m >> 2) & 1u)) + (m & 3u), rowi = m >> 3;
  NttWg w{tile, rowi % n_limbs, rowi / n_limbs, 0, 0, 0};
  w.y = __builtin_amdgcn_readfirstlane(w.y);
  w.z = __builtin_amdgcn_readfirstlane(w.z);
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  u64* __restrict__ X = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + (size_t)(w.pos - pos_off) * c.N;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const int64_t* M = SRC == SRC_MSG ? reb(c, f.msg, w.rep) + w.z * f.msg_stride : nullptr;
  if (q <= kSmallPrimeMax) strided4_body<true, INVERSE, SRC>(X, TW, lds, P, f, M, w.pos, w.tile, q, c.N);
  else                     strided4_body<false, INVERSE, SRC>(X, TW, lds, P, f, M, w.pos, w.tile, q, c.N);
}

// ---- contiguous narrow pass (stages 8..15): one wave per 256-coefficient block, `wl` = the wave's 256-word LDS region
// (+ 8 words of padding between the regions)
template <bool SMALL, bool INVERSE, int FUSE>
__device__ __forceinline__ void contig4_body(u64* __restrict__ Xb, const u64* __restrict__ Sb, const ulong2* __restrict__ TW, u64* wl,
                                             const DevPrime& P, const NttFuse& f, const u64* x_z, u64* out_z, u32 pos, u32 o, u64 q,
                                             size_t tail_off) {
  const BfK bk = bf_consts<SMALL>(q);
  const u32 l = threadIdx.x & 63;
  const u32 c2 = l >> 4, j2 = l & 15, c3 = l >> 2, j3 = l & 3;
  u64 x[4];
  asm volatile("" ::: "memory");
  // every round's twiddles are requested up front, next to the data
  const Tw3 t1 = load_tw3<SMALL, true>(TW, 8, o), t2 = load_tw3<SMALL, false>(TW, 10, (o << 2) + c2),
            t3 = load_tw3<SMALL, false>(TW, 12, (o << 4) + c3), t4 = load_tw3<SMALL, false>(TW, 14, (o << 6) + l);
  if (!INVERSE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = ntld(&Xb[l + 64 * k]);
    radix4_fwd<SMALL>(x, t1, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[l + 64 * k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[64 * c2 + j2 + 16 * k];
    radix4_fwd<SMALL>(x, t2, bk);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[64 * c2 + j2 + 16 * k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[16 * c3 + j3 + 4 * k];
    radix4_fwd<SMALL>(x, t3, bk);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[16 * c3 + j3 + 4 * k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[4 * l + k];
    radix4_fwd<SMALL>(x, t4, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = canon_fwd<SMALL>(x[k], q, P.prec128_hi);
    // the lane holds 4 contiguous coefficients: 32 bytes, two 16-byte accesses (Rescale / ModDown tails as in contig_pass)
    const u64* __restrict__ xin = FUSE ? x_z + tail_off + 4 * l : nullptr;
    u64* __restrict__ dst = FUSE ? out_z + tail_off + 4 * l : Xb + 4 * l;
    const u64 tw_w = FUSE ? f.w[pos] : 0, tw_p = FUSE ? f.wp[pos] : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ulong2 v{x[2 * h], x[2 * h + 1]};
      if (FUSE == 1) {
        const ulong2 xv = *reinterpret_cast<const ulong2*>(xin + 2 * h);
        v.x = add_mod(mul_shoup(xv.x, tw_w, tw_p, q), v.x, q);
        v.y = add_mod(mul_shoup(xv.y, tw_w, tw_p, q), v.y, q);
      } else if (FUSE == 2) {
        const ulong2 xv = *reinterpret_cast<const ulong2*>(xin + 2 * h);
        v.x = mul_shoup(sub_mod(xv.x, v.x, q), tw_w, tw_p, q);
        v.y = mul_shoup(sub_mod(xv.y, v.y, q), tw_w, tw_p, q);
      }
      *reinterpret_cast<ulong2*>(dst + 2 * h) = v;
    }
  } else {
    {  // 4 contiguous coefficients per lane from Sb (out of place when FUSE)
      const ulong2 v0 = *reinterpret_cast<const ulong2*>(Sb + 4 * l), v1 = *reinterpret_cast<const ulong2*>(Sb + 4 * l + 2);
      x[0] = v0.x;
      x[1] = v0.y;
      x[2] = v1.x;
      x[3] = v1.y;
    }
    radix4_inv<SMALL, true>(x, t4, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[4 * l + k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[16 * c3 + j3 + 4 * k];
    radix4_inv<SMALL, true>(x, t3, bk);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[16 * c3 + j3 + 4 * k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[64 * c2 + j2 + 16 * k];
    radix4_inv<SMALL, true>(x, t2, bk);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) wl[64 * c2 + j2 + 16 * k] = x[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = wl[l + 64 * k];
    radix4_inv<SMALL, true>(x, t1, bk);
#pragma unroll
    for (int k = 0; k < 4; ++k) ntst(&Xb[l + 64 * k], x[k]);  // lazy [0,lim): the strided pass follows
  }
  asm volatile("" ::: "memory");
}

template <bool INVERSE, int FUSE>
__global__ __launch_bounds__(256) void ntt4_contig_kernel(DevCtx c, u64* __restrict__ poly, size_t poly_stride, u32 level, u32 pos0,
                                                       u32 pos_off, u32 skip_alpha, NttFuse f, u32 n_limbs, u32 n_polys) {
  __shared__ u64 lds[4 * 264];
  const u32 b = blockIdx.x, tile = b & 63u, rowi = b >> 6;
  NttWg w{tile, rowi % n_limbs, rowi / n_limbs, 0, 0, 0};
  w.y = __builtin_amdgcn_readfirstlane(w.y);
  w.z = __builtin_amdgcn_readfirstlane(w.z);
  ntt_split_z(w, c, n_polys);
  if (!ntt_resolve(w, c, f, level, pos0, skip_alpha)) return;
  const DevPrime& P = c.primes[w.gi];
  const u64 q = uniform64(P.q);
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 o = w.tile * 4 + wave;  // 256-coefficient block of the limb
  const size_t limb_off = (size_t)(w.pos - pos_off) * c.N + (size_t)o * 256;
  u64* __restrict__ Xb = reb(c, f.polyz[0] ? f.polyz[w.z] : poly + w.z * poly_stride, w.rep) + limb_off;
  const u64* __restrict__ Sb = (INVERSE && FUSE) ? reb(c, w.z ? f.src1 : f.src0, w.rep) + limb_off : Xb;
  const u64* xin = (!INVERSE && FUSE) ? reb(c, w.z ? f.x1 : f.x0, w.rep) : nullptr;
  u64* xout = (!INVERSE && FUSE) ? reb(c, w.z ? f.out1 : f.out0, w.rep) : nullptr;
  const ulong2* __restrict__ TW = (INVERSE ? c.tw_inv : c.tw_fwd) + (size_t)w.gi * c.N;
  const size_t tail_off = (size_t)w.pos * c.N + (size_t)o * 256;
  u64* wl = lds + wave * 264;
  if (q <= kSmallPrimeMax) contig4_body<true, INVERSE, FUSE>(Xb, Sb, TW, wl, P, f, xin, xout, w.pos, o, q, tail_off);
  else                     contig4_body<false, INVERSE, FUSE>(Xb, Sb, TW, wl, P, f, xin, xout, w.pos, o, q, tail_off);
}

// small launches take the narrow passes: at most c.ntt_narrow_max_rows limb rows (limbs x polynomials)
static bool launch_ntt_narrow(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                              u32 n_polys, size_t poly_stride, u32 skip_alpha, const NttFuse& f) {
  if (n_limbs * n_polys * c.nrep > c.ntt_narrow_max_rows || f.conv != nullptr || f.km.nd != 0) return false;  // (the key inner product rides in the wide passes only)
  dim3 block(256), grid(64 * n_limbs * n_polys * c.nrep);
#define ACEHIP_N4_ARGS grid, block, 0, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys
  if (!inverse) {
    if (f.msg) hipLaunchKernelGGL((ntt4_strided_kernel<false, SRC_MSG>), ACEHIP_N4_ARGS);
    else       hipLaunchKernelGGL((ntt4_strided_kernel<false, SRC_MEM>), ACEHIP_N4_ARGS);
    if (f.epi == 1)      hipLaunchKernelGGL((ntt4_contig_kernel<false, 1>), ACEHIP_N4_ARGS);
    else if (f.epi == 2) hipLaunchKernelGGL((ntt4_contig_kernel<false, 2>), ACEHIP_N4_ARGS);
    else                 hipLaunchKernelGGL((ntt4_contig_kernel<false, 0>), ACEHIP_N4_ARGS);
  } else {
    if (f.src0) hipLaunchKernelGGL((ntt4_contig_kernel<true, 1>), ACEHIP_N4_ARGS);
    else        hipLaunchKernelGGL((ntt4_contig_kernel<true, 0>), ACEHIP_N4_ARGS);
    hipLaunchKernelGGL((ntt4_strided_kernel<true, SRC_MEM>), ACEHIP_N4_ARGS);
  }
#undef ACEHIP_N4_ARGS
  return true;
}

// the pipelined forms of both passes (T tiles per workgroup); false: this launch has a fused neighbour only the one-tile kernels know
template <int T>
static bool launch_ntt_pipe_t(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                              u32 n_polys, size_t poly_stride, u32 skip_alpha, const NttFuse& f, bool tw8) {
  dim3 block(256), grid(((c.N >> 12) / T) * n_limbs * n_polys * c.nrep);
#define ACEHIP_NP_ARGS grid, block, 0, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys
#define ACEHIP_NP_CONTIG(INV, FUSE)                                                                               \
  do {                                                                                                            \
    if (tw8) hipLaunchKernelGGL((ntt8_contig_pipe_kernel<INV, FUSE, true, T>), ACEHIP_NP_ARGS);                   \
    else     hipLaunchKernelGGL((ntt8_contig_pipe_kernel<INV, FUSE, false, T>), ACEHIP_NP_ARGS);                  \
  } while (0)
  // ACEHIP_NTT_PIPE_CONTIG=1: the contiguous pass walks tiles too (its twiddles depend on the tile: measured slower); default: only the strided pass
  static const bool contig_too = [] { const char* e = getenv("ACEHIP_NTT_PIPE_CONTIG"); return e && atoi(e) != 0; }();
  // ACEHIP_NTT_PIPE_INV=1: the inverse strided pass walks too (measured 5-8 % slower than one tile per workgroup, profiles/r06k_*)
  static const bool inverse_too = [] { const char* e = getenv("ACEHIP_NTT_PIPE_INV"); return e && atoi(e) != 0; }();
  if (inverse && !inverse_too && !contig_too) return false;
  dim3 grid1((c.N >> 12) * n_limbs * n_polys * c.nrep);
#define ACEHIP_N1_ARGS grid1, block, 0, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys
#define ACEHIP_N1_CONTIG(INV, CANON, FUSE)                                                                        \
  do {                                                                                                            \
    if (tw8) hipLaunchKernelGGL((ntt8_contig_kernel<INV, CANON, FUSE, true, true>), ACEHIP_N1_ARGS);              \
    else     hipLaunchKernelGGL((ntt8_contig_kernel<INV, CANON, FUSE, false, true>), ACEHIP_N1_ARGS);             \
  } while (0)
  if (!inverse) {
    if (f.msg || f.conv) return false;
    hipLaunchKernelGGL((ntt8_strided_pipe_kernel<false, T>), ACEHIP_NP_ARGS);
    if (contig_too) {
      if (f.epi == 1)      ACEHIP_NP_CONTIG(false, 1);
      else if (f.epi == 2) ACEHIP_NP_CONTIG(false, 2);
      else if (f.epi == 3) ACEHIP_NP_CONTIG(false, 3);
      else                 ACEHIP_NP_CONTIG(false, 0);
    } else {
      if (f.epi == 1)      ACEHIP_N1_CONTIG(false, true, 1);
      else if (f.epi == 2) ACEHIP_N1_CONTIG(false, true, 2);
      else if (f.epi == 3) ACEHIP_N1_CONTIG(false, true, 3);
      else                 ACEHIP_N1_CONTIG(false, true, 0);
    }
  } else {
    if (f.km.nd && contig_too) return false;
    if (contig_too) {
      if (f.src0) ACEHIP_NP_CONTIG(true, 1);
      else        ACEHIP_NP_CONTIG(true, 0);
    } else {
      if (f.km.nd)     ACEHIP_N1_CONTIG(true, false, 2);
      else if (f.src0) ACEHIP_N1_CONTIG(true, false, 1);
      else             ACEHIP_N1_CONTIG(true, false, 0);
    }
    if (inverse_too) hipLaunchKernelGGL((ntt8_strided_pipe_kernel<true, T>), ACEHIP_NP_ARGS);
    else             hipLaunchKernelGGL((ntt8_strided_kernel<true, SRC_MEM>), ACEHIP_N1_ARGS);
  }
#undef ACEHIP_N1_CONTIG
#undef ACEHIP_N1_ARGS
#undef ACEHIP_NP_CONTIG
#undef ACEHIP_NP_ARGS
  return true;
}
static bool launch_ntt_pipe(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                            u32 n_polys, size_t poly_stride, u32 skip_alpha, const NttFuse& f, bool tw8, u32 T) {
  if (c.logN != 16 || c.twp_fwd == nullptr || c.twp_inv == nullptr) return false;  // (the walk keeps companion-only twiddles)
#if NTT_PIPE_T4
  if (T == 4) return launch_ntt_pipe_t<4>(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha, f, tw8);
#endif
  return launch_ntt_pipe_t<2>(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha, f, tw8);
}

// logN >= 16 uses both fast passes when logN == 16; the contig pass alone serves the last 8 stages of
// any logN >= 13 (the generic LDS kernel does the leading logN-8 stages).
void launch_ntt_fast(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s,
                     u32 pos_off, u32 n_polys, size_t poly_stride, u32 skip_alpha) {
  launch_ntt_fused(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha, NttFuse{});
}

void launch_ntt_fused(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s, u32 pos_off,
                      u32 n_polys, size_t poly_stride, u32 skip_alpha, const NttFuse& f) {
  ACEHIP_ABLATE(ABL_NTT);
  if (n_limbs == 0) return;
  ntt_count((u64)n_limbs * n_polys * c.nrep);
  if (launch_ntt_narrow(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha, f)) return;
  dim3 block(256), grid((c.N >> 12) * n_limbs * n_polys * c.nrep);  // 1-D: ntt_block() maps it XCD-aware
  const bool tw8 = c.twp_fwd != nullptr && n_polys * c.nrep <= c.tw8_max_polys;  // few polynomials share the twiddles: 8-byte stream
  // ACEHIP_NTT_LDS_PAD (bytes of unused dynamic LDS per workgroup; experiment): caps the workgroups of these VALU-bound passes
  // per CU below the 4 their registers allow, which leaves registers for the memory-bound kernels of OTHER image streams to
  // run beside them on the same CU (0 = off)
  static const size_t lds_pad = [] { const char* e = getenv("ACEHIP_NTT_LDS_PAD"); return e ? (size_t)strtoul(e, nullptr, 0) : (size_t)0; }();
#define ACEHIP_NTT_ARGS grid, block, lds_pad, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys
  // ACEHIP_NTT_PIPE = 2 / 4: the pipelined forms (a workgroup walks that many tiles of a limb, the next tile's loads in flight during
  // the butterflies); 0: one tile per workgroup.  Bit-identical either way.
  static const u32 pipe = [] { const char* e = getenv("ACEHIP_NTT_PIPE"); const u32 v = e ? (u32)strtoul(e, nullptr, 0) : 0u; return (v == 2 || (v == 4 && NTT_PIPE_T4)) ? v : 0u; }();
  if (pipe && launch_ntt_pipe(c, poly, level, pos0, n_limbs, inverse, s, pos_off, n_polys, poly_stride, skip_alpha, f, tw8, pipe)) return;
  if (!inverse) {
    if (f.msg)                     hipLaunchKernelGGL((ntt8_strided_kernel<false, SRC_MSG>), ACEHIP_NTT_ARGS);
    else if (f.conv && f.conv_max_in <= 4)  hipLaunchKernelGGL((ntt8_strided_kernel<false, SRC_CONV4>), ACEHIP_NTT_ARGS);
    else if (f.conv && f.conv_max_in <= 8)  hipLaunchKernelGGL((ntt8_strided_kernel<false, SRC_CONV8>), ACEHIP_NTT_ARGS);
    else if (f.conv)               hipLaunchKernelGGL((ntt8_strided_kernel<false, SRC_CONV12>), ACEHIP_NTT_ARGS);
    else                           hipLaunchKernelGGL((ntt8_strided_kernel<false, SRC_MEM>), ACEHIP_NTT_ARGS);
    // (FP_OK: the strided pass above was this transform's first pass, so limbs of the FP class arrive as doubles)
    if (tw8) {
      if (f.epi == 1)      hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 1, true, true>), ACEHIP_NTT_ARGS);
      else if (f.epi == 2) hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 2, true, true>), ACEHIP_NTT_ARGS);
      else if (f.epi == 3) hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 3, true, true>), ACEHIP_NTT_ARGS);
      else                 hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 0, true, true>), ACEHIP_NTT_ARGS);
    } else {
      if (f.epi == 1)      hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 1, false, true>), ACEHIP_NTT_ARGS);
      else if (f.epi == 2) hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 2, false, true>), ACEHIP_NTT_ARGS);
      else if (f.epi == 3) hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 3, false, true>), ACEHIP_NTT_ARGS);
      else                 hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 0, false, true>), ACEHIP_NTT_ARGS);
    }
  } else {
    if (tw8) {
      if (f.km.nd)     hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 2, true, true>), ACEHIP_NTT_ARGS);
      else if (f.src0) hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 1, true, true>), ACEHIP_NTT_ARGS);
      else             hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 0, true, true>), ACEHIP_NTT_ARGS);
    } else {
      if (f.km.nd)     hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 2, false, true>), ACEHIP_NTT_ARGS);
      else if (f.src0) hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 1, false, true>), ACEHIP_NTT_ARGS);
      else             hipLaunchKernelGGL((ntt8_contig_kernel<true, false, 0, false, true>), ACEHIP_NTT_ARGS);
    }
    hipLaunchKernelGGL((ntt8_strided_kernel<true, SRC_MEM>), ACEHIP_NTT_ARGS);
  }
#undef ACEHIP_NTT_ARGS
}

void launch_ntt_contig8(const DevCtx& c, u64* poly, u32 level, u32 pos0, u32 n_limbs, bool inverse, hipStream_t s,
                        u32 pos_off, u32 n_polys, size_t poly_stride, u32 skip_alpha) {
  ACEHIP_ABLATE(ABL_NTT);
  dim3 block(256), grid((c.N >> 12) * n_limbs * n_polys * c.nrep);
  const NttFuse f{};
  if (!inverse) hipLaunchKernelGGL((ntt8_contig_kernel<false, true, 0>), grid, block, 0, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys);
  else          hipLaunchKernelGGL((ntt8_contig_kernel<true, true, 0>), grid, block, 0, s, c, poly, poly_stride, level, pos0, pos_off, skip_alpha, f, n_limbs, n_polys);
}

}  // namespace acehip
