// ntt_fp.hpp -- a second arithmetic for the butterflies of the register-tiled NTT passes (ntt_fast.hip): FP64 for primes below 2^50 + 2^43.
//
// The passes are bound by VALU issue as much as by memory: a forward butterfly of the SMALL integer class costs 15
// instructions, 9 of them v_mad_u64_u32.  The 48..50-bit scaling primes (33 of the 34 q-limbs of the generated ResNets) also fit the
// FP64 pipe, which issues at the same rate: residues are integers below 2^53, exactly representable, and
//     h = x*w;  l = fma(x, w, -h);            x*w = h + l exactly (one rounding, recovered by the fma)
//     f = rndne(h * (1/q));  r = fma(-f, q, h);  t = r + l      t = x*w - f*q: an integer congruent to x*w, every step exact
// is a twiddle product in 6 instructions.  Range, with rho = q / 2^50 <= 1.0079 (kFpPrimeMax) and |x| <= B*q: the computed quotient
// h*(1/q) carries three roundings of a value of size B*q, i.e. an absolute error of 3 * 2^-53 * B * q = 0.375*rho*B, so
// |t| <= (0.5 + 0.375*rho*B)*q; every intermediate must stay an integer below 2^53 = (8/rho)*q >= 7.93q.
// Forward (Cooley-Tukey): X' = X + t, Y' = X - t, no reduction for the FOUR stages of a radix-16 round.  The FIRST round of a transform
// starts from canonical residues in [0, q) (B = 1, not centred): 1.88q, 3.09q, 4.76q, 7.06q < 7.93q; every later round starts from
// |v| <= 0.51q: 1.21q, 2.16q, 3.49q, 5.31q.  Then v - rndne(v/q)*q (3 instructions) brings every value back to |v| <= 0.51q:
// 8 + 48/32 = 9.5 instructions per butterfly.  Inverse (Gentleman-Sande): sums double per stage, so the eight sums of the second stage
// are reduced as well -- from |v| <= 0.51q: 1.02q, 2.04q -> 0.51q, 2.55q, 5.09q.  From uncentred residues the last sums could reach
// 8.04q, so the first inverse round CENTRES its input (one fp_red per value on load: contig_inv_body_fp).
// (Round 4 admitted primes up to 1.125 * 2^50 with a bound that assumed centred inputs everywhere: the shipped primes -- 2^50 +- 2^26 --
// were safe, the admitted range was not; found in review, fixed by the cap and the centring above.)
// Measured register-resident against the integer class (tools/ubench_bf_fp64.hip, profiles/r04b_*): 1.52x the butterflies per second,
// bit-identical results.
//
// Between the two passes of a transform the intermediate lies in memory as FP64 bit patterns (|v| <= 0.51q); both passes decide the
// class from the prime alone, so they always agree.  The first pass converts canonical residues on load (2 v_cvt_f64_u32 + 1 fma), the
// last pass produces canonical u64 residues (or the centred lift) again: same bits as the integer classes and as the reference
// (ntt.c:190-353), which every parity test of the NTT checks with the class on and off (ACEHIP_NTT_FP=0).
#pragma once

namespace acehip {

constexpr u64 kFpPrimeMax = (1ull << 50) + (1ull << 43);  // rho <= 1.0079: 7.06q (forward from [0,q)) and 5.09q (inverse) stay below 2^53

struct FpK {
  double q, qinv;
};
__device__ __forceinline__ double fp_from_u64(u64 v) { return __builtin_fma((double)(u32)(v >> 32), 4294967296.0, (double)(u32)v); }
__device__ __forceinline__ u64 fp_bits(double v) { return __builtin_bit_cast(u64, v); }
__device__ __forceinline__ double fp_of_bits(u64 v) { return __builtin_bit_cast(double, v); }
__device__ __forceinline__ FpK fp_consts(u64 q) {
  const double qd = fp_from_u64(q);
  return FpK{qd, 1.0 / qd};  // (wave-uniform; the division is a handful of instructions once per workgroup)
}
// x*w - rndne(x*w/q)*q for integer-valued |x| < 2^53, 0 <= w < q
__device__ __forceinline__ double fp_mulmod(double x, double w, const FpK& k) {
  const double h = x * w;
  const double l = __builtin_fma(x, w, -h);
  const double f = __builtin_rint(h * k.qinv);
  const double r = __builtin_fma(-f, k.q, h);
  return r + l;
}
__device__ __forceinline__ double fp_red(double v, const FpK& k) { return __builtin_fma(-__builtin_rint(v * k.qinv), k.q, v); }
// canonical residue in [0,q) of an integer-valued |v| < 2^53, as u64
__device__ __forceinline__ double fp_canon_f(double v, const FpK& k) {
  v = fp_red(v, k);              // |v| <= 0.51q
  return v < 0 ? v + k.q : v;    // [0, q): v + q is an integer below 2^53, exact
}
__device__ __forceinline__ u64 fp_to_u64(double c) {  // integer-valued 0 <= c < 2^52: the integer is the mantissa of c + 2^52
  return fp_bits(c + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;
}
__device__ __forceinline__ void fp_bf_fwd(double& X, double& Y, double w, const FpK& k) {
  const double t = fp_mulmod(Y, w, k), x = X;
  X = x + t;
  Y = x - t;
}
__device__ __forceinline__ void fp_bf_inv(double& X, double& Y, double w, const FpK& k) {
  const double s = X + Y, d = X - Y;
  X = s;
  Y = fp_mulmod(d, w, k);
}

// 4 forward stages on 16 registers (same pairing and twiddle indices as radix16_fwd), inputs |v| <= 0.51q, outputs reduced
__device__ __forceinline__ void fp_radix16_fwd(double (&x)[16], double t0, const double (&t1)[2], const double (&t2)[4], const double (&t3)[8],
                                               const FpK& k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) fp_bf_fwd(x[i], x[i + 8], t0, k);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) fp_bf_fwd(x[8 * g + i], x[8 * g + i + 4], t1[g], k);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 2; ++i) fp_bf_fwd(x[4 * g + i], x[4 * g + i + 2], t2[g], k);
#pragma unroll
  for (int g = 0; g < 8; ++g) fp_bf_fwd(x[2 * g], x[2 * g + 1], t3[g], k);
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_red(x[i], k);
}
// inverse stages 3, 2, 1 (as radix16_inv_321); the sums of the second of them are reduced.  Inputs |v| <= 0.51q; on return
// |v| <= 2.53q: stage 0 follows (fp_radix16_inv_0, or the caller's folded last stage)
__device__ __forceinline__ void fp_radix16_inv_321(double (&x)[16], const double (&t1)[2], const double (&t2)[4], const double (&t3)[8],
                                                   const FpK& k) {
#pragma unroll
  for (int g = 0; g < 8; ++g) fp_bf_inv(x[2 * g], x[2 * g + 1], t3[g], k);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fp_bf_inv(x[4 * g + i], x[4 * g + i + 2], t2[g], k);
      x[4 * g + i] = fp_red(x[4 * g + i], k);
    }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) fp_bf_inv(x[8 * g + i], x[8 * g + i + 4], t1[g], k);
}
__device__ __forceinline__ void fp_radix16_inv_0(double (&x)[16], double t0, const FpK& k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) fp_bf_inv(x[i], x[i + 8], t0, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = fp_red(x[i], k);
}

// twiddles as doubles, [L+K][N] like the integer tables (same indices): per lane ...
__device__ __forceinline__ void fp_load_tw(const double* __restrict__ TW, u32 sbase, u32 prefix, double& t0, double (&t1)[2], double (&t2)[4],
                                           double (&t3)[8]) {
#if defined(NTT_EXP) && (NTT_EXP & 32)  // timing experiment (results are wrong): every per-lane twiddle load hits the same 2 KiB (what the twiddle stream costs)
#define ACEHIP_TWIDX(i) ((i) & 255u)
#else
#define ACEHIP_TWIDX(i) (i)
#endif
  t0 = TW[ACEHIP_TWIDX((1u << sbase) + prefix)];
#pragma unroll
  for (int i = 0; i < 2; ++i) t1[i] = TW[ACEHIP_TWIDX((2u << sbase) + (prefix << 1) + i)];
#pragma unroll
  for (int i = 0; i < 4; ++i) t2[i] = TW[ACEHIP_TWIDX((4u << sbase) + (prefix << 2) + i)];
#pragma unroll
  for (int i = 0; i < 8; ++i) t3[i] = TW[ACEHIP_TWIDX((8u << sbase) + (prefix << 3) + i)];
}
// ... and the 15 twiddles of stages 0..3, the same for every lane: scalar loads through the constant address space
typedef const __attribute__((address_space(4))) double* fp_ctw_ptr;
__device__ __forceinline__ void fp_load_tw_uniform(const double* __restrict__ TW, double& t0, double (&t1)[2], double (&t2)[4], double (&t3)[8]) {
  fp_ctw_ptr T = (fp_ctw_ptr)TW;
  t0 = T[1];
#pragma unroll
  for (int i = 0; i < 2; ++i) t1[i] = T[2 + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) t2[i] = T[4 + i];
#pragma unroll
  for (int i = 0; i < 8; ++i) t3[i] = T[8 + i];
}

}  // namespace acehip
