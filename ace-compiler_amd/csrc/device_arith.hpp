// device_arith.hpp -- 64-bit modular arithmetic for gfx950 (wave64, no MFMA: integer work).
//
// All routines return canonical residues in [0, q), so they are bit-identical to the reference's
// Barrett / Shoup code (fhe_utils.h:241-318) whatever the reduction strategy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace acehip {

using u64 = uint64_t;
using u32 = uint32_t;

// Per-prime constants in HBM (layout shared with host_params.hpp::PrimeConsts)
struct DevPrime {
  u64 q;
  u64 barrett_mu;
  u32 nbits;
  u32 pad;
  u64 prec128_lo, prec128_hi;
  u64 psi;
  u64 n_inv, n_inv_prec;
  u64 inv_w1_ninv, inv_w1_ninv_prec;
};

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q) {
  u64 s = a + b;
  return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Shoup multiplication by a constant w with companion wp = floor(w*2^64/q); a may be any 64-bit
// value, result canonical (reference Fast_mul_const_with_mod, fhe_utils.h:311-318).
__device__ __forceinline__ u64 mul_shoup(u64 a, u64 w, u64 wp, u64 q) {
  u64 hi = mulhi64(a, wp);
  u64 r  = a * w - hi * q;
  return r >= q ? r - q : r;
}
// lazy variant: result in [0, 2q)
__device__ __forceinline__ u64 mul_shoup_lazy(u64 a, u64 w, u64 wp, u64 q) {
  u64 hi = mulhi64(a, wp);
  return a * w - hi * q;
}

// a*b mod q for a,b < q < 2^61: single-word Barrett on the shifted product.
//   x = a*b < 2^(2n); x1 = x >> (n-1); qhat = mulhi(x1, mu << (63-n)) = floor(x1*mu / 2^(n+1));
//   x - qhat*q in [0, 3q)
// (full products are written as one unsigned __int128 multiply: the compiler then emits 4 v_mad_u64_u32 and shares
// the partial products between the halves; a separate a*b and __umul64hi(a,b) costs 2 extra quarter-rate v_mul_lo_u32)
__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, u64 q, u64 mu, u32 nbits) {
  const unsigned __int128 p = (unsigned __int128)a * b;
  u64 lo = (u64)p, hi = (u64)(p >> 64);
  u32 sh = nbits - 1;
  u64 x1 = (hi << (64 - sh)) | (lo >> sh);
  u64 qh = mulhi64(x1, mu);
  u64 r  = lo - qh * q;
  r = r >= 2 * q ? r - 2 * q : r;
  return r >= q ? r - q : r;
}
__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, const DevPrime& p) {
  return mul_mod(a, b, p.q, p.barrett_mu, p.nbits);
}

// 128-bit accumulator for the base-conversion sums
struct U128 {
  u64 lo, hi;
};
__device__ __forceinline__ void mac128(U128& acc, u64 a, u64 b) {
  unsigned __int128 v = ((unsigned __int128)acc.hi << 64) | acc.lo;
  v += (unsigned __int128)a * b;
  acc.lo = (u64)v;
  acc.hi = (u64)(v >> 64);
}

// (hi:lo) mod q with mu = floor(2^128/q) = (mh:ml), canonical result (the reference reduces the same sums with
// Mod_barrett_128, fhe_utils.h:241-280; any exact reduction gives the same residue).
// Generic form: the reference's quotient estimate, at most 2 below the true quotient.
__device__ __forceinline__ u64 reduce128_generic(U128 v, u64 q, u64 ml, u64 mh) {
  // qhat = floor( (v.hi:v.lo) * (mh:ml) / 2^128 ) up to the dropped low partial product
  const unsigned __int128 mid = (unsigned __int128)v.lo * mh + mulhi64(v.lo, ml);  // < 2^128
  const unsigned __int128 m2 = (unsigned __int128)v.hi * ml + (u64)mid;            // carries into bit 64
  u64 qhat = v.hi * mh + (u64)(mid >> 64) + (u64)(m2 >> 64);
  u64 r = v.lo - qhat * q;
  while (r >= q) r -= q;
  return r;
}
// Primes above 2^32 (every prime of a real parameter set) have mh < 2^32, and the base-conversion / BSGS kernels spend
// more instructions in this reduction than in their multiply-adds (profiles/r02y: base_conv_batch16_kernel issues VALU
// instructions 88 % of the time), so the quotient is assembled from the three partial products that matter:
//   floor(v*mu/2^128) = v1*mh + floor((v1*ml + v0*mh + floor(v0*ml/2^64)) / 2^64)
//                     = lo64(v1*mh) + hi64(v1*ml) + hi64(v0*mh) + c,  c in {0,1,2}   (mod 2^64; the true quotient may exceed 64
// bits, only its low word enters r), i.e. at most 2 + 2 below the true quotient: r = v0 - qhat*q lies in [0,5q) (q < 2^61,
// host_params) and three conditional subtractions make it canonical -- no 128-bit adds, no data-dependent loop.
__device__ __forceinline__ u64 reduce128(U128 v, u64 q, u64 ml, u64 mh) {
  if (mh >> 32) return reduce128_generic(v, q, ml, mh);  // wave-uniform: a property of the prime
  const u32 m = (u32)mh;
  const u32 v0l = (u32)v.lo, v0h = (u32)(v.lo >> 32), v1l = (u32)v.hi, v1h = (u32)(v.hi >> 32);
  u64 A = (u64)v1l * m;                          // lo64(v1 * mh)
  A += (u64)(u32)((u64)v1h * m) << 32;
  const u64 B = mulhi64(v.hi, ml);               // hi64(v1 * ml)
  const u64 t1 = (u64)v0l * m;                   // hi64(v0 * mh) = (v0h*m + hi32(v0l*m)) >> 32
  const u64 t2 = (u64)v0h * m + (t1 >> 32);
  const u64 qhat = A + B + (t2 >> 32);
  u64 r = v.lo - qhat * q;
  const u64 q4 = 4 * q, q2 = 2 * q;
  r = r >= q4 ? r - q4 : r;
  r = r >= q2 ? r - q2 : r;
  return r >= q ? r - q : r;
}

// Switch_modulus (fhe_utils.h:349-375): centred lift of v in [0,old_q) to [0,new_q)
__device__ __forceinline__ u64 switch_modulus(u64 v, u64 old_q, u64 new_q) {
  u64 half = old_q >> 1;
  if (new_q > old_q) return v > half ? v + (new_q - old_q) : v;
  u64 diff = new_q - (old_q % new_q);
  u64 r = v > half ? v + diff : v;
  return r >= new_q ? r % new_q : r;
}

}  // namespace acehip
