// rt_kernels.hip -- setup-side kernels used by key generation, encryption and encoding
// (one-time or per-input work, not the per-op hot path).
#include "kernels.hpp"

namespace acehip {

// out[pos][n] = vals[n] mod q (canonical), vals signed 64-bit
__global__ __launch_bounds__(256) void values_to_rns_kernel(DevCtx c, u64* out, const int64_t* vals, u32 level, u32 pos0) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  out = reb(c, out, c.rep0 + blockIdx.z);
  vals = reb(c, vals, c.rep0 + blockIdx.z);
  const DevPrime& P = c.primes[limb_prime(pos, level, c.L)];
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  if (n >= c.N) return;
  const int64_t v = vals[n];
  const u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
  u64 r = mag < P.q ? mag : reduce128(U128{mag, 0}, P.q, P.prec128_lo, P.prec128_hi);
  if (v < 0 && r != 0) r = P.q - r;
  out[(size_t)pos * c.N + n] = r;
}

// out[n] = centred representative of in[n] mod prime gi (as int64); in place allowed
__global__ __launch_bounds__(256) void center_kernel(DevCtx c, int64_t* out, const u64* in, u32 gi) {
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  if (n >= c.N || !owns(c, gi)) return;
  out = reb(c, out, c.rep0 + blockIdx.z);
  in = reb(c, in, c.rep0 + blockIdx.z);
  const u64 q = c.primes[gi].q, v = in[n];
  out[n] = (int64_t)(v > (q >> 1) ? v - q : v);
}
void launch_center(const DevCtx& c, int64_t* out, const u64* in, u32 gi, hipStream_t s) {
  hipLaunchKernelGGL(center_kernel, dim3((c.N + 255) / 256, 1, c.nrep), dim3(256), 0, s, c, out, in, gi);
}

void launch_values_to_rns(const DevCtx& c, u64* out, const int64_t* vals, u32 level, u32 pos0, u32 n_limbs, hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  if (n_limbs == 0) return;
  dim3 grid((c.N + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(values_to_rns_kernel, grid, block, 0, s, c, out, vals, level, pos0);
}

__device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// uniform in [0,q): 128 random bits reduced mod q (bias < 2^-60)
__global__ __launch_bounds__(256) void sample_uniform_kernel(DevCtx c, u64* out, u32 level, u32 pos0, u64 seed) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  out = reb(c, out, c.rep0 + blockIdx.z);  // (every replica gets the same sample: a launch over several replicas is a broadcast)
  const DevPrime& P = c.primes[limb_prime(pos, level, c.L)];
  const u32 n = blockIdx.x * 256 + threadIdx.x;
  if (n >= c.N) return;
  const u64 ctr = ((u64)pos << 32) | n;
  const u64 a = mix64(seed + 0x9E3779B97F4A7C15ull * (2 * ctr + 1));
  const u64 b = mix64(a ^ (seed * 0xD1342543DE82EF95ull + 2 * ctr + 2));
  out[(size_t)pos * c.N + n] = reduce128(U128{a, b >> 4}, P.q, P.prec128_lo, P.prec128_hi);
}

void launch_sample_uniform(const DevCtx& c, u64* out, u32 level, u32 pos0, u32 n_limbs, u64 seed, hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  if (n_limbs == 0) return;
  dim3 grid((c.N + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(sample_uniform_kernel, grid, block, 0, s, c, out, level, pos0, seed);
}

// uniform in [0,q) under a 256-bit key: one ChaCha20 block (RFC 8439 2.3: constants | key | counter | nonce, 20 rounds, + input) per four
// coefficients, each of its four 128-bit quarters reduced mod q (124 bits kept: bias < 2^-60).  counter = t (coefficients 4t .. 4t+3),
// nonce = (limb position, "UNIF", 0): every (key, position, t) names one block, whatever the launch shape.
__device__ __forceinline__ u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define ACEHIP_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
__global__ __launch_bounds__(256) void sample_uniform_keyed_kernel(DevCtx c, u64* out, u32 level, u32 pos0, ChaChaKey key) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  out = reb(c, out, c.rep0 + blockIdx.z);  // (every replica gets the same sample)
  const DevPrime& P = c.primes[limb_prime(pos, level, c.L)];
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (4 * t >= c.N) return;
  const u32 s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u, s13 = pos, s14 = 0x554E4946u, s15 = 0;
  u32 x0 = s0, x1 = s1, x2 = s2, x3 = s3, x4 = key.w[0], x5 = key.w[1], x6 = key.w[2], x7 = key.w[3], x8 = key.w[4], x9 = key.w[5],
      x10 = key.w[6], x11 = key.w[7], x12 = t, x13 = s13, x14 = s14, x15 = s15;
  for (int i = 0; i < 10; ++i) {
    ACEHIP_QR(x0, x4, x8, x12) ACEHIP_QR(x1, x5, x9, x13) ACEHIP_QR(x2, x6, x10, x14) ACEHIP_QR(x3, x7, x11, x15)
    ACEHIP_QR(x0, x5, x10, x15) ACEHIP_QR(x1, x6, x11, x12) ACEHIP_QR(x2, x7, x8, x13) ACEHIP_QR(x3, x4, x9, x14)
  }
  const u32 w[16] = {x0 + s0, x1 + s1, x2 + s2, x3 + s3, x4 + key.w[0], x5 + key.w[1], x6 + key.w[2], x7 + key.w[3], x8 + key.w[4],
                     x9 + key.w[5], x10 + key.w[6], x11 + key.w[7], x12 + t, x13 + s13, x14 + s14, x15 + s15};
  u64* o = out + (size_t)pos * c.N + 4 * (size_t)t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u64 lo = (u64)w[4 * j] | ((u64)w[4 * j + 1] << 32), hi = (u64)w[4 * j + 2] | ((u64)w[4 * j + 3] << 32);
    o[j] = reduce128(U128{lo, hi >> 4}, P.q, P.prec128_lo, P.prec128_hi);
  }
}
#undef ACEHIP_QR
void launch_sample_uniform_keyed(const DevCtx& c, u64* out, u32 level, u32 pos0, u32 n_limbs, const ChaChaKey& key, hipStream_t s) {
  ACEHIP_ABLATE(ABL_OTHER);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 4 + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(sample_uniform_keyed_kernel, grid, block, 0, s, c, out, level, pos0, key);
}

__global__ __launch_bounds__(256) void mul_scalars_kernel(DevCtx c, u64* r, const u64* a, LimbConsts w, u32 level, u32 pos0) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  r = reb(c, r, c.rep0 + blockIdx.z);
  a = reb(c, a, c.rep0 + blockIdx.z);
  const DevPrime P = c.primes[limb_prime(pos, level, c.L)];
  const u64 wl = w.w[blockIdx.y];
  const size_t base = (size_t)pos * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 v = *reinterpret_cast<const ulong2*>(a + base + i);
  v.x = mul_mod(v.x, wl, P);
  v.y = mul_mod(v.y, wl, P);
  *reinterpret_cast<ulong2*>(r + base + i) = v;
}

__global__ __launch_bounds__(256) void add_scalars_kernel(DevCtx c, u64* r, const u64* a, LimbConsts w, u32 level, u32 pos0) {
  const u32 pos = pos0 + blockIdx.y;
  if (!owns(c, limb_prime(pos, level, c.L))) return;
  r = reb(c, r, c.rep0 + blockIdx.z);
  a = reb(c, a, c.rep0 + blockIdx.z);
  const u64 q = c.primes[limb_prime(pos, level, c.L)].q;
  const u64 wl = w.w[blockIdx.y];
  const size_t base = (size_t)pos * c.N;
  const u32 i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= c.N) return;
  ulong2 v = *reinterpret_cast<const ulong2*>(a + base + i);
  v.x = add_mod(v.x, wl, q);
  v.y = add_mod(v.y, wl, q);
  *reinterpret_cast<ulong2*>(r + base + i) = v;
}

void launch_add_scalars(const DevCtx& c, u64* r, const u64* a, const LimbConsts& w, u32 level, u32 pos0, u32 n_limbs,
                        hipStream_t s) {
  ACEHIP_ABLATE(ABL_EW);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(add_scalars_kernel, grid, block, 0, s, c, r, a, w, level, pos0);
}

void launch_mul_scalars(const DevCtx& c, u64* r, const u64* a, const LimbConsts& w, u32 level, u32 pos0, u32 n_limbs,
                        hipStream_t s) {
  ACEHIP_ABLATE(ABL_EW);
  if (n_limbs == 0) return;
  dim3 grid((c.N / 2 + 255) / 256, n_limbs, c.nrep), block(256);
  hipLaunchKernelGGL(mul_scalars_kernel, grid, block, 0, s, c, r, a, w, level, pos0);
}

}  // namespace acehip
