// ubench_shoup.hip -- VALU cost of the twiddle product a*w mod q on gfx950: which instruction mix is cheapest.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_shoup.hip -o /tmp/ubench_shoup && /tmp/ubench_shoup
// Variants: 0 exact Shoup (compiler: 1 v_mul_hi_u32 + 5 v_mad_u64_u32 + 4 v_mul_lo_u32 + moves)
//           1 sloppy quotient (3 partial products), compiler-chosen multiplies (2 mul_hi + 3 mad + 4 mul_lo)
//           2 sloppy quotient, every product a v_mad_u64_u32 (half rate; v_mul_lo/hi_u32 are quarter rate)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
typedef uint32_t u32;
#define ITER 2048
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ u64 mul64(u32 a, u32 b) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ u64 shoup_exact(u64 a, u64 w, u64 p, u64 q) { return a * w - __umul64hi(a, p) * q; }
__device__ __forceinline__ u64 shoup4_c(u64 a, u64 w, u64 p, u64 q) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u64 h = (u64)a1 * p1 + __umulhi(a0, p1) + __umulhi(a1, p0);
  return a * w - h * q;
}
__device__ __forceinline__ u64 shoup4_mad(u64 a, u64 w, u64 p, u64 nq) {  // nq = 2^64 - q
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  const u64 m1 = mul64(a0, p1), m2 = mul64(a1, p0);
  const u64 h = mad64(a1, p1, (u64)(u32)(m1 >> 32)) + (u32)(m2 >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 t = mul64(a0, w0);
  t = mad64(h0, n0, t);
  u64 c = mul64(a0, w1);
  c = mad64(a1, w0, c);
  c = mad64(h0, n1, c);
  c = mad64(h1, n0, c);
  return t + (c << 32);
}
// hybrid: quotient from 2 v_mul_hi_u32 + 1 mad (their results pair with a zero register for free), low 64 bits of
// a*w + h*nq as two accumulation chains of v_mad_u64_u32 (no quarter-rate v_mul_lo_u32, no zero extension needed)
__device__ __forceinline__ u64 shoup4_hyb(u64 a, u64 w, u64 p, u64 nq) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  const u64 h = (u64)a1 * p1 + __umulhi(a0, p1) + __umulhi(a1, p0);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 t = mul64(a0, w0);
  t = mad64(h0, n0, t);
  u64 c = mul64(a0, w1);
  c = mad64(a1, w0, c);
  c = mad64(h0, n1, c);
  c = mad64(h1, n0, c);
  return (u64)(u32)t | ((u64)((u32)(t >> 32) + (u32)c) << 32);
}
// opaque barriers: no instruction, but the compiler must materialise the full 64-bit value, which stops it from narrowing
// a product to v_mul_lo_u32 / v_mul_hi_u32 (quarter rate) -- it emits v_mad_u64_u32 and allocates the pairs itself
#define OPQ(x) asm("" : "+v"(x))
__device__ __forceinline__ u64 shoup4_opq(u64 a, u64 w, u64 p, u64 nq) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 U = (u64)a0 * p1;
  OPQ(U);
  u64 V = (u64)a1 * p0;
  OPQ(V);
  const u64 h = (u64)a1 * p1 + (U >> 32) + (V >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 c = (u64)a0 * w1;
  c += (u64)a1 * w0;
  c += (u64)h0 * n1;
  c += (u64)h1 * n0;
  OPQ(c);
  u64 t = (u64)a0 * w0;
  t += (u64)h0 * n0;
  return t + ((u64)(u32)c << 32);
}
// quotient by mul_hi (compiler), low products as opaque mad chains
__device__ __forceinline__ u64 shoup4_opq2(u64 a, u64 w, u64 p, u64 nq) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  const u64 h = (u64)a1 * p1 + __umulhi(a0, p1) + __umulhi(a1, p0);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 c = (u64)a0 * w1;
  c += (u64)a1 * w0;
  c += (u64)h0 * n1;
  c += (u64)h1 * n0;
  OPQ(c);
  u64 t = (u64)a0 * w0;
  t += (u64)h0 * n0;
  return t + ((u64)(u32)c << 32);
}
template <int V> __global__ __launch_bounds__(256) void k(u64* out, u64 seed, int check) {
  const u64 q = 0x100000001a40001ull;  // a 57-bit prime of the C3 chain
  u64 a[4];
  for (int i = 0; i < 4; ++i) a[i] = (seed * (2 * i + 3) + threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x) & 0x7FFFFFFFFFFFFFFFull;
  u64 w = (seed * 0x2545F4914F6CDD1Dull + threadIdx.x) % q;
  const u64 p = (u64)(((unsigned __int128)w << 64) / q);
  const u64 nq = 0 - q;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u64 r;
      if (V == 0) r = shoup_exact(a[i], w, p, q);
      if (V == 1) r = shoup4_c(a[i], w, p, q);
      if (V == 2) r = shoup4_mad(a[i], w, p, nq);
      if (V == 3) r = shoup4_hyb(a[i], w, p, nq);
      if (V == 4) r = shoup4_opq(a[i], w, p, nq);
      if (V == 5) r = shoup4_opq2(a[i], w, p, nq);
      a[i] = r + it;  // keeps the chain dependent and the operand < 2^63
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
}
template <int V> double run(u64* d, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(4096), dim3(256), 0, 0, d, 12345ull, 0);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(4096), dim3(256), 0, 0, d, 12345ull + r, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double lane_ops = 4096.0 * 256 * ITER * 4;
  printf("%-40s %8.3f ms  %6.2f T lane-ops/s  %6.1f SIMD-cycles per wave-op at 2.4 GHz\n", name, ms, lane_ops / ms / 1e9,
         1024.0 * 2.4e9 / (lane_ops / 64 / (ms * 1e-3)));
  return ms;
}
int main() {
  u64 *d, *h1 = new u64[256 * 4096], *h2 = new u64[256 * 4096];
  hipMalloc(&d, 8 * 256 * 4096);
  run<0>(d, "exact Shoup (compiler)");
  run<1>(d, "sloppy Shoup (compiler multiplies)");
  hipMemcpy(h1, d, 8 * 256 * 4096, hipMemcpyDeviceToHost);
  run<2>(d, "sloppy Shoup (all v_mad_u64_u32)");
  hipMemcpy(h2, d, 8 * 256 * 4096, hipMemcpyDeviceToHost);
  run<3>(d, "sloppy Shoup (mul_hi quotient, mad chains)");
  hipMemcpy(h2, d, 8 * 256 * 4096, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < 256 * 4096; ++i) bad += h1[i] != h2[i];
  run<4>(d, "sloppy Shoup (opaque barriers, all mad)");
  hipMemcpy(h2, d, 8 * 256 * 4096, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < 256 * 4096; ++i) bad += h1[i] != h2[i];
  run<5>(d, "sloppy Shoup (mul_hi quotient, opaque mad)");
  hipMemcpy(h2, d, 8 * 256 * 4096, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < 256 * 4096; ++i) bad += h1[i] != h2[i];
  printf("variant 2 vs 1: %zu mismatching lanes\n", bad);
  return bad != 0;
}
