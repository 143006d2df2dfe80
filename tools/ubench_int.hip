// ubench_int.hip -- VALU integer-multiply throughput on gfx950 (feeds DESIGN.md's compute ceiling).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o /tmp/ubench_int && /tmp/ubench_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define ITER 4096
struct Tw { u64 w, p; };
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {
  u64 d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
__device__ __forceinline__ u64 shoup_lazy_mad(u64 a, Tw t, u64 nq) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)t.p, p1 = (u32)(t.p >> 32);
  const u32 w0 = (u32)t.w, w1 = (u32)(t.w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  const u64 t0 = mad64(a0, p0, 0);
  const u64 t1 = mad64(a0, p1, t0 >> 32);
  const u64 t2 = mad64(a1, p0, (u32)t1);
  const u64 h = mad64(a1, p1, t1 >> 32) + (t2 >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32);
  u64 acc = mad64(a0, w0, 0);
  acc = mad64(h0, n0, acc);
  u64 c = mad64(a0, w1, acc >> 32);
  c = mad64(a1, w0, (u32)c);
  c = mad64(h0, n1, (u32)c);
  c = mad64(h1, n0, (u32)c);
  return (c << 32) | (u32)acc;
}
template <int OP> __global__ __launch_bounds__(256) void k(u64* out, u64 seed) {
  u64 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 11;
  u64 b = seed * 0x9E3779B97F4A7C15ull + blockIdx.x;
  u32 x0 = (u32)a0, x1 = (u32)a1, x2 = (u32)a2, x3 = (u32)a3, y = (u32)b | 1;
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) { a0 = __umul64hi(a0, b) + i; a1 = __umul64hi(a1, b) + i; a2 = __umul64hi(a2, b) + i; a3 = __umul64hi(a3, b) + i; }
    if (OP == 1) { a0 = a0 * b + i; a1 = a1 * b + i; a2 = a2 * b + i; a3 = a3 * b + i; }
    if (OP == 2) { x0 = x0 * y + i; x1 = x1 * y + i; x2 = x2 * y + i; x3 = x3 * y + i; }
    if (OP == 3) { x0 = __umulhi(x0, y) + i; x1 = __umulhi(x1, y) + i; x2 = __umulhi(x2, y) + i; x3 = __umulhi(x3, y) + i; }
    if (OP == 4) { a0 = (u64)(u32)a0 * y + a0; a1 = (u64)(u32)a1 * y + a1; a2 = (u64)(u32)a2 * y + a2; a3 = (u64)(u32)a3 * y + a3; }  // v_mad_u64_u32
    if (OP == 5) { a0 = a0 + b + i; a1 = a1 + b + i; a2 = a2 + b + i; a3 = a3 + b + i; }  // 64-bit add chain
    if (OP == 6) { x0 = __mul24(x0, y) + i; x1 = __mul24(x1, y) + i; x2 = __mul24(x2, y) + i; x3 = __mul24(x3, y) + i; }
    if (OP == 7) {  // shoup modmul
      const u64 q = 0xFFFFFFFFFFC0001ull; u64 w = b % q, wp = b;
      u64 h = __umul64hi(a0, wp); a0 = a0 * w - h * q; a0 = a0 >= q ? a0 - q : a0;
      h = __umul64hi(a1, wp); a1 = a1 * w - h * q; a1 = a1 >= q ? a1 - q : a1;
      h = __umul64hi(a2, wp); a2 = a2 * w - h * q; a2 = a2 >= q ? a2 - q : a2;
      h = __umul64hi(a3, wp); a3 = a3 * w - h * q; a3 = a3 >= q ? a3 - q : a3;
    }
    if (OP == 8) {  // lazy shoup, compiler multiply
      const u64 q = 0xFFFFFFFFFFC0001ull; u64 w = b % q, wp = b;
      a0 = a0 * w - __umul64hi(a0, wp) * q; a1 = a1 * w - __umul64hi(a1, wp) * q;
      a2 = a2 * w - __umul64hi(a2, wp) * q; a3 = a3 * w - __umul64hi(a3, wp) * q;
    }
    if (OP == 9) {  // lazy shoup, v_mad_u64_u32 only
      const u64 q = 0xFFFFFFFFFFC0001ull; Tw t{b % q, b};
      a0 = shoup_lazy_mad(a0, t, 0 - q); a1 = shoup_lazy_mad(a1, t, 0 - q);
      a2 = shoup_lazy_mad(a2, t, 0 - q); a3 = shoup_lazy_mad(a3, t, 0 - q);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + x0 + x1 + x2 + x3;
}
template <int OP> void run(const char* name, double ops_per_iter) {
  u64* d; hipMalloc(&d, 8 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(4096), dim3(256), 0, 0, d, 12345ull);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(4096), dim3(256), 0, 0, d, 12345ull + r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double lane_ops = 4096.0 * 256 * ITER * ops_per_iter;
  printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.2f cycles per wave-op per SIMD @2.4GHz)\n", name, ms, lane_ops / ms / 1e9,
         1024.0 * 2.4e9 / (lane_ops / 64 / (ms * 1e-3)));
  hipFree(d);
}
int main() {
  run<0>("mulhi64", 4); run<1>("mullo64 (+add)", 4); run<2>("mul_lo_u32 (+add)", 4); run<3>("mul_hi_u32 (+add)", 4);
  run<4>("mad_u64_u32", 4); run<5>("add64 x2", 8); run<6>("mul24", 4); run<7>("shoup modmul", 4);
  run<8>("shoup lazy (compiler)", 4); run<9>("shoup lazy (mad asm)", 4);
  return 0;
}
