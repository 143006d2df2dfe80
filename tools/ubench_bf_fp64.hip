// ubench_bf_fp64.hip -- a second arithmetic for the NTT butterflies of the 50-bit scaling primes, measured against the product's.
//
// The register-tiled NTT passes (ace-compiler_amd/csrc/ntt_fast.hip) are bound by VALU issue as much as by memory: 15
// instructions per forward butterfly for the SMALL prime class, 9 of them v_mad_u64_u32.  Primes below 2^50.2 (33 of the 34 q-limbs of
// the generated ResNets) also fit FP64: residues are integers below 2^53 and a twiddle product needs
//     h = x*w; l = fma(x, w, -h);  qf = rndne(h * (1/q));  r = fma(-qf, q, h);  t = r + l          (6 instructions, t = x*w mod q, |t| < 2.6q)
// plus one add and one subtract for the butterfly; values may grow for FOUR stages before a reduction (rndne(v/q), fma: 3 instructions
// per value) brings them back to |v| <= q/2 -- 8 + 48/32 = 9.5 instructions per butterfly.  This program runs both arithmetics on the same
// inputs, register-resident (no memory traffic inside the timed loop): 16 values per lane, R rounds of four radix-2 stages (the radix-16
// round of the passes), and checks that both produce the same canonical residues.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ace-compiler_amd/csrc tools/ubench_bf_fp64.hip -o /tmp/ubench_bf_fp64 && /tmp/ubench_bf_fp64
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../ace-compiler_amd/csrc/ntt_fast.hip"  // the product's butterflies (bf_fwd<true>, canon_fwd<true>), as they are

namespace acehip {
void ntt_count(u64) {}  // (statistics hook of the library's launchers: they are compiled along with the butterflies, never called here)
}
using namespace acehip;

constexpr int kRounds = 64;  // radix-16 rounds per kernel; a canonical reduction after every 4 rounds (16 stages, as in a transform)

struct Params {
  u64 q, mu;        // mu = floor(2^64 / q)
  double qd, qinv;
  Tw tw[15];        // integer twiddles {w, floor(w*2^63/q)} (SMALL class layout)
  double twd[15];
};

__global__ __launch_bounds__(256) void int_kernel(u64* out, Params p, u64 seed) {
  const BfK k = bf_consts<true>(p.q);
  u64 x[16];
  const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (seed + tid * 16 + i) * 0x9E3779B97F4A7C15ull % p.q;
  Tw t0 = p.tw[0], t1[2] = {p.tw[1], p.tw[2]}, t2[4] = {p.tw[3], p.tw[4], p.tw[5], p.tw[6]},
     t3[8] = {p.tw[7], p.tw[8], p.tw[9], p.tw[10], p.tw[11], p.tw[12], p.tw[13], p.tw[14]};
  for (int r = 0; r < kRounds; ++r) {
    radix16_fwd<true>(x, t0, t1, t2, t3, k);
    if ((r & 3) == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = canon_fwd<true>(x[i], p.q, p.mu);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[tid * 16 + i] = x[i];
}

struct FK {
  double q, qinv;
};
__device__ __forceinline__ double mulmod_fp(double x, double w, const FK& k) {
  const double h = x * w;
  const double l = __builtin_fma(x, w, -h);
  const double qf = __builtin_rint(h * k.qinv);
  const double r = __builtin_fma(-qf, k.q, h);
  return r + l;
}
__device__ __forceinline__ void bf_fwd_fp(double& X, double& Y, double w, const FK& k) {
  const double t = mulmod_fp(Y, w, k), x = X;
  X = x + t;
  Y = x - t;
}
__device__ __forceinline__ double red_fp(double v, const FK& k) { return __builtin_fma(-__builtin_rint(v * k.qinv), k.q, v); }

__global__ __launch_bounds__(256) void fp_kernel(u64* out, Params p, u64 seed) {
  const FK k{p.qd, p.qinv};
  double x[16];
  const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const u64 v = (seed + tid * 16 + i) * 0x9E3779B97F4A7C15ull % p.q;
    x[i] = __builtin_fma((double)(u32)(v >> 32), 4294967296.0, (double)(u32)v);  // (what a pass would do on load: 2 cvt + 1 fma)
  }
  const double* w = p.twd;
  for (int r = 0; r < kRounds; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bf_fwd_fp(x[i], x[i + 8], w[0], k);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) bf_fwd_fp(x[8 * g + i], x[8 * g + i + 4], w[1 + g], k);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i) bf_fwd_fp(x[4 * g + i], x[4 * g + i + 2], w[3 + g], k);
#pragma unroll
    for (int g = 0; g < 8; ++g) bf_fwd_fp(x[2 * g], x[2 * g + 1], w[7 + g], k);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = red_fp(x[i], k);  // |v| < 6.8q -> |v| <= q/2 (+1 ulp of the quotient)
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double v = x[i];
    v = v < 0 ? v + k.q : v;  // canonical [0, q)
    v = v >= k.q ? v - k.q : v;
    const double m = v + 4503599627370496.0;  // 2^52: the integer sits in the mantissa
    u64 bits;
    __builtin_memcpy(&bits, &m, 8);
    out[tid * 16 + i] = bits & 0xFFFFFFFFFFFFFull;
  }
}

static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((unsigned __int128)a * b % q); }

int main() {
  const u64 q = 1125899906826241ull;  // a 50-bit NTT prime (2^50 - 16383 * 2^... style; any odd 50-bit modulus serves the arithmetic)
  Params p;
  p.q = q;
  p.mu = (u64)((((unsigned __int128)1) << 64) / q);
  p.qd = (double)q;
  p.qinv = 1.0 / (double)q;
  u64 w = 0x123456789ABCDull % q;
  for (int i = 0; i < 15; ++i) {
    w = mulmod(w, w + 12345, q);
    p.tw[i] = Tw{w, (u64)((((unsigned __int128)w) << 63) / q)};
    p.twd[i] = (double)w;
  }
  const int blocks = 256 * 16;  // 16 workgroups per CU over time, 4 resident at a time
  const size_t n = (size_t)blocks * 256 * 16;
  u64 *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms_int = 0, ms_fp = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(int_kernel, dim3(blocks), dim3(256), 0, 0, a, p, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_int, e0, e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fp_kernel, dim3(blocks), dim3(256), 0, 0, b, p, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_fp, e0, e1);
  }
  std::vector<u64> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += ha[i] != hb[i];
  const double bfly = (double)blocks * 256 * 8 * 4 * kRounds;  // butterflies per kernel
  printf("prime %llu (%.2f bits); %d workgroups x 256 lanes x 16 values, %d radix-16 rounds\n", (unsigned long long)q, log2((double)q), blocks, kRounds);
  printf("integer (product, SMALL class)  %8.3f ms  %7.1f G butterflies/s\n", ms_int, bfly / ms_int * 1e-6);
  printf("FP64 (mul/fma/rndne)            %8.3f ms  %7.1f G butterflies/s   x%.2f\n", ms_fp, bfly / ms_fp * 1e-6, ms_int / ms_fp);
  printf("results %s (%zu of %zu values differ)\n", bad ? "DIFFER" : "identical", bad, n);
  return bad ? 1 : 0;
}
