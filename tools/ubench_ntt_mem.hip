// ubench_ntt_mem.hip -- memory side of the two N=2^16 NTT passes without any arithmetic: which access pattern reaches
// which fraction of the HBM rate.  512 MiB batch (1024 limbs of 2^16 u64), in place, 256 lanes x 16 values per workgroup
// (the shape of ntt8_strided_kernel / ntt8_contig_kernel), 4 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ntt_mem.hip -o /tmp/ubench_ntt_mem && /tmp/ubench_ntt_mem
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
typedef uint32_t u32;
constexpr u32 N = 65536, LIMBS = 1024;

// P0: fully contiguous: 16 B per lane, 8 wave-wide 1 KiB transactions each way, no LDS
__global__ __launch_bounds__(256, 4) void p0_contig16(u64* X) {
  ulong2* p = reinterpret_cast<ulong2*>(X + (size_t)blockIdx.x * 4096);
  ulong2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p[threadIdx.x + 256 * i];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x += 1;
    p[threadIdx.x + 256 * i] = v[i];
  }
}
// P1: the strided pass: 16 columns x 256 rows, row pitch 2 KiB, 8 B per lane, 128-byte segments, same pattern back
template <bool LDS_X> __global__ __launch_bounds__(256, 4) void p1_strided(u64* X) {
  __shared__ u64 lds[256 * 17];
  const u32 tile = blockIdx.x & 15, limb = blockIdx.x >> 4, cc = threadIdx.x & 15, hg = threadIdx.x >> 4;
  u64* L = X + (size_t)limb * N + tile * 16 + cc;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = L[(size_t)(16 * k + hg) * 256];
  if (LDS_X) {
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[(16 * k + hg) * 17 + cc] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[(16 * hg + k) * 17 + cc] + 1;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) L[(size_t)(16 * hg + k) * 256] = x[k] + 1;
}
// P2: strided read, tile-contiguous write (a private intermediate layout [limb][tile][row][16]): the tile leaves as one
// 32 KiB block.  out of place within the limb is not possible in place, so this writes to a second buffer
__global__ __launch_bounds__(256, 4) void p2_strided_to_tile(const u64* X, u64* Y) {
  __shared__ u64 lds[256 * 17];
  const u32 tile = blockIdx.x & 15, limb = blockIdx.x >> 4, cc = threadIdx.x & 15, hg = threadIdx.x >> 4;
  const u64* L = X + (size_t)limb * N + tile * 16 + cc;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = L[(size_t)(16 * k + hg) * 256];
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[(16 * k + hg) * 17 + cc] = x[k];
  __syncthreads();
  ulong2* O = reinterpret_cast<ulong2*>(Y + (size_t)limb * N + tile * 4096);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 e = 2 * threadIdx.x + 512 * i, row = e >> 4, c = e & 15;
    ulong2 v;
    v.x = lds[row * 17 + c] + 1;
    v.y = lds[row * 17 + c + 1];
    O[threadIdx.x + 256 * i] = v;
  }
}
// P3: the contiguous pass as it is: 8-byte loads x[k] = X[b*256 + 16k + lo4] (128-byte segments), 16-byte stores through LDS
__global__ __launch_bounds__(256, 4) void p3_contig_now(u64* X) {
  __shared__ u64 lds[16 * 272];
  const u32 lo4 = threadIdx.x & 15, b = threadIdx.x >> 4;
  u64* T = X + (size_t)blockIdx.x * 4096;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = T[b * 256 + 16 * k + lo4];
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * 272 + 17 * k + lo4] = x[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[b * 272 + 17 * lo4 + k] + 1;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * 272 + 17 * lo4 + k] = x[k];
  __syncthreads();
  ulong2* O = reinterpret_cast<ulong2*>(T);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 e = 2 * threadIdx.x + 512 * i, bb = e >> 8, rho = e & 255;
    ulong2 v;
    v.x = lds[bb * 272 + rho + (rho >> 4)];
    v.y = lds[bb * 272 + rho + (rho >> 4) + 1];
    O[threadIdx.x + 256 * i] = v;
  }
}
// P4: the contiguous pass with a 16-byte twiddle stream per element (what the real pass reads from L2 / HBM)
__global__ __launch_bounds__(256, 4) void p4_contig_tw(u64* X, const ulong2* TW) {
  __shared__ u64 lds[16 * 272];
  const u32 lo4 = threadIdx.x & 15, b = threadIdx.x >> 4;
  const u32 tile = blockIdx.x & 15, limb = (blockIdx.x >> 4) & 31;  // 32 distinct twiddle limbs, shared by 32 polys
  u64* T = X + (size_t)blockIdx.x * 4096;
  const ulong2* W = TW + (size_t)limb * N + (size_t)(tile * 16 + b) * 256 + lo4 * 16;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = T[b * 256 + 16 * k + lo4];
  ulong2 t[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) t[k] = W[k];
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * 272 + 17 * k + lo4] = x[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = lds[b * 272 + 17 * lo4 + k] + (k < 15 ? t[k].x ^ t[k].y : 1);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[b * 272 + 17 * lo4 + k] = x[k];
  __syncthreads();
  ulong2* O = reinterpret_cast<ulong2*>(T);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 e = 2 * threadIdx.x + 512 * i, bb = e >> 8, rho = e & 255;
    ulong2 v;
    v.x = lds[bb * 272 + rho + (rho >> 4)];
    v.y = lds[bb * 272 + rho + (rho >> 4) + 1];
    O[threadIdx.x + 256 * i] = v;
  }
}
// P5: persistent variant of P0: 1024 workgroups, each streams 16 tiles with the next tile's loads issued before the stores
__global__ __launch_bounds__(256, 4) void p5_persistent(u64* X, u32 tiles_per_wg) {
  ulong2 v[8], n[8];
  ulong2* p = reinterpret_cast<ulong2*>(X + (size_t)blockIdx.x * tiles_per_wg * 4096);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p[threadIdx.x + 256 * i];
  for (u32 t = 0; t < tiles_per_wg; ++t) {
    ulong2* q = p + 2048;
    if (t + 1 < tiles_per_wg) {
#pragma unroll
      for (int i = 0; i < 8; ++i) n[i] = q[threadIdx.x + 256 * i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x += 1;
      p[threadIdx.x + 256 * i] = v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = n[i];
    p = q;
  }
}

template <class F> void run(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("%-58s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  u64 *X, *Y;
  ulong2* TW;
  const size_t bytes = (size_t)LIMBS * N * 8;
  hipMalloc(&X, bytes);
  hipMalloc(&Y, bytes);
  hipMalloc(&TW, (size_t)32 * N * 16);
  hipMemset(X, 1, bytes);
  hipMemset(Y, 1, bytes);
  hipMemset(TW, 2, (size_t)32 * N * 16);
  const dim3 g(LIMBS * 16), b(256);
  run("P0 contiguous 16 B/lane, no LDS", 2.0 * bytes, [&] { hipLaunchKernelGGL(p0_contig16, g, b, 0, 0, X); });
  run("P1 strided 8 B/lane (128 B segments), no LDS", 2.0 * bytes, [&] { hipLaunchKernelGGL(p1_strided<false>, g, b, 0, 0, X); });
  run("P1 strided + LDS transpose", 2.0 * bytes, [&] { hipLaunchKernelGGL(p1_strided<true>, g, b, 0, 0, X); });
  run("P2 strided read, tile-contiguous 16 B write (2 buffers)", 2.0 * bytes, [&] { hipLaunchKernelGGL(p2_strided_to_tile, g, b, 0, 0, X, Y); });
  run("P3 contiguous pass as is (8 B loads, LDS, 16 B stores)", 2.0 * bytes, [&] { hipLaunchKernelGGL(p3_contig_now, g, b, 0, 0, X); });
  run("P4 contiguous pass + 16 B twiddle per element", 2.0 * bytes, [&] { hipLaunchKernelGGL(p4_contig_tw, g, b, 0, 0, X, TW); });
  run("P5 persistent contiguous, 1024 wgs x 16 tiles, prefetch", 2.0 * bytes, [&] { hipLaunchKernelGGL(p5_persistent, dim3(1024), b, 0, 0, X, 16u); });
  run("P5 persistent contiguous, 2048 wgs x 8 tiles, prefetch", 2.0 * bytes, [&] { hipLaunchKernelGGL(p5_persistent, dim3(2048), b, 0, 0, X, 8u); });
  // working sets that fit the 256 MiB Infinity Cache: the same in-place stream over a smaller buffer
  for (size_t mb : {16, 32, 64, 128, 192, 256, 384}) {
    char name[96];
    snprintf(name, sizeof name, "P0 contiguous in place, %zu MiB working set", mb);
    const size_t b2 = mb << 20;
    run(name, 2.0 * b2, [&] { hipLaunchKernelGGL(p0_contig16, dim3(b2 / 32768), b, 0, 0, X); });
  }
  for (size_t mb : {32, 128}) {  // two kernels alternating over the same buffer: what pass 2 sees after pass 1
    char name[96];
    snprintf(name, sizeof name, "P1 strided then P3 contiguous, %zu MiB", mb);
    const size_t b2 = mb << 20;
    run(name, 4.0 * b2, [&] {
      hipLaunchKernelGGL(p1_strided<true>, dim3(b2 / 32768), b, 0, 0, X);
      hipLaunchKernelGGL(p3_contig_now, dim3(b2 / 32768), b, 0, 0, X);
    });
  }
  run("hipMemcpyDtoD 512 MiB", 2.0 * bytes, [&] { hipMemcpyAsync(Y, X, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
