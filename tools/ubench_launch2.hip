// ubench_launch2.hip -- T host threads, each launching n dependent kernels of `blocks` workgroups x `iters` spin iterations on its
// own stream: launches/s for a given (threads, blocks, iters), to probe runtime knobs (GPU_MAX_HW_QUEUES, HSA_ENABLE_INTERRUPT ...)
//   hipcc --offload-arch=gfx950 -O3 -fgpu-default-stream=per-thread tools/ubench_launch2.hip -o /tmp/ubench_launch2 -lpthread
//   /tmp/ubench_launch2 <threads> <blocks> <iters> [explicit_streams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long* p, int iters) {
  unsigned long long v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 6364136223846793005ull + 1442695040888963407ull;
  if (v == 42) p[0] = v;
}
static void worker(int n, int blocks, int iters, unsigned long long* buf, bool explicit_stream) {
  hipStream_t s = 0;
  if (explicit_stream) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, buf, iters);
  hipStreamSynchronize(s);
  if (explicit_stream) hipStreamDestroy(s);
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 4, blocks = argc > 2 ? atoi(argv[2]) : 512, iters = argc > 3 ? atoi(argv[3]) : 0;
  const bool ex = argc > 4 && atoi(argv[4]);
  unsigned long long* buf;
  hipMalloc(&buf, 1 << 20);
  const int n = 30000;
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(worker, 2000, blocks, iters, buf, ex);
  for (auto& x : th) x.join();
  th.clear();
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < T; ++t) th.emplace_back(worker, n, blocks, iters, buf, ex);
  for (auto& x : th) x.join();
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("threads %d blocks %4d iters %4d explicit %d: %8.0f launches/s total, %6.2f us per launch per thread\n", T, blocks, iters, (int)ex, T * n / s, s / n * 1e6);
  return 0;
}
