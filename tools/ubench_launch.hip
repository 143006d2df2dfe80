// ubench_launch.hip -- how many kernel launches per second can T host threads push through their own HIP streams?
// (the ResNet-20 workload is ~108 k small dependent launches per image; bench.py runs 4 image streams)
//   hipcc --offload-arch=gfx950 -O3 -fgpu-default-stream=per-thread tools/ubench_launch.hip -o /tmp/ubench_launch -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long* p, int iters) {
  unsigned long long v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 6364136223846793005ull + 1442695040888963407ull;
  if (v == 42) p[0] = v;
}
static void worker(int n, int blocks, int iters, unsigned long long* buf) {
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, buf, iters);
  hipStreamSynchronize(0);
}
int main(int argc, char** argv) {
  unsigned long long* buf;
  hipMalloc(&buf, 1 << 20);
  const int n = 40000;
  for (int iters : {0, 400}) {      // empty kernel / ~10 us of work per kernel
    for (int blocks : {1, 512}) {
      for (int T : {1, 2, 4, 8}) {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(worker, 2000, blocks, iters, buf);  // warm-up: streams, code load
        for (auto& x : th) x.join();
        th.clear();
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t) th.emplace_back(worker, n, blocks, iters, buf);
        for (auto& x : th) x.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("iters %4d blocks %4d threads %d: %8.0f launches/s total, %6.2f us per launch per thread\n", iters, blocks, T, T * n / s, s / n * 1e6);
      }
    }
  }
  return 0;
}
