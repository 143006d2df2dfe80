// Probe of HIP virtual memory management on the GPU box (round 6, owner-only key limbs): allocation granularity, one physical handle mapped
// at many addresses (the "sink" that backs limbs a rank does not own), cost per map call, kernels over a partly-sink range.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_vmm.hip -o /tmp/ubench_vmm && /tmp/ubench_vmm
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("FAILED %s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);                \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

__global__ void fill(unsigned long long* p, size_t words_per_limb, unsigned long long tag) {
  const size_t limb = blockIdx.y, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words_per_limb) p[limb * words_per_limb + i] = tag + limb;
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  int vmm = 0;
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  printf("virtual memory management supported: %d\n", vmm);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
  const size_t limb = 65536 * 8;  // one limb at N = 2^16
  if (limb % gmin) {
    printf("a 512 KiB limb is not a multiple of the minimum granularity: owner-only limbs impossible\n");
    return 0;
  }
  const int n_limbs = 270, world = argc > 1 ? atoi(argv[1]) : 8, rank = argc > 1 && atoi(argv[1]) == 1 ? 0 : 3;  // one switch key of the generated ResNets: 3 digits x 2 x 45 limbs
  size_t free0 = 0, free1 = 0, total = 0;
  CK(hipMemGetInfo(&free0, &total));
  void* base = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  CK(hipMemAddressReserve(&base, limb * n_limbs, 0, nullptr, 0));
  hipMemGenericAllocationHandle_t sink;
  CK(hipMemCreate(&sink, limb, &prop, 0));
  std::vector<hipMemGenericAllocationHandle_t> own;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  int mapped_own = 0;
  for (int k = 0; k < n_limbs; ++k) {
    void* at = (char*)base + (size_t)k * limb;
    if ((k % 45) % world == rank) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, limb, &prop, 0));
      CK(hipMemMap(at, limb, 0, h, 0));
      own.push_back(h);
      ++mapped_own;
    } else {
      CK(hipMemMap(at, limb, 0, sink, 0));  // the same physical limb behind every foreign position
    }
  }
  CK(hipMemSetAccess(base, limb * n_limbs, &acc, 1));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CK(hipMemGetInfo(&free1, &total));
  printf("%d limbs reserved, %d owned + 1 sink mapped in %.2f ms (%.1f us per limb); device memory used %.1f MiB (full key = %.1f MiB)\n", n_limbs,
         mapped_own, ms, ms * 1e3 / n_limbs, (free0 - free1) / 1048576.0, limb * n_limbs / 1048576.0);
  dim3 grid(65536 / 256, n_limbs);
  fill<<<grid, 256>>>((unsigned long long*)base, 65536, 1000);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(65536);
  int bad = 0;
  for (int k = 0; k < n_limbs; ++k) {
    CK(hipMemcpy(h.data(), (char*)base + (size_t)k * limb, limb, hipMemcpyDeviceToHost));
    const bool mine = (k % 45) % world == rank;
    if (mine && (h[0] != 1000ull + k || h[65535] != 1000ull + k)) ++bad;  // owned limbs hold what was written to them
  }
  printf("owned limbs hold their own values: %s; a kernel over the whole range (foreign limbs land in the sink) ran without a fault\n", bad ? "NO" : "yes");
  {  // ONE copy that spans many mappings (a key written to a file / read from one goes through such copies)
    std::vector<unsigned long long> all((size_t)65536 * 45);
    CK(hipMemcpy(all.data(), base, limb * 45, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int k = 0; k < 45; ++k)
      if (k % world == rank && all[(size_t)k * 65536 + 7] != 1000ull + k) ++bad2;
    for (auto& v : all) v = 77;
    CK(hipMemcpy(base, all.data(), limb * 45, hipMemcpyHostToDevice));
    CK(hipMemcpy(h.data(), (char*)base + (size_t)rank * limb, limb, hipMemcpyDeviceToHost));
    printf("one copy across 45 mappings: device -> host %s, host -> device %s\n", bad2 ? "WRONG" : "ok", h[123] == 77 ? "ok" : "WRONG");
  }
  for (int k = 0; k < n_limbs; ++k) CK(hipMemUnmap((char*)base + (size_t)k * limb, limb));
  for (auto hd : own) CK(hipMemRelease(hd));
  CK(hipMemRelease(sink));
  CK(hipMemAddressFree(base, limb * n_limbs));
  CK(hipMemGetInfo(&free1, &total));
  printf("after unmap / release: device memory back to within %.1f MiB\n", (double)((long long)free0 - (long long)free1) / 1048576.0);
  return 0;
}
