// ubench_ops.hip -- issue cost of the VALU instructions the integer kernels are made of, on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_ops.hip -o /tmp/ubench_ops && /tmp/ubench_ops
// Every kernel runs REP x 16 independent copies of one instruction (inline asm, 16 destination registers) in
// 256-thread workgroups, W workgroups per CU (W waves per SIMD), and reports shader cycles (s_memtime) per
// wave-instruction per SIMD = elapsed cycles / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint64_t u64; typedef uint32_t u32;
#define REP 2048

#define I16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int OP> __global__ __launch_bounds__(256) void k(u64* out, u64* cyc, u64 seed) {
  u64 a[16]; u32 x[16]; double f[16];
  u64 b = seed * 0x9E3779B97F4A7C15ull + threadIdx.x; u32 y = (u32)b | 1, z = (u32)(b >> 32);
  double g = 1.0 + (double)threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = b + i * 77; x[i] = y + i; f[i] = g + i; }
  __syncthreads();
  const u64 t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < REP; ++r) {
#define MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(z) : "vcc");
#define MADS(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(z) : "s20", "s21");
#define MAD0(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(y), "v"(x[i]) : "vcc");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b));
#define ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[i]) : "v"(y) : "vcc");
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y));
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define MUL64F(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define ADD64F(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y) : "vcc");
#define CMP64(i) asm volatile("v_cmp_le_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(a[i]));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(y));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
#define DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y), "v"(z));
#define PKMUL(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
#define PKMAD(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
#define MADU16(i) asm volatile("v_mad_u32_u16 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
#define SUBCO(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x[i]) : "v"(y) : "vcc");
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define MAD_I64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(y), "v"(z) : "vcc");
    if (OP == 0) { I16(MAD) }
    if (OP == 1) { I16(MADS) }
    if (OP == 2) { I16(MAD0) }
    if (OP == 3) { I16(MULLO) }
    if (OP == 4) { I16(MULHI) }
    if (OP == 5) { I16(ADD64) }
    if (OP == 6) { I16(ADDCO) }
    if (OP == 7) { I16(MOV) }
    if (OP == 8) { I16(ADD32) }
    if (OP == 9) { I16(FMA64) }
    if (OP == 10) { I16(MUL64F) }
    if (OP == 11) { I16(ADD64F) }
    if (OP == 12) { I16(MAD24) }
    if (OP == 13) { I16(MULHI24) }
    if (OP == 14) { I16(CNDM) }
    if (OP == 15) { I16(CMP64) }
    if (OP == 16) { I16(SHR64) }
    if (OP == 17) { I16(ALIGN) }
    if (OP == 18) { I16(ADD3) }
    if (OP == 19) { I16(DOT4) }
    if (OP == 20) { I16(PKMUL) }
    if (OP == 21) { I16(PKMAD) }
    if (OP == 22) { I16(MADU16) }
    if (OP == 23) { I16(SUBCO) }
    if (OP == 24) { I16(PKFMA) }
    if (OP == 25) { I16(MAD_I64) }
  }
  const u64 t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  u64 s = 0; double fs = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += a[i] + x[i]; fs += f[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s + (u64)fs;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP> void run(const char* name, int per_instr, int wpc) {
  const int blocks = 256 * wpc;
  u64 *d, *c; hipMalloc(&d, 8 * 256 * blocks); hipMalloc(&c, 16 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, c, 12345ull);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, c, 12345ull + r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<u64> h(2 * blocks); hipMemcpy(h.data(), c, 16 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> cy(blocks), ghz(blocks);
  for (int i = 0; i < blocks; ++i) { cy[i] = (double)h[2 * i]; ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
  std::sort(cy.begin(), cy.end()); std::sort(ghz.begin(), ghz.end());
  const double med = cy[blocks / 2], clk = ghz[blocks / 2];
  const double instr_per_wave = (double)REP * 16 * per_instr;
  // wall-based: all SIMDs busy for ms at clk GHz; total wave-instructions = blocks * 4 waves * instr_per_wave over 1024 SIMDs
  const double wall_cpi = (ms * 1e-3 * clk * 1e9) / (instr_per_wave * blocks * 4 / 1024.0);
  printf("%-30s W=%d %8.1f us  wave %9.0f cyc  in-kernel %5.2f  wall-based %5.2f cyc/wave-instr/SIMD  clock %.2f GHz\n", name, wpc,
         ms * 1e3, med, med / (instr_per_wave * wpc), wall_cpi, clk);
  hipFree(d); hipFree(c);
}
int main() {
  for (int w : {2, 4, 8}) {
    run<0>("v_mad_u64_u32 (vcc carry)", 1, w);
    run<1>("v_mad_u64_u32 (sgpr carry)", 1, w);
    run<2>("v_mad_u64_u32 addend 0", 1, w);
    run<25>("v_mad_i64_i32", 1, w);
    run<3>("v_mul_lo_u32", 1, w);
    run<4>("v_mul_hi_u32", 1, w);
    run<5>("v_lshl_add_u64", 1, w);
    run<6>("v_add_co + v_addc_co (pair)", 2, w);
    run<23>("v_sub_co_u32", 1, w);
    run<7>("v_mov_b32", 1, w);
    run<8>("v_add_u32", 1, w);
    run<18>("v_add3_u32", 1, w);
    run<9>("v_fma_f64", 1, w);
    run<10>("v_mul_f64", 1, w);
    run<11>("v_add_f64", 1, w);
    run<24>("v_pk_fma_f32", 1, w);
    run<12>("v_mad_u32_u24", 1, w);
    run<13>("v_mul_hi_u32_u24", 1, w);
    run<22>("v_mad_u32_u16", 1, w);
    run<14>("v_cndmask_b32", 1, w);
    run<15>("v_cmp_le_u64", 1, w);
    run<16>("v_lshrrev_b64", 1, w);
    run<17>("v_alignbit_b32", 1, w);
    run<19>("v_dot4_u32_u8", 1, w);
    run<20>("v_pk_mul_lo_u16", 1, w);
    run<21>("v_pk_mad_u16", 1, w);
  }
  return 0;
}
