// Issue-rate microbenchmarks for gfx950 FP64 VALU code (development tool).
// Occupancy is set with dynamic LDS: 160 KB -> 1 block (4 waves) per CU = 1 wave/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  extern __shared__ char lds[];
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x + i) * 1e-3;
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 0.5f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] = fma(x[i], a, b);
      if (MODE == 1) {  // one accvgpr write+read pair (32-bit) per DFMA
        float t;
        asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(t) : "v"(f[i]) : "a0");
        f[i] = t;
      } else if (MODE == 2) {  // two plain 32-bit movs per DFMA
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0" : "+v"(f[i]));
      } else if (MODE == 3) {  // one f32 fma per DFMA
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      } else if (MODE == 4) {  // four accvgpr moves per DFMA (accumulator round trip)
        float t;
        asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %0, a1" : "=v"(t) : "v"(f[i]) : "a0", "a1");
        f[i] = t;
      } else if (MODE == 5) {  // dependent chain: each DFMA depends on previous
        x[0] = fma(x[0], a, x[i]);
      }
    }
  }
  double s = 0;
  float sf = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += x[i]; sf += f[i]; }
  if (s == 123.456 || sf == 1.2345f) out[0] = s + sf + lds[threadIdx.x];
}

template <int MODE>
void run(const char* name, double* d, int lds_bytes, int blocks_per_cu) {
  const int iters = 2048, blocks = 256 * blocks_per_cu;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds_bytes, 0, d, iters, 0.999999, 1e-7);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  // wave-instructions of the DFMA stream per SIMD: blocks*4 waves / 1024 SIMDs * iters*16
  double dfma_per_simd = (double)blocks * 4 / 1024.0 * iters * 16;
  double ns_per_dfma = best * 1e6 / dfma_per_simd;
  printf("%-34s waves/SIMD=%d  %.3f ms  %.2f ns per DFMA(+extras) per SIMD  (%.1f TFLOP/s DFMA)\n", name, blocks_per_cu, best,
         ns_per_dfma, 2.0 * 64 * dfma_per_simd * 1024 / (best * 1e-3) * 1e-12);
}

int main() {
  double* d; CK(hipMalloc(&d, 1 << 20));
  const int cfg[3][2] = {{160 * 1024, 1}, {80 * 1024, 2}, {40 * 1024, 4}};
  for (auto& c : cfg) {
    run<0>("dfma only", d, c[0], c[1]);
    run<5>("dfma + dependent dfma chain", d, c[0], c[1]);
    run<1>("dfma + accvgpr write/read pair", d, c[0], c[1]);
    run<4>("dfma + 4 accvgpr moves", d, c[0], c[1]);
    run<2>("dfma + 2 v_mov_b32", d, c[0], c[1]);
    run<3>("dfma + v_fma_f32", d, c[0], c[1]);
  }
  return 0;
}
