// Shader clock under light load (development tool): a single wavefront runs a dependent
// FMA chain and reports shader cycles (clock64) against the constant 100 MHz counter
// (wall_clock64), alone and right after a device-filling kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clk.hip -o tools/ubench/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_chain(int n, double* out, long long* t) {
  double x = 1.0 + threadIdx.x * 1e-9;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) x = fma(x, 1.0000001, 1e-9);
  const long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void k_busy(int n, double* out) {
  double x = 1.0 + threadIdx.x * 1e-9, y = 2.0;
  for (int i = 0; i < n; ++i) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 0.9999999, 1e-9); }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
int main() {
  double* out; long long* t;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  hipMalloc(&t, 16);
  long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, 20000, out, t);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    std::printf("alone:      %lld shader cycles, %lld x 10 ns -> %.0f MHz, %.1f cycles / dependent f64 fma\n", h[0], h[1],
                h[0] / (h[1] * 0.01), h[0] / 20000.0);
    hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, 0, 200000, out);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, 20000, out, t);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    std::printf("after busy: %lld shader cycles, %lld x 10 ns -> %.0f MHz, %.1f cycles / dependent f64 fma\n", h[0], h[1],
                h[0] / (h[1] * 0.01), h[0] / 20000.0);
  }
  // a train of small kernels, as a solver loop issues them
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, 2000, out, t);
  hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, 20000, out, t);
  hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  std::printf("after 2000 small kernels: %.0f MHz\n", h[0] / (h[1] * 0.01));
  return 0;
}
