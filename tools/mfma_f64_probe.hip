// Maps the lane layout of v_mfma_f64_4x4x4_4b_f64 with one-hot operands and times it against
// v_mfma_f64_16x16x4_f64 (MI355X).  Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_map(double* out) {
  const int p = blockIdx.x, q = blockIdx.y, lane = threadIdx.x;
  double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[((size_t)p * 64 + q) * 64 + lane] = d;
}
template <int SHAPE>
__global__ void k_time(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  if (SHAPE == 4) {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  } else {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  }
}
int main() {
  double* d;
  hipMalloc(&d, 64 * 64 * 64 * 8);
  hipLaunchKernelGGL(k_map, dim3(64, 64), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  for (int p = 0; p < 64; ++p)
    for (int q = 0; q < 64; ++q)
      for (int l = 0; l < 64; ++l)
        if (h[((size_t)p * 64 + q) * 64 + l] != 0.0) printf("MAP %d %d %d %g\n", p, q, l, h[((size_t)p * 64 + q) * 64 + l]);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 4096;
  for (int shape = 0; shape < 2; ++shape)
    for (int wpb = 1; wpb <= 2; ++wpb) {   // waves per SIMD: 256 or 512 threads per CU
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k_time<4>, dim3(256), dim3(256 * wpb), 0, 0, d, iters);
        else hipLaunchKernelGGL(k_time<16>, dim3(256), dim3(256 * wpb), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("TIME shape=%s waves_per_simd=%d: %.3f ms, %.1f ns per MFMA and SIMD (%d per wave)\n", shape ? "16x16x4" : "4x4x4_4b", wpb, ms,
                        ms * 1e6 / (4.0 * iters * wpb), 4 * iters);
      }
    }
  return 0;
}
