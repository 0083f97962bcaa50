// VALU issue-rate microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32 (and the
// quarter-rate ops the Jacobi rotation uses) at 1..8 waves per SIMD.  Decides
// whether the tile kernels should be written for packed or scalar FP32.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 2048;

template <int NACC>
__global__ __launch_bounds__(64) void k_fma(float* out, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (float)threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(64) void k_pkfma(float* out, float a, float b) {
  v2f acc[NACC];
  v2f va = {a, a * 0.5f}, vb = {b, b * 0.25f};
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v2f{(float)threadIdx.x + i, 1.0f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
  if (s == 12345.678f) out[0] = s;
}

// transcendental mix: one v_rcp/v_rsq/v_sqrt per N fma
__global__ __launch_bounds__(64) void k_trans(float* out, float a) {
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 1.0f + (float)threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
      asm volatile("v_rsq_f32 %0, %0" : "+v"(acc[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 12345.678f) out[0] = s;
}

template <typename K>
static double run(K kern, int waves_per_simd, float* d, double ops_per_thread, const char* name) {
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  kern(blocks, d);                                  // warm-up
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) kern(blocks, d);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double lane_ops = (double)blocks * 64 * ops_per_thread;
  const double per_simd_clk = lane_ops / (ms * 1e-3) / (256.0 * 4) / 2.4e9;   // lane-ops / clk / SIMD @2.4GHz
  printf("%-28s waves/SIMD=%d  %8.3f ms  %7.2f T lane-op/s  %6.2f lane-op/clk/SIMD(@2.4GHz)\n", name,
         waves_per_simd, ms, lane_ops / (ms * 1e-3) / 1e12, per_simd_clk);
  return ms;
}

int main() {
  float* d; CHK(hipMalloc(&d, 1024));
  const double n = (double)ITERS * 16;
  for (int w : {1, 2, 3, 4, 8}) {
    run([&](int b, float* o) { hipLaunchKernelGGL(k_fma<16>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n, "v_fma_f32 16acc");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_fma<4>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n, "v_fma_f32 4acc");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_fma<1>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n, "v_fma_f32 1acc(dep)");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_pkfma<16>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n * 2, "v_pk_fma_f32 16acc (x2 lanes)");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_pkfma<4>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n * 2, "v_pk_fma_f32 4acc (x2 lanes)");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_pkfma<1>, dim3(b), dim3(64), 0, 0, o, 1.0001f, 0.5f); }, w, d, n * 2, "v_pk_fma_f32 1acc(dep)");
    run([&](int b, float* o) { hipLaunchKernelGGL(k_trans, dim3(b), dim3(64), 0, 0, o, 1.0f); }, w, d, (double)ITERS * 16, "v_rcp+v_rsq 8acc");
  }
  CHK(hipFree(d));
  return 0;
}
