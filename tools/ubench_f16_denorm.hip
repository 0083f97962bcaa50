// Does v_mfma_f32_32x32x16_f16 keep f16 subnormal operands (or flush them to zero)?  One wave, A = a constant in every
// element, B = 1: every output is 16 * a when the operand is honoured.  hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float a, const float b, float* out) {
  h8_t av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (_Float16)a; bv[i] = (_Float16)b; }
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)av[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float vals[] = {1.0f, 6.2e-5f, 3.0e-5f, 3.0e-6f, 6.0e-8f};
  for (float a : vals) {
    float h[2];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1.0f, d); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e as A operand (f16 %.6e): acc = %.6e (16 a = %.6e)\n", a, h[1], h[0], 16.0 * h[1]);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1.0f, a, d); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e as B operand: acc = %.6e\n", a, h[0]);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.0f, d); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e times 1024: acc = %.6e (expected %.6e)\n", a, h[0], 16.0 * 1024.0 * h[1]);
  }
  return 0;
}
