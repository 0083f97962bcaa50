// Probe of v_mfma_f32_32x32x2_f32 operand/result layouts on gfx950 (development aid for the
// full-frame gram/apply tiles): D = A(32x2) * B(2x32), A[i][k] in lane i + 32k, B[k][j] in lane j + 32k,
// D[8*(v/4) + 4*(lane/32) + v%4][lane%32] in accumulator register v.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/mfma_probe tools/mfma_probe.hip && tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* D) {   // A [32][2], B [2][32], D raw [16][64]
  const int l = threadIdx.x;
  const float a = A[(l & 31) * 2 + (l >> 5)];
  const float b = B[(l >> 5) * 32 + (l & 31)];
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[v * 64 + l] = acc[v];
}

int main() {
  float hA[64], hB[64], hD[1024], ref[32][32];
  srand(3);
  for (int i = 0; i < 64; ++i) { hA[i] = (rand() % 200 - 100) / 7.0f; hB[i] = (rand() % 200 - 100) / 11.0f; }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) ref[i][j] = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
  float *dA, *dB, *dD;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int v = 0; v < 16; ++v) for (int l = 0; l < 64; ++l) {
    const int i = 8 * (v / 4) + 4 * (l / 32) + v % 4, j = l % 32;
    worst = fmax(worst, fabs(hD[v * 64 + l] - ref[i][j]));
  }
  printf("max |D - ref| under the assumed layout: %g\n", worst);
  return worst < 1e-4 ? 0 : 1;
}
