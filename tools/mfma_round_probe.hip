// How does v_mfma_f32_32x32x2_f32 round its accumulation on gfx950?  (round 3: the full-frame mode's "scale drift" and a
// 4e-5 low DC coefficient of the 8K DCT pointed at the matrix core, not at v_rsq_f32.)
//   test 1: C = 1.0, a*b summed over the instruction's two k = 0.75 ulp(1.0)   -> RNE: 1 + ulp, RTZ: 1.0
//   test 2: C = 1.0, each k contributes 0.375 ulp (sum 0.75 ulp): are the two products added to C one after the
//           other (each < 0.5 ulp -> both lost under RNE) or summed first?
//   test 3: sum of N positive terms through a chain of MFMAs against the float64 sum and a v_fma_f32 chain
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/mfma_round_probe tools/mfma_round_probe.hip && tools/bin/mfma_round_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_one(float a0, float a1, float b0, float b1, float c, float* D) {
  const int l = threadIdx.x;
  const float a = (l >> 5) ? a1 : a0, b = (l >> 5) ? b1 : b0;
  v16f acc;
  for (int v = 0; v < 16; ++v) acc[v] = c;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  if (l == 0) D[0] = acc[0];
}

// chain: lane (i, h) holds A[i][k + h] = x[k + h], B = 1: every output = sum_k x[k]
__global__ void k_chain(const float* x, int n, float* D) {
  const int l = threadIdx.x, h = l >> 5;
  v16f acc = {0};
  float f = 0.0f;
  for (int k = 0; k < n; k += 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k + h], 1.0f, acc, 0, 0, 0);
    f = __builtin_fmaf(x[k], 1.0f, f); f = __builtin_fmaf(x[k + 1], 1.0f, f);
  }
  if (l == 0) { D[0] = acc[0]; D[1] = f; }
}

int main() {
  float* dD; hipMalloc(&dD, 64);
  float h[2];
  const float ulp = ldexpf(1.0f, -23);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, 0.75f * ulp, 0.0f, 1.0f, 1.0f, 1.0f, dD);
  hipMemcpy(h, dD, 4, hipMemcpyDeviceToHost);
  printf("test 1: 1 + 0.75 ulp            -> 1 + %.2f ulp   (RNE 1.00, RTZ 0.00)\n", (h[0] - 1.0f) / ulp);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, 0.375f * ulp, 0.375f * ulp, 1.0f, 1.0f, 1.0f, dD);
  hipMemcpy(h, dD, 4, hipMemcpyDeviceToHost);
  printf("test 2: 1 + 0.375 ulp + 0.375   -> 1 + %.2f ulp   (products summed first + RNE: 1.00; one by one or RTZ: 0.00)\n", (h[0] - 1.0f) / ulp);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, -0.25f * ulp, 0.0f, 1.0f, 1.0f, 1.0f, dD);
  hipMemcpy(h, dD, 4, hipMemcpyDeviceToHost);
  printf("test 2b: 1 - 0.25 ulp (= 1 - 0.5 ulp of the binade below) -> 1 %+.2f ulp   (RNE: 0.00 (tie to even) ; RTZ: -0.50)\n", (h[0] - 1.0f) / ulp);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, -0.2f * ulp, 0.0f, 1.0f, 1.0f, 1.0f, dD);
  hipMemcpy(h, dD, 4, hipMemcpyDeviceToHost);
  printf("test 2c: 1 - 0.2 ulp            -> 1 %+.2f ulp   (RNE: 0.00; RTZ: -0.50)\n", (h[0] - 1.0f) / ulp);
  for (int n : {512, 2048, 7680}) {
    float* x = (float*)malloc(n * 4); double ref = 0; srand(1);
    for (int i = 0; i < n; ++i) { x[i] = 100.0f + (rand() % 1000) / 7.0f; ref += x[i]; }
    float* dx; hipMalloc(&dx, n * 4); hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dx, n, dD);
    hipMemcpy(h, dD, 8, hipMemcpyDeviceToHost);
    printf("test 3: sum of %d positive terms: MFMA chain rel err %+.3e, v_fma_f32 chain %+.3e\n", n, (h[0] - ref) / ref, (h[1] - ref) / ref);
    // zero-mean terms: error relative to the rms partial sum
    double ref2 = 0, rms = 0;
    for (int i = 0; i < n; ++i) { x[i] = (rand() % 2001 - 1000) / 7.0f; ref2 += x[i]; rms += (double)x[i] * x[i]; }
    hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dx, n, dD);
    hipMemcpy(h, dD, 8, hipMemcpyDeviceToHost);
    printf("        zero-mean terms (sum %.1f, sqrt(sum x^2) %.1f): MFMA err %+.3e, v_fma err %+.3e of sqrt(sum x^2)\n", ref2, sqrt(rms),
           (h[0] - ref2) / sqrt(rms), (h[1] - ref2) / sqrt(rms));
    hipFree(dx); free(x);
  }
  return 0;
}
