// Microbenchmark: what v_mfma_f32_32x32x2_f32 sustains per SIMD under the operand-feeding patterns the full-frame GEMM kernels use.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_mfma_f32 tools/ubench_mfma_f32.hip && ./ubench_mfma_f32
// Variants (each: one workgroup per CU x WPS waves per SIMD, ITER chunks of 16 k-steps x 4 MFMAs = 2 x 2 tiles of 32 x 32):
//   0 regs      operands fixed in registers (the pipe's own rate)
//   1 lds_b32   operands by ds_read_b32 from a k-major image [k][row] (k_hgram as first written)
//   2 lds_b128  operands by ds_read_b128 from a row-major image [row][k], pitch 36: lane (j, h) takes k = 8 m + 4 h + s
//   3 lds_b32 with the whole chunk's operands read before its MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR>
__global__ void k(float* out, const float* in, int iters) {
  __shared__ float As[32][129];
  __shared__ float Ar[128][36];
  const int t = threadIdx.x, lane = t & 63, j = lane & 31, h = lane >> 5, wv = t >> 6;
  for (int i = t; i < 32 * 129; i += blockDim.x) (&As[0][0])[i] = in[i & 1023];
  for (int i = t; i < 128 * 36; i += blockDim.x) (&Ar[0][0])[i] = in[i & 1023];
  __syncthreads();
  v16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
  const int wi = (wv >> 1) & 1, wj = wv & 1;
  float ra = in[t], rb = in[t + 7];
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, rb, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, rb, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, ra, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(rb, rb, c11, 0, 0, 0);
      }
    } else if (VAR == 1) {
      const float* pa = &As[h][64 * wi + j];
      const float* pb = &As[h][64 * wj + j];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float a0 = pa[2 * kk * 129], a1 = pa[2 * kk * 129 + 32], b0 = pb[2 * kk * 129], b1 = pb[2 * kk * 129 + 32];
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
      }
    } else if (VAR == 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 a0 = *reinterpret_cast<const float4*>(&Ar[64 * wi + j][8 * m + 4 * h]);
        const float4 a1 = *reinterpret_cast<const float4*>(&Ar[64 * wi + 32 + j][8 * m + 4 * h]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Ar[64 * wj + j][8 * m + 4 * h]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Ar[64 * wj + 32 + j][8 * m + 4 * h]);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], c00, 0, 0, 0);
          c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], c01, 0, 0, 0);
          c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], c10, 0, 0, 0);
          c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], c11, 0, 0, 0);
        }
      }
    } else {
      const float* pa = &As[h][64 * wi + j];
      const float* pb = &As[h][64 * wj + j];
      float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { a0[kk] = pa[2 * kk * 129]; a1[kk] = pa[2 * kk * 129 + 32]; b0[kk] = pb[2 * kk * 129]; b1[kk] = pb[2 * kk * 129 + 32]; }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b1[kk], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b0[kk], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], c11, 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  }
  v16 s = c00 + c01 + c10 + c11;
  float r = 0; for (int v = 0; v < 16; ++v) r += s[v];
  out[blockIdx.x * blockDim.x + t] = r;
}

template <int VAR> void run(const char* name, int wps, float* out, float* in) {
  const int iters = 2000, ncu = 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<VAR>, dim3(ncu), dim3(256 * wps), 0, 0, out, in, 10);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<VAR>, dim3(ncu), dim3(256 * wps), 0, 0, out, in, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma = (double)ncu * 4 * wps * iters * 64;
  printf("%-10s %d wave(s)/SIMD: %7.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, wps, ms, mfma * 4096 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 64.0 * wps));
}

int main() {
  float *out, *in; CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&in, 1 << 16));
  CK(hipMemset(in, 0, 1 << 16));
  for (int wps = 1; wps <= 2; ++wps) {
    run<0>("regs", wps, out, in); run<1>("lds_b32", wps, out, in); run<2>("lds_b128", wps, out, in); run<3>("lds_pre", wps, out, in);
  }
  return 0;
}
