// Are stream order and kernel-boundary visibility (across XCDs: the reader's `shift`) kept while several HIP streams run concurrently?  Per stream and iteration: kernel W writes the iteration number
// into every word of the stream's buffer (workgroups of uneven length), kernel C - next in the SAME stream - counts words that do
// not hold it.  hipcc -O3 --offload-arch=gfx950 -o tools/bin/stream_order_probe tools/stream_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_w(int* buf, int n, int iter, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int s = (blockIdx.x * 37 % 11) * spin;                       // uneven workgroup durations
  float x = 1.0f;
  for (int k = 0; k < s; ++k) x = x * 1.000001f + 1e-7f;
  if (i < n) buf[i] = iter + (x < 0.0f ? 1 : 0);
}
__global__ void k_c(const int* buf, int n, int iter, int* errs, int shift) {
  // workgroup b checks what workgroup (b + shift) % gridDim.x wrote: shift = 0 the same XCD (b % 8 on both sides), else another one
  const int i = ((blockIdx.x + shift) % gridDim.x) * blockDim.x + threadIdx.x;
  if (i < n && buf[i] != iter) atomicAdd(errs, 1);
}
int main(int argc, char** argv) {
  const int NS = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 3000, wgs = argc > 3 ? atoi(argv[3]) : 72, spin = argc > 4 ? atoi(argv[4]) : 200, shift = argc > 5 ? atoi(argv[5]) : 0;
  const int n = wgs * 256;
  hipStream_t st[8]; int* buf[8]; int* errs;
  hipMalloc(&errs, 8 * 4); hipMemset(errs, 0, 8 * 4);
  for (int s = 0; s < NS; ++s) { hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking); hipMalloc(&buf[s], n * 4); hipMemset(buf[s], 0xff, n * 4); }
  hipDeviceSynchronize();
  for (int it = 0; it < iters; ++it)
    for (int s = 0; s < NS; ++s) {
      hipLaunchKernelGGL(k_w, dim3(wgs), dim3(256), 0, st[s], buf[s], n, it, spin);
      hipLaunchKernelGGL(k_c, dim3(wgs), dim3(256), 0, st[s], buf[s], n, it, errs + s, shift);
    }
  hipDeviceSynchronize();
  int h[8]; hipMemcpy(h, errs, 8 * 4, hipMemcpyDeviceToHost);
  printf("streams %d iterations %d workgroups %d shift %d: words that did not hold the iteration number:", NS, iters, wgs, shift);
  for (int s = 0; s < NS; ++s) printf(" %d", h[s]);
  printf("\n");
  return 0;
}
