// ubench_sector.hip - does a miss in the vector L1 / L2 move a whole 128-byte line or a 64-byte half?
// Reads ONE float per `stride` bytes over a 4 GiB buffer (every access a different line / half line); if half lines were
// fetched, stride 128 would move half the bytes of stride 64 and take half as long per GiB touched.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sector.hip -o /tmp/ubench_sector && /tmp/ubench_sector
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void touch(const float* __restrict__ p, size_t n_access, size_t stride_f, float* out)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n_access; i += (size_t)gridDim.x * blockDim.x) acc += p[i * stride_f];
  if (acc == 123.456f) *out = acc;
}

int main()
{
  const size_t bytes = 4ull << 30;
  float *buf, *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (size_t stride : { (size_t)4, (size_t)32, (size_t)64, (size_t)128, (size_t)256 }) {
    const size_t n = bytes / stride;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(touch, dim3(256 * 32), dim3(256), 0, 0, buf, n, stride / 4, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("stride %4zu B: %8.3f ms for %zu accesses over 4 GiB  -> %.2f TB/s if whole 128-B lines move, %.2f TB/s if only touched bytes/64-B halves (%zu B each)\n", stride, best, n,
           (double)(stride <= 128 ? bytes : n * 128) / best / 1e9, (double)(n * (stride < 64 ? stride : 64)) / best / 1e9, stride < 64 ? stride : (size_t)64);
  }
  return 0;
}
