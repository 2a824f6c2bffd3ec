// Which XCD does workgroup i of a 1-D grid run on?  Reads HW_REG_XCC_ID (gfx940+) per workgroup and prints the mapping for a grid
// shaped like the march's (256 threads, 32 KiB of LDS, 168 VGPRs are not reproduced).   hipcc --offload-arch=gfx950 tools/ubench_xcc.hip -o /tmp/ubench_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned int* out, unsigned long long* when)
{
  extern __shared__ unsigned char lds[];
  unsigned int id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) { out[blockIdx.x] = id; when[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); lds[0] = 1; }
  // stay resident for a while so that the grid does not drain as fast as it is dispatched
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000ull) {}
}
int main()
{
  const int n = 32400;
  unsigned int* d; unsigned long long* w;
  hipMalloc(&d, n * 4); hipMalloc(&w, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 32768, 0, d, w);
  hipDeviceSynchronize();
  std::vector<unsigned int> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  printf("first 64 workgroups -> XCC_ID & 0xf:");
  for (int i = 0; i < 64; ++i) printf(" %u", h[i] & 0xfu);
  printf("\n");
  int match = 0, hist[16] = {0};
  for (int i = 0; i < n; ++i) { match += ((h[i] & 0xfu) == (unsigned)(i % 8)); hist[h[i] & 0xfu]++; }
  printf("workgroups with XCC == blockIdx %% 8: %d of %d\nper-XCC counts:", match, n);
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\nraw register of workgroup 0: 0x%08x\n", h[0]);
  return 0;
}
