// micro-benchmark: throughput of gather-style global loads on gfx950 (guides the volume layout).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o gpurun_out/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f2a __attribute__((ext_vector_type(2), aligned(8)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// mode 0: aligned dwordx2 ; 1: misaligned dwordx2 (offset odd) ; 2: two dword loads ; 3: one dword ; 4: misaligned dwordx4 ; 5: 4 dwords
template <int MODE>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ base, const unsigned* __restrict__ idx, float* out, int iters, unsigned mask)
{
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned o = idx[tid];
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const unsigned off = (o & mask);
    if (MODE == 0) { f2a v = *(const f2a*)(base + (off & ~1u)); acc += v.x + v.y; }
    if (MODE == 1) { f2u v = *(const f2u*)(base + (off | 1u)); acc += v.x + v.y; }
    if (MODE == 2) { acc += base[off] + base[off + 1]; }
    if (MODE == 3) { acc += base[off]; }
    if (MODE == 4) { f4u v = *(const f4u*)(base + (off | 1u)); acc += v.x + v.y + v.z + v.w; }
    if (MODE == 5) { acc += base[off] + base[off + 1] + base[off + 2] + base[off + 3]; }
    o = o * 1664525u + 1013904223u + (unsigned)(acc == 123.456f);
  }
  out[tid] = acc;
}

template <int MODE>
double run(const float* d, const unsigned* idx, float* out, int blocks, int iters, unsigned mask)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<MODE><<<blocks, 256>>>(d, idx, out, 4, mask);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  gather<MODE><<<blocks, 256>>>(d, idx, out, iters, mask);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main()
{
  const size_t N = 1u << 28; // 1 GiB of floats
  float* d; CK(hipMalloc(&d, N * 4 + 64)); CK(hipMemset(d, 0, N * 4 + 64));
  const int blocks = 256 * 8, iters = 256;
  const size_t T = (size_t)blocks * 256;
  // index patterns: lanes of a wave spread with a given stride (in floats) around a per-wave random base
  unsigned* idx; CK(hipMalloc(&idx, T * 4));
  float* out; CK(hipMalloc(&out, T * 4));
  std::vector<unsigned> h(T);
  const char* names[] = {"aligned x2", "misaligned x2", "2 x dword", "1 x dword", "misaligned x4", "4 x dword"};
  // footprints: working set mask (bytes): 16 KiB (L1), 2 MiB (L2), 128 MiB (MALL), 1 GiB (HBM)
  const unsigned masks[] = {(1u << 12) - 1, (1u << 19) - 1, (1u << 25) - 1, (1u << 28) - 1};
  const char* mnames[] = {"16KiB", "2MiB", "128MiB", "1GiB"};
  for (int mi = 0; mi < 4; ++mi) {
    for (size_t i = 0; i < T; ++i) h[i] = (unsigned)rand() * 2654435761u;
    CK(hipMemcpy(idx, h.data(), T * 4, hipMemcpyHostToDevice));
    printf("== fully random per lane, working set %s\n", mnames[mi]);
    double ms[6];
    ms[0] = run<0>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    ms[1] = run<1>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    ms[2] = run<2>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    ms[3] = run<3>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    ms[4] = run<4>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    ms[5] = run<5>(d, idx, out, blocks, iters, masks[mi] & ~3u);
    for (int m = 0; m < 6; ++m) {
      const double winstr = (double)T / 64 * iters; // wave-iterations
      printf("  %-14s %8.3f ms  %7.1f clk/wave-iter/CU (2.4GHz)  %.2f G lane-iter/s\n", names[m], ms[m],
             ms[m] * 1e-3 * 2.4e9 / (winstr / 256), (double)T * iters / ms[m] / 1e6);
    }
  }
  return 0;
}
