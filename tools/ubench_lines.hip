// micro-benchmark: cost of one 64-lane gather instruction (L1-resident data) as a function of how many DISTINCT 128-byte
// lines its lanes touch and of how the sharing lanes are arranged.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(16)));

template <int W>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ base, const unsigned* __restrict__ idx, float* out, int iters)
{
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned o = idx[tid];
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (W == 1) acc += base[o];
    else if (W == 4) { f4u v = *(const f4u*)(base + (o & ~3u)); acc += v.x + v.y + v.z + v.w; } // 16-byte aligned (round 3: the quad layout's loads)
    else { f2u v = *(const f2u*)(base + o); acc += v.x + v.y; }
    o = (o + 32u * 97u + (unsigned)(acc == 123.456f)) & 4095u; // move every lane by 97 lines inside a 16 KiB window: pattern preserved
  }
  out[tid] = acc;
}

int main()
{
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  const int blocks = 256 * 4, iters = 512;
  const size_t T = (size_t)blocks * 256;
  unsigned* idx; CK(hipMalloc(&idx, T * 4));
  float* out; CK(hipMalloc(&out, T * 4));
  std::vector<unsigned> h(T);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("lines/instr  arrangement      dword clk/instr   dwordx2 clk/instr   dwordx4 clk/instr\n");
  for (int consecutive = 1; consecutive >= 0; --consecutive)
    for (int nl : {1, 2, 4, 8, 16, 32, 64}) {
      // nl distinct lines per wave; lanes sharing a line are consecutive lanes (consecutive=1) or strided (lane % nl)
      for (size_t t = 0; t < T; ++t) {
        const unsigned lane = t & 63, wave = (unsigned)(t >> 6);
        const unsigned per = 64 / nl;
        const unsigned line = consecutive ? lane / per : lane % nl;
        const unsigned within = consecutive ? lane % per : lane / nl;
        h[t] = ((wave * 7u + line * 5u) % 128u) * 32u + (within % 30u);
      }
      CK(hipMemcpy(idx, h.data(), T * 4, hipMemcpyHostToDevice));
      float ms1, ms2, ms4;
      gather<4><<<blocks, 256>>>(d, idx, out, 8); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); gather<4><<<blocks, 256>>>(d, idx, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms4, a, b));
      gather<1><<<blocks, 256>>>(d, idx, out, 8); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); gather<1><<<blocks, 256>>>(d, idx, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms1, a, b));
      gather<2><<<blocks, 256>>>(d, idx, out, 8); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); gather<2><<<blocks, 256>>>(d, idx, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms2, a, b));
      const double winstr = (double)T / 64 * iters / 256; // wave-instructions per CU
      printf("%6d       %-14s  %8.1f          %8.1f          %8.1f\n", nl, consecutive ? "consecutive" : "interleaved", ms1 * 1e-3 * 2.4e9 / winstr, ms2 * 1e-3 * 2.4e9 / winstr,
             ms4 * 1e-3 * 2.4e9 / winstr);
    }
  return 0;
}
