// micro-benchmark: does the ALIGNMENT of a gather's per-lane address cost texture-addresser time?  64-lane loads from L1-resident data, every quad of 4
// consecutive lanes inside one 128-byte line (the march's pattern), loads of 2 / 4 / 8 / 16 bytes at byte offsets 0 ... 7 from an 8-byte boundary.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_align.hip -o /tmp/ubench_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned short us1 __attribute__((aligned(1)));
typedef unsigned int u1 __attribute__((aligned(1)));
typedef unsigned int u2v __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned int u4v __attribute__((ext_vector_type(4), aligned(1)));

template <int W>
__global__ __launch_bounds__(256) void gather(const unsigned char* __restrict__ base, const unsigned* __restrict__ idx, unsigned* out, int iters)
{
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned o = idx[tid]; // byte offset
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (W == 2) acc += *(const us1*)(base + o);
    else if (W == 4) acc += *(const u1*)(base + o);
    else if (W == 8) { u2v v = *(const u2v*)(base + o); acc += v.x + v.y; }
    else { u4v v = *(const u4v*)(base + o); acc += v.x + v.y + v.z + v.w; }
    o = (o + 128u * 97u + (acc == 0x12345u)) & 16383u; // move every lane by 97 lines inside a 16 KiB window: pattern and alignment preserved
  }
  out[tid] = acc;
}

template <int W>
float run(const unsigned char* d, const unsigned* idx, unsigned* out, int blocks, int iters)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<W><<<blocks, 256>>>(d, idx, out, 8); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); gather<W><<<blocks, 256>>>(d, idx, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main()
{
  unsigned char* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  const int blocks = 256 * 4, iters = 512;
  const size_t T = (size_t)blocks * 256;
  unsigned* idx; CK(hipMalloc(&idx, T * 4));
  unsigned* out; CK(hipMalloc(&out, T * 4));
  std::vector<unsigned> h(T);
  printf("quads per line: 1 (16 lines per instruction).  clocks per 64-lane instruction and CU at 2.4 GHz\n");
  printf("byte offset in 8-byte row   2-byte   4-byte   8-byte   16-byte\n");
  for (int off = 0; off < 8; ++off) {
    for (size_t t = 0; t < T; ++t) {
      const unsigned lane = t & 63, wave = (unsigned)(t >> 6);
      const unsigned line = lane / 4, within = lane % 4;
      // 4 lanes of a quad at rows 0, 2, 4, 6 of a 128-byte line (16-byte spacing), each at byte `off` of its row
      h[t] = ((wave * 7u + line * 5u) % 128u) * 128u + within * 32u + (unsigned)off;
    }
    CK(hipMemcpy(idx, h.data(), T * 4, hipMemcpyHostToDevice));
    const double winstr = (double)T / 64 * iters / 256; // wave-instructions per CU
    const float m2 = run<2>(d, idx, out, blocks, iters), m4 = run<4>(d, idx, out, blocks, iters), m8 = run<8>(d, idx, out, blocks, iters), m16 = run<16>(d, idx, out, blocks, iters);
    printf("%8d                  %7.1f  %7.1f  %7.1f  %7.1f\n", off, m2 * 1e-3 * 2.4e9 / winstr, m4 * 1e-3 * 2.4e9 / winstr, m8 * 1e-3 * 2.4e9 / winstr, m16 * 1e-3 * 2.4e9 / winstr);
  }
  printf("\nspacing of a quad's 4 lanes inside ONE line (bytes), offset 0:   2-byte   4-byte   8-byte   16-byte\n");
  for (int sp : {4, 8, 16, 32}) {
    for (size_t t = 0; t < T; ++t) {
      const unsigned lane = t & 63, wave = (unsigned)(t >> 6);
      h[t] = ((wave * 7u + (lane / 4) * 5u) % 128u) * 128u + (lane % 4) * (unsigned)sp;
    }
    CK(hipMemcpy(idx, h.data(), T * 4, hipMemcpyHostToDevice));
    const double winstr = (double)T / 64 * iters / 256;
    const float m2 = run<2>(d, idx, out, blocks, iters), m4 = run<4>(d, idx, out, blocks, iters), m8 = run<8>(d, idx, out, blocks, iters), m16 = sp >= 16 ? run<16>(d, idx, out, blocks, iters) : 0.f;
    printf("%8d                                                      %7.1f  %7.1f  %7.1f  %7.1f\n", sp, m2 * 1e-3 * 2.4e9 / winstr, m4 * 1e-3 * 2.4e9 / winstr, m8 * 1e-3 * 2.4e9 / winstr, m16 * 1e-3 * 2.4e9 / winstr);
  }
  printf("\nlines a quad's 4 lanes touch (8-byte aligned loads):            4-byte   8-byte   16-byte\n");
  for (int nl : {1, 2, 4}) {
    for (size_t t = 0; t < T; ++t) {
      const unsigned lane = t & 63, wave = (unsigned)(t >> 6);
      const unsigned q = lane / 4, w = lane % 4;
      h[t] = ((wave * 7u + q * 5u + (w % nl) * 31u) % 128u) * 128u + (w / nl) * 32u;
    }
    CK(hipMemcpy(idx, h.data(), T * 4, hipMemcpyHostToDevice));
    const double winstr = (double)T / 64 * iters / 256;
    const float m4 = run<4>(d, idx, out, blocks, iters), m8 = run<8>(d, idx, out, blocks, iters), m16 = run<16>(d, idx, out, blocks, iters);
    printf("%8d                                                      %7.1f  %7.1f  %7.1f\n", nl, m4 * 1e-3 * 2.4e9 / winstr, m8 * 1e-3 * 2.4e9 / winstr, m16 * 1e-3 * 2.4e9 / winstr);
  }
  return 0;
}
