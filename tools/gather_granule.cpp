// What does a scattered 8-byte load cost the memory side on gfx950: a whole 128-byte line, or the 64-byte half it lies in?
// (C4's rays are 3.4-4.8 voxels apart: most of every 128-byte brick they fetch is never used.  A layout of 64-byte bricks would only pay if the
// memory side moved 64-byte pieces.)
//   a: one 8-byte load at the start of a random 128-byte line
//   b: two loads in ONE random line (offsets 0 and 64)
//   c: two loads in two random lines (offset 0 each)
//   d: two loads in two random lines, the second one in its upper half (offset 64)
//   e: the march's shape - the four lanes of a ray read 8 bytes each from ONE random line (16 lines per wave instruction)
// whole lines  =>  b ~ a, c ~ d ~ 2a;   64-byte pieces  =>  b ~ c ~ d ~ 2a (in bytes: a moves 64, not 128)
// build: hipcc --offload-arch=gfx950 -O2 tools/gather_granule.cpp -o gpurun_out/gather_granule ; run under rocprofv3 --pmc for the byte counts
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t v)
{
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(256) gather(const char* __restrict__ table, uint64_t n_lines, int per_thread, uint64_t seed, uint64_t* __restrict__ out)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        const uint64_t who = MODE == 4 ? tid >> 2 : tid;
        const uint64_t h = mix(seed + who * 0x9e3779b97f4a7c15ull + (uint64_t)i * 0xd1b54a32d192ed03ull);
        const uint64_t l0 = h % n_lines, l1 = mix(h) % n_lines;
        acc += *reinterpret_cast<const uint64_t*>(table + l0 * 128 + (MODE == 4 ? (tid & 3) * 32 : 0));
        if (MODE == 1) acc += *reinterpret_cast<const uint64_t*>(table + l0 * 128 + 64);
        if (MODE == 2) acc += *reinterpret_cast<const uint64_t*>(table + l1 * 128);
        if (MODE == 3) acc += *reinterpret_cast<const uint64_t*>(table + l1 * 128 + 64);
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;   // never true for the fill pattern: keeps the loads alive
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;   // 16 GiB: every line from HBM; 0.0625: from the 256 MB memory-side cache; 0.002: from the L2s
    printf("table of %g GiB\n", gib);
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)), n_lines = bytes / 128;
    char* table; uint64_t* out;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(table, 1, bytes));
    const int blocks = 256 * 64, per_thread = 16;
    const double accesses = (double)blocks * 256 * per_thread;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"a one load per line", "b two loads in one line (0, 64)", "c two loads in two lines (0, 0)", "d two loads in two lines (0, 64)", "e four lanes per line, one line each"};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            CHECK(hipEventRecord(e0));
            const uint64_t seed = 1000 * rep + mode;
            if (mode == 0) gather<0><<<blocks, 256>>>(table, n_lines, per_thread, seed, out);
            if (mode == 1) gather<1><<<blocks, 256>>>(table, n_lines, per_thread, seed, out);
            if (mode == 2) gather<2><<<blocks, 256>>>(table, n_lines, per_thread, seed, out);
            if (mode == 3) gather<3><<<blocks, 256>>>(table, n_lines, per_thread, seed, out);
            if (mode == 4) gather<4><<<blocks, 256>>>(table, n_lines, per_thread, seed, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double lines = accesses * (mode == 2 || mode == 3 ? 2 : mode == 4 ? 0.25 : 1);
            if (rep) printf("%-40s %8.3f ms  %7.2f G lines/s  %6.2f G wave-loads/s\n", names[mode], ms, lines / ms * 1e-6, accesses * (mode >= 1 && mode <= 3 ? 2 : 1) / 64 / ms * 1e-6);
        }
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
