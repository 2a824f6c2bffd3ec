// ubench_valu.hip - issue cost of the vector instructions the shadow march is made of, relative to v_fma_f32 (4 clocks per wave64 on
// one SIMD, MI355X_MICROARCH.md): is v_pk_fma_f32 one issue slot for two fmas, what do v_log / v_exp / v_med3 / v_cvt / LDS reads cost.
// Every kernel runs N x 32 copies of ONE instruction on independent registers; W waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

// independent accumulators: 8 registers cycled by the 32 copies (REP8 x 4) would make chains; each asm names its own operands instead
#define D1 float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f; float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7
#define S1 if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.f) out[threadIdx.x] = r0
#define EIGHT(OP)                                                                                          \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                             \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c) : "vcc", "s10", "s11");

#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %8, %" #i "\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CVT(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define OP_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define OP_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_CMP(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_SUB(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define OP_MIN(i) "v_min_f32 %" #i ", %8, %" #i "\n"
#define OP_MAXE64(i) "v_max_f32_e64 %" #i ", %8, %" #i "\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define OP_ADDC(i) "v_addc_co_u32 %" #i ", vcc, 0, %" #i ", vcc\n"
#define OP_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_CNDMASKC(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[10:11]\n"
#define OP_CMP64(i) "v_cmp_gt_f32_e64 s[10:11], %" #i ", %8\n"
#define OP_MOV64(i) "v_mov_b64 %" #i ", %8\n"
#define OP_FMACLAMP(i) "v_fma_f32 %" #i ", %8, %9, %" #i " clamp\n"

__global__ __launch_bounds__(256) void k_fma(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_FMA) EIGHT(OP_FMA) EIGHT(OP_FMA) EIGHT(OP_FMA) } S1; }
__global__ __launch_bounds__(256) void k_add(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_ADD) EIGHT(OP_ADD) EIGHT(OP_ADD) EIGHT(OP_ADD) } S1; }
__global__ __launch_bounds__(256) void k_mov(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_MOV) EIGHT(OP_MOV) EIGHT(OP_MOV) EIGHT(OP_MOV) } S1; }
__global__ __launch_bounds__(256) void k_med3(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_MED3) EIGHT(OP_MED3) EIGHT(OP_MED3) EIGHT(OP_MED3) } S1; }
__global__ __launch_bounds__(256) void k_cvt(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_CVT) EIGHT(OP_CVT) EIGHT(OP_CVT) EIGHT(OP_CVT) } S1; }
__global__ __launch_bounds__(256) void k_fract(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_FRACT) EIGHT(OP_FRACT) EIGHT(OP_FRACT) EIGHT(OP_FRACT) } S1; }
__global__ __launch_bounds__(256) void k_log(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_LOG) EIGHT(OP_LOG) EIGHT(OP_LOG) EIGHT(OP_LOG) } S1; }
__global__ __launch_bounds__(256) void k_exp(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_EXP) EIGHT(OP_EXP) EIGHT(OP_EXP) EIGHT(OP_EXP) } S1; }
__global__ __launch_bounds__(256) void k_rsq(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_RSQ) EIGHT(OP_RSQ) EIGHT(OP_RSQ) EIGHT(OP_RSQ) } S1; }
__global__ __launch_bounds__(256) void k_lshladd(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_LSHLADD) EIGHT(OP_LSHLADD) EIGHT(OP_LSHLADD) EIGHT(OP_LSHLADD) } S1; }
__global__ __launch_bounds__(256) void k_add3(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_ADD3) EIGHT(OP_ADD3) EIGHT(OP_ADD3) EIGHT(OP_ADD3) } S1; }
__global__ __launch_bounds__(256) void k_cndmask(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_CNDMASK) EIGHT(OP_CNDMASK) EIGHT(OP_CNDMASK) EIGHT(OP_CNDMASK) } S1; }
__global__ __launch_bounds__(256) void k_cmp(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_CMP) EIGHT(OP_CMP) EIGHT(OP_CMP) EIGHT(OP_CMP) } S1; }
__global__ __launch_bounds__(256) void k_fmaclamp(float* out, int n, float seed) { D1; for (int i = 0; i < n; ++i) { EIGHT(OP_FMACLAMP) EIGHT(OP_FMACLAMP) EIGHT(OP_FMACLAMP) EIGHT(OP_FMACLAMP) } S1; }

#define KERN1(NAME, OP) __global__ __launch_bounds__(256) void NAME(float* out, int n, float seed) { D1; asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[10:11], 0x3333" ::: "vcc", "s10", "s11"); for (int i = 0; i < n; ++i) { EIGHT(OP) EIGHT(OP) EIGHT(OP) EIGHT(OP) } S1; }
KERN1(k_fmac, OP_FMAC) KERN1(k_mul, OP_MUL) KERN1(k_sub, OP_SUB) KERN1(k_min, OP_MIN) KERN1(k_maxe64, OP_MAXE64) KERN1(k_addu, OP_ADDU) KERN1(k_addc, OP_ADDC)
KERN1(k_cndmask64, OP_CNDMASK64) KERN1(k_cndmaskc, OP_CNDMASKC) KERN1(k_cmp64, OP_CMP64) KERN1(k_cndmask_init, OP_CNDMASK)

// packed f32: 8 independent register pairs
#define D2 f2 b = { seed, seed * 0.5f }, c = { 1.0001f, 0.9999f }; f2 r0 = { seed + threadIdx.x, 1 }, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f
#define S2 if (r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y == 12345.f) out[threadIdx.x] = r0.x
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_PKFMA_SEL(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel_hi:[0,1,1]\n"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n"
#define OP_PKADD_NEG(i) "v_pk_add_f32 %" #i ", %8, %" #i " neg_lo:[0,1] neg_hi:[0,1]\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_PKMOV(i) "v_pk_mov_b32 %" #i ", %8, %9 op_sel:[1,0]\n"
__global__ __launch_bounds__(256) void k_pkfma(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKFMA) EIGHT(OP_PKFMA) EIGHT(OP_PKFMA) EIGHT(OP_PKFMA) } S2; }
__global__ __launch_bounds__(256) void k_pkfma_sel(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKFMA_SEL) EIGHT(OP_PKFMA_SEL) EIGHT(OP_PKFMA_SEL) EIGHT(OP_PKFMA_SEL) } S2; }
__global__ __launch_bounds__(256) void k_pkadd(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKADD) EIGHT(OP_PKADD) EIGHT(OP_PKADD) EIGHT(OP_PKADD) } S2; }
__global__ __launch_bounds__(256) void k_pkadd_neg(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKADD_NEG) EIGHT(OP_PKADD_NEG) EIGHT(OP_PKADD_NEG) EIGHT(OP_PKADD_NEG) } S2; }
__global__ __launch_bounds__(256) void k_pkmul(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKMUL) EIGHT(OP_PKMUL) EIGHT(OP_PKMUL) EIGHT(OP_PKMUL) } S2; }
__global__ __launch_bounds__(256) void k_mov64(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_MOV64) EIGHT(OP_MOV64) EIGHT(OP_MOV64) EIGHT(OP_MOV64) } S2; }
__global__ __launch_bounds__(256) void k_pkmov(float* out, int n, float seed) { D2; for (int i = 0; i < n; ++i) { EIGHT(OP_PKMOV) EIGHT(OP_PKMOV) EIGHT(OP_PKMOV) EIGHT(OP_PKMOV) } S2; }

// dependent chains (latency, one wave): the same register through 32 instructions
#define CH(OP) asm volatile(OP OP OP OP OP OP OP OP : "+v"(r0) : "v"(b), "v"(c));
__global__ __launch_bounds__(64) void k_fma_chain(float* out, int n, float seed) { float r0 = seed + threadIdx.x, b = seed, c = 1.0001f; for (int i = 0; i < n; ++i) { CH("v_fma_f32 %0, %1, %2, %0\n") CH("v_fma_f32 %0, %1, %2, %0\n") CH("v_fma_f32 %0, %1, %2, %0\n") CH("v_fma_f32 %0, %1, %2, %0\n") } if (r0 == 12345.f) out[threadIdx.x] = r0; }
__global__ __launch_bounds__(64) void k_pkfma_chain(float* out, int n, float seed) { f2 r0 = { seed + threadIdx.x, 1 }, b = { seed, seed }, c = { 1.0001f, 1.f }; for (int i = 0; i < n; ++i) { CH("v_pk_fma_f32 %0, %1, %2, %0\n") CH("v_pk_fma_f32 %0, %1, %2, %0\n") CH("v_pk_fma_f32 %0, %1, %2, %0\n") CH("v_pk_fma_f32 %0, %1, %2, %0\n") } if (r0.x + r0.y == 12345.f) out[threadIdx.x] = r0.x; }
__global__ __launch_bounds__(64) void k_log_chain(float* out, int n, float seed) { float r0 = seed + threadIdx.x, b = seed, c = 1.0001f; for (int i = 0; i < n; ++i) { CH("v_log_f32 %0, %0\n") CH("v_log_f32 %0, %0\n") CH("v_log_f32 %0, %0\n") CH("v_log_f32 %0, %0\n") } if (r0 == 12345.f) out[threadIdx.x] = r0; (void)b; (void)c; }

// LDS reads: ds_read_b32 / ds_read2_b32 with per-lane addresses spread over a 4 KiB table (bank conflicts as the TF lookups have them)
__global__ __launch_bounds__(256) void k_ds_read(float* out, int n, float seed)
{
  __shared__ float tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = seed + i;
  __syncthreads();
  unsigned a0 = ((threadIdx.x * 37u) & 1023u) * 4u, a1 = ((threadIdx.x * 53u + 7u) & 1023u) * 4u;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) {
    float v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %8 offset:4\n ds_read_b32 %3, %9 offset:4\n"
                 "ds_read_b32 %4, %8 offset:8\n ds_read_b32 %5, %9 offset:8\n ds_read_b32 %6, %8 offset:12\n ds_read_b32 %7, %9 offset:12\n s_waitcnt lgkmcnt(0)\n"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a0), "v"(a1));
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    a0 = (a0 + 148u) & 4092u; a1 = (a1 + 212u) & 4092u;
  }
  if (acc == 12345.f) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_ds_read2(float* out, int n, float seed)
{
  __shared__ float tab[1024 + 8];
  for (int i = threadIdx.x; i < 1032; i += 256) tab[i] = seed + i;
  __syncthreads();
  unsigned a0 = ((threadIdx.x * 37u) & 1023u) * 4u, a1 = ((threadIdx.x * 53u + 7u) & 1023u) * 4u;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) {
    f2 v0, v1, v2, v3;
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %5 offset1:1\n ds_read2_b32 %2, %4 offset0:2 offset1:3\n ds_read2_b32 %3, %5 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)\n"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a0), "v"(a1));
    acc += v0.x + v1.y + v2.x + v3.y;
    a0 = (a0 + 148u) & 4092u; a1 = (a1 + 212u) & 4092u;
  }
  if (acc == 12345.f) out[threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, int, float);
static double run(kern_t k, int blocks, int threads, int n, float* out)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 16, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, n, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  float* out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  const int n = 20000;
  struct { const char* name; kern_t k; int per_iter; } ks[] = {
    { "v_fma_f32", k_fma, 32 }, { "v_add_f32", k_add, 32 }, { "v_mov_b32", k_mov, 32 }, { "v_med3_f32", k_med3, 32 }, { "v_cvt_i32_f32", k_cvt, 32 },
    { "v_fract_f32", k_fract, 32 }, { "v_log_f32", k_log, 32 }, { "v_exp_f32", k_exp, 32 }, { "v_rsq_f32", k_rsq, 32 }, { "v_lshl_add_u32", k_lshladd, 32 },
    { "v_add3_u32", k_add3, 32 }, { "v_cndmask_b32", k_cndmask, 32 }, { "v_cmp_gt_f32", k_cmp, 32 }, { "v_fma_f32 clamp", k_fmaclamp, 32 },
    { "v_pk_fma_f32", k_pkfma, 32 }, { "v_pk_fma_f32 op_sel_hi:[0,1,1]", k_pkfma_sel, 32 }, { "v_pk_add_f32", k_pkadd, 32 }, { "v_pk_add_f32 neg", k_pkadd_neg, 32 },
    { "v_pk_mul_f32", k_pkmul, 32 }, { "v_mov_b64", k_mov64, 32 }, { "v_fmac_f32", k_fmac, 32 }, { "v_mul_f32", k_mul, 32 }, { "v_sub_f32", k_sub, 32 }, { "v_min_f32", k_min, 32 },
    { "v_max_f32_e64", k_maxe64, 32 }, { "v_add_u32", k_addu, 32 }, { "v_addc_co_u32 vcc", k_addc, 32 }, { "v_cndmask_b32_e64 sgpr mask", k_cndmask64, 32 },
    { "v_cndmask_b32_e64 0, v, sgpr mask", k_cndmaskc, 32 }, { "v_cmp_gt_f32_e64 -> sgpr", k_cmp64, 32 }, { "v_cndmask_b32 vcc (vcc initialised)", k_cndmask_init, 32 }, { "v_pk_mov_b32", k_pkmov, 32 }, { "ds_read_b32 (scattered)", k_ds_read, 8 }, { "ds_read2_b32 (scattered)", k_ds_read2, 4 },
  };
  printf("%d CUs; time of N x 32 independent copies, W waves per SIMD on every CU; cost relative to v_fma_f32 at the same W\n", cus);
  for (int w : { 1, 2, 4 }) {
    double base = 0;
    for (auto& k : ks) {
      const double ms = run(k.k, cus * w, 256, n, out);
      const double per = ms * 1e6 / ((double)n * k.per_iter * w); // ns per instruction per wave-slot
      if (k.k == (kern_t)k_fma) base = per;
      printf("W=%d %-34s %8.3f ms  %7.4f ns/instr  x%.2f of v_fma_f32\n", w, k.name, ms, per, per / base);
    }
  }
  {
    const double a = run(k_fma_chain, cus * 4, 64, n, out), b = run(k_pkfma_chain, cus * 4, 64, n, out), c = run(k_log_chain, cus * 4, 64, n, out);
    printf("dependent chains, one wave per SIMD: v_fma_f32 %.4f ns  v_pk_fma_f32 %.4f ns  v_log_f32 %.4f ns per instruction\n", a * 1e6 / (n * 32.0), b * 1e6 / (n * 32.0),
           c * 1e6 / (n * 32.0));
  }
  hipFree(out);
  return 0;
}
