// ovr_hip_device.h - device code of the ray-march path that depends on the voxel type: helpers, the bricked voxel access,
// raymarch_kernel / shade_pool_kernel and their launch dispatch.  Included by ovr_hip_kernels.hip (type-independent kernels
// and the launch interface) and by ovr_hip_march_<type>.hip (one explicit instantiation of launch_v per voxel type).
// See ovr_hip_kernels.hip for the overview.
#pragma once
#include "ovr_hip_kernels.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace ovrhip {

// ------------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float3_& a) { return mk3(a.x, a.y, a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// clamp(x, lo, hi) for lo <= hi as ONE instruction: v_med3_f32 returns the median of its operands, and min3 of them when one is NaN - with
// IEEE v_min that is the smallest non-NaN operand, lo - so the result equals fminf(fmaxf(x, lo), hi) for every input, NaN -> lo like CUDA's
// fminf / fmaxf (the two-instruction form was ~9 % of the shadow march's instructions, which bound the all-shaded frames)
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); } // NaN -> 0 like CUDA fminf/fmaxf (folds into a clamp modifier)
__device__ __forceinline__ float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
// gdt normalize (extern/gdt/gdt/math/vec.h:443-448) with the hardware reciprocal square root (1 ulp)
// OVR_PARITY_EXACT = 1 builds the PARITY INSTRUMENT (libovr_hip_parity.so; never the product, nothing loads it by default): the four places where the
// kernels deliberately depart from the oracle's float arithmetic take the oracle's form instead - (1) the opacity correction's __powf is
// the machine-independent det_powf below instead of v_exp_f32(y * v_log_f32(x)), (2) the three per-sample normalisations divide by sqrtf instead of
// multiplying with v_rsq_f32, (3) the gradient divides by the step instead of multiplying with its reciprocal, (4) 8-bit voxels are normalised one by
// one before the filter instead of once behind it.  tests/test_parity_exact_gpu.py: frames and every counter then equal the oracle's (mode "det") - which
// pins those four as the ONLY sources of the product's tolerated differences.
#ifndef OVR_PARITY_EXACT
#define OVR_PARITY_EXACT 0
#endif
__device__ __forceinline__ f3 normalize3(f3 v)
{
#if OVR_PARITY_EXACT
  const float l = sqrtf(dot3(v, v)); // gdt normalize: (v * 1.f) / sqrt(dot(v, v)), extern/gdt/gdt/math/vec.h:443-448
  return mk3(v.x / l, v.y / l, v.z / l);
#endif
  const float r = __builtin_amdgcn_rsqf(dot3(v, v));
  return mk3(v.x * r, v.y * r, v.z * r);
}
// IEEE-exact normalize for the once-per-ray direction
__device__ __forceinline__ f3 normalize3_exact(f3 v)
{
  const float l = sqrtf(dot3(v, v));
  return mk3(v.x / l, v.y / l, v.z / l);
}

// opacity correction, shaders_raymarching.cu:118-122: 1 - __powf(1 - a, base*dt); __powf == exp2(y * log2(x)).
// BF: branch-free form (bit select instead of the exec-mask branch the compiler builds around the two transcendentals);
// same value - adj == 1 keeps a exactly as the reference's branch does.  Only the skipping shadow march gains from it
// (0.85 -> 0.71 ms on C3); the other kernels are measurably slower with it, so they keep the branch.
// A log2 / exp2 pair that is the same float arithmetic on every machine (fmaf Horner chains, integer exponent handling; ~1 ulp each, the accuracy
// class of v_log_f32 / v_exp_f32).  NOT the product's pow: a library built with -DOVR_PARITY_EXACT=1 (libovr_hip_parity.so) evaluates __powf with it, and so
// does the CPU oracle in its mode 2 - with the same pow on both sides every sample count equals the oracle's exactly (tests/test_parity_exact_gpu.py): the
// last bit of the transcendentals is all that the parity tests' remaining tolerances come from.  Constants: tests/golden/make_detpow.py.
__device__ __forceinline__ float det_log2f(float x)
{
  if (!(x > 0.f)) return x == 0.f ? -__builtin_inff() : __builtin_nanf("");
  if (x > 3.402823466e+38f) return x;
  unsigned int ix = __float_as_uint(x);
  int e = (int)(ix >> 23) - 127;
  if (e == -127) { ix = __float_as_uint(x * 8388608.f); e = (int)(ix >> 23) - 127 - 23; }
  float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
  if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.f;
  float p = -0x1.b8f078p-4f;
  p = fmaf(p, f, 0x1.7aec18p-3f);
  p = fmaf(p, f, -0x1.881ca4p-3f);
  p = fmaf(p, f, 0x1.a37bc2p-3f);
  p = fmaf(p, f, -0x1.eab168p-3f);
  p = fmaf(p, f, 0x1.277a9ap-2f);
  p = fmaf(p, f, -0x1.715a9p-2f);
  p = fmaf(p, f, 0x1.ec70a8p-2f);
  p = fmaf(p, f, -0x1.71547p-1f);
  p = fmaf(p, f, 0x1.715476p+0f);
  return fmaf(f, p, (float)e);
}
__device__ __forceinline__ float det_exp2f(float m)
{
  if (m != m) return m;
  if (m >= 128.f) return __builtin_inff();
  if (m < -126.f) return 0.f;
  const float n = floorf(m + 0.5f);
  const float r = m - n;
  float p = 0x1.00c0e4p-16f;
  p = fmaf(p, r, 0x1.446c7ap-13f);
  p = fmaf(p, r, 0x1.5d8776p-10f);
  p = fmaf(p, r, 0x1.3b29d8p-7f);
  p = fmaf(p, r, 0x1.c6b08ep-5f);
  p = fmaf(p, r, 0x1.ebfbep-3f);
  p = fmaf(p, r, 0x1.62e43p-1f);
  p = fmaf(p, r, 1.f);
  const int in = (int)n;
  const int h = in / 2;
  const float s = p * __uint_as_float((unsigned int)(h + 127) << 23) * __uint_as_float((unsigned int)(in - h + 127) << 23);
  return s < 1.175494351e-38f ? 0.f : s;
}
__device__ __forceinline__ float det_powf(float x, float y) { return det_exp2f(y * det_log2f(x)); }
template <bool BF>
__device__ __forceinline__ float opacity_correction(float a, float adj)
{
#if OVR_PARITY_EXACT
  if (!(fabsf(adj - 1.f) < 1e-7f)) a = clamp01(1.f - det_powf(1.f - a, adj));
  return a;
#endif
  if (BF) {
    const float pw = __builtin_amdgcn_exp2f(adj * __builtin_amdgcn_logf(1.f - a));
    const float c = clamp01(1.f - pw);
    const unsigned int m = (fabsf(adj - 1.f) < 1e-7f) ? 0u : ~0u;
    return __uint_as_float((__float_as_uint(c) & m) | (__float_as_uint(a) & ~m));
  }
  if (!(fabsf(adj - 1.f) < 1e-7f)) {
    const float pw = __builtin_amdgcn_exp2f(adj * __builtin_amdgcn_logf(1.f - a));
    a = clamp01(1.f - pw);
  }
  return a;
}

// ------------------------------------------------------------------------------------------------------------------
// voxel access.  Volume layout in HBM: 128-byte bricks with an x-apron, inside macro blocks (see ovr_hip_kernels.h).
// One L1/L2 line is one brick.  A brick stores CX+1 voxels along x (the last one duplicates the first of its +x neighbour),
// so the two x-neighbours of a trilinear tap always sit next to each other in ONE brick and a tap is 4 pair loads
// (f32: 4 x 8 bytes) instead of 8 scalar loads.  The texture addresser spends ~41 clocks on a 64-lane gather instruction
// whatever its width (tools/ubench_gather.hip), so halving the instruction count halves the cost of the path's bottleneck;
// the price is 4/3 (8/7 for 8-bit) of the memory.   f32: (3+1)x4x2   u16/i16: (3+1)x4x4   u8/i8: (7+1)x4x4
// The element offset is separable: off(x,y,z) = X(x) + Y(y) + Z(z).
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned short u16x2_u __attribute__((ext_vector_type(2), aligned(2)));
typedef short i16x2_u __attribute__((ext_vector_type(2), aligned(2)));
typedef unsigned char u8x2_u __attribute__((ext_vector_type(2), aligned(1)));
typedef signed char i8x2_u __attribute__((ext_vector_type(2), aligned(1)));

template <int VT> struct Vox;
// f32 brick shape (experiment switches; the default is what the A/B runs of profiles/r02_notes.md keep)
#ifndef OVR_F32_CX
#define OVR_F32_CX 3
#define OVR_F32_MBX 10
#define OVR_F32_BY 2
#define OVR_F32_BZ 1
#endif
template <> struct Vox<VOX_F32> {
  typedef float T; typedef f32x2_u P;
  static constexpr int cx = OVR_F32_CX, mbx = OVR_F32_MBX, by = OVR_F32_BY, bz = OVR_F32_BZ; // cells per brick in x, bricks per macro block in x, log2 brick y/z
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
#ifndef OVR_U16_CX
#define OVR_U16_CX 3
#define OVR_U16_MBX 10
#define OVR_U16_BY 2
#define OVR_U16_BZ 2
#endif
template <> struct Vox<VOX_U16> {
  typedef unsigned short T; typedef u16x2_u P;
  static constexpr int cx = OVR_U16_CX, mbx = OVR_U16_MBX, by = OVR_U16_BY, bz = OVR_U16_BZ;
  static constexpr bool kScale = false, kClamp = false; // u16 is sampled as RAW float (array.cpp:335-338)
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
// Thin replicas (view-dependent layout choice, see VoxelType in ovr_hip_kernels.h): 1 cell + apron along the pair axis, 4 x 4
// (f32) or 4 x 8 (u16) voxels across.  *_TT is stored with x and y exchanged, so its pair axis is the volume's y.
#ifndef OVR_U16T_BY
#define OVR_U16T_BY 2
#define OVR_U16T_BZ 3
#endif
template <> struct Vox<VOX_F32_T> {
  typedef float T; typedef f32x2_u P;
  static constexpr int cx = 1, mbx = 32, by = 2, bz = 2;
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
template <> struct Vox<VOX_F32_TT> : Vox<VOX_F32_T> { static constexpr bool kTransposed = true; };
template <> struct Vox<VOX_U16_T> {
  typedef unsigned short T; typedef u16x2_u P;
  static constexpr int cx = 1, mbx = 32, by = OVR_U16T_BY, bz = OVR_U16T_BZ;
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
template <> struct Vox<VOX_U16_TT> : Vox<VOX_U16_T> { static constexpr bool kTransposed = true; };
template <> struct Vox<VOX_I16> {
  typedef short T; typedef i16x2_u P;
  static constexpr int cx = 3, mbx = 10, by = 2, bz = 2;
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
template <> struct Vox<VOX_U8> {
  typedef unsigned char T; typedef u8x2_u P;
  static constexpr int cx = 7, mbx = 4, by = 2, bz = 2;
  static constexpr bool kScale = true, kClamp = false; // normalized read: v / 255 (array.cpp:304-306)
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};
template <> struct Vox<VOX_I8> {
  typedef signed char T; typedef i8x2_u P;
  static constexpr int cx = 7, mbx = 4, by = 2, bz = 2;
  static constexpr bool kScale = true, kClamp = true; // max(v / 127, -1)
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = false;
};

// Quad replicas (VoxelType in ovr_hip_kernels.h): a cell stores its 2 x 2 (x, y) voxels as one 4-vector Q; a 128-byte brick holds
// 2^lx x 2^ly x 2^lz cells.  (cx / mbx / by / bz describe the brick to the code that is shared with the other layouts; P is not used)
typedef float f32x4_q __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_q __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4_q __attribute__((ext_vector_type(4)));
template <> struct Vox<VOX_F32_Q> {
  typedef float T; typedef f32x2_u P; typedef f32x4_q Q;
  static constexpr int lx = 1, ly = 1, lz = 1;
  static constexpr int cx = 2, mbx = 16, by = 1, bz = 1;
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = true;
};
template <> struct Vox<VOX_U16_Q> {
  typedef unsigned short T; typedef u16x2_u P; typedef u16x4_q Q;
  static constexpr int lx = 2, ly = 1, lz = 1;
  static constexpr int cx = 4, mbx = 8, by = 1, bz = 1;
  static constexpr bool kScale = false, kClamp = false;
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = true;
};
template <> struct Vox<VOX_U8_Q> {
  typedef unsigned char T; typedef u8x2_u P; typedef u8x4_q Q;
  static constexpr int lx = 2, ly = 2, lz = 1;
  static constexpr int cx = 4, mbx = 8, by = 2, bz = 1;
  static constexpr bool kScale = true, kClamp = false; // normalized read like VOX_U8
  static constexpr bool kTransposed = false;
  static constexpr bool kQuad = true;
};

template <int VT> struct BrickMap {
  typedef Vox<VT> V;
  static constexpr unsigned SX = V::cx + 1;                           // stored voxels per brick row (4 or 8)
  static constexpr unsigned BV = SX << (V::by + V::bz);               // stored voxels per brick (128 bytes)
  static constexpr unsigned sby = V::mbx * BV, sbz = (32u >> V::by) * V::mbx * BV;
  static constexpr unsigned MV = (32u >> V::bz) * sbz;                // stored voxels per macro block
  static constexpr unsigned MCX = V::cx * V::mbx;                     // cells per macro block along x (30 or 28)
  static_assert(BV * sizeof(typename V::T) == 128, "a brick is exactly one 128-byte L1/L2 line");
  // exact for x < 65536: q = floor(x / d) = mulhi(x, ceil(2^32 / d))
  static __host__ __device__ __forceinline__ unsigned div_cx(unsigned x) { return (unsigned)(((unsigned long long)x * ((0xffffffffull / V::cx) + 1ull)) >> 32); }
  static __host__ __device__ __forceinline__ unsigned div_mbx(unsigned b) { return (unsigned)(((unsigned long long)b * ((0xffffffffull / V::mbx) + 1ull)) >> 32); }
  // offset of STORED POSITION u as the LOWER member of a pair.  Stored position = voxel index + 1 (round 4): position 0 is a copy of
  // voxel 0, positions past voxel n - 1 repeat it - the pair of voxel index i in [-1, n - 1] is positions (i + 1, i + 2), which is the
  // reference's clamp-to-edge pair (clamp(i), clamp(i + 1)) without a clamp in the tap (shaders_common.h:186-193, cuda_buffer.h:248-287)
  static __host__ __device__ __forceinline__ unsigned X(unsigned x)
  {
    const unsigned b = div_cx(x), xr = x - b * V::cx;
    const unsigned m = div_mbx(b), bm = b - m * V::mbx;
    return xr + bm * BV + m * MV;
  }
  static __host__ __device__ __forceinline__ unsigned Y(unsigned y, unsigned macro_y_stride)
  {
    return (y & ((1u << V::by) - 1u)) * SX + ((y >> V::by) & ((32u >> V::by) - 1u)) * sby + (y >> 5) * macro_y_stride;
  }
  static __host__ __device__ __forceinline__ unsigned Zlo(unsigned z)
  {
    return ((z & ((1u << V::bz) - 1u)) << V::by) * SX + ((z >> V::bz) & ((32u >> V::bz) - 1u)) * sbz;
  }
};

// quad replicas: offsets in voxels (elements of T); cell (u, v, z) -> 4 elements at X(u) + Y(v) + Z(z); cells x-fastest inside a brick, bricks
// x-fastest inside macro blocks of 32^3 cells, macro blocks x-fastest.  Cell (u, v) = (x + 1, y + 1) holds the voxels (clamp(x), clamp(x + 1)) x
// (clamp(y), clamp(y + 1)) for x in [-1, nx - 1], y in [-1, ny - 1] (round 4: the lower faces' clamp-to-edge pairs are cells of their own)
template <int VT> struct QuadMap {
  typedef Vox<VT> V;
  static constexpr unsigned SX = 2, BV = 128u / (unsigned)sizeof(typename V::T);   // elements per brick
  static constexpr unsigned MV = 32u * 32u * 32u * 4u, MCX = 32;                     // elements per macro block; cells per macro block along x
  static constexpr unsigned bx_ = 32u >> V::lx, by_ = 32u >> V::ly;                  // bricks per macro block along x / y
  static_assert((4u << (V::lx + V::ly + V::lz)) == BV, "a brick is exactly one 128-byte L1/L2 line");
  static __host__ __device__ __forceinline__ unsigned div_cx(unsigned x) { return x >> V::lx; }
  static __host__ __device__ __forceinline__ unsigned X(unsigned x) { return (x & ((1u << V::lx) - 1u)) * 4u + ((x >> V::lx) & (bx_ - 1u)) * BV + (x >> 5) * MV; }
  static __host__ __device__ __forceinline__ unsigned Y(unsigned y, unsigned macro_y_stride)
  {
    return (y & ((1u << V::ly) - 1u)) * (4u << V::lx) + ((y >> V::ly) & (by_ - 1u)) * (bx_ * BV) + (y >> 5) * macro_y_stride;
  }
  static __host__ __device__ __forceinline__ unsigned Zlo(unsigned z)
  {
    return (z & ((1u << V::lz) - 1u)) * (4u << (V::lx + V::ly)) + ((z >> V::lz) & ((32u >> V::lz) - 1u)) * (bx_ * by_ * BV);
  }
};
template <> struct BrickMap<VOX_F32_Q> : QuadMap<VOX_F32_Q> {};
template <> struct BrickMap<VOX_U16_Q> : QuadMap<VOX_U16_Q> {};
template <> struct BrickMap<VOX_U8_Q> : QuadMap<VOX_U8_Q> {};

struct VolConsts {
  const void* data;
  // per-axis offset tables in LDS (AM 0 / 1 / 2), bytes (AM 0) or elements (AM 1 / 2).  The pointers are BIASED by one entry: tab_x[i] is
  // valid for the voxel index i = -1 ... n - 1 (lower member of the pair), tab_y[i] / tab_z[i] for i = -1 ... n; index -1 addresses a copy of
  // voxel 0 and index n a copy of voxel n - 1, so (i, i + 1) is the reference's clamp-to-edge pair for every i = floor(x) without a select
  const unsigned int *tab_x, *tab_y, *tab_z;
  const unsigned long long* tab_z64; // AM 2: z offsets need 64 bits (>= 2^32 stored voxels)
  int nx1, ny1, nz1; // n - 1
  const float* majorant; // per-macrocell max TF opacity (null: empty-space skipping off)
  const unsigned char* occupancy; // per 4^3 macrocells: 1 if one of them, or a macrocell next to them, has majorant > 0
  const unsigned char* occupancy_fine; // the same per macrocell
  int mcx1, mcy1, mcz1;  // macrocell grid dims - 1
  unsigned int macro_y;          // stored elements between macro rows: MV * macros_x
  unsigned long long macro_z;    // stored elements between macro layers: MV * macros_x * macros_y
  f3 cs, cb;
  float vscale, vmin;
};

// (int)floor(x) as ONE instruction (the compiler only selects v_cvt_flr_i32_f32 under no-NaNs math; x is never NaN here)
__device__ __forceinline__ int floor_to_int(float x)
{
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
// One axis of tex3D's linear filter with clamp addressing (shaders_common.h:186-193; cuda_buffer.h:248-287): x = p * N - 0.5 lies in
// [-0.5, N - 0.5], the texels are (clamp(i0), clamp(i0 + 1)) for i0 = floor(x) in [-1, N - 1], the weight is x - i0.  Every layout stores a
// copy of voxel 0 at index -1 and of voxel N - 1 at index N (in the data along the pair axis, in the offset tables along the others), so
// neither the coordinate nor the indices are clamped (round 4; before, x was clamped to [0, N - 1] and the half voxel outside the first
// centre read (voxel 0, voxel 1) with weight 0 where the reference reads (0, 0): 0 x NaN for a non-finite voxel 1).  In that zone the weight
// is immaterial - lerp(a, a, f) = fma(f, a - a, a) is a for finite a and NaN otherwise, whatever f in [0, 1) - and for x >= 0 v_fract_f32 is the
// exact x - floor(x) of the reference's filter (in full precision: the oracle's convention).
__device__ __forceinline__ void axis_tap(float po, float cs, float cb, int& i0, float& f)
{
  const float p = clamp01(po);                       // sample_volume_object_space clamps p to [0,1] (NaN -> 0)
  const float x = fmaf(p, cs, cb);                   // cell-centred: p*N - 0.5
  f = __builtin_amdgcn_fractf(x);
  i0 = floor_to_int(x);
}

// One trilinear tap, split in two so that several taps can be in flight before the first is consumed
// (software pipelining: the march is latency-bound otherwise).  issue: 4 pair loads; finish: 7 lerps.
struct Tap {
  float c000, c100, c010, c110, c001, c101, c011, c111;
  float fx, fy, fz;
  int x0, y0, z0; // lower corner of the footprint = floor of the texel coordinate, -1 ... n - 1 (live only between tap_coords and tap_loads)
};

// macrocell (16^3 voxels, reference accel/spatial_partition.h:24) whose value range covers the footprint (i0, i0 + 1) on
// every axis: cell c holds voxels [16c - 1, 16c + 15] (sp_singlemc.cu:36-42), i.e. c = (i0 + 1) >> 4

__device__ __forceinline__ void tap_coords(const VolConsts& vc, f3 p, Tap& t)
{
  axis_tap(p.x, vc.cs.x, vc.cb.x, t.x0, t.fx);
  axis_tap(p.y, vc.cs.y, vc.cb.y, t.y0, t.fy);
  axis_tap(p.z, vc.cs.z, vc.cb.z, t.z0, t.fz);
}

__device__ __forceinline__ unsigned int tap_cell(const VolConsts& vc, const Tap& t)
{
  const int cx = min((t.x0 + 1) >> 4, vc.mcx1), cy = min((t.y0 + 1) >> 4, vc.mcy1), cz = min((t.z0 + 1) >> 4, vc.mcz1);
  return (unsigned int)cx + (unsigned int)(vc.mcx1 + 1) * ((unsigned int)cy + (unsigned int)(vc.mcy1 + 1) * (unsigned int)cz);
}

// The pair (x0, x0 + 1) of a brick row.  The texture addresser merges the lanes of a quad that read one 128-byte line only for loads of 8
// bytes or more: a 64-lane gather whose quads each stay inside one line costs 18 clocks as dwordx2 / dwordx4 and 66 - as if every lane had a
// line of its own - as dword or ushort, whatever the alignment (tools/ubench_align.hip, profiles/r05_notes.md section 10; a dwordx2 at a 4-byte
// boundary - the 32-bit bricks' odd pairs - costs the 18; dwords merge only when the quad's four are consecutive).  The 16-bit layouts' pairs are
// 4 bytes: OVR_ROW_LOADS loads the ALIGNED 8 bytes around the pair (a brick row of the general layout, two rows of a thin replica) and shifts the
// pair out of them - the same voxels, so the same frame.  C4: shade 3.21 -> 2.87 ms, frame 7.5 -> 7.2 ms; front view (thin replica) 3.57 -> 3.29;
// a 512^3 u16 volume at 512^2 0.353 -> 0.335; the 8-bit layouts' 2-byte pairs likewise (a 1024^3 u8 volume at C3's settings 1.80 -> 1.70 ms, front view 1.085 ->
// 0.97) - where the layout is large: C1 (256^3 u8) is bound by vector issue and the shifts cost it 3 % (0.183 -> 0.188 ms), see RowLoads.
#ifndef OVR_ROW_LOADS
#define OVR_ROW_LOADS 1
#endif
// Which launches take the row loads: the 16-bit and 8-bit volumes whose layout is too large for the caches to serve (addressing modes 1 and 2, and mode 4 = mode 0's
// 32-bit byte offsets + row loads: launch_vs upgrades a 16-bit or 8-bit layout of more than 128 MB).  Small 16-bit volumes are bound by vector issue and keep the
// 4-byte loads (a 256 x 256 x 226 u16 volume, all samples shaded in place: 0.88 -> 1.08 ms WITH the row loads; the 1024 x 1024 x 1080 one 34 -> 24 ms).
template <int VT, int AM> struct RowLoads { static constexpr bool on = OVR_ROW_LOADS && sizeof(typename Vox<VT>::T) <= 2 && !Vox<VT>::kQuad && (AM == 1 || AM == 2 || AM == 4); };
template <int VT, int AM, typename B>
__device__ __forceinline__ typename Vox<VT>::P load_pair(const B* base, unsigned long long off) // off in units of B (bytes for char, else elements)
{
  typedef typename Vox<VT>::T T;
  typedef typename Vox<VT>::P P;
  if constexpr (RowLoads<VT, AM>::on) {
    constexpr unsigned per = 8u / (unsigned)sizeof(B);                 // units of B per 8 bytes
    const unsigned long long al = off & ~(unsigned long long)(per - 1u);
    const unsigned sh = ((unsigned)off & (per - 1u)) * (8u * (unsigned)sizeof(B));
    const uint2 row = *reinterpret_cast<const uint2*>(base + al);
    const unsigned w = (unsigned)((((unsigned long long)row.y << 32) | (unsigned long long)row.x) >> sh); // (a pair never crosses its 8 bytes: rows are 8 bytes or 4)
    P r;
    if constexpr (sizeof(T) == 2) { r.x = (T)(w & 0xffffu); r.y = (T)(w >> 16); }
    else { r.x = (T)(w & 0xffu); r.y = (T)((w >> 8) & 0xffu); }
    return r;
  }
  else return *reinterpret_cast<const P*>(base + off);
}

// (A non-temporal hint on these loads - `global_load ... nt` - was measured: the march takes 3.50 instead of 1.51 ms, the shade
// kernel 3.17 instead of 1.13 ms on C3: neighbouring quads and consecutive rounds do re-use the lines; profiles/r02_notes.md §7.)
template <int VT, int AM>
__device__ __forceinline__ void tap_loads(const VolConsts& vc, Tap& t)
{
  typedef BrickMap<VT> M;
  typedef typename Vox<VT>::T T;
  typedef typename Vox<VT>::P P;
  if constexpr (Vox<VT>::kQuad) { // quad replica: the (x, y) footprint of a z slice is ONE load (16 / 8 / 4 bytes)
    typedef typename Vox<VT>::Q Q;
    Q q0, q1;
    if (AM == 3) {
      const unsigned z0 = (unsigned)max(t.z0, 0), z1 = (unsigned)min(t.z0 + 1, vc.nz1);
      const unsigned o = M::X((unsigned)(t.x0 + 1)) + M::Y((unsigned)(t.y0 + 1), vc.macro_y);
      const T* base = static_cast<const T*>(vc.data);
      const unsigned long long oz0 = (unsigned long long)M::Zlo(z0) + (unsigned long long)(z0 >> 5) * vc.macro_z;
      const unsigned long long oz1 = (unsigned long long)M::Zlo(z1) + (unsigned long long)(z1 >> 5) * vc.macro_z;
      q0 = *reinterpret_cast<const Q*>(base + (oz0 + o)); q1 = *reinterpret_cast<const Q*>(base + (oz1 + o));
    }
    else if (AM == 2) {
      const unsigned o = vc.tab_x[t.x0] + vc.tab_y[t.y0];
      const unsigned long long oz0 = vc.tab_z64[t.z0], oz1 = vc.tab_z64[t.z0 + 1];
      const T* base = static_cast<const T*>(vc.data);
      q0 = *reinterpret_cast<const Q*>(base + (oz0 + o)); q1 = *reinterpret_cast<const Q*>(base + (oz1 + o));
    }
    else {
      const unsigned o = vc.tab_x[t.x0] + vc.tab_y[t.y0];
      const unsigned oz0 = vc.tab_z[t.z0], oz1 = vc.tab_z[t.z0 + 1];
      if (AM == 1) {
        const T* base = static_cast<const T*>(vc.data);
        q0 = *reinterpret_cast<const Q*>(base + (oz0 + o)); q1 = *reinterpret_cast<const Q*>(base + (oz1 + o));
      }
      else {
        const char* cb = static_cast<const char*>(vc.data);
        q0 = *reinterpret_cast<const Q*>(cb + (oz0 + o)); q1 = *reinterpret_cast<const Q*>(cb + (oz1 + o));
      }
    }
    t.c000 = (float)q0.x; t.c100 = (float)q0.y; t.c010 = (float)q0.z; t.c110 = (float)q0.w;
    t.c001 = (float)q1.x; t.c101 = (float)q1.y; t.c011 = (float)q1.z; t.c111 = (float)q1.w;
    return;
  }
  // layout coordinates (a, b, c): a = the pair axis (the volume's x; its y in a transposed replica), b = the other of the two
  constexpr bool TR = Vox<VT>::kTransposed;
  const int a0 = TR ? t.y0 : t.x0, b0 = TR ? t.x0 : t.y0, z0 = t.z0;
  P p00, p10, p01, p11;
  if (AM == 3) { // > 2^32 elements and axis tables too large for LDS: 64-bit element offsets, computed arithmetically (explicit clamps)
    const int nb1 = TR ? vc.nx1 : vc.ny1;
    const unsigned bc0 = (unsigned)max(b0, 0), bc1 = (unsigned)min(b0 + 1, nb1), zc0 = (unsigned)max(z0, 0), zc1 = (unsigned)min(z0 + 1, vc.nz1);
    const unsigned ox = M::X((unsigned)(a0 + 1));
    const unsigned o0 = ox + M::Y(bc0, vc.macro_y), o1 = ox + M::Y(bc1, vc.macro_y);
    const T* base = static_cast<const T*>(vc.data);
    const unsigned long long oz0 = (unsigned long long)M::Zlo(zc0) + (unsigned long long)(zc0 >> 5) * vc.macro_z;
    const unsigned long long oz1 = (unsigned long long)M::Zlo(zc1) + (unsigned long long)(zc1 >> 5) * vc.macro_z;
    p00 = *reinterpret_cast<const P*>(base + (oz0 + o0)); p10 = *reinterpret_cast<const P*>(base + (oz0 + o1));
    p01 = *reinterpret_cast<const P*>(base + (oz1 + o0)); p11 = *reinterpret_cast<const P*>(base + (oz1 + o1));
  }
  else if (AM == 2) { // > 2^32 elements: a and b offsets (inside one macro layer) stay 32-bit, the z table is 64-bit
    const unsigned ox = vc.tab_x[a0];
    const unsigned o0 = ox + vc.tab_y[b0], o1 = ox + vc.tab_y[b0 + 1];
    const unsigned long long oz0 = vc.tab_z64[z0], oz1 = vc.tab_z64[z0 + 1];
    const T* base = static_cast<const T*>(vc.data);
    p00 = load_pair<VT, AM>(base, oz0 + o0); p10 = load_pair<VT, AM>(base, oz0 + o1);
    p01 = load_pair<VT, AM>(base, oz1 + o0); p11 = load_pair<VT, AM>(base, oz1 + o1);
  }
  else {
    // three LDS lookups (one b32 + two adjacent pairs) replace ~40 bit-field / multiply instructions per tap: the march is
    // VALU-bound once its gathers coalesce, and the LDS pipe is otherwise nearly idle
    const unsigned ox = vc.tab_x[a0];
    const unsigned oy0 = vc.tab_y[b0], oy1 = vc.tab_y[b0 + 1];
    const unsigned oz0 = vc.tab_z[z0], oz1 = vc.tab_z[z0 + 1];
    const unsigned o0 = ox + oy0, o1 = ox + oy1;
    if (AM == 1) { // < 2^32 elements: 32-bit element offsets, one 64-bit shift-add per load
      const T* base = static_cast<const T*>(vc.data);
      p00 = load_pair<VT, AM>(base, (unsigned long long)(oz0 + o0)); p10 = load_pair<VT, AM>(base, (unsigned long long)(oz0 + o1));
      p01 = load_pair<VT, AM>(base, (unsigned long long)(oz1 + o0)); p11 = load_pair<VT, AM>(base, (unsigned long long)(oz1 + o1));
    }
    else { // the whole volume is <= 4 GiB: 32-bit BYTE offsets, the loads use the SGPR-base + 32-bit-VGPR-offset form
      const char* cb = static_cast<const char*>(vc.data);
      p00 = load_pair<VT, AM>(cb, (unsigned long long)(oz0 + o0)); p10 = load_pair<VT, AM>(cb, (unsigned long long)(oz0 + o1));
      p01 = load_pair<VT, AM>(cb, (unsigned long long)(oz1 + o0)); p11 = load_pair<VT, AM>(cb, (unsigned long long)(oz1 + o1));
    }
  }
#ifdef OVR_EXP_HALF_LOADS /* timing experiment only (wrong pictures): what would half the gather instructions buy? */
  p10 = p00; p11 = p01;
#endif
  // p(b, c) = the pair along a at (b0 + b, z0 + c); the corners keep their volume-axis names, so the lerp order (x, then y,
  // then z) and with it every bit of the result is the same for every layout
  if (!TR) {
    t.c000 = (float)p00.x; t.c100 = (float)p00.y; t.c010 = (float)p10.x; t.c110 = (float)p10.y;
    t.c001 = (float)p01.x; t.c101 = (float)p01.y; t.c011 = (float)p11.x; t.c111 = (float)p11.y;
  }
  else {
    t.c000 = (float)p00.x; t.c010 = (float)p00.y; t.c100 = (float)p10.x; t.c110 = (float)p10.y;
    t.c001 = (float)p01.x; t.c011 = (float)p01.y; t.c101 = (float)p11.x; t.c111 = (float)p11.y;
  }
}

template <int VT, int AM>
__device__ __forceinline__ void tap_issue(const VolConsts& vc, f3 p, Tap& t)
{
  tap_coords(vc, p, t);
  tap_loads<VT, AM>(vc, t);
}

template <int VT>
__device__ __forceinline__ float tap_finish(const VolConsts& vc, Tap t)
{
#if OVR_PARITY_EXACT
  if (Vox<VT>::kScale) { // the texture's normalized read, voxel by voxel: v / 255, max(v / 127, -1) (array.cpp:304-306, array.h:68-106)
    float* c = &t.c000;
    for (int k = 0; k < 8; ++k) c[k] = Vox<VT>::kClamp ? fmaxf(c[k] / 127.f, -1.f) : c[k] / 255.f;
  }
#else
  if (Vox<VT>::kClamp) {
    t.c000 = fmaxf(t.c000, vc.vmin); t.c100 = fmaxf(t.c100, vc.vmin); t.c010 = fmaxf(t.c010, vc.vmin); t.c110 = fmaxf(t.c110, vc.vmin);
    t.c001 = fmaxf(t.c001, vc.vmin); t.c101 = fmaxf(t.c101, vc.vmin); t.c011 = fmaxf(t.c011, vc.vmin); t.c111 = fmaxf(t.c111, vc.vmin);
  }
#endif
  const float c00 = lerpf(t.c000, t.c100, t.fx), c10 = lerpf(t.c010, t.c110, t.fx);
  const float c01 = lerpf(t.c001, t.c101, t.fx), c11 = lerpf(t.c011, t.c111, t.fx);
  const float c0 = lerpf(c00, c10, t.fy), c1 = lerpf(c01, c11, t.fy);
  float s = lerpf(c0, c1, t.fz);
  if (Vox<VT>::kScale && !OVR_PARITY_EXACT) s *= vc.vscale;
  return s;
}

// trilinear tap at object-space p (shaders_common.h:186-193); returns what tex3D<float> returns
template <int VT, int AM>
__device__ __forceinline__ float sample_volume(const VolConsts& vc, f3 p)
{
  Tap t;
  tap_issue<VT, AM>(vc, p, t);
  return tap_finish<VT>(vc, t);
}

// ------------------------------------------------------------------------------------------------------------------
// Quad-cooperative taps (round 5).  The texture addresser works through a gather quad by quad - 4 consecutive lanes - and a quad costs by the
// number of 128-byte lines its lanes touch: 1.1 clocks for one line, 2.8 for two, 4.4 for four (tools/ubench_align.hip, 8-byte loads).  In the
// shade kernel a quad is 4 consecutive steps of one primary ray; when every lane loads the pairs of its OWN tap, one instruction reads pair j of
// four taps that lie 1 voxel apart along the view direction - 2.7 bricks on the oblique view (3 x 4 x 2 cells a brick).  Transposed: instruction i
// reads the FOUR pairs of request i's tap, lane j the pair j = (y0 + (j & 1), z0 + (j >> 1)) - one tap's pairs share a brick unless the tap
// straddles a y or z face: 1.9 lines (1.6 for the 16-bit bricks).  The lerps follow the data: lane j lerps its pair along x, the even lanes along
// y with their neighbour's value, lane 0 along z, and lane i takes the result of request i - the same fma of the same operands as tap_finish
// (lerp(a, b, f) = fma(f, b - a, a), x then y then z), so frames stay bit-identical, with 6 instead of 7 x 2 lerp instructions per tap and lane.
// All 4 lanes of a quad must be active around these calls (the callers enter per quad, with a per-lane `use` flag).
// Not for the transposed and quad replicas (their pair axis / cell differs) and not without LDS tables (addressing mode 3): those keep tap_loads.
// MEASURED (profiles/r05_notes.md section 11): bit-identical (211 parity tests with it on) and SLOWER - C3 shade 0.92 -> 1.08 ms, C4 2.85 -> 3.38, C5 2.80 ->
// 3.53: the transposition is paid in vector instructions.  Per tap and lane the addresses take 12 v_mov_dpp + 12 v_lshl_add + 12 ds_read_b32 (each lane
// looks up the table entries of FOUR taps) where the own tap took 3 + 3 + 6 adds, and the lerps 12 instructions per request (the weights' broadcasts are not
// folded into v_fmac as DPP operands, the final select branches): 2339 instead of 1545 vector instructions in the kernel, and the shade kernel was at 0.58 of
// vector issue before.  What the texture addresser saves the ALUs spend.  Off; a transposition through LDS (ds_write_b64 x 4 / ds_read_b128 x 2 per tap, no
// vector instructions) is the untried alternative - it would put the LDS pipe at ~50-70 clocks per wave and tap beside the addresser's 100-150.
// ------------------------------------------------------------------------------------------------------------------
#ifndef OVR_COOP_TAPS
#define OVR_COOP_TAPS 0
#endif
template <int VT, int AM> struct Coop { static constexpr bool ok = OVR_COOP_TAPS && !Vox<VT>::kQuad && !Vox<VT>::kTransposed && AM != 3; };
template <int B> __device__ __forceinline__ int qb_i(int x) { return __builtin_amdgcn_update_dpp(0, x, B * 0x55, 0xf, 0xf, true); } // lane B of the quad
template <int B> __device__ __forceinline__ float qb_f(float x) { return __int_as_float(qb_i<B>(__float_as_int(x))); }
template <int CTRL> __device__ __forceinline__ float qperm_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true)); }

// lane `sub` of the quad loads pair (sub & 1, sub >> 1) of the tap of the quad's lane I (its corner t.x0 / y0 / z0); use == 0: the tap is not
// wanted - the four lanes read the volume's first bytes instead (one hot line)
template <int VT, int AM, int I>
__device__ __forceinline__ typename Vox<VT>::P coop_load(const VolConsts& vc, const Tap& t, int use, int sub)
{
  typedef typename Vox<VT>::T T;
  const int xi = qb_i<I>(t.x0), yi = qb_i<I>(t.y0) + (sub & 1), zi = qb_i<I>(t.z0) + (sub >> 1);
  const unsigned int m = (unsigned int)qb_i<I>(use);
  const unsigned int oxy = vc.tab_x[xi] + vc.tab_y[yi];
  if (AM == 2) {
    const unsigned long long off = (vc.tab_z64[zi] + oxy) & (((unsigned long long)m << 32) | m);
    return load_pair<VT, AM>(static_cast<const T*>(vc.data), off);
  }
  const unsigned int off = (vc.tab_z[zi] + oxy) & m;
  if (AM == 1) return load_pair<VT, AM>(static_cast<const T*>(vc.data), (unsigned long long)off);
  return load_pair<VT, AM>(static_cast<const char*>(vc.data), (unsigned long long)off);
}
// the value of request I's tap from the four lanes' pairs; returned in every lane of the quad (fx / fy / fz: the OWN tap's weights of each lane)
template <int VT, int I>
__device__ __forceinline__ float coop_lerp(const VolConsts& vc, typename Vox<VT>::P p, float fx, float fy, float fz)
{
  float lo = (float)p.x, hi = (float)p.y;
#if OVR_PARITY_EXACT
  if (Vox<VT>::kScale) { lo = Vox<VT>::kClamp ? fmaxf(lo / 127.f, -1.f) : lo / 255.f; hi = Vox<VT>::kClamp ? fmaxf(hi / 127.f, -1.f) : hi / 255.f; }
#else
  if (Vox<VT>::kClamp) { lo = fmaxf(lo, vc.vmin); hi = fmaxf(hi, vc.vmin); }
#endif
  const float cx = fmaf(qb_f<I>(fx), hi - lo, lo);                       // lane j: c(y_j, z_j) = lerp along x
  const float cy = fmaf(qb_f<I>(fy), qperm_f<0xB1>(cx) - cx, cx);        // lanes 0, 2: lerp(c(y0, z), c(y1, z), fy)   (quad_perm 1,0,3,2)
  float cz = fmaf(qb_f<I>(fz), qperm_f<0x4E>(cy) - cy, cy);              // lane 0: lerp(c(z0), c(z1), fz)               (quad_perm 2,3,0,1)
  if (Vox<VT>::kScale && !OVR_PARITY_EXACT) cz *= vc.vscale;
  return qb_f<0>(cz);
}
// one tap per lane of the quad, all four in flight: issue, then finish (own result per lane; garbage where use == 0)
template <int VT, int AM>
struct CoopTap {
  typename Vox<VT>::P p[4];
  float fx, fy, fz;
  __device__ __forceinline__ void issue(const VolConsts& vc, const Tap& t, int use, int sub)
  {
    fx = t.fx; fy = t.fy; fz = t.fz;
    p[0] = coop_load<VT, AM, 0>(vc, t, use, sub); p[1] = coop_load<VT, AM, 1>(vc, t, use, sub);
    p[2] = coop_load<VT, AM, 2>(vc, t, use, sub); p[3] = coop_load<VT, AM, 3>(vc, t, use, sub);
  }
  __device__ __forceinline__ float finish(const VolConsts& vc, int sub) const
  {
    const float s0 = coop_lerp<VT, 0>(vc, p[0], fx, fy, fz), s1 = coop_lerp<VT, 1>(vc, p[1], fx, fy, fz);
    const float s2 = coop_lerp<VT, 2>(vc, p[2], fx, fy, fz), s3 = coop_lerp<VT, 3>(vc, p[3], fx, fy, fz);
    return sub == 0 ? s0 : sub == 1 ? s1 : sub == 2 ? s2 : s3;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// packed FP32: two taps of ONE lane side by side in 64-bit register pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: each half
// is the IEEE operation of the scalar instruction, so results are bit-identical to the scalar form).  Measured issue cost relative
// to v_fma_f32 (tools/ubench_valu.hip): v_pk_fma_f32 1.3 for two fmas, v_pk_add / v_pk_mul 1.1 for two.  Only what is naturally
// a pair is packed: the loaded voxel pairs are (x, x+1) of one brick row - their x-lerp needs hi - lo of ONE register pair, which a
// packed instruction cannot do without moves (the compiler's own SLP packing of those lerps costs 3 v_mov per 2 lerps: the
// library is built with -fno-slp-vectorize) - so the x-lerps stay scalar and write their results side by side for the packed y / z lerps.
// ------------------------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 splat2(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 lerp2(f2 a, f2 b, f2 f) { return fma2(f, b - a, a); }

// axis_tap for two taps.  The clamp of p to [0, 1] (sample_volume_object_space) is applied to x instead: x = fma(p, cs, cb) is monotone
// in p (cs > 0), p = 0 maps to cb and p = 1 to cs + cb (exact: n - 0.5 or n - 1), so clamping x to [cb, cs + cb] gives the value clamping p
// first gives, for every p (NaN -> cb either way: clamp01(NaN) = 0 -> cb, and med3(NaN, cb, cs + cb) = cb)
// a * sc.x + sc.y on both halves; sc = (scale, offset) lives in one register pair whose halves the instruction broadcasts (op_sel)
__device__ __forceinline__ f2 fma2_sc(f2 a, f2 sc) { return fma2(a, __builtin_shufflevector(sc, sc, 0, 0), __builtin_shufflevector(sc, sc, 1, 1)); }
__device__ __forceinline__ void axis_tap2(f2 po, f2 csb, int& ia, int& ib, float& fa, float& fb)
{
  const f2 x = fma2_sc(po, csb);
  const float hi = csb.x + csb.y;
  const float xa = clampf(x.x, csb.y, hi), xb = clampf(x.y, csb.y, hi);
  fa = __builtin_amdgcn_fractf(xa); ia = floor_to_int(xa);
  fb = __builtin_amdgcn_fractf(xb); ib = floor_to_int(xb);
}
__device__ __forceinline__ void tap_coords2(const VolConsts& vc, f2 px, f2 py, f2 pz, f2 cx, f2 cy, f2 cz, Tap& a, Tap& b)
{
  axis_tap2(px, cx, a.x0, b.x0, a.fx, b.fx);
  axis_tap2(py, cy, a.y0, b.y0, a.fy, b.fy);
  axis_tap2(pz, cz, a.z0, b.z0, a.fz, b.fz);
}
// tap_finish for two taps: 8 scalar x-lerps, then the y and z lerps of both taps packed (6 instructions instead of 12)
template <int VT>
__device__ __forceinline__ f2 tap_finish2(const VolConsts& vc, Tap a, Tap b)
{
#if OVR_PARITY_EXACT
  return mk2(tap_finish<VT>(vc, a), tap_finish<VT>(vc, b));
#endif
  if (Vox<VT>::kClamp) {
    a.c000 = fmaxf(a.c000, vc.vmin); a.c100 = fmaxf(a.c100, vc.vmin); a.c010 = fmaxf(a.c010, vc.vmin); a.c110 = fmaxf(a.c110, vc.vmin);
    a.c001 = fmaxf(a.c001, vc.vmin); a.c101 = fmaxf(a.c101, vc.vmin); a.c011 = fmaxf(a.c011, vc.vmin); a.c111 = fmaxf(a.c111, vc.vmin);
    b.c000 = fmaxf(b.c000, vc.vmin); b.c100 = fmaxf(b.c100, vc.vmin); b.c010 = fmaxf(b.c010, vc.vmin); b.c110 = fmaxf(b.c110, vc.vmin);
    b.c001 = fmaxf(b.c001, vc.vmin); b.c101 = fmaxf(b.c101, vc.vmin); b.c011 = fmaxf(b.c011, vc.vmin); b.c111 = fmaxf(b.c111, vc.vmin);
  }
  const f2 c00 = mk2(lerpf(a.c000, a.c100, a.fx), lerpf(b.c000, b.c100, b.fx)), c10 = mk2(lerpf(a.c010, a.c110, a.fx), lerpf(b.c010, b.c110, b.fx));
  const f2 c01 = mk2(lerpf(a.c001, a.c101, a.fx), lerpf(b.c001, b.c101, b.fx)), c11 = mk2(lerpf(a.c011, a.c111, a.fx), lerpf(b.c011, b.c111, b.fx));
  const f2 fy = mk2(a.fy, b.fy), fz = mk2(a.fz, b.fz);
  const f2 c0 = lerp2(c00, c10, fy), c1 = lerp2(c01, c11, fy);
  f2 s = lerp2(c0, c1, fz);
  if (Vox<VT>::kScale) s = s * splat2(vc.vscale);
  return s;
}

// ------------------------------------------------------------------------------------------------------------------
// transfer function in LDS (or global when it does not fit)
// ------------------------------------------------------------------------------------------------------------------
struct TfConsts {
  const float4* color; // LDS or global
  const float* alpha;
  int nc1, na1;
  float fnc1, fna1;
  float lower, upper, scale;
};

__device__ __forceinline__ float tf_coord(const TfConsts& tf, float sample)
{
  return clamp01((fminf(fmaxf(sample, tf.lower), tf.upper) - tf.lower) * tf.scale); // shaders_common.h:363, :316
}
// (both tables carry one more entry, a copy of their last one - stage_tf - so that (i0, i0 + 1) is the reference's clamp-to-edge pair
// without min(i0 + 1, n - 1): i0 = n - 1 only for v = 1, where the fraction is 0 and lerp(a, a, 0) = a; the two alphas arrive with
// one ds_read2_b32.  Three instructions fewer per lookup, i.e. per shadow sample.)
// PAD = false keeps the two separate reads with min(i0 + 1, n - 1): the POOLED march of the headline (bound by L1 fills, not by instructions)
// is 4 % slower with the paired read - 1.512 -> 1.577 ms on C3, same box, four alternating runs (profiles/r03_ab/r03_ab_regress2.txt) - while every
// instruction-bound kernel gains 1-4 % from it (shadow march at rate 4, the in-place kernels of the shipped scenes, C2, C5)
template <bool PAD = true>
__device__ __forceinline__ float tf_alpha(const TfConsts& tf, float v)
{
  const float x = v * tf.fna1;                      // v in [0, 1]: x >= 0, see axis_tap
  const int i0 = (int)x;
  if (!PAD) return lerpf(tf.alpha[i0], tf.alpha[min(i0 + 1, tf.na1)], __builtin_amdgcn_fractf(x));
  return lerpf(tf.alpha[i0], tf.alpha[i0 + 1], __builtin_amdgcn_fractf(x));
}
__device__ __forceinline__ f3 tf_color(const TfConsts& tf, float v)
{
  const float x = v * tf.fnc1;
  const int i0 = (int)x;
  const float f = __builtin_amdgcn_fractf(x);
  const float4 a = tf.color[i0], b = tf.color[i0 + 1];
  return mk3(lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f));
}

// opacity of two samples: tf_coord + tf_alpha packed.  Two of the scalar form's instructions per sample are dropped without changing a bit:
//  * the clamp of the coordinate to [0, 1]: v = (clamp(s, lower, upper) - lower) * scale is never negative and exceeds 1 by at most a
//    rounding step; such a v indexes the table's last entry with a small fraction, and the entry after the last is a copy of the
//    last (stage_tf), so the lerp returns A[n - 1] exactly as v = 1 does (update_tfn_range keeps scale finite);
//  * min(i0 + 1, n - 1): the same copy makes (i0, i0 + 1) clamp-to-edge, and the two entries arrive with one ds_read2_b32.
__device__ __forceinline__ f2 tf_alpha2(const TfConsts& tf, f2 s)
{
  const f2 sc = mk2(fminf(fmaxf(s.x, tf.lower), tf.upper), fminf(fmaxf(s.y, tf.lower), tf.upper));
  const f2 x = ((sc - splat2(tf.lower)) * splat2(tf.scale)) * splat2(tf.fna1);
  const int ia = (int)x.x, ib = (int)x.y;
  const float a0 = tf.alpha[ia], a1 = tf.alpha[ia + 1], b0 = tf.alpha[ib], b1 = tf.alpha[ib + 1];
  return mk2(lerpf(a0, a1, __builtin_amdgcn_fractf(x.x)), lerpf(b0, b1, __builtin_amdgcn_fractf(x.y)));
}

// ------------------------------------------------------------------------------------------------------------------
// box test vs [0,1]^3, shaders_common.h:156-184 (__frcp_rn restated as an IEEE divide)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool intersect_unit_box(float& t0, float& t1, f3 o, f3 d)
{
  const bool sx = fabsf(d.x) < FLT_MIN, sy = fabsf(d.y) < FLT_MIN, sz = fabsf(d.z) < FLT_MIN;
  const float rx = 1.f / d.x, ry = 1.f / d.y, rz = 1.f / d.z;
  const float lx = sx ? FLT_MAX : (0.f - o.x) * rx, ly = sy ? FLT_MAX : (0.f - o.y) * ry, lz = sz ? FLT_MAX : (0.f - o.z) * rz;
  const float hx = sx ? -FLT_MAX : (1.f - o.x) * rx, hy = sy ? -FLT_MAX : (1.f - o.y) * ry, hz = sz ? -FLT_MAX : (1.f - o.z) * rz;
  t0 = fmaxf(t0, fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fminf(lz, hz)));
  t1 = fminf(t1, fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fmaxf(lz, hz)));
  return t1 > t0;
}
// The reference's quirk, restated above: a direction component below FLT_MIN switches its slab OFF - such a ray "hits" the box wherever
// its origin lies along that axis.  True for a ray that takes this path with its origin outside the ignored slab: the only rays that can
// hit the box from outside its silhouette (the host needs to know: mapframe(HOST) copies the silhouette's rectangle, ovr_hip_api.cpp)
__device__ __forceinline__ bool ignored_slab_outside(f3 o, f3 d)
{
  return (fabsf(d.x) < FLT_MIN && !(o.x >= 0.f && o.x <= 1.f)) || (fabsf(d.y) < FLT_MIN && !(o.y >= 0.f && o.y <= 1.f)) ||
         (fabsf(d.z) < FLT_MIN && !(o.z >= 0.f && o.z <= 1.f));
}

// ------------------------------------------------------------------------------------------------------------------
// TEA, random.h:146-188
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tea16(unsigned int& v0, unsigned int& v1)
{
  unsigned int sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    sum += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
  }
}
#define OVR_TEA_TOFLOAT 2.3283064365386962890625e-10f

// ------------------------------------------------------------------------------------------------------------------
// per-frame constants shared by the primary and the shadow march
// ------------------------------------------------------------------------------------------------------------------
struct MarchConsts {
  f3 inv_scale, wto_p, otw_it, light;
  f3 gstep;     // one voxel in normalized object coordinates
  f3 ginv;      // 1 / gstep
  float step, base, shadow_stride;
};

__device__ __forceinline__ f3 to_object(const MarchConsts& mc, f3 p)
{
  return mk3(fmaf(p.x, mc.inv_scale.x, mc.wto_p.x), fmaf(p.y, mc.inv_scale.y, mc.wto_p.y), fmaf(p.z, mc.inv_scale.z, mc.wto_p.z));
}

// Does the ray of pixel (ix, iy) - one sample per pixel, no jitter - meet the volume's box?  The expressions are the march's own
// (raymarch_kernel: screen position, ray direction, object-space ray, box test), so the answer is the march's answer bit for bit, the
// ignored-slab quirk of the reference's box test included.  launch_schedule uses it to find the 8x8 blocks NONE of whose rays hits.
__device__ __forceinline__ bool pixel_ray_hits_box(const RayMarchParams& P, const MarchConsts& mc, int ix, int iy)
{
  const float rsx = 1.f / (float)P.width, rsy = 1.f / (float)P.height;
  const float sx = ((float)ix + .5f) * rsx, sy = ((float)iy + .5f) * rsy;
  const float ux = sx - 0.5f, uy = sy - 0.5f;
  const f3 cdir = ld3(P.cam_dir), chor = ld3(P.cam_hor), cver = ld3(P.cam_ver);
  const f3 dir = normalize3_exact(mk3(cdir.x + ux * chor.x + uy * cver.x, cdir.y + ux * chor.y + uy * cver.y, cdir.z + ux * chor.z + uy * cver.z));
  const f3 od = mk3(dir.x * mc.inv_scale.x, dir.y * mc.inv_scale.y, dir.z * mc.inv_scale.z);
  const f3 oo = to_object(mc, ld3(P.cam_pos));
  float t0 = 0.f, t1 = FLT_MAX;
  return intersect_unit_box(t0, t1, oo, od);
}

// Empty-space skipping, per ray: the t interval outside of which every sample lies in a macrocell with majorant 0.
// skip_walk: one lane walks the ray's [ta, tb] through an occupancy grid (3-D DDA; accel/dda.h is the reference's walker for
// its path tracer) and widens [first, last] by the entry / exit of every set entry it crosses.  FINE = false: the coarse grid
// (4^3 macrocells = 64 voxels per entry), FINE = true: one entry per macrocell (16 voxels); both are dilated by one macrocell.
// Sample coordinates: x = p * cs + cb (tap_coords), macrocell = (floor(x) + 1) >> 4 (tap_cell), i.e. the regular 16-voxel
// grid in w = x + 1.
// Round 3 - any number of occupied intervals per ray: besides the hull [first, last] the fine walk of a primary ray marks, in a 64-bit mask,
// which of 64 equal segments of the span [seg0, seg0 + 64 / seg_scale] it walked (the coarse hull) meet a set entry (one segment of
// margin on either side).  A round of the march whose t range touches no marked segment lies in empty macrocells although it is inside the
// hull - interior voids take the 45-instruction bulk path too (accel/dda.h walks such gaps cell by cell for the reference's path tracer).
struct SkipSpan {
  float first, last;            // hull: every sample outside it lies in a macrocell with majorant 0
  float seg0, seg_scale;        // segment i covers t in [seg0 + i / seg_scale, seg0 + (i + 1) / seg_scale)
  unsigned int mask_lo, mask_hi; // segments that can hold a sample with majorant > 0 (all set: no refinement)
};
__device__ __forceinline__ SkipSpan skipspan_all() { SkipSpan s; s.first = -FLT_MAX; s.last = FLT_MAX; s.seg0 = 0.f; s.seg_scale = 0.f; s.mask_lo = s.mask_hi = ~0u; return s; }
__device__ __forceinline__ SkipSpan skipspan_none() { SkipSpan s; s.first = FLT_MAX; s.last = -FLT_MAX; s.seg0 = 0.f; s.seg_scale = 0.f; s.mask_lo = s.mask_hi = 0u; return s; }
__device__ __forceinline__ int skipspan_seg(const SkipSpan& s, float t) { return min(max((int)((t - s.seg0) * s.seg_scale), 0), 63); }
// can a sample at t lie in a macrocell with majorant > 0?
__device__ __forceinline__ bool skipspan_inside(const SkipSpan& s, float t)
{
  const int g = skipspan_seg(s, t);
  const unsigned int w = g < 32 ? s.mask_lo : s.mask_hi;
  return t >= s.first && t <= s.last && ((w >> (g & 31)) & 1u) != 0u;
}
// may the t range [ta, tb] (ta <= tb) hold such a sample?
__device__ __forceinline__ bool skipspan_touches(const SkipSpan& s, float ta, float tb)
{
  if (tb < s.first || ta > s.last) return false;
  const int lo = skipspan_seg(s, ta), hi = skipspan_seg(s, tb);
  const unsigned long long m = ((unsigned long long)s.mask_hi << 32) | s.mask_lo;
  const unsigned long long range = (hi >= 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
  return (m & range) != 0ull;
}

template <bool FINE, bool MARK = false>
__device__ __forceinline__ void skip_walk(const VolConsts& vc, f3 oo, f3 od, float ta, float tb, float& first, float& last, float seg0 = 0.f, float seg_scale = 0.f,
                                          unsigned long long* mask = nullptr)
{
  const float w0[3] = { fmaf(oo.x, vc.cs.x, vc.cb.x + 1.f), fmaf(oo.y, vc.cs.y, vc.cb.y + 1.f), fmaf(oo.z, vc.cs.z, vc.cb.z + 1.f) };
  const float dw[3] = { od.x * vc.cs.x, od.y * vc.cs.y, od.z * vc.cs.z };
  const int m1[3] = { FINE ? vc.mcx1 : vc.mcx1 >> 2, FINE ? vc.mcy1 : vc.mcy1 >> 2, FINE ? vc.mcz1 : vc.mcz1 >> 2 }; // grid dims - 1
  const unsigned char* grid = FINE ? vc.occupancy_fine : vc.occupancy;
  constexpr float G = FINE ? 16.f : 64.f;
  int ci[3], st[3];
  float tmax[3], tdel[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float w = fmaf(ta, dw[k], w0[k]);
    ci[k] = min(max((int)floorf(w * (1.f / G)), 0), m1[k]);
    st[k] = dw[k] > 0.f ? 1 : -1;
    const bool moves = fabsf(dw[k]) > 1e-20f;
    tdel[k] = moves ? G / fabsf(dw[k]) : FLT_MAX;
    tmax[k] = moves ? ((float)(ci[k] + (dw[k] > 0.f ? 1 : 0)) * G - w0[k]) / dw[k] : FLT_MAX;
  }
  float t = ta;
  const int limit = m1[0] + m1[1] + m1[2] + 8; // a ray crosses at most this many entries: every lane leaves the loop
  for (int it = 0; it < limit && t < tb; ++it) {
    const bool occ = grid[(size_t)ci[0] + (size_t)(m1[0] + 1) * ((size_t)ci[1] + (size_t)(m1[1] + 1) * (size_t)ci[2])] != 0;
    const int ax = (tmax[0] <= tmax[1]) ? (tmax[0] <= tmax[2] ? 0 : 2) : (tmax[1] <= tmax[2] ? 1 : 2);
    const float tn = ax == 0 ? tmax[0] : ax == 1 ? tmax[1] : tmax[2];
    if (occ) {
      first = fminf(first, t); last = fmaxf(last, fminf(tn, tb));
      if (MARK) {
        const int lo = max((int)((t - seg0) * seg_scale) - 1, 0), hi = min((int)((fminf(tn, tb) - seg0) * seg_scale) + 1, 63);
        if (hi >= lo) *mask |= (hi >= 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
      }
    }
    t = fmaxf(t, tn);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k == ax) {
        const int nxt = ci[k] + st[k];
        if (nxt < 0 || nxt > m1[k]) tmax[k] = FLT_MAX; // the clamped coordinate stays in the border entry
        else { ci[k] = nxt; tmax[k] += tdel[k]; }
      }
  }
  if (t < tb) { first = fminf(first, t); last = tb; if (MARK) *mask = ~0ull; } // safety limit hit (never expected): treat the rest as occupied
}

// raymarching_shadow, shaders_raymarching.cu:44-85 (+ :205-229): alpha-only march toward the light.
// KS taps are issued before the first one is consumed; taps past the end of the march or past the early-termination
// point are speculative (their coordinates are clamped, so the loads are always in bounds) and simply dropped.
template <int VT, int AM, int KS, bool SKIP>
__device__ __forceinline__ float march_shadow(const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, f3 org, unsigned int& n_shadow,
                                              unsigned int& n_shadow_skipped)
{
  const f3 oo = to_object(mc, org);
  const f3 od = mk3(mc.light.x * mc.inv_scale.x, mc.light.y * mc.inv_scale.y, mc.light.z * mc.inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  float alpha = 0.f;
  if (!intersect_unit_box(t0, t1, oo, od)) return alpha;
  float tx = t0, ty = fminf(t1, t0 + mc.shadow_stride);
  bool live = true;
  // empty-space skipping: the shadow ray's own skip interval (a handful of coarse entries: the ray is a few hundred voxels)
  float skip_first = FLT_MAX, skip_last = -FLT_MAX;
  if (SKIP) skip_walk<false>(vc, oo, od, t0, t1, skip_first, skip_last);
  while (live) {
    Tap taps[KS];
    float dts[KS], mj[KS];
    bool valid[KS];
    bool any_inside = false;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      valid[k] = ty > tx;
      dts[k] = ty - tx;
      const float tm = 0.5f * (tx + ty);
      const bool inside = !SKIP || (tm >= skip_first && tm <= skip_last);
      if (SKIP) taps[k] = Tap{};
      mj[k] = SKIP ? 0.f : 1.f;
      if (inside) {
        const f3 pos = mk3(fmaf(tm, mc.light.x, org.x), fmaf(tm, mc.light.y, org.y), fmaf(tm, mc.light.z, org.z));
        tap_coords(vc, to_object(mc, pos), taps[k]);
        if (SKIP) mj[k] = vc.majorant[tap_cell(vc, taps[k])]; // empty-space skipping: max TF opacity of the macrocell
      }
      any_inside = any_inside || (mj[k] > 0.f);
      tx = ty;
      ty = fminf(tx + mc.shadow_stride, t1);
    }
    if (SKIP && __ballot(any_inside) == 0ull) { // nothing to fetch for any lane of the wave: bookkeeping only
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        live = live && valid[k] && (alpha < 0.9999f);
        n_shadow_skipped += live ? 1u : 0u;
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < KS; ++k)
      if (!SKIP || mj[k] > 0.f) tap_loads<VT, AM>(vc, taps[k]);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      // branch-free on purpose: a conditional use would let the compiler sink this tap's loads into the branch and
      // serialise the taps again (seen in the ISA); dead lanes just compute a value that is not used
      const float s = tap_finish<VT>(vc, taps[k]);
      float a = tf_alpha(tf, tf_coord(tf, s));
#ifndef OVR_SHADOW_BF
#define OVR_SHADOW_BF 0 /* branch-free opacity correction in the plain shadow march too (measurement switch) */
#endif
      a = opacity_correction<SKIP || OVR_SHADOW_BF>(a, mc.base * dts[k]);
      if (SKIP) a = mj[k] > 0.f ? a : 0.f; // a macrocell whose majorant is 0 holds no sample with opacity > 0
      live = live && valid[k] && (alpha < 0.9999f);
      alpha = live ? fmaf(1.f - alpha, a, alpha) : alpha;
      n_shadow += (live && (!SKIP || mj[k] > 0.f)) ? 1u : 0u;
      if (SKIP) n_shadow_skipped += (live && !(mj[k] > 0.f)) ? 1u : 0u;
    }
  }
  return alpha;
}

// The same march with quad-cooperative taps (CoopTap): entered by all 4 lanes of a quad with a request to shade in any of them (`want`: this
// lane's own), left when none of the quad's rays is live any more; a lane whose ray has ended keeps loading its share of the others' taps.
template <int VT, int AM, int KS, bool SKIP>
__device__ __forceinline__ float march_shadow_coop(const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, f3 org, bool want, unsigned int& n_shadow,
                                                   unsigned int& n_shadow_skipped)
{
  const int sub = (int)(threadIdx.x & 3u);
  const unsigned int qshift = threadIdx.x & 60u; // first lane of the quad within the wave
  const f3 oo = to_object(mc, org);
  const f3 od = mk3(mc.light.x * mc.inv_scale.x, mc.light.y * mc.inv_scale.y, mc.light.z * mc.inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  float alpha = 0.f;
  bool live = intersect_unit_box(t0, t1, oo, od) && want;
  float tx = t0, ty = fminf(t1, t0 + mc.shadow_stride);
  float skip_first = FLT_MAX, skip_last = -FLT_MAX;
  if (SKIP && live) skip_walk<false>(vc, oo, od, t0, t1, skip_first, skip_last);
  for (;;) {
    if (((unsigned int)(__ballot(live) >> qshift) & 0xfu) == 0u) break; // no live ray in this quad
    Tap taps[KS];
    float dts[KS], mj[KS];
    bool valid[KS];
    bool any_inside = false;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      valid[k] = ty > tx;
      dts[k] = ty - tx;
      const float tm = 0.5f * (tx + ty);
      const bool inside = live && (!SKIP || (tm >= skip_first && tm <= skip_last));
      taps[k] = Tap{};
      mj[k] = SKIP ? 0.f : 1.f;
      if (inside || !SKIP) {
        const f3 pos = mk3(fmaf(tm, mc.light.x, org.x), fmaf(tm, mc.light.y, org.y), fmaf(tm, mc.light.z, org.z));
        tap_coords(vc, to_object(mc, pos), taps[k]);
        if (SKIP) mj[k] = vc.majorant[tap_cell(vc, taps[k])]; // empty-space skipping: max TF opacity of the macrocell
      }
      any_inside = any_inside || (live && mj[k] > 0.f);
      tx = ty;
      ty = fminf(tx + mc.shadow_stride, t1);
    }
    if (SKIP && __ballot(any_inside) == 0ull) { // nothing to fetch for any lane of the wave: bookkeeping only
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        live = live && valid[k] && (alpha < 0.9999f);
        n_shadow_skipped += live ? 1u : 0u;
      }
      continue;
    }
    CoopTap<VT, AM> ct[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) ct[k].issue(vc, taps[k], (live && mj[k] > 0.f) ? -1 : 0, sub);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float s = ct[k].finish(vc, sub);
      float a = tf_alpha(tf, tf_coord(tf, s));
      a = opacity_correction<true>(a, mc.base * dts[k]);
      if (SKIP) a = mj[k] > 0.f ? a : 0.f; // a macrocell whose majorant is 0 holds no sample with opacity > 0
      live = live && valid[k] && (alpha < 0.9999f);
      alpha = live ? fmaf(1.f - alpha, a, alpha) : alpha;
      n_shadow += (live && (!SKIP || mj[k] > 0.f)) ? 1u : 0u;
      if (SKIP) n_shadow_skipped += (live && !(mj[k] > 0.f)) ? 1u : 0u;
    }
  }
  return alpha;
}

// The same march with its taps in pairs (packed FP32, see f2 above): the all-shaded frames - most of the reference's shipped scenes, and
// every frame at the scene files' sampling rate 4 - are bound by this loop's instruction stream (profiles/r03_notes.md: VALU ~100 % busy).
// Per pair of steps: positions, object coordinates and cell coordinates as 9 v_pk_fma_f32 (18 v_fma_f32 in the scalar form), the y / z lerps,
// the transfer-function coordinate and the opacity correction's multiplies packed; the redundant clamps dropped (axis_tap2, tf_alpha2).
// Every value is computed by the same IEEE operations in the same order as in march_shadow: bit-identical results.
#ifndef OVR_SHADOW_PACKED
#define OVR_SHADOW_PACKED 0
#endif
#ifndef OVR_PACKED_LERP
#define OVR_PACKED_LERP 0
#endif
template <int VT, int AM, int KS>
__device__ __forceinline__ float march_shadow_packed(const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, f3 org, unsigned int& n_shadow)
{
  static_assert(KS % 2 == 0, "taps come in pairs");
  constexpr int KP = KS / 2;
  const f3 oo = to_object(mc, org);
  const f3 od = mk3(mc.light.x * mc.inv_scale.x, mc.light.y * mc.inv_scale.y, mc.light.z * mc.inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  float alpha = 0.f;
  if (!intersect_unit_box(t0, t1, oo, od)) return alpha;
  float tx = t0, ty = fminf(t1, t0 + mc.shadow_stride);
  bool live = true;
  const f2 lx = splat2(mc.light.x), ly = splat2(mc.light.y), lz = splat2(mc.light.z);
  const f2 ox = splat2(org.x), oy = splat2(org.y), oz = splat2(org.z);
  // (scale, offset) of the two affine maps per axis, each pair in ONE 64-bit scalar register: a packed fma may read one scalar operand
  const f2 wx = mk2(mc.inv_scale.x, mc.wto_p.x), wy = mk2(mc.inv_scale.y, mc.wto_p.y), wz = mk2(mc.inv_scale.z, mc.wto_p.z);
  const f2 cx = mk2(vc.cs.x, vc.cb.x), cy = mk2(vc.cs.y, vc.cb.y), cz = mk2(vc.cs.z, vc.cb.z);
  while (live) {
    Tap taps[KS];
    f2 dts[KP];
    bool valid[KS];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      const float txa = tx, tya = ty;
      tx = ty; ty = fminf(tx + mc.shadow_stride, t1);
      const float txb = tx, tyb = ty;
      tx = ty; ty = fminf(tx + mc.shadow_stride, t1);
      valid[2 * p] = tya > txa; valid[2 * p + 1] = tyb > txb;
      const f2 vtx = mk2(txa, txb), vty = mk2(tya, tyb);
      dts[p] = vty - vtx;
      const f2 tm = (vtx + vty) * splat2(0.5f);
      // pos = org + tm * light; to_object (its clamp to [0, 1] is subsumed, axis_tap2); cell coordinates
      const f2 px = fma2(tm, lx, ox), py = fma2(tm, ly, oy), pz = fma2(tm, lz, oz);
      const f2 qx = fma2_sc(px, wx), qy = fma2_sc(py, wy), qz = fma2_sc(pz, wz);
      tap_coords2(vc, qx, qy, qz, cx, cy, cz, taps[2 * p], taps[2 * p + 1]);
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) tap_loads<VT, AM>(vc, taps[k]);
#pragma unroll
    for (int p = 0; p < KP; ++p) {
#if OVR_PACKED_LERP
      const f2 s = tap_finish2<VT>(vc, taps[2 * p], taps[2 * p + 1]);
#else
      // the 7 lerps of a tap stay scalar: v_sub + v_fmac cost 1.4 v_fma_f32 issue slots, the packed pair 2.45 for two - and the pairs would
      // have to be assembled with moves first (tools/ubench_valu.hip, profiles/r03_notes.md)
      const f2 s = mk2(tap_finish<VT>(vc, taps[2 * p]), tap_finish<VT>(vc, taps[2 * p + 1]));
#endif
      const f2 a = tf_alpha2(tf, s);
      // opacity correction (shaders_raymarching.cu:118-122), branch-free: both transcendentals always run (the shadow stride is never
      // 1 / base in practice), the select keeps a exactly where the reference's branch does
      const f2 adj = splat2(mc.base) * dts[p];
      const f2 lg = mk2(__builtin_amdgcn_logf(1.f - a.x), __builtin_amdgcn_logf(1.f - a.y));
      const f2 pl = adj * lg;
      const float ca = clamp01(1.f - __builtin_amdgcn_exp2f(pl.x)), cb = clamp01(1.f - __builtin_amdgcn_exp2f(pl.y));
      const float aa = (fabsf(adj.x - 1.f) < 1e-7f) ? a.x : ca, ab = (fabsf(adj.y - 1.f) < 1e-7f) ? a.y : cb;
      live = live && valid[2 * p] && (alpha < 0.9999f);
      alpha = live ? fmaf(1.f - alpha, aa, alpha) : alpha;
      n_shadow += live ? 1u : 0u;
      live = live && valid[2 * p + 1] && (alpha < 0.9999f);
      alpha = live ? fmaf(1.f - alpha, ab, alpha) : alpha;
      n_shadow += live ? 1u : 0u;
    }
  }
  return alpha;
}

// ------------------------------------------------------------------------------------------------------------------
// the ray-march kernels
//
// Work decomposition: four lanes per ray (a quad = 4 consecutive steps), 16 rays = a 4x4 pixel tile per wave64, four waves
// = 8x8 pixels per workgroup; workgroups are launched longest rays first (launch_schedule).  Details at raymarch_kernel.
//  * primary march: K instructions x 4 steps per round, all their voxel loads in flight before the first is consumed.
//  * deferred, compacted shading (SHADE != 0): a sample whose opacity is > 0 is not shaded by its own lane; the lane
//    pushes a 32-byte request into its wave's queue in LDS (slot = tail + prefix-of-ballot, v_mbcnt) and keeps
//    marching - alpha does not depend on shading, so early termination is unaffected.  Requests are shaded 64 at a
//    time, one request per lane (gradient taps, normals, shadow march toward the light - the expensive, otherwise
//    badly divergent part); each owner applies its colour contributions in sample order (the requests of one lane
//    form a linked list), so the result is bit-identical to shading in place.
//  * two ways to shade a batch:
//      in place   (raymarch_kernel)  the wave that owns the tile shades its own batches and gets the contributions
//                                    back with ds_bpermute.  Used when shading is off and as the reference pipeline.
//      pooled     (raymarch_kernel<POOLED> -> shade_pool_kernel -> composite_kernel)  the tile's wave spills each full
//                                    batch as a 2 KiB chunk into a global pool; a second, persistent kernel shades
//                                    chunks from ALL tiles with perfect load balance (the shadow work of a frame sits in
//                                    a few hundred tiles: in place, their waves ran alone for 10 ms of a 13 ms kernel);
//                                    a third kernel walks each tile's chunks in order and composites.
//  * counters: per-workgroup partial sums, reduced by a tiny kernel (no same-address atomics).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

#ifndef OVR_SHADOW_K
#define OVR_SHADOW_K 4
#endif
constexpr int kShadowTaps = OVR_SHADOW_K; // shadow-march taps in flight per lane
// chunks a tile reserves at a time: consecutive chunks of one tile are shaded by ONE workgroup, one chunk per wave
// (L1/L2 reuse: measured 2.4 -> 1.5 ms for the shading kernel at C3; 8 / 16 / 32 are slower - imbalance)
constexpr int kRun = 4;
#ifndef OVR_RUN_MAX
#define OVR_RUN_MAX 16
#endif
#ifndef OVR_TICKET_RUNS
#define OVR_TICKET_RUNS 8
#endif
constexpr int kTicketRuns = OVR_TICKET_RUNS; // shade kernel: most runs a workgroup takes per ticket
#ifndef OVR_TICKET_BLOCK
#define OVR_TICKET_BLOCK 8
#endif
constexpr unsigned int kTicketBlock = OVR_TICKET_BLOCK; // shade kernel: consecutive tickets that stay in one sub-pool
constexpr int kRunMax = OVR_RUN_MAX; // largest reservation (a multiple of kRun: the shade kernel takes kRun chunks per workgroup)

struct ShadeReq { // 32 bytes; after shading the same slot holds the result (cx,cy,cz,gx,gy,gz,a,next)
  float px, py, pz; // world-space sample position          | colour contribution  tr*clamp01(rgb*shade)
  float s;          // sample value                           | gradient contribution tr*clamp01(n_c) .x
  float v;          // transfer-function coordinate           | .y
  float tr;         // transmittance before the sample        | .z
  float a;          // corrected opacity
  int next;         // stream position of the owner's next request (valid once that request exists)
};

__device__ __forceinline__ float bperm(int src_lane, float x)
{
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(x)));
}
__device__ __forceinline__ int bperm_i(int src_lane, int x) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, x); }

// shade order by light beams (PoolDesc): the beam a world-space position projects into, permuted so that the beams of one XCD list are
// contiguous: list = (cu + 3 cv) % 8 - neighbours in both directions go to different lists -, index = list * G*G/8 + cv * G/8 + cu / 8
__device__ __forceinline__ unsigned int beam_index(const PoolDesc& Q, float px, float py, float pz)
{
  const int G = Q.order_grid;
  const int cu = min(max((int)fmaf(px, Q.order_u[0], fmaf(py, Q.order_u[1], fmaf(pz, Q.order_u[2], Q.order_u[3]))), 0), G - 1); // (NaN -> 0)
  const int cv = min(max((int)fmaf(px, Q.order_v[0], fmaf(py, Q.order_v[1], fmaf(pz, Q.order_v[2], Q.order_v[3]))), 0), G - 1);
  return ((unsigned int)(cu + 3 * cv) & 7u) * (unsigned int)(G * G / 8) + (unsigned int)(cv * (G / 8) + (cu >> 3));
}

__device__ __forceinline__ void setup_consts(const RayMarchParams& P, VolConsts& vc, MarchConsts& mc)
{
  vc.data = P.vol.data;
  vc.nx1 = P.vol.nx - 1; vc.ny1 = P.vol.ny - 1; vc.nz1 = P.vol.nz - 1;
  vc.macro_y = P.vol.macro_elems * (unsigned int)P.vol.macros_x;
  vc.macro_z = (unsigned long long)P.vol.macro_elems * (unsigned long long)P.vol.macros_x * (unsigned long long)P.vol.macros_y;
  vc.cs = ld3(P.coord_scale); vc.cb = ld3(P.coord_bias);
  vc.vscale = P.vol.value_scale; vc.vmin = P.vol.value_min_clamp;
  vc.majorant = P.majorant;
  vc.occupancy = P.occupancy;
  vc.occupancy_fine = P.occupancy_fine;
  vc.mcx1 = (P.vol.nx + 15) / 16 - 1; vc.mcy1 = (P.vol.ny + 15) / 16 - 1; vc.mcz1 = (P.vol.nz + 15) / 16 - 1;
  mc.inv_scale = ld3(P.inv_scale); mc.wto_p = ld3(P.wto_p); mc.otw_it = ld3(P.otw_it); mc.light = ld3(P.light);
  mc.gstep = ld3(P.grad_step);
  mc.ginv = mk3(1.f / mc.gstep.x, 1.f / mc.gstep.y, 1.f / mc.gstep.z);
  mc.step = P.step; mc.base = P.base; mc.shadow_stride = P.shadow_stride;
}

// copy the layout's per-axis offset tables (VolumeDesc::axis_ab / axis_z, built once per volume by axis_tables_kernel) into LDS,
// scaled to what the addressing mode adds to its base (all threads of the workgroup); returns the bytes used
template <int VT, int AM>
__device__ __forceinline__ size_t stage_tables(const RayMarchParams& P, unsigned char* base, VolConsts& vc)
{
  vc.tab_x = vc.tab_y = vc.tab_z = nullptr;
  vc.tab_z64 = nullptr;
  if (AM == 3) return 0;
  // layout axes: a = pair axis, b = the other one (x and y, exchanged in a transposed replica)
  const int na = Vox<VT>::kTransposed ? P.vol.ny : P.vol.nx, nb = Vox<VT>::kTransposed ? P.vol.nx : P.vol.ny;
  const unsigned int* __restrict__ gab = P.vol.axis_ab;
  const unsigned long long* __restrict__ gz = P.vol.axis_z;
  const int ea = axis_a_entries(na), nab = ea + axis_b_entries(nb), ez = axis_z_entries(P.vol.nz);
  // the table pointers handed to the taps are biased by one entry: index -1 is entry 0 (see VolConsts)
  if (AM == 2) { // [z: ez x u64][a: ea x u32][b: eb x u32], element offsets: the global tables as they are
    unsigned long long* tz = reinterpret_cast<unsigned long long*>(base);
    unsigned int* tx = reinterpret_cast<unsigned int*>(tz + ez);
    for (int i = threadIdx.x; i < nab; i += kBlock) tx[i] = gab[i];
    for (int i = threadIdx.x; i < ez; i += kBlock) tz[i] = gz[i];
    vc.tab_x = tx + 1; vc.tab_y = tx + ea + 1; vc.tab_z64 = tz + 1;
    return (size_t)ez * sizeof(unsigned long long) + (size_t)nab * sizeof(unsigned int);
  }
  // AM 0: byte offsets (the volume is <= 4 GiB), AM 1: element offsets (< 2^32 stored voxels) - 32 bits hold every entry
  unsigned int* tx = reinterpret_cast<unsigned int*>(base);
  unsigned int* tz = tx + nab;
  const unsigned int mul = (AM == 0 || AM == 4) ? (unsigned int)sizeof(typename Vox<VT>::T) : 1u;
  for (int i = threadIdx.x; i < nab; i += kBlock) tx[i] = gab[i] * mul;
  for (int i = threadIdx.x; i < ez; i += kBlock) tz[i] = (unsigned int)gz[i] * mul;
  vc.tab_x = tx + 1; vc.tab_y = tx + ea + 1; vc.tab_z = tz + 1;
  return (size_t)(nab + ez) * sizeof(unsigned int);
}
// Addressing mode of a volume layout: 0 = 32-bit byte offsets (<= 4 GiB), 1 = 32-bit element offsets (< 2^32 stored voxels), 2 = 64-bit z
// table, 3 = computed 64-bit offsets, no tables.  Modes 0-2 keep the per-axis tables in LDS next to the transfer function and the request
// queues (32 KiB in the in-place march): a volume with one very long axis - small in bytes, tens of thousands of voxels long - whose
// tables do not fit in the 160 KiB of a CU takes mode 3 whatever its size (before round 3 only mode 2 fell back, and such a volume was
// accepted by ovr_hip_set_volume and failed at its first launch).
__host__ inline int addressing_mode(const VolumeDesc& vd, int n_color, int n_alpha)
{
  int am = vd.bytes <= 0x100000000ull ? 0 : (vd.bytes / voxel_size(vd.type) < 0xffffffffull) ? 1 : 2;
  const size_t ab = (size_t)(vd.nx + vd.ny + 3), ez = (size_t)axis_z_entries(vd.nz); // a: n + 1 entries, b: n + 2 (whichever of x / y is the pair axis)
  const size_t tables = am == 2 ? ez * sizeof(unsigned long long) + ab * sizeof(unsigned int) : (ab + ez) * sizeof(unsigned int);
  const size_t fixed = raymarch_lds_bytes(n_color, n_alpha) + (size_t)kWaves * 256 * 32 + 1024; // TF + the largest request queues + slack
  if ((am == 2 && tables > 64 * 1024) || tables + 16 + fixed > 160 * 1024) am = 3;
  return am;
}
__host__ inline size_t table_lds_bytes(const RayMarchParams& p, int am)
{
  if (am == 3) return 0;
  const size_t ab = (size_t)(p.vol.nx + p.vol.ny + 3), ez = (size_t)axis_z_entries(p.vol.nz);
  if (am == 2) return (ez * sizeof(unsigned long long) + ab * sizeof(unsigned int) + 15) & ~(size_t)15;
  return ((ab + ez) * sizeof(unsigned int) + 15) & ~(size_t)15;
}

// stage the transfer function in LDS (all threads of the workgroup); color may be skipped by alpha-only kernels
__device__ __forceinline__ void stage_tf(const RayMarchParams& P, unsigned char* tf_base, bool with_color, TfConsts& tf)
{
  float4* lc = reinterpret_cast<float4*>(tf_base);
  float* la = reinterpret_cast<float*>(tf_base + (with_color ? (size_t)(P.n_color + 1) * sizeof(float4) : 0));
  if (with_color) {
    const float4* gc = reinterpret_cast<const float4*>(P.tf_color);
    for (int i = threadIdx.x; i <= P.n_color; i += kBlock) lc[i] = gc[min(i, P.n_color - 1)]; // one more entry, a copy of the last (tf_color)
  }
  for (int i = threadIdx.x; i < P.n_alpha; i += kBlock) la[i] = P.tf_alpha[i];
  if (threadIdx.x == 0) la[P.n_alpha] = P.tf_alpha[P.n_alpha - 1]; // one more entry, a copy of the last: (i, i + 1) is clamp-to-edge (tf_alpha)
  __syncthreads();
  tf.color = lc;
  tf.alpha = la;
  tf.nc1 = P.n_color - 1; tf.na1 = P.n_alpha - 1;
  tf.fnc1 = (float)tf.nc1; tf.fna1 = (float)tf.na1;
  tf.lower = P.tf_lower; tf.upper = P.tf_upper; tf.scale = P.tf_scale;
}

// accumulation + framebuffer write, shaders_raymarching.cu:389-409
__device__ __forceinline__ void write_pixel(const RayMarchParams& P, unsigned int pixel_index, f3 o_c, float o_a, f3 o_g)
{
  float4 out = make_float4(o_c.x, o_c.y, o_c.z, o_a);
  float4* fb = reinterpret_cast<float4*>(P.rgba) + pixel_index;
  if (P.accumulate) {
    float4* ac = reinterpret_cast<float4*>(P.accum) + pixel_index;
    if (P.frame_index == 1) {
      *ac = out;
    }
    else {
      float4 acc = *ac;
      acc.x += out.x; acc.y += out.y; acc.z += out.z; acc.w += out.w;
      *ac = acc;
      const float fi = (float)P.frame_index;
      out = make_float4(acc.x / fi, acc.y / fi, acc.z / fi, acc.w / fi);
    }
  }
  *fb = out;
  if (P.grad) {
    float* pg = P.grad + 3ull * pixel_index;
    pg[0] = o_g.x; pg[1] = o_g.y; pg[2] = o_g.z;
  }
}

// shade one request: gradient (shaders_common.h:195-215), normals, shadow march, Lambert-ish term
// (shaders_raymarching.cu:124-158).  Writes the result over the request.
template <int VT, int SHADE, int AM, bool SKIP>
__device__ __forceinline__ void shade_request(const RayMarchParams& P, const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, ShadeReq& r, bool want,
                                              unsigned int& n_shadow, unsigned int& n_shadow_skipped)
{
  // (entered by whole quads: `want` says whether this lane's request is a real one - the others only help with the quad-cooperative taps and
  // compute on whatever their slot holds; their result is not stored)
  constexpr bool COOP = Coop<VT, AM>::ok;
  if (!COOP && !want) return;
  const f3 pos = mk3(r.px, r.py, r.pz);
  const f3 po = to_object(mc, pos);
  // one-sided differences, flipped at the upper bound; the three taps are issued together
  const bool flx = (po.x + mc.gstep.x) > 1.f, fly = (po.y + mc.gstep.y) > 1.f, flz = (po.z + mc.gstep.z) > 1.f;
  const f3 pgx = mk3(po.x + (flx ? -mc.gstep.x : mc.gstep.x), po.y, po.z), pgy = mk3(po.x, po.y + (fly ? -mc.gstep.y : mc.gstep.y), po.z);
  const f3 pgz = mk3(po.x, po.y, po.z + (flz ? -mc.gstep.z : mc.gstep.z));
  float sgx, sgy, sgz;
  f3 rgb;
  if constexpr (COOP) {
    const int sub = (int)(threadIdx.x & 3u), use = want ? -1 : 0;
    Tap tx_, ty_, tz_;
    tap_coords(vc, pgx, tx_); tap_coords(vc, pgy, ty_); tap_coords(vc, pgz, tz_);
    CoopTap<VT, AM> cgx, cgy, cgz;
    cgx.issue(vc, tx_, use, sub); cgy.issue(vc, ty_, use, sub); cgz.issue(vc, tz_, use, sub);
    rgb = tf_color(tf, want ? r.v : 0.f);
    sgx = cgx.finish(vc, sub); sgy = cgy.finish(vc, sub); sgz = cgz.finish(vc, sub);
  }
  else {
    Tap tgx, tgy, tgz;
    tap_issue<VT, AM>(vc, pgx, tgx);
    tap_issue<VT, AM>(vc, pgy, tgy);
    tap_issue<VT, AM>(vc, pgz, tgz);
    rgb = tf_color(tf, r.v);
    sgx = tap_finish<VT>(vc, tgx); sgy = tap_finish<VT>(vc, tgy); sgz = tap_finish<VT>(vc, tgz);
  }
  f3 g;
#if OVR_PARITY_EXACT
  g.x = (sgx - r.s) / (flx ? -mc.gstep.x : mc.gstep.x); // (sample(c + stp) - v) / stp, shaders_common.h:195-215
  g.y = (sgy - r.s) / (fly ? -mc.gstep.y : mc.gstep.y);
  g.z = (sgz - r.s) / (flz ? -mc.gstep.z : mc.gstep.z);
#else
  g.x = (sgx - r.s) * (flx ? -mc.ginv.x : mc.ginv.x);
  g.y = (sgy - r.s) * (fly ? -mc.ginv.y : mc.ginv.y);
  g.z = (sgz - r.s) * (flz ? -mc.ginv.z : mc.ginv.z);
#endif
  const f3 gn = normalize3(g);
  const f3 n_o = mk3(-gn.x, -gn.y, -gn.z);
  const f3 n_w = normalize3(mk3(n_o.x * mc.otw_it.x, n_o.y * mc.otw_it.y, n_o.z * mc.otw_it.z));
  f3 n_c = mk3(0, 0, 0);
  if (P.grad) {
    const float* m = P.wtc_it;
    n_c = normalize3(mk3(fmaf(n_w.x, m[0], fmaf(n_w.y, m[3], n_w.z * m[6])), fmaf(n_w.x, m[1], fmaf(n_w.y, m[4], n_w.z * m[7])),
                         fmaf(n_w.x, m[2], fmaf(n_w.y, m[5], n_w.z * m[8]))));
  }
  float shadow = 0.f;
  if (SHADE == 2) {
    if constexpr (COOP) shadow = march_shadow_coop<VT, AM, kShadowTaps, SKIP>(vc, tf, mc, pos, want, n_shadow, n_shadow_skipped);
    else if (OVR_SHADOW_PACKED && !SKIP) shadow = march_shadow_packed<VT, AM, kShadowTaps>(vc, tf, mc, pos, n_shadow);
    else shadow = march_shadow<VT, AM, kShadowTaps, SKIP>(vc, tf, mc, pos, n_shadow, n_shadow_skipped);
  }
  const float cosNL = fabsf(dot3(mc.light, n_w));
  const float shade = 0.5f + 0.5f * cosNL * 2.f * (1.f - shadow); // shaders_raymarching.cu:156-157
  const float tr = r.tr;
  r.px = tr * clamp01(rgb.x * shade);
  r.py = tr * clamp01(rgb.y * shade);
  r.pz = tr * clamp01(rgb.z * shade);
  r.s = tr * clamp01(n_c.x);
  r.v = tr * clamp01(n_c.y);
  r.tr = tr * clamp01(n_c.z);
}

// hand the contributions of one shaded batch (stream positions [base, base + n), result of position base + j in lane j)
// back to the owning lanes: every owner walks its own requests of this batch in sample order
__device__ __forceinline__ void apply_batch(const ShadeReq& res, unsigned int base, unsigned int n, int lane, int& pend, unsigned int& first, f3& color,
                                            f3& gradient)
{
  for (;;) {
    const bool has = (pend > 0) && ((first - base) < n);
    if (__ballot(has) == 0ull) break;
    const int j = has ? (int)(first - base) : lane;
    const float tcx = bperm(j, res.px), tcy = bperm(j, res.py), tcz = bperm(j, res.pz);
    const float tgx = bperm(j, res.s), tgy = bperm(j, res.v), tgz = bperm(j, res.tr);
    const float ta = bperm(j, res.a);
    const int tn = bperm_i(j, res.next);
    if (has) {
      color.x = fmaf(tcx, ta, color.x);
      color.y = fmaf(tcy, ta, color.y);
      color.z = fmaf(tcz, ta, color.z);
      gradient.x = fmaf(tgx, ta, gradient.x);
      gradient.y = fmaf(tgy, ta, gradient.y);
      gradient.z = fmaf(tgz, ta, gradient.z);
      first = (unsigned int)tn;
      --pend;
    }
  }
}

constexpr int kNC = kBlockCounters; // counters: rays, samples, shaded, shadow, active pixels, skipped samples, skipped shadow samples, hits through an ignored slab (ignored_slab_outside)
// per-wave counters -> LDS -> one plain store of the workgroup's partial sums (lds must hold kWaves*kNC uints)
__device__ __forceinline__ void store_block_counters(const RayMarchParams& P, unsigned int* red, int lane, int wave, unsigned int n_rays,
                                                     unsigned int n_samples, unsigned int n_shaded, unsigned int n_shadow, unsigned int n_active,
                                                     unsigned int n_skipped, unsigned int n_shadow_skipped, unsigned int n_outside_hits)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_rays += __shfl_down(n_rays, off);
    n_samples += __shfl_down(n_samples, off);
    n_shaded += __shfl_down(n_shaded, off);
    n_shadow += __shfl_down(n_shadow, off);
    n_active += __shfl_down(n_active, off);
    n_skipped += __shfl_down(n_skipped, off);
    n_shadow_skipped += __shfl_down(n_shadow_skipped, off);
    n_outside_hits += __shfl_down(n_outside_hits, off);
  }
  if (!P.block_counters) return;
  __syncthreads(); // LDS is dead at this point: reuse its front
  if (lane == 0) {
    red[wave * kNC + 0] = n_rays; red[wave * kNC + 1] = n_samples; red[wave * kNC + 2] = n_shaded;
    red[wave * kNC + 3] = n_shadow; red[wave * kNC + 4] = n_active; red[wave * kNC + 5] = n_skipped; red[wave * kNC + 6] = n_shadow_skipped;
    red[wave * kNC + 7] = n_outside_hits;
  }
  __syncthreads();
  if (threadIdx.x < kNC) {
    const unsigned int bid = blockIdx.x;
    unsigned int sum = 0;
    for (int w = 0; w < kWaves; ++w) sum += red[w * kNC + threadIdx.x];
    P.block_counters[(size_t)bid * kNC + threadIdx.x] = sum;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// the march kernel (both pipelines)
//
// Lane mapping - "four lanes per ray": a wave handles 16 rays (a 4x4 pixel tile); the 4 lanes of a quad are 4 CONSECUTIVE
// STEPS of one ray.  The texture addresser coalesces a gather only inside quads of 4 consecutive lanes (16 clocks per
// instruction when a quad shares a 128-byte line, 66 when its lanes hit 4 lines - tools/ubench_lines.hip); with one lane
// per pixel (2 voxels apart at the bench's resolution) a quad almost never shared a brick, with 4 lanes per ray its taps
// are 1 voxel apart and nearly always do.  A round = K instructions x 4 steps; every lane recomputes the ray's t sequence
// (tx, ty recurrence, shaders_raymarching.cu:168-169) and the alpha recurrence (:165) for all 4K steps, fetching the
// other lanes' opacities with DPP quad broadcasts, so the arithmetic and its order are exactly the reference's.
//   POOLED = false: shade queued requests in place (raymarch pipeline 1)      POOLED = true: spill them to the pool
// ------------------------------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ float quad_bcast(float x) // value of lane B of this lane's quad
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), B * 0x55, 0xf, 0xf, true));
}
// a_sub for the lane's sub = lane & 3.  As nested conditionals the compiler builds this from compares and exec-mask branches (5 vector + 7
// scalar instructions per select, 8 selects per round of the march); with the two lane masks m1 = -(sub & 1), m2 = -((sub >> 1) & 1) kept in
// registers it is three v_bfi_b32 (bit-field insert: (m & b) | (~m & a)) - the same bits
#ifndef OVR_SEL4_BFI
#define OVR_SEL4_BFI 1
#endif
struct SubMask { unsigned int m1, m2; int sub; };
__device__ __forceinline__ SubMask make_submask(int sub) { SubMask m; m.m1 = 0u - (unsigned int)(sub & 1); m.m2 = 0u - (unsigned int)((sub >> 1) & 1); m.sub = sub; return m; }
__device__ __forceinline__ unsigned int bfi(unsigned int m, unsigned int b, unsigned int a) { return (m & b) | (~m & a); }
__device__ __forceinline__ float sel4(float a0, float a1, float a2, float a3, const SubMask& m)
{
  if (!OVR_SEL4_BFI) return m.sub == 0 ? a0 : m.sub == 1 ? a1 : m.sub == 2 ? a2 : a3;
  const unsigned int lo = bfi(m.m1, __float_as_uint(a1), __float_as_uint(a0)), hi = bfi(m.m1, __float_as_uint(a3), __float_as_uint(a2));
  return __uint_as_float(bfi(m.m2, hi, lo));
}

// Blue-noise pixel jitter (P.jitter_mode == 1; BASELINE C5, north_star): sample k of frame f takes slice
// t = ((f - 1) * spp + k) % 64 of the noise tile (lookup as blue_noise.h:95-99; the tile is stored transposed, [t][y][x]):
// xi0 = tile[t][iy % xy][ix % xy], xi1 = the same slice shifted by half a tile in x and y.  In dense mode the workgroup
// stages the 2 x 64 variates of its 8x8 pixels for the first kJitStaged samples in LDS; later samples and sparse-mode
// pixels read the tile directly.
constexpr int kJitStaged = 4;
__device__ __forceinline__ int jitter_slice(const RayMarchParams& P, int k) { return (int)((((long long)P.frame_index - 1) * P.spp + k) % 64); }
__device__ __forceinline__ void jitter_global(const RayMarchParams& P, int ix, int iy, int k, float& x0, float& x1)
{
  const int xy = P.jitter_xy, h = xy >> 1;
  const float* slice = P.jitter_noise + (size_t)jitter_slice(P, k) * xy * xy;
  x0 = slice[(size_t)(iy % xy) * xy + (ix % xy)];
  x1 = slice[(size_t)((iy + h) % xy) * xy + ((ix + h) % xy)];
}

// which pixel does this QUAD own?  (4x4 pixels per wave, 8x8 per workgroup; sparse mode: 64 list entries per workgroup)
// Dense mode: workgroup s of the 1-D grid renders the 8x8 block P.schedule[s] = bx | by << 16 - the blocks this rank owns,
// longest rays first (launch_schedule) - compute_screen_position of the reference (shaders_common.h:394-451) is the
// identity on the launch index, which fixes neither an order nor a grouping.
__device__ __forceinline__ bool assign_pixel_quad(const RayMarchParams& P, int lane, int wave, int& ix, int& iy)
{
  const int ray = lane >> 2;
  bool active;
  if (P.sparse_xy) {
    const unsigned long long i = (unsigned long long)blockIdx.x * (kBlock / 4) + (unsigned int)(threadIdx.x >> 2);
    active = (2ull * i) < *P.sparse_count;
    ix = active ? P.sparse_xy[2 * i] : 0;
    iy = active ? P.sparse_xy[2 * i + 1] : 0;
  }
  else {
    const unsigned int e = P.schedule[blockIdx.x];
    ix = (int)(e & 0xffffu) * 8 + (wave & 1) * 4 + (ray & 3);
    iy = (int)(e >> 16) * 8 + (wave >> 1) * 4 + (ray >> 2);
    active = ix < P.width && iy < P.height;
  }
  if (P.world > 1 && active) active = ((ix / P.tile_w + iy / P.tile_h) % P.world) == P.rank;
  return active;
}

// the primary rays' form: the 4 lanes of the quad each walk a quarter of [t0, t1]; min / max over the quad.  Two levels: the
// coarse grid over the whole ray, then - only inside the coarse interval - the per-macrocell grid: a primary ray is long (its
// coarse interval is up to 80 voxels too wide at either end, and a ray that only grazes the dilated coarse entries gets an
// interval although it meets nothing), and every round inside the interval costs ~330 instead of ~45 instructions.
#ifndef OVR_SKIP_FINE
#define OVR_SKIP_FINE 1
#endif
template <bool FINE>
__device__ __forceinline__ void skip_interval_level(const VolConsts& vc, f3 oo, f3 od, float t0, float t1, int sub, bool live, float& t_first, float& t_last)
{
  float first = FLT_MAX, last = -FLT_MAX;
  if (live) {
    const float len = t1 - t0;
    const float ta = fmaf((float)sub * 0.25f, len, t0), tb = sub == 3 ? t1 : fmaf((float)(sub + 1) * 0.25f, len, t0);
    skip_walk<FINE>(vc, oo, od, ta, tb, first, last);
  }
  t_first = fminf(fminf(quad_bcast<0>(first), quad_bcast<1>(first)), fminf(quad_bcast<2>(first), quad_bcast<3>(first)));
  t_last = fmaxf(fmaxf(quad_bcast<0>(last), quad_bcast<1>(last)), fmaxf(quad_bcast<2>(last), quad_bcast<3>(last)));
}
// Measured (profiles/r03_notes.md, r03_ab_segments.txt): bit-identical frames, the march with skipping 2-4 % SLOWER on every bench
// configuration and on the shipped scenes' shapes (C3 0.452 -> 0.467 ms, C2 0.334 -> 0.348, C5 1.70 -> 1.72) - their occupied macrocells are
// one blob per ray, the hull already is the interval, and the mask costs registers and a 64-bit test per round.  Data with real interior
// voids (the datasets do not ship) is where it would pay; off by default, -DOVR_SKIP_SEGMENTS=1 builds it.
#ifndef OVR_SKIP_SEGMENTS
#define OVR_SKIP_SEGMENTS 0
#endif
__device__ __forceinline__ SkipSpan skip_interval(const VolConsts& vc, f3 oo, f3 od, float t0, float t1, int sub, bool live)
{
  SkipSpan sp = skipspan_none();
  sp.mask_lo = sp.mask_hi = ~0u;
  skip_interval_level<false>(vc, oo, od, t0, t1, sub, live, sp.first, sp.last);
  if (OVR_SKIP_FINE) {
    const float c0 = sp.first, c1 = sp.last; // quad-uniform
    if (__ballot(live && c0 <= c1) != 0ull) {
      const bool walk = live && c0 <= c1;
      if (!OVR_SKIP_SEGMENTS) skip_interval_level<true>(vc, oo, od, c0, c1, sub, walk, sp.first, sp.last);
      else {
        // the fine walk over the coarse hull [c0, c1], a quarter per lane: hull and segment mask in one pass
        float first = FLT_MAX, last = -FLT_MAX;
        unsigned long long m = 0ull;
        const float len = c1 - c0;
        const float scale = len > 0.f ? 64.f / len : 0.f;
        if (walk) {
          const float ta = fmaf((float)sub * 0.25f, len, c0), tb = sub == 3 ? c1 : fmaf((float)(sub + 1) * 0.25f, len, c0);
          skip_walk<true, true>(vc, oo, od, ta, tb, first, last, c0, scale, &m);
          if (!(len > 0.f)) m = ~0ull;
        }
        sp.first = fminf(fminf(quad_bcast<0>(first), quad_bcast<1>(first)), fminf(quad_bcast<2>(first), quad_bcast<3>(first)));
        sp.last = fmaxf(fmaxf(quad_bcast<0>(last), quad_bcast<1>(last)), fmaxf(quad_bcast<2>(last), quad_bcast<3>(last)));
        const float lo = __uint_as_float((unsigned int)m), hi = __uint_as_float((unsigned int)(m >> 32));
        sp.mask_lo = __float_as_uint(quad_bcast<0>(lo)) | __float_as_uint(quad_bcast<1>(lo)) | __float_as_uint(quad_bcast<2>(lo)) | __float_as_uint(quad_bcast<3>(lo));
        sp.mask_hi = __float_as_uint(quad_bcast<0>(hi)) | __float_as_uint(quad_bcast<1>(hi)) | __float_as_uint(quad_bcast<2>(hi)) | __float_as_uint(quad_bcast<3>(hi));
        sp.seg0 = c0;
        sp.seg_scale = scale;
      }
    }
  }
  return sp;
}

template <int SHADE, bool POOLED> struct QCfg {
#ifndef OVR_POOLED_K
#define OVR_POOLED_K 4
#endif
  static constexpr int K = POOLED ? OVR_POOLED_K : (SHADE == 0 ? 4 : 3);   // instructions (x4 steps) per round
  static constexpr int QCAP = SHADE == 0 ? 0 : (POOLED ? 128 : 256);       // pooled: spills after every instruction
};

// Register budget: at most 3 waves per SIMD (up to 168 VGPRs).  Left alone the compiler squeezes the kernel into 128 VGPRs
// for a 4th wave by serialising the K tap groups it is supposed to keep in flight - measured 1.98 instead of 1.53 ms on C3.
#ifndef OVR_MARCH_WPE
#define OVR_MARCH_WPE 3
#endif
// LDSB = true: the "LDS-staged bricks" variant (north_star; measured in profiles/r02_notes.md).  Once per round the workgroup
// copies the bricks its 64 rays touch in the round's 16 steps - the brick-aligned bounding box of the block's four corner
// rays over the round's t range - from HBM / L2 into LDS with whole-line 16-byte loads (8 lanes per 128-byte brick), and the
// round's taps read LDS (ds_read2_b32 pairs) instead of going through the texture addresser.  A round whose box exceeds the LDS
// budget, and a tap that falls outside the staged box, take the ordinary path - the result is bit-identical either way.
// Built for the in-place, unshaded march of the general f32 layout (BASELINE C2, where rays are denser than voxels).
constexpr int kLdsBrickCap = 384;                       // 48 KiB of bricks per workgroup: two workgroups per CU
struct LdsRegion { int bx0, by0, bz0, ebx, eby, ebz, nbr, ok; };
// DEEP = true (plain pooled march only): 6 instead of 4 instructions (x 4 steps) per round - 24 pair loads in flight per lane at 2
// waves per SIMD (198 VGPRs) instead of 16 at 3.  On a full frame the two are within 1 % of each other (and 4 is better with dense
// transfer functions and on axis views), but a ray's chain of dependent rounds is the floor of an image SHARD's march, and there 6
// wins: 8-way shard of C3 0.295 -> 0.250 ms, 4-way 0.451 -> 0.423 (profiles/r02_notes.md).  launch_vsbs picks it for small shards (use_deep_rounds).
constexpr int kDeepK = 6;
template <int VT, int SHADE, int AM, bool POOLED, bool SKIP, bool LDSB = false, bool DEEP = false>
#ifndef OVR_PIN_AM
#define OVR_PIN_AM 1 /* the 64-bit addressing modes would spill to scratch under the pin */
#endif
// (the skipping pooled march sits at the edge of the 3-waves budget: 169 VGPRs - one too many - cost it 15 %; it is pinned to 3)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((SKIP && POOLED && (AM == 4 ? 0 : AM) <= OVR_PIN_AM) ? OVR_MARCH_WPE : 1, OVR_MARCH_WPE))) void raymarch_kernel(const RayMarchParams P)
{
  static_assert(!LDSB || (SHADE == 0 && !POOLED && !SKIP && AM <= 1 && !Vox<VT>::kTransposed), "LDS-staged bricks: unshaded in-place march only");
  using Cfg = QCfg<SHADE, POOLED>;
  static_assert(!DEEP || (POOLED && !SKIP && !LDSB), "the deep variant exists for the plain pooled march");
  constexpr int K = DEEP ? kDeepK : Cfg::K;
  constexpr int QCAP = Cfg::QCAP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 3;               // this lane's step inside each group of 4 consecutive steps
  const int qbase = lane & ~3;            // first lane of the quad
  const bool owner = sub == 0;            // the quad's lane that keeps the pixel's colour / request list
  const SubMask subm = make_submask(sub);
  const unsigned long long t_start = P.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

  int ix, iy;
  const bool active = assign_pixel_quad(P, lane, wave, ix, iy);
  unsigned int n_rays = 0, n_samples = 0, n_shaded = 0, n_shadow = 0, n_skipped = 0, n_shadow_skipped = 0, n_outside_hits = 0;
  VolConsts vc;
  MarchConsts mc;
  setup_consts(P, vc, mc);

  // ---- LDS carve: [request queues][offset tables][TF colour (not needed by the pooled march)][TF alpha]
  // A workgroup none of whose rays can hit the volume (most of the image outside the silhouette) stages nothing; with
  // empty-space skipping, neither does one whose rays only cross empty macrocells.
  ShadeReq* const queue = reinterpret_cast<ShadeReq*>(lds_raw) + (size_t)wave * (QCAP > 0 ? QCAP : 1);
  TfConsts tf;
  SkipSpan pro_span = skipspan_none(); // skipping, spp == 1: the ray's skip hull and segment mask, found here once
  const float rsx = 1.f / (float)P.width, rsy = 1.f / (float)P.height;
  const float scx = ((float)ix + .5f) * rsx, scy = ((float)iy + .5f) * rsy;
  // pooled: one launch per sample-per-pixel generation (P.spp_index), in place: all here.  min(P.spp, 1) is 1, but as a
  // run-time value: with a constant trip count of 1 the compiler restructures the kernel into a schedule that keeps fewer
  // taps in flight (126 instead of 153 VGPRs) and the C3 march takes 1.98 instead of 1.53 ms
  const int spp = POOLED ? min(P.spp, 1) : P.spp;
  // blue-noise jitter: the variates of this block's pixels, staged in LDS (dense mode) for the samples this launch renders
  __shared__ float jit_lds[kJitStaged][2][64];
  const bool jit_staged = P.jitter_mode == 1 && !P.sparse_xy;
  if (jit_staged) {
    const unsigned int e = P.schedule[blockIdx.x];
    const int xy = P.jitter_xy, h = xy >> 1;
    const int nk = min(spp, kJitStaged);
    for (int i = threadIdx.x; i < nk * 128; i += kBlock) {
      const int k = i >> 7, d = (i >> 6) & 1, p = i & 63;
      const int px = (int)(e & 0xffffu) * 8 + (p & 7) + d * h, py = (int)(e >> 16) * 8 + (p >> 3) + d * h;
      jit_lds[k][d][p] = P.jitter_noise[(size_t)jitter_slice(P, (POOLED ? P.spp_index : 0) + k) * xy * xy + (size_t)(py % xy) * xy + (px % xy)];
    }
    __syncthreads();
  }
  // the two jitter variates of sample k (k counts the samples of this launch)
  auto jitter = [&](int k, float& x0, float& x1) {
    if (jit_staged && k < kJitStaged) {
      const int ray = lane >> 2;
      const int p = ((wave >> 1) * 4 + (ray >> 2)) * 8 + (wave & 1) * 4 + (ray & 3);
      x0 = jit_lds[k][0][p];
      x1 = jit_lds[k][1][p];
    }
    else jitter_global(P, ix, iy, (POOLED ? P.spp_index : 0) + k, x0, x1);
  };
  bool staged; // workgroup-uniform: the offset tables and the transfer function are in LDS
  {
    bool need = active;
    if (P.spp == 1 && active) { // spp == 1: the ray is known - test it (the same expressions as the march below uses)
      float sx0 = scx, sy0 = scy;
      if (P.jitter_mode == 1) {
        float j0, j1;
        jitter(0, j0, j1);
        sx0 += (j0 - 0.5f) * rsx;
        sy0 += (j1 - 0.5f) * rsy;
      }
      const float ux0 = sx0 - 0.5f, uy0 = sy0 - 0.5f;
      const f3 c0 = ld3(P.cam_dir), h0 = ld3(P.cam_hor), v0 = ld3(P.cam_ver);
      const f3 d0 = normalize3_exact(mk3(c0.x + ux0 * h0.x + uy0 * v0.x, c0.y + ux0 * h0.y + uy0 * v0.y, c0.z + ux0 * h0.z + uy0 * v0.z));
      float a0 = 0.f, b0 = FLT_MAX;
      const f3 oo0 = to_object(mc, ld3(P.cam_pos)), od0 = mk3(d0.x * mc.inv_scale.x, d0.y * mc.inv_scale.y, d0.z * mc.inv_scale.z);
      need = intersect_unit_box(a0, b0, oo0, od0);
      if (SKIP && need) { // skipping: a ray that meets no occupied macrocell never fetches a voxel or a TF entry either
        pro_span = skip_interval(vc, oo0, od0, a0, b0, sub, true);
        need = pro_span.first <= pro_span.last;
      }
    }
    staged = __syncthreads_or(need ? 1 : 0) != 0;
    if (staged) {
      unsigned char* base = lds_raw + (size_t)kWaves * QCAP * sizeof(ShadeReq);
      const size_t tb = (stage_tables<VT, AM>(P, base, vc) + 15) & ~(size_t)15;
      stage_tf(P, base + tb, !POOLED, tf);
    }
    else {
      tf = TfConsts{};
      vc.tab_x = vc.tab_y = vc.tab_z = nullptr;
      vc.tab_z64 = nullptr;
    }
  }
  const PoolDesc& Q = P.pool;
  const unsigned int tile = blockIdx.x * kWaves + wave;
  // LDS-staged bricks: [.. tables, TF ..][bricks: kLdsBrickCap x 128 B][region descriptor][corner rays 4 x float3][t range]
  float* const lds_bricks = LDSB ? reinterpret_cast<float*>(lds_raw + P.lds_brick_offset) : nullptr;
  LdsRegion* const lds_region = reinterpret_cast<LdsRegion*>(lds_bricks + (size_t)kLdsBrickCap * 32);
  float* const lds_corner = reinterpret_cast<float*>(lds_region + 1);
  int* const lds_trange = reinterpret_cast<int*>(lds_corner + 12);
  unsigned int n_lds_fb_taps = 0, n_lds_fb_rounds = 0; // diagnostics: taps outside the staged box, rounds whose box did not fit
  if (LDSB && threadIdx.x < 4) { // directions of the block's corner rays, half a pixel outside the corner pixels (covers any jitter)
    const unsigned int e = P.schedule[blockIdx.x];
    const float cx_ = ((float)((int)(e & 0xffffu) * 8 + ((threadIdx.x & 1) ? 8 : 0))) / (float)P.width - 0.5f;
    const float cy_ = ((float)((int)(e >> 16) * 8 + ((threadIdx.x & 2) ? 8 : 0))) / (float)P.height - 0.5f;
    const f3 c0 = ld3(P.cam_dir), h0 = ld3(P.cam_hor), v0 = ld3(P.cam_ver);
    const f3 d = normalize3_exact(mk3(c0.x + cx_ * h0.x + cy_ * v0.x, c0.y + cx_ * h0.y + cy_ * v0.y, c0.z + cx_ * h0.z + cy_ * v0.z));
    lds_corner[3 * threadIdx.x] = d.x; lds_corner[3 * threadIdx.x + 1] = d.y; lds_corner[3 * threadIdx.x + 2] = d.z;
  }

  const unsigned int pixel_index = (unsigned int)ix + (unsigned int)iy * (unsigned int)P.width;
  unsigned int v0 = (unsigned int)P.frame_index, v1 = pixel_index; // RandomTEA(frame_index, pixel_index)
  const f3 org = ld3(P.cam_pos), cdir = ld3(P.cam_dir), chor = ld3(P.cam_hor), cver = ld3(P.cam_ver);
  const f3 oo = to_object(mc, org);

  float o_a = 0.f;
  f3 o_c = mk3(0, 0, 0), o_g = mk3(0, 0, 0);
  if (POOLED && P.spp > 1)
    for (int i = 0; i < P.spp_index; ++i) tea16(v0, v1); // RandomTEA state of this generation (random.h:146-188)
  // wave-uniform queue cursors (stream positions; slot = position & (QCAP - 1))
  unsigned int q_head = 0, q_tail = 0;
  // pooled: the tile's current reservation of consecutive chunks - kRun at first, doubling up to kRunMax with every further
  // reservation: a tile that pushes a lot (dense transfer function: every sample is shaded) would otherwise hit the one
  // pool counter every round - 48 k same-address returning atomics per C3 frame, which serialise in L2
  unsigned int run_base = 0, run_left = 0, run_size = 0, run_next = kRun;
  bool run_ok = true; // the current reservation lies inside its sub-pool
  int prev_chunk = -1;
  if (POOLED && lane == 0) Q.tile_first[tile] = -1;
  // per-ray request list (identical in the 4 lanes of the quad; the owner lane applies the contributions)
  int pend = 0;
  unsigned int first = 0, last = 0, last_gidx = 0;
  float alpha = 0.f;
  f3 color = mk3(0, 0, 0), gradient = mk3(0, 0, 0);

  // pooled: spill the n oldest queued requests as one chunk of the global pool
  auto spill = [&](unsigned int n) {
    if (run_left == 0) {
      const unsigned int sub = blockIdx.x & (unsigned int)(kPoolSubs - 1); // the workgroup's sub-pool (PoolDesc)
      unsigned int c0 = 0;
      if (lane == 0) c0 = atomicAdd(&Q.ctrl[32u * (sub + 1u)], run_next);
      c0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)c0);
      run_left = run_size = run_next;
      run_ok = c0 + run_size <= Q.sub_capacity;
      if (!run_ok && lane == 0) Q.ctrl[1] = 1u; // the counter keeps counting: the host sees how much was asked for
      run_base = sub * Q.sub_capacity + c0;
      run_next = min(run_next * 2u, (unsigned int)kRunMax);
    }
    const unsigned int c = run_base + (run_size - run_left);
    --run_left;
    if (run_ok) {
      __builtin_amdgcn_wave_barrier();
      if ((unsigned int)lane < n) Q.reqs[(size_t)c * 64 + lane] = queue[(q_head + lane) & (QCAP - 1)];
      if (lane == 0) {
        Q.chunk_n[c] = n;
        if (prev_chunk >= 0) Q.chunk_next[prev_chunk] = (int)c; else Q.tile_first[tile] = (int)c;
        if (Q.order && (c & (unsigned int)(kRun - 1)) == 0u) { // the first chunk of a run: its beam (shade order, PoolDesc)
          const ShadeReq& r0 = queue[q_head & (QCAP - 1)];
          const unsigned int key = beam_index(Q, r0.px, r0.py, r0.pz);
          Q.order_key[c / (unsigned int)kRun] = key;
          atomicAdd(&Q.order_ws[kOrderHist + key], 1u);
        }
      }
      if (pend > 0 && (last - q_head) < n) last_gidx = c * 64u + (last - q_head);
      prev_chunk = (int)c;
    }
    // beyond the sub-pool: it is exhausted; its counter keeps counting so the host knows how much was needed, re-sizes
    // the pool and renders the frame again (ovr_hip_api.cpp) - nothing of this frame is used
    q_head += n;
  };

  for (int k_spp = 0; k_spp < spp; ++k_spp) { // uniform trip count: every lane of the wave runs every round
    float sx = scx, sy = scy;
    if (P.jitter_mode == 1) {
      float j0, j1;
      jitter(k_spp, j0, j1);
      sx += (j0 - 0.5f) * rsx;
      sy += (j1 - 0.5f) * rsy;
    }
    else if (P.spp > 1) {
      tea16(v0, v1);
      sx += ((float)v0 * OVR_TEA_TOFLOAT - 0.5f) * rsx;
      sy += ((float)v1 * OVR_TEA_TOFLOAT - 0.5f) * rsy;
    }
    const float ux = sx - 0.5f, uy = sy - 0.5f;
    const f3 dir = normalize3_exact(mk3(cdir.x + ux * chor.x + uy * cver.x, cdir.y + ux * chor.y + uy * cver.y,
                                        cdir.z + ux * chor.z + uy * cver.z));
    // ---- __intersection__volume: object-space ray, direction not renormalised so t is shared
    const f3 od = mk3(dir.x * mc.inv_scale.x, dir.y * mc.inv_scale.y, dir.z * mc.inv_scale.z);
    float t0 = 0.f, t1 = FLT_MAX;
    alpha = 0.f;
    color = mk3(0, 0, 0);
    gradient = mk3(0, 0, 0);
    bool live = active && intersect_unit_box(t0, t1, oo, od);
    if (active && owner) ++n_rays;
    if (live && owner && ignored_slab_outside(oo, od)) ++n_outside_hits;
    SkipSpan span = skipspan_all(); // samples outside the hull, or in an unmarked segment of it, are in empty macrocells
    if (SKIP) {
      if (P.spp == 1) span = pro_span; // same ray, same [t0, t1] as in the prologue
      else span = skip_interval(vc, oo, od, t0, t1, sub, live);
    }
    // `staged` guards the prologue's decision (it tests the same ray with the same expressions): without the tables and the TF
    // in LDS no sample may be fetched - a skipping ray then only counts its (all empty) steps, any other ray is dead
    if (SKIP) { if (!staged) span = skipspan_none(); }
    else live = live && staged;
    float tx = t0, ty = fminf(t1, t0 + mc.step);
    pend = 0;
    unsigned int lds_round = 0;
    if (LDSB) { // the workgroup's range of entry distances: round r of every ray lies in [tmin + 16 r step, tmax + 16 (r + 1) step]
      __syncthreads();
      if (threadIdx.x == 0) { lds_trange[0] = 0x7f7fffff; lds_trange[1] = 0; }
      __syncthreads();
      if (live) { atomicMin(&lds_trange[0], __float_as_int(t0)); atomicMax(&lds_trange[1], __float_as_int(t0)); } // t0 >= 0: ordered as ints
      __syncthreads();
    }

    for (;;) {
      // LDSB: the rounds are workgroup-synchronous (the staging below has barriers): all four waves run until no ray of the block is live
      const bool any_live = LDSB ? (__syncthreads_or(live ? 1 : 0) != 0) : (__ballot(live) != 0ull);
      // ---- (1) in place: shade queued requests, a full batch whenever 64 are queued, the remainder once no ray is live
      if (SHADE != 0 && !POOLED) {
        while ((q_tail - q_head) >= 64u || (!any_live && q_tail != q_head)) {
          const unsigned int n = min(q_tail - q_head, 64u);
          __builtin_amdgcn_wave_barrier(); // requests were written by other lanes of this wave (LDS ops are in order)
          ShadeReq r;
          r.px = r.py = r.pz = r.s = r.v = r.tr = r.a = 0.f; r.next = 0;
          if ((unsigned int)lane < n) {
            r = queue[(q_head + lane) & (QCAP - 1)];
            const bool want = r.a > 0.f; // a == 0: null request
            if (Coop<VT, AM>::ok) { // whole quads enter (n is a multiple of 4): quad-cooperative taps, CoopTap
              if (((unsigned int)(__ballot(want) >> (lane & 60)) & 0xfu) != 0u) {
                ShadeReq rr = r;
                shade_request<VT, SHADE, AM, SKIP>(P, vc, tf, mc, rr, want, n_shadow, n_shadow_skipped);
                if (want) r = rr;
              }
            }
            else if (want) shade_request<VT, SHADE, AM, SKIP>(P, vc, tf, mc, r, true, n_shadow, n_shadow_skipped);
          }
          int opend = owner ? pend : 0;
          apply_batch(r, q_head, n, lane, opend, first, color, gradient);
          // the quad's lanes keep identical list cursors: take the owner's
          pend = __builtin_amdgcn_ds_bpermute(qbase << 2, opend);
          first = (unsigned int)__builtin_amdgcn_ds_bpermute(qbase << 2, (int)first);
          q_head += n;
        }
      }
      if (!any_live) break;

      // ---- (1b) empty-space skipping, bulk form: a round whose 4K steps all lie before (after) the ray's skip interval
      //      needs neither the lanes' own sample positions nor per-step bookkeeping - only the (tx, ty) recurrence, which
      //      has to be run step by step (its rounding is part of the result), and one validity test: the steps' midpoints
      //      lie in [tx_0, tx_4K], and validity (ty > tx) is monotone, so the last step being valid makes all of them valid.
      //      ~45 instructions per round instead of ~330 for the per-step form below.
      if (SKIP) {
        float ntx = tx, nty = ty, ptx = tx;
#pragma unroll
        for (int b = 0; b < 4 * K; ++b) {
          ptx = ntx;
          ntx = nty;
          nty = fminf(ntx + mc.step, t1);
        }
        const bool all_valid = ntx > ptx;                       // the round's last step: ty_last (= ntx) > tx_last (= ptx)
        const bool outside = !skipspan_touches(span, tx, ntx);
        const bool go = live && (alpha < 0.9999f);
        if (__ballot(live && !(outside && all_valid)) == 0ull) {
          n_skipped += go ? (unsigned int)K : 0u;               // this lane's K steps of the round
          live = go;
          tx = ntx; ty = nty;
          continue;
        }
      }
      // ---- (2) the ray's next 4K steps: every lane runs the (tx, ty) recurrence, keeps its own K steps and one validity
      //      bit per step (ty > tx, the first half of the reference's loop condition)
      unsigned int vmask = 0;
      Tap taps[K];
      f3 poss[K];
      float dts[K], mj[K], tms[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float txq[4], tyq[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          txq[b] = tx; tyq[b] = ty;
          vmask |= (ty > tx) ? (1u << (4 * k + b)) : 0u;
          tx = ty;
          ty = fminf(tx + mc.step, t1);
        }
        const float mtx = sel4(txq[0], txq[1], txq[2], txq[3], subm);
        const float mty = sel4(tyq[0], tyq[1], tyq[2], tyq[3], subm);
        dts[k] = mty - mtx;
        tms[k] = 0.5f * (mtx + mty);
      }
      // empty-space fast path (wave-uniform): every sample of this round lies in a macrocell whose majorant is 0, so all
      // opacities are exactly 0, alpha does not move and nothing is pushed - only liveness and the counters advance
      auto skip_round = [&]() {
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            live = live && ((vmask >> (4 * k + b)) & 1u) != 0u && (alpha < 0.9999f);
            n_skipped += (live && sub == b) ? 1u : 0u;
          }
        }
      };
      if (SKIP) {
        // (a) by the ray's skip interval: no coordinates, no majorant lookups
        bool inside = false;
#pragma unroll
        for (int k = 0; k < K; ++k) inside = inside || skipspan_inside(span, tms[k]);
        if (__ballot(inside && live) == 0ull) { skip_round(); continue; }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        poss[k] = mk3(fmaf(tms[k], dir.x, org.x), fmaf(tms[k], dir.y, org.y), fmaf(tms[k], dir.z, org.z));
        tap_coords(vc, to_object(mc, poss[k]), taps[k]);
        mj[k] = SKIP ? vc.majorant[tap_cell(vc, taps[k])] : 1.f; // empty-space skipping: the macrocell's max TF opacity
      }
      if (SKIP) {
        // (b) by the majorants of this round's own macrocells
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) any = any || (mj[k] > 0.f);
        if (__ballot(any && live) == 0ull) { skip_round(); continue; }
      }
      if (LDSB) {
        typedef BrickMap<VT> M;
        // (a) wave 0: the brick-aligned box of this round - 4 corner rays x {t_lo, t_hi} (the rays of the block and the round's
        //     t range span a convex set between them), one voxel of margin, one more for the taps' upper neighbours
        if (wave == 0) {
          // (no safety margins: a tap that falls outside the box is caught below and read the ordinary way)
          const float t_lo = __int_as_float(lds_trange[0]) + (float)(lds_round * (unsigned)(4 * K)) * mc.step;
          const float t_hi = __int_as_float(lds_trange[1]) + (float)((lds_round + 1u) * (unsigned)(4 * K)) * mc.step;
          const int c = lane & 3;
          const float tt = (lane & 4) ? t_hi : t_lo;
          const f3 pw = mk3(fmaf(tt, lds_corner[3 * c], org.x), fmaf(tt, lds_corner[3 * c + 1], org.y), fmaf(tt, lds_corner[3 * c + 2], org.z));
          const f3 po = to_object(mc, pw);
          float lo[3] = { fmaf(po.x, vc.cs.x, vc.cb.x), fmaf(po.y, vc.cs.y, vc.cb.y), fmaf(po.z, vc.cs.z, vc.cb.z) };
          float hi[3] = { lo[0], lo[1], lo[2] };
#pragma unroll
          for (int off = 1; off < 8; off <<= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
              hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
            }
          if (lane == 0) {
            const int n1[3] = { vc.nx1, vc.ny1, vc.nz1 };
            int l[3], h[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              l[a] = min(max((int)floorf(fminf(fmaxf(lo[a], -4.f), 1e9f)), 0), n1[a]);
              h[a] = min(max((int)floorf(fminf(fmaxf(hi[a], -4.f), 1e9f)) + 1, 0), n1[a]); // + 1: the taps' upper neighbours
            }
            // along x the bricks are addressed by STORED position (voxel + 1): the pairs' lower members span [floor(lo) + 1, floor(hi) + 1]
            const int lu = min(max((int)floorf(fminf(fmaxf(lo[0], -4.f), 1e9f)) + 1, 0), n1[0] + 1);
            const int hu = min(max((int)floorf(fminf(fmaxf(hi[0], -4.f), 1e9f)) + 1, 0), n1[0] + 1);
            LdsRegion g;
            g.bx0 = (int)M::div_cx((unsigned)lu); g.by0 = l[1] >> Vox<VT>::by; g.bz0 = l[2] >> Vox<VT>::bz;
            g.ebx = (int)M::div_cx((unsigned)hu) - g.bx0 + 1; g.eby = (h[1] >> Vox<VT>::by) - g.by0 + 1; g.ebz = (h[2] >> Vox<VT>::bz) - g.bz0 + 1;
            g.nbr = g.ebx * g.eby * g.ebz;
            g.ok = g.nbr <= kLdsBrickCap ? 1 : 0;
            *lds_region = g;
          }
        }
        __syncthreads(); // also: every wave has finished reading the previous round's bricks
        const LdsRegion g = *lds_region;
        ++lds_round;
        n_shadow += (lane == 0 && wave == 0) ? 1u : 0u; // diagnostics: workgroup rounds (the unshaded march has no shadow samples)
        if (g.ok) {
          // (b) copy the bricks: 8 lanes x 16 bytes per brick, whole 128-byte lines.  (Issuing all of a thread's up to 12 loads
          //     into a register array before the first store was measured slower: 1.48 instead of 1.02 ms on C2 front.)
          const int row = threadIdx.x & 7;
          const float rcp_x = 1.f / (float)g.ebx, rcp_xy = 1.f / (float)(g.ebx * g.eby);
          for (int i = threadIdx.x >> 3; i < g.nbr; i += kBlock / 8) {
            const int iz = (int)(((float)i + 0.5f) * rcp_xy), rem = i - iz * g.ebx * g.eby;
            const int iy = (int)(((float)rem + 0.5f) * rcp_x), ixb = rem - iy * g.ebx;
            // (tab_x is indexed by voxel = stored position - 1; tab_y / tab_z by voxel)
            const unsigned int off = vc.tab_x[(g.bx0 + ixb) * Vox<VT>::cx - 1] + vc.tab_y[(g.by0 + iy) << Vox<VT>::by] + vc.tab_z[(g.bz0 + iz) << Vox<VT>::bz];
            const float4 v = AM == 0 ? *reinterpret_cast<const float4*>(static_cast<const char*>(vc.data) + off + row * 16)
                                     : *reinterpret_cast<const float4*>(static_cast<const float*>(vc.data) + off + row * 4);
            reinterpret_cast<float4*>(lds_bricks)[i * 8 + row] = v;
          }
          __syncthreads();
          // (c) the taps: region-relative brick index + position inside the brick; anything outside the box goes the ordinary way
          const int sxy = g.ebx * g.eby;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            Tap& t = taps[k];
            const int y0 = max(t.y0, 0), z0 = max(t.z0, 0); // clamp-to-edge: index -1 reads voxel 0
            const int y1 = min(t.y0 + 1, vc.ny1), z1 = min(t.z0 + 1, vc.nz1);
            const int xu = t.x0 + 1;                          // stored position of the pair's lower member
            const int bxr = (int)M::div_cx((unsigned)xu), xr = xu - bxr * Vox<VT>::cx;
            const int cy0 = (y0 >> Vox<VT>::by) - g.by0, cy1 = (y1 >> Vox<VT>::by) - g.by0, cz0 = (z0 >> Vox<VT>::bz) - g.bz0, cz1 = (z1 >> Vox<VT>::bz) - g.bz0;
            const int cxr = bxr - g.bx0;
            const bool inside = (unsigned)cxr < (unsigned)g.ebx && (unsigned)cy0 < (unsigned)g.eby && (unsigned)cy1 < (unsigned)g.eby &&
                                (unsigned)cz0 < (unsigned)g.ebz && (unsigned)cz1 < (unsigned)g.ebz;
            if (inside) {
              constexpr int ym = (1 << Vox<VT>::by) - 1, zm = (1 << Vox<VT>::bz) - 1, SX = (int)M::SX;
              const int ox = cxr * (int)M::BV + xr;
              const int oy0 = cy0 * g.ebx * (int)M::BV + (y0 & ym) * SX, oy1 = cy1 * g.ebx * (int)M::BV + (y1 & ym) * SX;
              const int oz0 = cz0 * sxy * (int)M::BV + ((z0 & zm) << Vox<VT>::by) * SX, oz1 = cz1 * sxy * (int)M::BV + ((z1 & zm) << Vox<VT>::by) * SX;
              const float* b0 = lds_bricks + ox + oz0;
              const float* b1 = lds_bricks + ox + oz1;
              t.c000 = b0[oy0]; t.c100 = b0[oy0 + 1]; t.c010 = b0[oy1]; t.c110 = b0[oy1 + 1];
              t.c001 = b1[oy0]; t.c101 = b1[oy0 + 1]; t.c011 = b1[oy1]; t.c111 = b1[oy1 + 1];
            }
            else {
              tap_loads<VT, AM>(vc, t);
              n_lds_fb_taps += live ? 1u : 0u;
            }
          }
        }
        else {
          n_lds_fb_rounds += (lane == 0 && wave == 0) ? 1u : 0u;
#pragma unroll
          for (int k = 0; k < K; ++k) tap_loads<VT, AM>(vc, taps[k]);
        }
      }
      else {
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (!SKIP || mj[k] > 0.f) tap_loads<VT, AM>(vc, taps[k]);
      }
      // ---- (3) own samples: value, TF coordinate, corrected opacity (and colour when shading is off)
      float sa[K], va[K], aa[K];
      f3 ca[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        sa[k] = tap_finish<VT>(vc, taps[k]);
        va[k] = tf_coord(tf, sa[k]);
#ifndef OVR_MARCH_TF_PAD
#define OVR_MARCH_TF_PAD 0 /* 1: the pooled march takes the paired read too (measurements) */
#endif
        aa[k] = opacity_correction<false>(tf_alpha<!POOLED || OVR_MARCH_TF_PAD>(tf, va[k]), mc.base * dts[k]);
        if (SKIP) aa[k] = mj[k] > 0.f ? aa[k] : 0.f; // a macrocell whose majorant is 0 holds no sample with opacity > 0
      }
      // ---- (3b) transparent round (wave-uniform; ~9 of 10 rounds with a sparse transfer function): no live sample of any
      //      ray of the wave has opacity > 0, so every step adds exactly 0 to alpha and colour (fma(tr, 0, x) == x) and
      //      pushes nothing - only liveness and the counters move.  The validity bits are monotone (once ty == tx == t1 it
      //      stays), so step i is live iff the ray was live at the start of the round, alpha < 0.9999 and bit i is set.
#ifndef OVR_FAST_SKIP
#define OVR_FAST_SKIP 0 /* the skipping kernels have their own, earlier fast path; this one costs them 30 VGPRs */
#endif
      if (OVR_FAST_SKIP || !SKIP) {
        bool opaque = false;
#pragma unroll
        for (int k = 0; k < K; ++k) opaque = opaque || (aa[k] > 0.f);
        if (__ballot(opaque && live) == 0ull) {
          const bool go = live && (alpha < 0.9999f);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const bool ml = go && ((vmask >> (4 * k + sub)) & 1u) != 0u;
            n_samples += (ml && (!SKIP || mj[k] > 0.f)) ? 1u : 0u;
            if (SKIP) n_skipped += (ml && !(mj[k] > 0.f)) ? 1u : 0u;
          }
          live = go && ((vmask >> (4 * K - 1)) & 1u) != 0u;
          continue;
        }
      }
      // (round 3) the colours of an unshaded march are only needed from here on: ~9 of 10 rounds leave through the transparent-round exit
      // above and never look them up (15 vector instructions and two 12-byte LDS reads per sample; C2 is bound by vector issue)
      if (SHADE == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const f3 rgb = tf_color(tf, va[k]);
          ca[k] = mk3(clamp01(rgb.x), clamp01(rgb.y), clamp01(rgb.z));
        }
      }
      // ---- (4) the ray's alpha recurrence over the 4K steps, in order; every lane of the quad computes all of it
      //      (a dead or zero-opacity step feeds a = 0: fma(tr, 0, alpha) == alpha exactly, so no select is needed)
      bool mlive[K], mpush[K];
      float mtr[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float al[4];   // alpha BEFORE each of the 4 steps
        bool lv[4];    // the reference's loop condition at each step, shaders_raymarching.cu:110
#define OVR_STEP(B)                                                                                                            \
        {                                                                                                                      \
          const float aj = quad_bcast<B>(aa[k]);                                                                               \
          live = live && ((vmask >> (4 * k + B)) & 1u) != 0u && (alpha < 0.9999f);                                             \
          lv[B] = live;                                                                                                        \
          al[B] = alpha;                                                                                                       \
          const float ae = live ? aj : 0.f;                                                                                    \
          const float trj = 1.f - alpha;                                                                                       \
          if (SHADE == 0) {                                                                                                    \
            const float cxj = quad_bcast<B>(ca[k].x), cyj = quad_bcast<B>(ca[k].y), czj = quad_bcast<B>(ca[k].z);              \
            color.x = fmaf(trj * cxj, ae, color.x);                                                                            \
            color.y = fmaf(trj * cyj, ae, color.y);                                                                            \
            color.z = fmaf(trj * czj, ae, color.z);                                                                            \
          }                                                                                                                    \
          alpha = fmaf(trj, ae, alpha);                                                                                        \
        }
        OVR_STEP(0) OVR_STEP(1) OVR_STEP(2) OVR_STEP(3)
#undef OVR_STEP
        mlive[k] = sub == 0 ? lv[0] : sub == 1 ? lv[1] : sub == 2 ? lv[2] : lv[3];
        mtr[k] = 1.f - sel4(al[0], al[1], al[2], al[3], subm);
        // a sample whose corrected opacity is exactly 0 adds exactly 0 to colour, gradient and alpha: nothing is shaded
        mpush[k] = mlive[k] && (aa[k] > 0.f);
      }
      // ---- (5) count; queue the samples that need shading (slot = tail + prefix of the ballot, lane order = step order)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        n_samples += (mlive[k] && (!SKIP || mj[k] > 0.f)) ? 1u : 0u;
        if (SKIP) n_skipped += (mlive[k] && !(mj[k] > 0.f)) ? 1u : 0u;
        n_shaded += mpush[k] ? 1u : 0u;
        if (SHADE != 0) {
          // Quad-granular compaction: if any of a ray's 4 steps needs shading the ray takes 4 consecutive slots (the steps
          // that do not are written as null requests, a == 0, and cost the shader nothing but an idle lane).  Stream
          // positions stay multiples of 4, so in every 64-request chunk the 4 lanes of a quad shade 4 consecutive steps of
          // ONE ray: their gradient and shadow taps are 1 voxel apart and share bricks (texture-addresser coalescing).
          const bool push = mpush[k];
          const unsigned long long mp = __ballot(push);
          if (mp != 0ull) {
            const unsigned int quad_bits = (unsigned int)(mp >> qbase) & 0xfu;
            const bool qpush = quad_bits != 0u;
            const unsigned long long m = __ballot(qpush); // whole quads
            const unsigned int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (qpush) {
              const unsigned int pos_q = q_tail + below;            // = quad_first + sub
              const unsigned int quad_first = pos_q - (unsigned int)sub;
              const unsigned int higher = quad_bits >> (sub + 1);   // later steps of this ray pushed by this instruction
              ShadeReq r;
              r.px = poss[k].x; r.py = poss[k].y; r.pz = poss[k].z;
              r.s = sa[k]; r.v = va[k]; r.tr = mtr[k];
              r.a = push ? aa[k] : 0.f;
              r.next = (push && higher != 0u) ? (int)(pos_q + 1u + (unsigned int)__builtin_ctz(higher)) : 0;
              queue[pos_q & (QCAP - 1)] = r;
              const unsigned int first_sub = (unsigned int)__builtin_ctz(quad_bits), last_sub = 31u - (unsigned int)__builtin_clz(quad_bits);
              if (push && (unsigned int)sub == first_sub && pend > 0) { // link the ray's previous request to this one
                if (!POOLED || (int)(last - q_head) >= 0) queue[last & (QCAP - 1)].next = (int)pos_q; // still in LDS
                else Q.reqs[last_gidx].next = (int)pos_q;                                              // already spilled
              }
              if (pend == 0) first = quad_first + first_sub;
              last = quad_first + last_sub;
              pend += (int)__popc(quad_bits);
            }
            q_tail += (unsigned int)__popcll(m);
            if (POOLED && (q_tail - q_head) >= 64u) spill(64u);
          }
        }
      }
    }

    // render_raymarching / alpha_blend with an always-missing background (shaders_raymarching.cu:260-321)
    if (!POOLED) {
      o_a += alpha;
      if (alpha > 0.f) {
        o_c.x += color.x / alpha; o_c.y += color.y / alpha; o_c.z += color.z / alpha;
        o_g.x += gradient.x / alpha; o_g.y += gradient.y / alpha; o_g.z += gradient.z / alpha;
      }
    }
  }

  if (POOLED) {
    if (q_tail != q_head) spill(q_tail - q_head); // the tile's last, partial chunk
    if (lane == 0 && run_ok)
      for (unsigned int i = run_size - run_left; i < run_size && run_left != 0; ++i) { // unused tail of the reservation
        Q.chunk_n[run_base + i] = 0;
        if (Q.order && ((run_base + i) & (unsigned int)(kRun - 1)) == 0u) Q.order_key[(run_base + i) / (unsigned int)kRun] = kOrderNoKey;
      }
    if (lane == 0) Q.tile_count[tile] = q_tail;
    if (active && owner) Q.pix_state[pixel_index] = make_float4(alpha, __uint_as_float(first), __int_as_float(pend), 0.f);
  }
  else if (active && owner) {
    const float rspp = 1.f / (float)spp;
    o_a *= rspp;
    o_c.x *= rspp; o_c.y *= rspp; o_c.z *= rspp;
    o_g.x *= rspp; o_g.y *= rspp; o_g.z *= rspp;
    write_pixel(P, pixel_index, o_c, o_a, o_g);
  }

  if (P.trace && lane == 0) { // diagnostic only (OVR_HIP_TRACE): per-wave residency interval and work, never read by the kernel
    unsigned long long* t = P.trace + (size_t)tile * 4;
    t[0] = t_start; t[1] = __builtin_amdgcn_s_memrealtime();
    t[2] = ((unsigned long long)n_samples << 32) | n_shaded; t[3] = n_shadow;
  }
  // LDSB (never combined with skipping): the two skip counters carry the staging diagnostics instead
  store_block_counters(P, reinterpret_cast<unsigned int*>(lds_raw), lane, wave, n_rays, n_samples, n_shaded, n_shadow,
                       (active && owner && (!POOLED || P.spp_index == 0)) ? 1u : 0u, LDSB ? n_lds_fb_taps : n_skipped,
                       LDSB ? n_lds_fb_rounds : n_shadow_skipped, n_outside_hits);
}

// ------------------------------------------------------------------------------------------------------------------
// pooled pipeline, kernel B: persistent waves shade chunks from all tiles; one returning atomic per chunk
// ------------------------------------------------------------------------------------------------------------------
template <int VT, int SHADE, int AM, bool SKIP>
__global__ __launch_bounds__(kBlock) void shade_pool_kernel(const RayMarchParams P)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TfConsts tf;
  VolConsts vc;
  MarchConsts mc;
  setup_consts(P, vc, mc);
  const size_t tb = (stage_tables<VT, AM>(P, lds_raw, vc) + 15) & ~(size_t)15; // [offset tables][TF]
  stage_tf(P, lds_raw + tb, true, tf);
  const PoolDesc& Q = P.pool;
  // runs per sub-pool; tickets enumerate them block-cyclically - kTicketBlock consecutive tickets are kTicketBlock consecutive
  // runs of ONE sub-pool (usually of one tile), the next kTicketBlock tickets belong to the next sub-pool; a ticket beyond its
  // sub-pool's runs is idle
  __shared__ unsigned int s_runs[kPoolSubs];
  __shared__ unsigned int s_tot[2];
  if (threadIdx.x < (unsigned int)kPoolSubs) s_runs[threadIdx.x] = min(Q.ctrl[32u * (threadIdx.x + 1u)], Q.sub_capacity) / (unsigned int)kRun;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int mx = 0, sum = 0;
    for (int i = 0; i < kPoolSubs; ++i) { mx = max(mx, s_runs[i]); sum += s_runs[i]; }
    s_tot[0] = mx; s_tot[1] = sum;
  }
  __syncthreads();
  const bool overflow = Q.ctrl[1] != 0u; // a sub-pool ran out: the frame is re-rendered, nothing to shade
  const unsigned int total_runs = overflow ? 0u : s_tot[1];
  const unsigned int n_runs = overflow ? 0u : ((s_tot[0] + kTicketBlock - 1) / kTicketBlock) * kTicketBlock * (unsigned int)kPoolSubs; // tickets
  unsigned int n_shadow = 0, n_shadow_skipped = 0;
  __shared__ unsigned int s_run;
  // Guided self-scheduling: a workgroup takes a BATCH of consecutive runs per ticket - (runs left) / (2 x workgroups), at most
  // kTicketRuns, at least 1 - because the ticket is a same-address returning atomic and those serialise in L2 (~30 clocks each:
  // with one run per ticket the dense-transfer-function frame spent its whole shade time on 116 k tickets).  Large batches while
  // there is plenty of work, single runs at the end for balance.  Inside a batch the 4 waves walk their chunks independently
  // (run r, chunk `wave`): consecutive depth steps of ONE tile, whose gradient and shadow taps fall into the same bricks.
  // The cap follows the frame: 1 run per ticket up to 32 runs per workgroup (sparse transfer functions: 36 k runs per C3 frame -
  // there batches only cost balance, measured +3 ... 13 %), up to kTicketRuns beyond (dense: 174 k runs, shade 2.11 -> 1.13 ms)
  // Without the shadow march (SHADE == 1) a run is three gradient taps per request - so cheap that even 36 k single-run tickets ARE the
  // kernel's time (0.45 ms on C3: ~16 ns per same-address atomic); there the cap is always kTicketRuns (shade 0.45 -> 0.20 ms, profiles/r02_notes.md §11)
  // (with the shadow march the batches cost more in balance and locality than the tickets do - also in the skipping kernel: 0.668 -> 0.729 ms)
  const unsigned int ticket_cap = SHADE == 1 ? (unsigned int)kTicketRuns : min((unsigned int)kTicketRuns, max(1u, total_runs / (32u * gridDim.x)));
  auto shade_chunk = [&](unsigned int c) {
    const unsigned int n = Q.chunk_n[c];
    if ((unsigned int)lane < n) {
      ShadeReq r = Q.reqs[(size_t)c * 64 + lane];
      const bool want = r.a > 0.f; // a == 0: null request (a step of the quad that needs no shading)
      // whole quads enter (a chunk holds whole quads): the taps are quad-cooperative where the layout allows (CoopTap)
      if (Coop<VT, AM>::ok ? (((unsigned int)(__ballot(want) >> (lane & 60)) & 0xfu) != 0u) : want) {
        shade_request<VT, SHADE, AM, SKIP>(P, vc, tf, mc, r, want, n_shadow, n_shadow_skipped);
        if (want) Q.reqs[(size_t)c * 64 + lane] = r;
      }
    }
  };
  if (Q.order) {
    // (the next generation's march counts into a zeroed histogram; the order kernel's fill counters likewise)
    for (unsigned int i = blockIdx.x * kBlock + threadIdx.x; i < 2u * (unsigned int)kOrderMaxKeys; i += gridDim.x * kBlock) {
      const unsigned int e = i < (unsigned int)kOrderMaxKeys ? i : i - (unsigned int)kOrderMaxKeys;
      if (e < (unsigned int)(Q.order_grid * Q.order_grid)) Q.order_ws[i] = 0u;
    }
    // Runs sorted by light beam (PoolDesc::order, shade_order_kernel): 8 lists with a ticket counter each; the workgroups of an XCD start on
    // "their" list, so the bricks a beam's shadow rays sweep are fetched into ONE L2 and hit there by the beam's later runs; a workgroup whose
    // list is done helps with the others (the cursors only grow: every workgroup passes every list once and leaves)
    const unsigned int* ls = Q.order_ws + kOrderListStart;
    unsigned int* tick = Q.order_ws + kOrderTickets;
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const unsigned int total = overflow ? 0u : ls[kOrderLists];
    const unsigned int cap = SHADE == 1 ? (unsigned int)kTicketRuns : min((unsigned int)kTicketRuns, max(1u, total / (32u * gridDim.x)));
    const unsigned int wgs = max(1u, gridDim.x / (unsigned int)kOrderLists);
    for (unsigned int li = 0; li < (unsigned int)kOrderLists; ++li) {
      const unsigned int l = (xcc + li) & (unsigned int)(kOrderLists - 1);
      const unsigned int beg = ls[l], n_l = overflow ? 0u : ls[l + 1] - beg;
      unsigned int seen = 0;
      for (;;) {
        const unsigned int left = n_l > seen ? n_l - seen : 0u;
        const unsigned int batch = min(cap, max(1u, left / (2u * wgs)));
        __syncthreads();
        if (threadIdx.x == 0) s_run = atomicAdd(&tick[32u * l], batch);
        __syncthreads();
        const unsigned int run0 = s_run;
        if (run0 >= n_l) break;
        seen = run0 + batch;
        const unsigned int run1 = min(run0 + batch, n_l);
        for (unsigned int ticket = run0; ticket < run1; ++ticket) {
          if (ticket != run0) __syncthreads(); // the 4 waves stay on ONE run
          const unsigned int c0 = Q.order[beg + ticket] * (unsigned int)kRun;
          for (unsigned int i = (unsigned int)wave; i < (unsigned int)kRun; i += kWaves) shade_chunk(c0 + i);
        }
      }
    }
  }
  else {
  unsigned int seen = 0; // a lower bound of the global cursor: the end of this workgroup's last batch
  for (;;) {
    const unsigned int left = n_runs > seen ? n_runs - seen : 0u;
    const unsigned int batch = min(ticket_cap, max(1u, left / (2u * gridDim.x)));
    __syncthreads();
    if (threadIdx.x == 0) s_run = atomicAdd(&Q.ctrl[0], batch);
    __syncthreads();
    const unsigned int run0 = s_run;
    if (run0 >= n_runs) break; // every workgroup reaches this: the cursor only grows
    seen = run0 + batch;
    const unsigned int run1 = min(run0 + batch, n_runs);
    for (unsigned int ticket = run0; ticket < run1; ++ticket) {
#ifndef OVR_TICKET_LOCKSTEP
#define OVR_TICKET_LOCKSTEP 1
#endif
      if (OVR_TICKET_LOCKSTEP && ticket != run0) __syncthreads(); // the 4 waves stay on ONE run: its chunks share their bricks in L1
      const unsigned int grp = ticket / kTicketBlock, sub = grp & (unsigned int)(kPoolSubs - 1);
      const unsigned int run = (grp / (unsigned int)kPoolSubs) * kTicketBlock + ticket % kTicketBlock;
      if (run >= s_runs[sub]) continue; // idle ticket (workgroup-uniform)
      for (unsigned int i = (unsigned int)wave; i < (unsigned int)kRun; i += kWaves) shade_chunk(sub * Q.sub_capacity + run * kRun + i);
    }
  }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_shadow += __shfl_down(n_shadow, off);
    n_shadow_skipped += __shfl_down(n_shadow_skipped, off);
  }
  __syncthreads();
  unsigned int* red = reinterpret_cast<unsigned int*>(lds_raw);
  if (lane == 0) { red[wave] = n_shadow; red[kWaves + wave] = n_shadow_skipped; }
  __syncthreads();
  if (threadIdx.x == 0 && Q.shade_counters) {
    Q.shade_counters[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    Q.shade_counters[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
  }
}


constexpr int kReduceBlocks = 64;

// the two kernels of the pooled pipeline that do not depend on the voxel type live in ovr_hip_kernels.hip
hipError_t launch_composite(const RayMarchParams& q, dim3 grid, hipStream_t stream);
hipError_t launch_shade_order(const RayMarchParams& q, hipStream_t stream);
hipError_t launch_reduce_counters(const unsigned int* partials, int n_blocks, const unsigned int* shade_partials, int n_shade_blocks,
                                  unsigned long long* counters, unsigned int* pool_ctrl, unsigned long long* publish, unsigned int* done, hipStream_t stream);

inline dim3 raymarch_grid(const RayMarchParams& p)
{
  return dim3((unsigned)raymarch_grid_blocks(p));
}

template <typename KernT>
inline hipError_t set_lds(KernT kern, size_t lds)
{
  if (lds > 64 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return hipSuccess;
}

constexpr int kShadeBlocks = 1024; // persistent shade grid: at most 4 workgroups per CU (size of the shade_counters workspace)
// workgroups of the persistent shade kernel for this frame; OVR_HIP_SHADE_BLOCKS overrides it (measurements)
inline int shade_grid_blocks(const RayMarchParams& p)
{
  static const int forced = getenv("OVR_HIP_SHADE_BLOCKS") ? atoi(getenv("OVR_HIP_SHADE_BLOCKS")) : 0;
  if (forced > 0) return std::min(forced, kShadeBlocks);
  return p.shade_blocks > 0 ? std::min(p.shade_blocks, kShadeBlocks) : kShadeBlocks;
}

// The deep variant of the pooled march (6 instead of 4 instructions per round, 2 instead of 3 waves per SIMD) pays when the launch is
// bound by its longest ray's chain of dependent rounds rather than by throughput: image shards with few blocks.  Measured
// (profiles/r02_ab/r02b_deep.txt, march ms plain -> deep): C3 4-way 0.443 -> 0.422, 8-way 0.288 -> 0.254, 2-way (16 200 blocks) equal;
// but C5 (4K: 16 200 blocks even 8-way, throughput-bound) 0.508 -> 0.607 and C4 8-way (64-bit addressing) 0.604 -> 0.632: so only
// shards of at most 10 000 blocks with 32-bit addressing take it.  OVR_HIP_DEEP=0|1 overrides the choice (measurements).
#ifndef OVR_DEEP_MAX_BLOCKS
#define OVR_DEEP_MAX_BLOCKS 10000
#endif
inline bool use_deep_rounds(const RayMarchParams& p)
{
  static const int forced = getenv("OVR_HIP_DEEP") ? atoi(getenv("OVR_HIP_DEEP")) : -1;
  if (forced >= 0) return forced != 0;
  // sparse (foveated) frames: the kept rays are few and concentrated where the rays are long - the same floor (round 3: march 1.12 ->
  // 1.01 ms at the app's default focus); the host passes the previous frame's pixel count, the list's length is only known on the device
  if (p.sparse_xy) return p.world == 1 && p.sparse_hint_pixels > 0 && p.sparse_hint_pixels <= 64ull * OVR_DEEP_MAX_BLOCKS;
  // (the threshold was measured on the number of blocks a shard OWNS, launched or not: n_blocks_owned - n_schedule may be smaller since
  // round 4, when blocks none of whose rays hits the box are no longer launched)
  return p.world > 1 && p.n_blocks_owned <= (unsigned int)OVR_DEEP_MAX_BLOCKS;
}

template <int VT, int SHADE, int AM, bool SKIP>
inline hipError_t launch_vsbs(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  const size_t tf_lds = raymarch_lds_bytes(p.n_color, p.n_alpha);
  if (tf_lds == 0) return hipErrorInvalidValue;
  if (!p.sparse_xy && p.n_schedule > 0 && !p.schedule) return hipErrorInvalidValue;
  const dim3 grid = raymarch_grid(p), block(kBlock);
  hipError_t e;
  const bool pooled = (SHADE != 0) && p.pool.reqs != nullptr;
  if (!pooled) {
    const size_t lds = std::max<size_t>(tf_lds + table_lds_bytes(p, AM) + (size_t)kWaves * QCfg<SHADE, false>::QCAP * sizeof(ShadeReq), (size_t)kWaves * kNC * sizeof(unsigned int)); // the counter reduction reuses it
    bool launched = false;
    if constexpr (VT == VOX_F32 && SHADE == 0 && AM <= 1 && !SKIP) {
      if (p.lds_staging && !p.sparse_xy) { // LDS-staged bricks (see raymarch_kernel): the bricks follow the tables and the TF
        RayMarchParams q = p;
        q.lds_brick_offset = (unsigned int)((lds + 15) & ~(size_t)15);
        const size_t lds2 = q.lds_brick_offset + (size_t)kLdsBrickCap * 128 + sizeof(LdsRegion) + 12 * sizeof(float) + 2 * sizeof(int) + 16;
        auto kern = raymarch_kernel<VT, SHADE, AM, false, SKIP, true>;
        if ((e = set_lds(kern, lds2)) != hipSuccess) return e;
        if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds2, stream, q);
        launched = true;
      }
    }
    if (!launched) {
      auto kern = raymarch_kernel<VT, SHADE, AM, false, SKIP>;
      if ((e = set_lds(kern, lds)) != hipSuccess) return e;
      if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev && ev[1] && ev[2]) { (void)hipEventRecord(ev[1], stream); (void)hipEventRecord(ev[2], stream); }
    if (p.block_counters && p.counters) {
      if (!p.publish || p.zero_first)
        if ((e = hipMemsetAsync(p.counters, 0, 8 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
      if ((e = launch_reduce_counters(p.block_counters, (int)raymarch_grid_blocks(p), nullptr, 0, p.counters, nullptr, p.publish, p.reduce_done, stream)) != hipSuccess) return e;
    }
    return hipGetLastError();
  }
  if constexpr (SHADE != 0) { // (no pooled kernels are built for SHADE == 0: the in-place march above is its only pipeline)
    // ---- pooled pipeline: march -> shade -> composite, once per sample-per-pixel generation
    // (the last frame's final reduction left the control words and the counters zeroed - RayMarchParams::publish)
    const bool self_cleaning = p.publish && p.block_counters && p.counters && !p.zero_first;
    if (!self_cleaning) {
      if ((e = hipMemsetAsync(p.pool.ctrl, 0, (size_t)kPoolCtrlWords * sizeof(unsigned int), stream)) != hipSuccess) return e;
      if (p.block_counters && p.counters)
        if ((e = hipMemsetAsync(p.counters, 0, 8 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
    }
    RayMarchParams q = p;
    if (SHADE != 2) q.pool.order = nullptr; // no shadow rays: creation order (its tickets' batches, profiles/r02_notes.md section 11)
    for (int g = 0; g < p.spp; ++g) {
      q.spp_index = g;
      if (g > 0 && (e = hipMemsetAsync(p.pool.ctrl, 0, (size_t)32 * (kPoolSubs + 1) * sizeof(unsigned int), stream)) != hipSuccess) return e; // all but the frame's maximum
      {
        constexpr int SH = SHADE;
        const size_t lds = (size_t)kWaves * QCfg<SH, true>::QCAP * sizeof(ShadeReq) + table_lds_bytes(p, AM) + (size_t)p.n_alpha * sizeof(float) + 64;
        bool launched = false;
        if constexpr (!SKIP && (AM <= 1 || AM == 4)) {
          if (use_deep_rounds(p)) { // a small image shard: the longest ray's chain of rounds is the floor - deeper rounds
            auto kern = raymarch_kernel<VT, SH, AM, true, SKIP, false, true>;
            if ((e = set_lds(kern, lds)) != hipSuccess) return e;
            if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds, stream, q);
            launched = true;
          }
        }
        if (!launched) {
          auto kern = raymarch_kernel<VT, SH, AM, true, SKIP>;
          if ((e = set_lds(kern, lds)) != hipSuccess) return e;
          if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds, stream, q);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
      }
      if (ev && ev[1] && g == p.spp - 1) (void)hipEventRecord(ev[1], stream);
      if (q.pool.order && (e = launch_shade_order(q, stream)) != hipSuccess) return e; // runs sorted by light beam (PoolDesc)
      {
        const size_t lds = std::max<size_t>(tf_lds + table_lds_bytes(p, AM), 64);
        auto kern = shade_pool_kernel<VT, SHADE, AM, SKIP>;
        if ((e = set_lds(kern, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)shade_grid_blocks(p)), block, lds, stream, q);
        if ((e = hipGetLastError()) != hipSuccess) return e;
      }
      if (ev && ev[2] && g == p.spp - 1) (void)hipEventRecord(ev[2], stream);
      if (grid.x > 0 && (e = launch_composite(q, grid, stream)) != hipSuccess) return e;
      if (p.block_counters && p.counters)
        if ((e = launch_reduce_counters(p.block_counters, (int)raymarch_grid_blocks(p), (const unsigned int*)p.pool.shade_counters, shade_grid_blocks(p), p.counters,
                                        p.pool.ctrl, g == p.spp - 1 ? p.publish : nullptr, p.reduce_done, stream)) != hipSuccess) return e; // the frame's last generation publishes
    }
    return hipGetLastError();
  }
  else return hipErrorInvalidValue;
}

template <int VT, int SHADE, int AM>
inline hipError_t launch_vsb(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // empty-space skipping is a separate instantiation: the non-skipping kernels stay exactly as they are
  if (p.majorant) return launch_vsbs<VT, SHADE, AM, true>(p, stream, ev);
  return launch_vsbs<VT, SHADE, AM, false>(p, stream, ev);
}

template <int VT, int SHADE>
inline hipError_t launch_vs(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // addressing mode: 0 = 32-bit byte offsets (volume <= 4 GiB; largest byte offset = bytes - sizeof(voxel)),
  //                  1 = 32-bit element offsets (< 2^32 stored voxels), 2 = 64-bit z table in LDS,
  //                  3 = 64-bit, computed (axis tables would not fit in LDS next to the queues: a dimension beyond ~8000)
  int am = addressing_mode(p.vol, p.n_color, p.n_alpha);
  if (const char* f = getenv("OVR_HIP_ADDRESSING")) am = std::max(am, atoi(f)); // diagnostic: a more general mode than needed (tests)
  if (am < 3 && (!p.vol.axis_ab || !p.vol.axis_z)) return hipErrorInvalidValue; // the layout's offset tables (launch_axis_tables)
  if constexpr (sizeof(typename Vox<VT>::T) <= 2 && !Vox<VT>::kQuad && OVR_ROW_LOADS) {
    // mode 4 = mode 0 with the 16-bit pairs read as aligned 8-byte rows (RowLoads): layouts the caches do not serve; OVR_HIP_ROW_LOADS=0|1, read when a renderer is created, forces (tests, measurements)
    const int forced = p.row_loads - 1; // RayMarchParams::row_loads: 0 = by size, 1 = never, 2 = always
    if (am == 0 && (forced >= 0 ? forced != 0 : p.vol.bytes > (128ull << 20))) return launch_vsb<VT, SHADE, 4>(p, stream, ev);
  }
  switch (am) {
  case 0: return launch_vsb<VT, SHADE, 0>(p, stream, ev);
  case 1: return launch_vsb<VT, SHADE, 1>(p, stream, ev);
  case 2: return launch_vsb<VT, SHADE, 2>(p, stream, ev);
  default: return launch_vsb<VT, SHADE, 3>(p, stream, ev);
  }
}

// one explicit instantiation per voxel type, each in its own translation unit (ovr_hip_march_*.hip): the ~60 kernel variants of
// a type compile in parallel with the other types
template <int VT>
hipError_t launch_v(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  switch (p.shading) {
  case 0: return launch_vs<VT, 0>(p, stream, ev);
  case 1: return launch_vs<VT, 1>(p, stream, ev);
  default: return launch_vs<VT, 2>(p, stream, ev);
  }
}


} // namespace ovrhip
