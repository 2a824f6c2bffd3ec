// ovr_hip_kernels.hip - gfx950 (MI355X, CDNA4) kernels of the OVR ray-marching path.
//
// Hand-written for wave64 / 256 CUs / 160 KiB LDS; no CUDA dual path.  Semantics follow the reference's in-tree ray
// marcher (citations relative to the reference tree):
//   raygen + accumulation ........ ovr/devices/optix7/shaders_raymarching.cu:323-413
//   box test ..................... ovr/devices/optix7/shaders_common.h:156-184,379-392
//   marching loop ................ shaders_raymarching.cu:87-171   shadow march :44-85,205-229
//   volume tap / gradient / TF ... shaders_common.h:186-215,311-319,356-367
//   TEA RNG ...................... ovr/common/random/random.h:146-188
//   sparse-sampling mask ......... ovr/common/generate_mask.cu:55-120, ovr/common/random/blue_noise.h:81-102
//
// Kernel shape: FOUR LANES PER RAY - the 4 lanes of a quad are 4 consecutive steps of one ray - so a wave64 marches 16 rays
// (a 4x4-pixel tile) and a workgroup of four waves an 8x8-pixel block; workgroups run longest rays first (schedule_kernel).
// Shaded samples become 32-byte requests that a second, persistent kernel shades from a global pool and a third kernel
// composites in ray order (pooled pipeline), or that the tile's own wave shades (in-place pipeline).  The transfer
// function (colour float4 table + alpha table, 20 KiB at the shipped resolution of 1024), the per-axis brick-offset tables,
// the request queues and the blue-noise jitter of the block's pixels live in LDS; the volume is read from the bricked layout
// described in ovr_hip_kernels.h (one 128-byte line = one 3-D brick).  Built with -ffp-contract=off: every fused
// multiply-add below is explicit so the operation order is the one the CPU oracle (oracle/ovr_oracle.c) restates.
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
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); } // NaN -> 0 like CUDA fminf/fmaxf
__device__ __forceinline__ float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
// gdt normalize (extern/gdt/gdt/math/vec.h:443-448) with the hardware reciprocal square root (1 ulp)
__device__ __forceinline__ f3 normalize3(f3 v)
{
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
template <bool BF>
__device__ __forceinline__ float opacity_correction(float a, float adj)
{
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
};
template <> struct Vox<VOX_U16> {
  typedef unsigned short T; typedef u16x2_u P;
  static constexpr int cx = 3, mbx = 10, by = 2, bz = 2;
  static constexpr bool kScale = false, kClamp = false; // u16 is sampled as RAW float (array.cpp:335-338)
};
template <> struct Vox<VOX_I16> {
  typedef short T; typedef i16x2_u P;
  static constexpr int cx = 3, mbx = 10, by = 2, bz = 2;
  static constexpr bool kScale = false, kClamp = false;
};
template <> struct Vox<VOX_U8> {
  typedef unsigned char T; typedef u8x2_u P;
  static constexpr int cx = 7, mbx = 4, by = 2, bz = 2;
  static constexpr bool kScale = true, kClamp = false; // normalized read: v / 255 (array.cpp:304-306)
};
template <> struct Vox<VOX_I8> {
  typedef signed char T; typedef i8x2_u P;
  static constexpr int cx = 7, mbx = 4, by = 2, bz = 2;
  static constexpr bool kScale = true, kClamp = true; // max(v / 127, -1)
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
  static __host__ __device__ __forceinline__ unsigned X(unsigned x) // offset of voxel x as the LOWER member of a pair
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

struct VolConsts {
  const void* data;
  // per-axis offset tables in LDS (AM 0 / 1): tx[x] = X(x), ty[y] = Y(y), tz[z] = Z(z), bytes (AM 0) or elements (AM 1);
  // ty / tz hold one extra entry equal to the last one, so (i, i + 1) is clamp-to-edge without a select
  const unsigned int *tab_x, *tab_y, *tab_z;
  const unsigned long long* tab_z64; // AM 2: z offsets need 64 bits (>= 2^32 stored voxels)
  int nx1, ny1, nz1; // n - 1
  const float* majorant; // per-macrocell max TF opacity (null: empty-space skipping off)
  const unsigned char* occupancy; // per 4^3 macrocells: 1 if one of them, or a macrocell next to them, has majorant > 0
  int mcx1, mcy1, mcz1;  // macrocell grid dims - 1
  unsigned int macro_y;          // stored elements between macro rows: MV * macros_x
  unsigned long long macro_z;    // stored elements between macro layers: MV * macros_x * macros_y
  float fx1, fy1, fz1;
  f3 cs, cb;
  float vscale, vmin;
};

__device__ __forceinline__ void axis_tap(float po, float cs, float cb, float fn1, int n1, int& i0, int& i1, float& f)
{
  const float p = clamp01(po);                       // sample_volume_object_space clamps p to [0,1]
  float x = fmaf(p, cs, cb);                         // cell-centred: p*N - 0.5
  x = fminf(fmaxf(x, 0.f), fn1);                     // == clamp-to-edge addressing: taps (i0, min(i0+1, n-1))
  const float fl = floorf(x);
  f = x - fl;
  i0 = (int)fl;
  i1 = min(i0 + 1, n1);
}

// One trilinear tap, split in two so that several taps can be in flight before the first is consumed
// (software pipelining: the march is latency-bound otherwise).  issue: 4 pair loads; finish: 7 lerps.
struct Tap {
  float c000, c100, c010, c110, c001, c101, c011, c111;
  float fx, fy, fz;
  int x0, y0, z0; // lower corner of the footprint (live only between tap_coords and tap_loads)
};

// macrocell (16^3 voxels, reference accel/spatial_partition.h:24) whose value range covers the footprint (i0, i0 + 1) on
// every axis: cell c holds voxels [16c - 1, 16c + 15] (sp_singlemc.cu:36-42), i.e. c = (i0 + 1) >> 4

__device__ __forceinline__ void tap_coords(const VolConsts& vc, f3 p, Tap& t)
{
  int x1, y1, z1;
  axis_tap(p.x, vc.cs.x, vc.cb.x, vc.fx1, vc.nx1, t.x0, x1, t.fx);
  axis_tap(p.y, vc.cs.y, vc.cb.y, vc.fy1, vc.ny1, t.y0, y1, t.fy);
  axis_tap(p.z, vc.cs.z, vc.cb.z, vc.fz1, vc.nz1, t.z0, z1, t.fz);
  (void)x1; (void)y1; (void)z1;
}

__device__ __forceinline__ unsigned int tap_cell(const VolConsts& vc, const Tap& t)
{
  const int cx = min((t.x0 + 1) >> 4, vc.mcx1), cy = min((t.y0 + 1) >> 4, vc.mcy1), cz = min((t.z0 + 1) >> 4, vc.mcz1);
  return (unsigned int)cx + (unsigned int)(vc.mcx1 + 1) * ((unsigned int)cy + (unsigned int)(vc.mcy1 + 1) * (unsigned int)cz);
}

template <int VT, int AM>
__device__ __forceinline__ void tap_loads(const VolConsts& vc, Tap& t)
{
  typedef BrickMap<VT> M;
  typedef typename Vox<VT>::T T;
  typedef typename Vox<VT>::P P;
  const int x0 = t.x0, y0 = t.y0, z0 = t.z0;
  const int y1 = min(y0 + 1, vc.ny1), z1 = min(z0 + 1, vc.nz1);
  P p00, p10, p01, p11;
  if (AM == 3) { // > 2^32 elements and axis tables too large for LDS: 64-bit element offsets, computed arithmetically
    const unsigned ox = M::X((unsigned)x0);
    const unsigned o0 = ox + M::Y((unsigned)y0, vc.macro_y), o1 = ox + M::Y((unsigned)y1, vc.macro_y);
    const T* base = static_cast<const T*>(vc.data);
    const unsigned long long oz0 = (unsigned long long)M::Zlo((unsigned)z0) + (unsigned long long)((unsigned)z0 >> 5) * vc.macro_z;
    const unsigned long long oz1 = (unsigned long long)M::Zlo((unsigned)z1) + (unsigned long long)((unsigned)z1 >> 5) * vc.macro_z;
    p00 = *reinterpret_cast<const P*>(base + (oz0 + o0)); p10 = *reinterpret_cast<const P*>(base + (oz0 + o1));
    p01 = *reinterpret_cast<const P*>(base + (oz1 + o0)); p11 = *reinterpret_cast<const P*>(base + (oz1 + o1));
  }
  else if (AM == 2) { // > 2^32 elements: x and y offsets (inside one macro layer) stay 32-bit, the z table is 64-bit
    (void)y1; (void)z1;
    const unsigned ox = vc.tab_x[x0];
    const unsigned o0 = ox + vc.tab_y[y0], o1 = ox + vc.tab_y[y0 + 1];
    const unsigned long long oz0 = vc.tab_z64[z0], oz1 = vc.tab_z64[z0 + 1];
    const T* base = static_cast<const T*>(vc.data);
    p00 = *reinterpret_cast<const P*>(base + (oz0 + o0)); p10 = *reinterpret_cast<const P*>(base + (oz0 + o1));
    p01 = *reinterpret_cast<const P*>(base + (oz1 + o0)); p11 = *reinterpret_cast<const P*>(base + (oz1 + o1));
  }
  else {
    // three LDS lookups (one b32 + two adjacent pairs) replace ~40 bit-field / multiply instructions per tap: the march is
    // VALU-bound once its gathers coalesce, and the LDS pipe is otherwise nearly idle
    (void)y1; (void)z1;
    const unsigned ox = vc.tab_x[x0];
    const unsigned oy0 = vc.tab_y[y0], oy1 = vc.tab_y[y0 + 1];
    const unsigned oz0 = vc.tab_z[z0], oz1 = vc.tab_z[z0 + 1];
    const unsigned o0 = ox + oy0, o1 = ox + oy1;
    if (AM == 1) { // < 2^32 elements: 32-bit element offsets, one 64-bit shift-add per load
      const T* base = static_cast<const T*>(vc.data);
      p00 = *reinterpret_cast<const P*>(base + (oz0 + o0)); p10 = *reinterpret_cast<const P*>(base + (oz0 + o1));
      p01 = *reinterpret_cast<const P*>(base + (oz1 + o0)); p11 = *reinterpret_cast<const P*>(base + (oz1 + o1));
    }
    else { // the whole volume is <= 4 GiB: 32-bit BYTE offsets, the loads use the SGPR-base + 32-bit-VGPR-offset form
      const char* cb = static_cast<const char*>(vc.data);
      p00 = *reinterpret_cast<const P*>(cb + (oz0 + o0)); p10 = *reinterpret_cast<const P*>(cb + (oz0 + o1));
      p01 = *reinterpret_cast<const P*>(cb + (oz1 + o0)); p11 = *reinterpret_cast<const P*>(cb + (oz1 + o1));
    }
  }
  t.c000 = (float)p00.x; t.c100 = (float)p00.y; t.c010 = (float)p10.x; t.c110 = (float)p10.y;
  t.c001 = (float)p01.x; t.c101 = (float)p01.y; t.c011 = (float)p11.x; t.c111 = (float)p11.y;
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
  if (Vox<VT>::kClamp) {
    t.c000 = fmaxf(t.c000, vc.vmin); t.c100 = fmaxf(t.c100, vc.vmin); t.c010 = fmaxf(t.c010, vc.vmin); t.c110 = fmaxf(t.c110, vc.vmin);
    t.c001 = fmaxf(t.c001, vc.vmin); t.c101 = fmaxf(t.c101, vc.vmin); t.c011 = fmaxf(t.c011, vc.vmin); t.c111 = fmaxf(t.c111, vc.vmin);
  }
  const float c00 = lerpf(t.c000, t.c100, t.fx), c10 = lerpf(t.c010, t.c110, t.fx);
  const float c01 = lerpf(t.c001, t.c101, t.fx), c11 = lerpf(t.c011, t.c111, t.fx);
  const float c0 = lerpf(c00, c10, t.fy), c1 = lerpf(c01, c11, t.fy);
  float s = lerpf(c0, c1, t.fz);
  if (Vox<VT>::kScale) s *= vc.vscale;
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
__device__ __forceinline__ float tf_alpha(const TfConsts& tf, float v)
{
  const float x = v * tf.fna1;
  const float fl = floorf(x);
  const int i0 = (int)fl, i1 = min(i0 + 1, tf.na1);
  return lerpf(tf.alpha[i0], tf.alpha[i1], x - fl);
}
__device__ __forceinline__ f3 tf_color(const TfConsts& tf, float v)
{
  const float x = v * tf.fnc1;
  const float fl = floorf(x);
  const int i0 = (int)fl, i1 = min(i0 + 1, tf.nc1);
  const float f = x - fl;
  const float4 a = tf.color[i0], b = tf.color[i1];
  return mk3(lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f));
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

// Empty-space skipping, per ray: the t interval outside of which every sample lies in a macrocell with majorant 0.
// skip_walk: one lane walks the ray's [ta, tb] through the coarse occupancy grid (3-D DDA; accel/dda.h is the reference's
// walker for its path tracer) and widens [first, last] by the entry / exit of every set entry it crosses.
// Sample coordinates: x = p * cs + cb (tap_coords), macrocell = (floor(x) + 1) >> 4 (tap_cell), i.e. the regular 16-voxel
// grid in w = x + 1; a coarse entry is 64 voxels of w.
__device__ __forceinline__ void skip_walk(const VolConsts& vc, f3 oo, f3 od, float ta, float tb, float& first, float& last)
{
  const float w0[3] = { fmaf(oo.x, vc.cs.x, vc.cb.x + 1.f), fmaf(oo.y, vc.cs.y, vc.cb.y + 1.f), fmaf(oo.z, vc.cs.z, vc.cb.z + 1.f) };
  const float dw[3] = { od.x * vc.cs.x, od.y * vc.cs.y, od.z * vc.cs.z };
  const int m1[3] = { vc.mcx1 >> 2, vc.mcy1 >> 2, vc.mcz1 >> 2 }; // coarse grid dims - 1
  constexpr float G = 64.f;
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
    const bool occ = vc.occupancy[(size_t)ci[0] + (size_t)(m1[0] + 1) * ((size_t)ci[1] + (size_t)(m1[1] + 1) * (size_t)ci[2])] != 0;
    const int ax = (tmax[0] <= tmax[1]) ? (tmax[0] <= tmax[2] ? 0 : 2) : (tmax[1] <= tmax[2] ? 1 : 2);
    const float tn = ax == 0 ? tmax[0] : ax == 1 ? tmax[1] : tmax[2];
    if (occ) { first = fminf(first, t); last = fmaxf(last, fminf(tn, tb)); }
    t = fmaxf(t, tn);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k == ax) {
        const int nxt = ci[k] + st[k];
        if (nxt < 0 || nxt > m1[k]) tmax[k] = FLT_MAX; // the clamped coordinate stays in the border entry
        else { ci[k] = nxt; tmax[k] += tdel[k]; }
      }
  }
  if (t < tb) { first = fminf(first, t); last = tb; } // safety limit hit (never expected): treat the rest as occupied
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
  if (SKIP) skip_walk(vc, oo, od, t0, t1, skip_first, skip_last);
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
      a = opacity_correction<SKIP>(a, mc.base * dts[k]);
      if (SKIP) a = mj[k] > 0.f ? a : 0.f; // a macrocell whose majorant is 0 holds no sample with opacity > 0
      live = live && valid[k] && (alpha < 0.9999f);
      alpha = live ? fmaf(1.f - alpha, a, alpha) : alpha;
      n_shadow += (live && (!SKIP || mj[k] > 0.f)) ? 1u : 0u;
      if (SKIP) n_shadow_skipped += (live && !(mj[k] > 0.f)) ? 1u : 0u;
    }
  }
  return alpha;
}

// ------------------------------------------------------------------------------------------------------------------
// the ray-march kernels
//
// Work decomposition: four lanes per ray (a quad = 4 consecutive steps), 16 rays = a 4x4 pixel tile per wave64, four waves
// = 8x8 pixels per workgroup; workgroups are launched longest rays first (schedule_kernel).  Details at raymarch_kernel.
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

__device__ __forceinline__ void setup_consts(const RayMarchParams& P, VolConsts& vc, MarchConsts& mc)
{
  vc.data = P.vol.data;
  vc.nx1 = P.vol.nx - 1; vc.ny1 = P.vol.ny - 1; vc.nz1 = P.vol.nz - 1;
  vc.fx1 = (float)vc.nx1; vc.fy1 = (float)vc.ny1; vc.fz1 = (float)vc.nz1;
  vc.macro_y = P.vol.macro_elems * (unsigned int)P.vol.macros_x;
  vc.macro_z = (unsigned long long)P.vol.macro_elems * (unsigned long long)P.vol.macros_x * (unsigned long long)P.vol.macros_y;
  vc.cs = ld3(P.coord_scale); vc.cb = ld3(P.coord_bias);
  vc.vscale = P.vol.value_scale; vc.vmin = P.vol.value_min_clamp;
  vc.majorant = P.majorant;
  vc.occupancy = P.occupancy;
  vc.mcx1 = (P.vol.nx + 15) / 16 - 1; vc.mcy1 = (P.vol.ny + 15) / 16 - 1; vc.mcz1 = (P.vol.nz + 15) / 16 - 1;
  mc.inv_scale = ld3(P.inv_scale); mc.wto_p = ld3(P.wto_p); mc.otw_it = ld3(P.otw_it); mc.light = ld3(P.light);
  mc.gstep = ld3(P.grad_step);
  mc.ginv = mk3(1.f / mc.gstep.x, 1.f / mc.gstep.y, 1.f / mc.gstep.z);
  mc.step = P.step; mc.base = P.base; mc.shadow_stride = P.shadow_stride;
}

// build the per-axis offset tables in LDS (all threads of the workgroup); returns the bytes used
template <int VT, int AM>
__device__ __forceinline__ size_t stage_tables(const RayMarchParams& P, unsigned char* base, VolConsts& vc)
{
  typedef BrickMap<VT> M;
  vc.tab_x = vc.tab_y = vc.tab_z = nullptr;
  vc.tab_z64 = nullptr;
  if (AM == 3) return 0;
  if (AM == 2) { // [z: nz + 1 x u64][x: nx x u32][y: ny + 1 x u32], element offsets
    unsigned long long* tz = reinterpret_cast<unsigned long long*>(base);
    unsigned int* tx = reinterpret_cast<unsigned int*>(tz + P.vol.nz + 1);
    unsigned int* ty = tx + P.vol.nx;
    for (int i = threadIdx.x; i < P.vol.nx; i += kBlock) tx[i] = M::X((unsigned)i);
    for (int i = threadIdx.x; i <= P.vol.ny; i += kBlock) ty[i] = M::Y((unsigned)min(i, P.vol.ny - 1), vc.macro_y);
    for (int i = threadIdx.x; i <= P.vol.nz; i += kBlock) {
      const unsigned z = (unsigned)min(i, P.vol.nz - 1);
      tz[i] = (unsigned long long)M::Zlo(z) + (unsigned long long)(z >> 5) * vc.macro_z;
    }
    vc.tab_x = tx; vc.tab_y = ty; vc.tab_z64 = tz;
    return (size_t)(P.vol.nz + 1) * sizeof(unsigned long long) + (size_t)(P.vol.nx + P.vol.ny + 1) * sizeof(unsigned int);
  }
  unsigned int* tx = reinterpret_cast<unsigned int*>(base);
  unsigned int* ty = tx + P.vol.nx;
  unsigned int* tz = ty + P.vol.ny + 1;
  const unsigned int mul = AM == 0 ? (unsigned int)sizeof(typename Vox<VT>::T) : 1u;
  for (int i = threadIdx.x; i < P.vol.nx; i += kBlock) tx[i] = M::X((unsigned)i) * mul;
  for (int i = threadIdx.x; i <= P.vol.ny; i += kBlock) ty[i] = M::Y((unsigned)min(i, P.vol.ny - 1), vc.macro_y) * mul;
  for (int i = threadIdx.x; i <= P.vol.nz; i += kBlock) {
    const unsigned z = (unsigned)min(i, P.vol.nz - 1);
    tz[i] = (M::Zlo(z) + (z >> 5) * (unsigned)vc.macro_z) * mul;
  }
  vc.tab_x = tx; vc.tab_y = ty; vc.tab_z = tz;
  return (size_t)(P.vol.nx + P.vol.ny + P.vol.nz + 2) * sizeof(unsigned int);
}
__host__ inline size_t table_lds_bytes(const RayMarchParams& p, int am)
{
  if (am == 3) return 0;
  if (am == 2) return ((size_t)(p.vol.nz + 1) * sizeof(unsigned long long) + (size_t)(p.vol.nx + p.vol.ny + 1) * sizeof(unsigned int) + 15) & ~(size_t)15;
  return ((size_t)(p.vol.nx + p.vol.ny + p.vol.nz + 2) * sizeof(unsigned int) + 15) & ~(size_t)15;
}

// stage the transfer function in LDS (all threads of the workgroup); color may be skipped by alpha-only kernels
__device__ __forceinline__ void stage_tf(const RayMarchParams& P, unsigned char* tf_base, bool with_color, TfConsts& tf)
{
  float4* lc = reinterpret_cast<float4*>(tf_base);
  float* la = reinterpret_cast<float*>(tf_base + (with_color ? (size_t)P.n_color * sizeof(float4) : 0));
  if (with_color) {
    const float4* gc = reinterpret_cast<const float4*>(P.tf_color);
    for (int i = threadIdx.x; i < P.n_color; i += kBlock) lc[i] = gc[i];
  }
  for (int i = threadIdx.x; i < P.n_alpha; i += kBlock) la[i] = P.tf_alpha[i];
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
__device__ __forceinline__ void shade_request(const RayMarchParams& P, const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, ShadeReq& r,
                                              unsigned int& n_shadow, unsigned int& n_shadow_skipped)
{
  const f3 pos = mk3(r.px, r.py, r.pz);
  const f3 po = to_object(mc, pos);
  // one-sided differences, flipped at the upper bound; the three taps are issued together
  const bool flx = (po.x + mc.gstep.x) > 1.f, fly = (po.y + mc.gstep.y) > 1.f, flz = (po.z + mc.gstep.z) > 1.f;
  Tap tgx, tgy, tgz;
  tap_issue<VT, AM>(vc, mk3(po.x + (flx ? -mc.gstep.x : mc.gstep.x), po.y, po.z), tgx);
  tap_issue<VT, AM>(vc, mk3(po.x, po.y + (fly ? -mc.gstep.y : mc.gstep.y), po.z), tgy);
  tap_issue<VT, AM>(vc, mk3(po.x, po.y, po.z + (flz ? -mc.gstep.z : mc.gstep.z)), tgz);
  const f3 rgb = tf_color(tf, r.v);
  f3 g;
  g.x = (tap_finish<VT>(vc, tgx) - r.s) * (flx ? -mc.ginv.x : mc.ginv.x);
  g.y = (tap_finish<VT>(vc, tgy) - r.s) * (fly ? -mc.ginv.y : mc.ginv.y);
  g.z = (tap_finish<VT>(vc, tgz) - r.s) * (flz ? -mc.ginv.z : mc.ginv.z);
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
  if (SHADE == 2) shadow = march_shadow<VT, AM, kShadowTaps, SKIP>(vc, tf, mc, pos, n_shadow, n_shadow_skipped);
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

constexpr int kNC = 7; // counters: rays, samples, shaded, shadow, active pixels, skipped samples, skipped shadow samples
// per-wave counters -> LDS -> one plain store of the workgroup's partial sums (lds must hold kWaves*kNC uints)
__device__ __forceinline__ void store_block_counters(const RayMarchParams& P, unsigned int* red, int lane, int wave, unsigned int n_rays,
                                                     unsigned int n_samples, unsigned int n_shaded, unsigned int n_shadow, unsigned int n_active,
                                                     unsigned int n_skipped, unsigned int n_shadow_skipped)
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
  }
  if (!P.block_counters) return;
  __syncthreads(); // LDS is dead at this point: reuse its front
  if (lane == 0) {
    red[wave * kNC + 0] = n_rays; red[wave * kNC + 1] = n_samples; red[wave * kNC + 2] = n_shaded;
    red[wave * kNC + 3] = n_shadow; red[wave * kNC + 4] = n_active; red[wave * kNC + 5] = n_skipped; red[wave * kNC + 6] = n_shadow_skipped;
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
__device__ __forceinline__ float sel4(float a0, float a1, float a2, float a3, int sub) { return sub == 0 ? a0 : sub == 1 ? a1 : sub == 2 ? a2 : a3; }

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
// longest rays first (schedule_kernel) - compute_screen_position of the reference (shaders_common.h:394-451) is the
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

// the primary rays' form: the 4 lanes of the quad each walk a quarter of [t0, t1]; min / max over the quad
__device__ __forceinline__ void skip_interval(const VolConsts& vc, f3 oo, f3 od, float t0, float t1, int sub, bool live, float& t_first, float& t_last)
{
  float first = FLT_MAX, last = -FLT_MAX;
  if (live) {
    const float len = t1 - t0;
    const float ta = fmaf((float)sub * 0.25f, len, t0), tb = sub == 3 ? t1 : fmaf((float)(sub + 1) * 0.25f, len, t0);
    skip_walk(vc, oo, od, ta, tb, first, last);
  }
  t_first = fminf(fminf(quad_bcast<0>(first), quad_bcast<1>(first)), fminf(quad_bcast<2>(first), quad_bcast<3>(first)));
  t_last = fmaxf(fmaxf(quad_bcast<0>(last), quad_bcast<1>(last)), fmaxf(quad_bcast<2>(last), quad_bcast<3>(last)));
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
template <int VT, int SHADE, int AM, bool POOLED, bool SKIP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, OVR_MARCH_WPE))) void raymarch_kernel(const RayMarchParams P)
{
  using Cfg = QCfg<SHADE, POOLED>;
  constexpr int K = Cfg::K;
  constexpr int QCAP = Cfg::QCAP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 3;               // this lane's step inside each group of 4 consecutive steps
  const int qbase = lane & ~3;            // first lane of the quad
  const bool owner = sub == 0;            // the quad's lane that keeps the pixel's colour / request list
  const unsigned long long t_start = P.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

  int ix, iy;
  const bool active = assign_pixel_quad(P, lane, wave, ix, iy);
  unsigned int n_rays = 0, n_samples = 0, n_shaded = 0, n_shadow = 0, n_skipped = 0, n_shadow_skipped = 0;
  VolConsts vc;
  MarchConsts mc;
  setup_consts(P, vc, mc);

  // ---- LDS carve: [request queues][offset tables][TF colour (not needed by the pooled march)][TF alpha]
  // A workgroup none of whose rays can hit the volume (most of the image outside the silhouette) stages nothing; with
  // empty-space skipping, neither does one whose rays only cross empty macrocells.
  ShadeReq* const queue = reinterpret_cast<ShadeReq*>(lds_raw) + (size_t)wave * (QCAP > 0 ? QCAP : 1);
  TfConsts tf;
  float pro_first = FLT_MAX, pro_last = -FLT_MAX; // skipping, spp == 1: the ray's skip interval, found here once
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
        skip_interval(vc, oo0, od0, a0, b0, sub, true, pro_first, pro_last);
        need = pro_first <= pro_last;
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
  // pooled: the tile's current reservation of kRun consecutive chunks
  unsigned int run_base = 0, run_left = 0;
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
      unsigned int c0 = 0;
      if (lane == 0) c0 = atomicAdd(&Q.ctrl[0], (unsigned int)kRun);
      run_base = (unsigned int)__builtin_amdgcn_readfirstlane((int)c0);
      run_left = kRun;
    }
    const unsigned int c = run_base + (kRun - run_left);
    --run_left;
    if (run_base + kRun <= Q.capacity) {
      __builtin_amdgcn_wave_barrier();
      if ((unsigned int)lane < n) Q.reqs[(size_t)c * 64 + lane] = queue[(q_head + lane) & (QCAP - 1)];
      if (lane == 0) {
        Q.chunk_n[c] = n;
        if (prev_chunk >= 0) Q.chunk_next[prev_chunk] = (int)c; else Q.tile_first[tile] = (int)c;
      }
      if (pend > 0 && (last - q_head) < n) last_gidx = c * 64u + (last - q_head);
      prev_chunk = (int)c;
    }
    // beyond capacity: the pool is exhausted; ctrl[0] keeps counting so the host knows how much was needed, re-sizes
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
    float skip_first = -FLT_MAX, skip_last = FLT_MAX; // samples outside [skip_first, skip_last] are in empty macrocells
    if (SKIP) {
      if (P.spp == 1) { skip_first = pro_first; skip_last = pro_last; } // same ray, same [t0, t1] as in the prologue
      else skip_interval(vc, oo, od, t0, t1, sub, live, skip_first, skip_last);
    }
    // `staged` guards the prologue's decision (it tests the same ray with the same expressions): without the tables and the TF
    // in LDS no sample may be fetched - a skipping ray then only counts its (all empty) steps, any other ray is dead
    if (SKIP) { if (!staged) { skip_first = FLT_MAX; skip_last = -FLT_MAX; } }
    else live = live && staged;
    float tx = t0, ty = fminf(t1, t0 + mc.step);
    pend = 0;

    for (;;) {
      const bool any_live = __ballot(live) != 0ull;
      // ---- (1) in place: shade queued requests, a full batch whenever 64 are queued, the remainder once no ray is live
      if (SHADE != 0 && !POOLED) {
        while ((q_tail - q_head) >= 64u || (!any_live && q_tail != q_head)) {
          const unsigned int n = min(q_tail - q_head, 64u);
          __builtin_amdgcn_wave_barrier(); // requests were written by other lanes of this wave (LDS ops are in order)
          ShadeReq r;
          r.px = r.py = r.pz = r.s = r.v = r.tr = r.a = 0.f; r.next = 0;
          if ((unsigned int)lane < n) {
            r = queue[(q_head + lane) & (QCAP - 1)];
            if (r.a > 0.f) shade_request<VT, SHADE, AM, SKIP>(P, vc, tf, mc, r, n_shadow, n_shadow_skipped); // a == 0: null request
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
        const bool outside = (ntx < skip_first) || (tx > skip_last);
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
        const float mtx = sel4(txq[0], txq[1], txq[2], txq[3], sub);
        const float mty = sel4(tyq[0], tyq[1], tyq[2], tyq[3], sub);
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
        for (int k = 0; k < K; ++k) inside = inside || (tms[k] >= skip_first && tms[k] <= skip_last);
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
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (!SKIP || mj[k] > 0.f) tap_loads<VT, AM>(vc, taps[k]);
      // ---- (3) own samples: value, TF coordinate, corrected opacity (and colour when shading is off)
      float sa[K], va[K], aa[K];
      f3 ca[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        sa[k] = tap_finish<VT>(vc, taps[k]);
        va[k] = tf_coord(tf, sa[k]);
        aa[k] = opacity_correction<false>(tf_alpha(tf, va[k]), mc.base * dts[k]);
        if (SKIP) aa[k] = mj[k] > 0.f ? aa[k] : 0.f; // a macrocell whose majorant is 0 holds no sample with opacity > 0
        if (SHADE == 0) {
          const f3 rgb = tf_color(tf, va[k]);
          ca[k] = mk3(clamp01(rgb.x), clamp01(rgb.y), clamp01(rgb.z));
        }
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
        mtr[k] = 1.f - sel4(al[0], al[1], al[2], al[3], sub);
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
    if (lane == 0 && run_base + kRun <= Q.capacity)
      for (unsigned int i = kRun - run_left; i < (unsigned int)kRun && run_left != 0; ++i) Q.chunk_n[run_base + i] = 0; // unused tail of the reservation
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
  store_block_counters(P, reinterpret_cast<unsigned int*>(lds_raw), lane, wave, n_rays, n_samples, n_shaded, n_shadow,
                       (active && owner && (!POOLED || P.spp_index == 0)) ? 1u : 0u, n_skipped,
                       n_shadow_skipped);
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
  const unsigned int n_runs = Q.ctrl[0] > Q.capacity ? 0u : Q.ctrl[0] / (unsigned int)kRun; // overflow: the frame is re-rendered
  unsigned int n_shadow = 0, n_shadow_skipped = 0;
  __shared__ unsigned int s_run;
  for (;;) {
    // one returning atomic per workgroup and run; the 4 waves shade the run's chunks (consecutive depth steps of ONE
    // tile: their gradient and shadow taps fall into the same bricks, which the CU's L1 and the XCD's L2 now keep)
    __syncthreads();
    if (threadIdx.x == 0) s_run = atomicAdd(&Q.ctrl[1], 1u);
    __syncthreads();
    const unsigned int run = s_run;
    if (run >= n_runs) break; // every workgroup reaches this: the cursor only grows
    for (unsigned int i = (unsigned int)wave; i < (unsigned int)kRun; i += kWaves) {
      const unsigned int c = run * kRun + i;
      const unsigned int n = Q.chunk_n[c];
      if ((unsigned int)lane < n) {
        ShadeReq r = Q.reqs[(size_t)c * 64 + lane];
        if (r.a > 0.f) { // a == 0: null request (a step of the quad that needs no shading)
          shade_request<VT, SHADE, AM, SKIP>(P, vc, tf, mc, r, n_shadow, n_shadow_skipped);
          Q.reqs[(size_t)c * 64 + lane] = r;
        }
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

// ------------------------------------------------------------------------------------------------------------------
// pooled pipeline, kernel C: every tile walks its chunks in order, composites and writes its pixels
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void composite_kernel(const RayMarchParams P)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ix, iy;
  const bool active = assign_pixel_quad(P, lane, wave, ix, iy) && (lane & 3) == 0; // one owner lane per ray, as in the march
  const PoolDesc& Q = P.pool;
  if (Q.ctrl[0] > Q.capacity) return; // pool overflow: the host re-renders this frame with a larger pool, nothing may be written
  const unsigned int tile = blockIdx.x * kWaves + wave;
  const unsigned int pixel_index = (unsigned int)ix + (unsigned int)iy * (unsigned int)P.width;
  float alpha = 0.f;
  unsigned int first = 0;
  int pend = 0;
  if (active) {
    const float4 st = Q.pix_state[pixel_index];
    alpha = st.x; first = __float_as_uint(st.y); pend = __float_as_int(st.z);
  }
  f3 color = mk3(0, 0, 0), gradient = mk3(0, 0, 0);
  // The tile's chunks come in runs of kRun consecutive, kRun-aligned pool slots (the march reserves them that way), so
  // inside a run the next chunk is c + 1 and only the run-to-run link (chunk_next of the run's last slot) is chased - it
  // is fetched when the run is entered, a whole run ahead of its use.  The next chunk's requests are loaded before the
  // current ones are applied, so the walk no longer pays two dependent memory latencies per chunk.
  const unsigned int total = Q.tile_count[tile];
  auto load_chunk = [&](int c, unsigned int n) {
    ShadeReq r;
    r.px = r.py = r.pz = r.s = r.v = r.tr = r.a = 0.f; r.next = 0;
    if ((unsigned int)lane < n) r = Q.reqs[(size_t)c * 64 + lane];
    return r;
  };
  int c = total > 0u ? Q.tile_first[tile] : 0;
  int link = total > 0u ? Q.chunk_next[c | (kRun - 1)] : 0;
  ShadeReq r = load_chunk(c, min(total, 64u));
  for (unsigned int base = 0; base < total; base += 64u) {
    const unsigned int n = min(total - base, 64u);
    const bool more = base + 64u < total;
    const bool run_end = (c & (kRun - 1)) == kRun - 1;
    const int c_next = run_end ? link : c + 1;
    ShadeReq r_next = r;
    if (more) {
      r_next = load_chunk(c_next, min(total - base - 64u, 64u));
      if (run_end) link = Q.chunk_next[c_next | (kRun - 1)];
    }
    apply_batch(r, base, n, lane, pend, first, color, gradient);
    r = r_next;
    c = c_next;
  }
  if (active) {
    f3 o_c = mk3(0, 0, 0), o_g = mk3(0, 0, 0);
    if (alpha > 0.f) {
      o_c = mk3(color.x / alpha, color.y / alpha, color.z / alpha);
      o_g = mk3(gradient.x / alpha, gradient.y / alpha, gradient.z / alpha);
    }
    float o_a = alpha;
    if (P.spp > 1) {
      // one launch per sample-per-pixel generation: sum the generations in order (shaders_raymarching.cu:351-376)
      float4* sr = reinterpret_cast<float4*>(P.spp_sum_rgba) + pixel_index;
      float* sg = P.spp_sum_grad + 3ull * pixel_index;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      f3 accg = mk3(0, 0, 0);
      if (P.spp_index > 0) { acc = *sr; accg = mk3(sg[0], sg[1], sg[2]); }
      acc.x += o_c.x; acc.y += o_c.y; acc.z += o_c.z; acc.w += alpha;
      accg.x += o_g.x; accg.y += o_g.y; accg.z += o_g.z;
      if (P.spp_index + 1 < P.spp) {
        *sr = acc; sg[0] = accg.x; sg[1] = accg.y; sg[2] = accg.z;
        return;
      }
      if (Q.ctrl[3] > Q.capacity) return; // an earlier generation overflowed the pool: the whole frame is re-rendered
      const float rspp = 1.f / (float)P.spp;
      o_a = acc.w * rspp;
      o_c = mk3(acc.x * rspp, acc.y * rspp, acc.z * rspp);
      o_g = mk3(accg.x * rspp, accg.y * rspp, accg.z * rspp);
    }
    // spp == 1: (x + 0) * (1 / 1) is exact, so this equals the in-place kernel's o_c * rspp bit for bit
    write_pixel(P, pixel_index, o_c, o_a, o_g);
  }
}

// sums the per-workgroup partials into counters[0..4]: each workgroup reduces a slice and adds its 5 sums with one
// atomic each (integer sums: the result does not depend on the order); counters are zeroed by a memset node before
__global__ __launch_bounds__(256) void reduce_counters_kernel(const unsigned int* __restrict__ partials, int n_blocks, const unsigned int* __restrict__ shade_partials,
                                                             int n_shade_blocks, unsigned long long* counters, unsigned int* pool_ctrl)
{
  if (pool_ctrl && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&pool_ctrl[3], pool_ctrl[0]); // most chunks any generation asked for
  __shared__ unsigned long long red[4][kNC];
  unsigned long long acc[kNC] = { 0, 0, 0, 0, 0, 0, 0 };
  const int stride = gridDim.x * 256;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < n_blocks; b += stride) {
#pragma unroll
    for (int c = 0; c < kNC; ++c) acc[c] += partials[(size_t)b * kNC + c];
  }
  if (shade_partials)
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n_shade_blocks; b += stride) {
      acc[3] += shade_partials[2 * b];
      acc[6] += shade_partials[2 * b + 1];
    }
#pragma unroll
  for (int c = 0; c < kNC; ++c) {
    for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_down(acc[c], off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < kNC; ++c) red[wave][c] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x < kNC) atomicAdd(&counters[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// ------------------------------------------------------------------------------------------------------------------
// launch order of the march workgroups: longest rays first
// A wave needs as long as its longest ray (~1700 dependent steps through the volume's diagonal), however empty the machine
// is; in launch-index order those waves sit in the middle of the grid and the kernel ends with a tail of a few hundred
// microseconds at a fraction of the occupancy (per-wave trace: the last 15 % of the march's time ran < 1/3 of the waves;
// on an image shard, where the kernel is 4-8 x shorter, the tail was half of it).  Sorting the owned 8x8 blocks by the
// length of their rays' box intersection (counting sort, 64 classes, descending) starts the long ones first and lets the
// short ones fill the gaps.  One workgroup; runs when the camera, the framebuffer size, the volume's box or the shard changes.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSchedClasses = 64;
__device__ __forceinline__ unsigned int schedule_class(const RayMarchParams& P, const MarchConsts& mc, unsigned int e)
{
  const f3 c0 = ld3(P.cam_dir), h0 = ld3(P.cam_hor), v0 = ld3(P.cam_ver);
  const f3 oo = to_object(mc, ld3(P.cam_pos));
  const int bx = (int)(e & 0xffffu) * 8, by = (int)(e >> 16) * 8;
  float longest = -1.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) { // the block's centre and corners
    const int ix = min(bx + (k == 0 ? 4 : (k & 1) ? 0 : 7), P.width - 1), iy = min(by + (k == 0 ? 4 : (k & 2) ? 0 : 7), P.height - 1);
    const float ux = ((float)ix + .5f) / (float)P.width - 0.5f, uy = ((float)iy + .5f) / (float)P.height - 0.5f;
    const f3 d = normalize3_exact(mk3(c0.x + ux * h0.x + uy * v0.x, c0.y + ux * h0.y + uy * v0.y, c0.z + ux * h0.z + uy * v0.z));
    float a = 0.f, b = FLT_MAX;
    if (intersect_unit_box(a, b, oo, mk3(d.x * mc.inv_scale.x, d.y * mc.inv_scale.y, d.z * mc.inv_scale.z))) longest = fmaxf(longest, b - a);
  }
  if (!(longest >= 0.f)) return 0u; // no ray of the block meets the volume: last
  const float rel = longest / (P.long_ray_steps * P.step); // 1 = the volume's diagonal
  return (unsigned int)min(kSchedClasses - 1, 1 + (int)(rel * (float)(kSchedClasses - 2)));
}

__global__ __launch_bounds__(1024) void schedule_kernel(const RayMarchParams P, const unsigned int* __restrict__ src, unsigned int n,
                                                        unsigned int* __restrict__ dst)
{
  __shared__ unsigned int count[kSchedClasses], cursor[kSchedClasses];
  VolConsts vc;
  MarchConsts mc;
  setup_consts(P, vc, mc);
  if (threadIdx.x < kSchedClasses) count[threadIdx.x] = 0u;
  __syncthreads();
  for (unsigned int i = threadIdx.x; i < n; i += 1024u) atomicAdd(&count[schedule_class(P, mc, src[i])], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int at = 0u;
    for (int c = kSchedClasses - 1; c >= 0; --c) { cursor[c] = at; at += count[c]; }
  }
  __syncthreads();
  // Stable scatter, 1024 entries at a time: inside a class the blocks keep their list order (the host lists them supertile
  // by supertile, 4x4 blocks = 32x32 pixels), so blocks that run at the same time are compact squares of the image and
  // share their bricks in L2 / Infinity Cache.
  // OVR_SCHED_XCD=1 (experiment, off): transpose the final slot inside aligned groups of 128 so that the 16 blocks of a
  // supertile land on ONE XCD (workgroup s -> XCD s % 8, each XCD has its own L2): sorted position g*128 + x*16 + j -> slot
  // g*128 + j*8 + x.  Measured: C3 march 1.63 instead of 1.57 ms - the march is bound by L1 lookups, not by L2 misses.
#ifndef OVR_SCHED_XCD
#define OVR_SCHED_XCD 0
#endif
  __shared__ unsigned int wave_count[16][kSchedClasses];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int n_full = n & ~127u; // the transposition is applied to whole groups only
  for (unsigned int c0 = 0; c0 < n; c0 += 1024u) {
    for (int k = threadIdx.x; k < 16 * kSchedClasses; k += 1024) (&wave_count[0][0])[k] = 0u;
    __syncthreads();
    const unsigned int i = c0 + threadIdx.x;
    const bool valid = i < n;
    const unsigned int e = valid ? src[i] : 0u;
    const unsigned int cls = valid ? schedule_class(P, mc, e) : 0u;
    // rank among the lanes of this wave with the same class (lower lanes first), wave totals to LDS
    unsigned int rank = 0u;
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {
      const int leader = __builtin_ctzll(todo);
      const unsigned int lc = (unsigned int)__builtin_amdgcn_readlane((int)cls, leader);
      const unsigned long long same = __ballot(valid && cls == lc);
      if (valid && cls == lc) rank = (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) wave_count[wave][lc] = (unsigned int)__popcll(same);
      todo &= ~same;
    }
    __syncthreads();
    // exclusive prefix over the 16 waves, per class; the class cursor moves on by the chunk's total
    if (threadIdx.x < kSchedClasses) {
      unsigned int run = cursor[threadIdx.x];
      for (int w = 0; w < 16; ++w) { const unsigned int t = wave_count[w][threadIdx.x]; wave_count[w][threadIdx.x] = run; run += t; }
      cursor[threadIdx.x] = run;
    }
    __syncthreads();
    if (valid) {
      unsigned int pos = wave_count[wave][cls] + rank;
      if (OVR_SCHED_XCD && pos < n_full) pos = (pos & ~127u) | ((pos & 15u) << 3) | ((pos >> 4) & 7u);
      dst[pos] = e;
    }
    __syncthreads();
  }
}

hipError_t launch_schedule(const RayMarchParams& p, const unsigned int* src, unsigned int n, unsigned int* dst, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, stream, p, src, n, dst);
  return hipGetLastError();
}

constexpr int kReduceBlocks = 64;

size_t raymarch_lds_bytes(int n_color, int n_alpha)
{
  // the transfer function always lives in LDS; 0 = does not fit next to the request queues (caller reports an error)
  const size_t need = (size_t)n_color * sizeof(float4) + (size_t)n_alpha * sizeof(float);
  return need <= 96 * 1024 ? need : 0;
}

size_t raymarch_grid_blocks(const RayMarchParams& p)
{
  if (p.sparse_xy) return ((size_t)p.width * p.height + kBlock / 4 - 1) / (kBlock / 4);
  return p.n_schedule;
}

static dim3 raymarch_grid(const RayMarchParams& p)
{
  return dim3((unsigned)raymarch_grid_blocks(p));
}

template <typename KernT>
static hipError_t set_lds(KernT kern, size_t lds)
{
  if (lds > 64 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return hipSuccess;
}

constexpr int kShadeBlocks = 1024; // persistent shade grid: 4 workgroups per CU

template <int VT, int SHADE, int AM, bool SKIP>
static hipError_t launch_vsbs(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  const size_t tf_lds = raymarch_lds_bytes(p.n_color, p.n_alpha);
  if (tf_lds == 0) return hipErrorInvalidValue;
  if (!p.sparse_xy && p.n_schedule > 0 && !p.schedule) return hipErrorInvalidValue;
  const dim3 grid = raymarch_grid(p), block(kBlock);
  hipError_t e;
  const bool pooled = (SHADE != 0) && p.pool.reqs != nullptr;
  if (!pooled) {
    const size_t lds = std::max<size_t>(tf_lds + table_lds_bytes(p, AM) + (size_t)kWaves * QCfg<SHADE, false>::QCAP * sizeof(ShadeReq), 64); // >= 64 B: the counter reduction reuses it
    auto kern = raymarch_kernel<VT, SHADE, AM, false, SKIP>;
    if ((e = set_lds(kern, lds)) != hipSuccess) return e;
    if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev) { (void)hipEventRecord(ev[1], stream); (void)hipEventRecord(ev[2], stream); }
    if (p.block_counters && p.counters) {
      if ((e = hipMemsetAsync(p.counters, 0, 8 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
      hipLaunchKernelGGL(reduce_counters_kernel, dim3(kReduceBlocks), dim3(256), 0, stream, p.block_counters, (int)raymarch_grid_blocks(p), (const unsigned int*)nullptr, 0, p.counters, (unsigned int*)nullptr);
    }
    return hipGetLastError();
  }
  // ---- pooled pipeline: march -> shade -> composite, once per sample-per-pixel generation
  if ((e = hipMemsetAsync(p.pool.ctrl, 0, 4 * sizeof(unsigned int), stream)) != hipSuccess) return e;
  if (p.block_counters && p.counters)
    if ((e = hipMemsetAsync(p.counters, 0, 8 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
  RayMarchParams q = p;
  for (int g = 0; g < p.spp; ++g) {
    q.spp_index = g;
    if (g > 0 && (e = hipMemsetAsync(p.pool.ctrl, 0, 2 * sizeof(unsigned int), stream)) != hipSuccess) return e;
    {
      constexpr int SH = SHADE == 0 ? 1 : SHADE; // (never instantiated for SHADE == 0: pooled is false)
      const size_t lds = (size_t)kWaves * QCfg<SH, true>::QCAP * sizeof(ShadeReq) + table_lds_bytes(p, AM) + (size_t)p.n_alpha * sizeof(float) + 64;
      auto kern = raymarch_kernel<VT, SH, AM, true, SKIP>;
      if ((e = set_lds(kern, lds)) != hipSuccess) return e;
      if (grid.x > 0) hipLaunchKernelGGL(kern, grid, block, lds, stream, q);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev && g == p.spp - 1) (void)hipEventRecord(ev[1], stream);
    {
      const size_t lds = std::max<size_t>(tf_lds + table_lds_bytes(p, AM), 64);
      auto kern = shade_pool_kernel<VT, SHADE, AM, SKIP>;
      if ((e = set_lds(kern, lds)) != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(kShadeBlocks), block, lds, stream, q);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev && g == p.spp - 1) (void)hipEventRecord(ev[2], stream);
    if (grid.x > 0) hipLaunchKernelGGL(composite_kernel, grid, block, 0, stream, q);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (p.block_counters && p.counters)
      hipLaunchKernelGGL(reduce_counters_kernel, dim3(kReduceBlocks), dim3(256), 0, stream, p.block_counters, (int)raymarch_grid_blocks(p), (const unsigned int*)p.pool.shade_counters,
                         kShadeBlocks, p.counters, p.pool.ctrl);
  }
  return hipGetLastError();
}

template <int VT, int SHADE, int AM>
static hipError_t launch_vsb(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // empty-space skipping is a separate instantiation: the non-skipping kernels stay exactly as they are
  if (p.majorant) return launch_vsbs<VT, SHADE, AM, true>(p, stream, ev);
  return launch_vsbs<VT, SHADE, AM, false>(p, stream, ev);
}

template <int VT, int SHADE>
static hipError_t launch_vs(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // addressing mode: 0 = 32-bit byte offsets (volume <= 4 GiB; largest byte offset = bytes - sizeof(voxel)),
  //                  1 = 32-bit element offsets (< 2^32 stored voxels), 2 = 64-bit z table in LDS,
  //                  3 = 64-bit, computed (axis tables would not fit in LDS next to the queues: a dimension beyond ~8000)
  int am = p.vol.bytes <= 0x100000000ull ? 0 : (p.vol.bytes / voxel_size(p.vol.type) < 0xffffffffull) ? 1 : 2;
  if (const char* f = getenv("OVR_HIP_ADDRESSING")) am = std::max(am, atoi(f)); // diagnostic: a more general mode than needed (tests)
  if (am == 2 && table_lds_bytes(p, 2) > 64 * 1024) am = 3;
  switch (am) {
  case 0: return launch_vsb<VT, SHADE, 0>(p, stream, ev);
  case 1: return launch_vsb<VT, SHADE, 1>(p, stream, ev);
  case 2: return launch_vsb<VT, SHADE, 2>(p, stream, ev);
  default: return launch_vsb<VT, SHADE, 3>(p, stream, ev);
  }
}

template <int VT>
static hipError_t launch_v(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  switch (p.shading) {
  case 0: return launch_vs<VT, 0>(p, stream, ev);
  case 1: return launch_vs<VT, 1>(p, stream, ev);
  default: return launch_vs<VT, 2>(p, stream, ev);
  }
}

size_t pool_shade_blocks() { return kShadeBlocks; }

hipError_t launch_raymarch(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // ev (optional): ev[0] before the first kernel, ev[1] after the march, ev[2] after the shade kernel, ev[3] at the end
  if (ev) (void)hipEventRecord(ev[0], stream);
  hipError_t e;
  switch (p.vol.type) {
  case VOX_U8: e = launch_v<VOX_U8>(p, stream, ev); break;
  case VOX_I8: e = launch_v<VOX_I8>(p, stream, ev); break;
  case VOX_U16: e = launch_v<VOX_U16>(p, stream, ev); break;
  case VOX_I16: e = launch_v<VOX_I16>(p, stream, ev); break;
  case VOX_F32: e = launch_v<VOX_F32>(p, stream, ev); break;
  default: e = hipErrorInvalidValue;
  }
  if (ev) (void)hipEventRecord(ev[3], stream);
  return e;
}

// ------------------------------------------------------------------------------------------------------------------
// volume relayout: linear (x fastest) -> 128-byte x-apron bricks in macro blocks
// ------------------------------------------------------------------------------------------------------------------
int device_voxel_type(int t)
{
  switch (t) {
  case 100: return VOX_U8;
  case 101: return VOX_I8;
  case 200: return VOX_U16;
  case 201: return VOX_I16;
  case 300: case 301: case 400: case 500: return VOX_F32; // u32 / i32 -> normalized f32, f64 -> f32 at upload
  default: return -1;
  }
}
size_t voxel_size(int vt)
{
  switch (vt) {
  case VOX_U8: case VOX_I8: return 1;
  case VOX_U16: case VOX_I16: return 2;
  default: return 4;
  }
}

template <typename TI, typename TO> struct Conv { static __device__ __forceinline__ TO cv(TI v) { return (TO)v; } };
template <> struct Conv<unsigned int, float> { // integer_normalize<float, uint32_t>, array.h:68-76
  static __device__ __forceinline__ float cv(unsigned int v) { return (float)v / (float)0xffffffffu; }
};
template <> struct Conv<int, float> { // array.h:78-90
  static __device__ __forceinline__ float cv(int v) { const float n = (float)v / (float)0x7fffffff; return n < -1.f ? -1.f : n; }
};

template <typename TI, typename TO, int VT>
__global__ __launch_bounds__(256) void relayout_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int nx, int ny, int bricks_x, unsigned int macro_y,
                                                      unsigned long long macro_z, int z0, int nz_chunk)
{
  // grid: x = ceil(bricks_x * SX / 256) over STORED x positions, y = ny, z = nz_chunk ; src holds slices [z0, z0 + nz_chunk)
  typedef BrickMap<VT> M;
  const unsigned sx = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, zl = blockIdx.z;
  if (sx >= (unsigned)bricks_x * M::SX || zl >= nz_chunk) return;
  const unsigned b = sx / M::SX, xr = sx - b * M::SX;  // brick and position inside its row (xr == cx is the apron)
  const unsigned x = b * Vox<VT>::cx + xr;
  const unsigned xs = min(x, (unsigned)(nx - 1));      // beyond the grid: replicate the last voxel (clamp addressing)
  const unsigned z = (unsigned)(z0 + zl);
  const TI v = src[(size_t)xs + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)zl)];
  const unsigned m = M::div_mbx(b), bm = b - m * Vox<VT>::mbx;
  const unsigned long long off = (unsigned long long)(xr + bm * M::BV + m * M::MV) + M::Y((unsigned)y, macro_y) + M::Zlo(z) + (unsigned long long)(z >> 5) * macro_z;
  dst[off] = Conv<TI, TO>::cv(v);
}

template <typename TI, typename TO, int VT>
static hipError_t relayout_t(const void* src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  typedef BrickMap<VT> M;
  const int bricks_x = vd.macros_x * Vox<VT>::mbx;
  dim3 grid((unsigned)((bricks_x * M::SX + 255) / 256), (unsigned)vd.ny, (unsigned)nzc);
  hipLaunchKernelGGL((relayout_kernel<TI, TO, VT>), grid, dim3(256), 0, stream, (const TI*)src, (TO*)dst, vd.nx, vd.ny, bricks_x,
                     vd.macro_elems * (unsigned)vd.macros_x, (unsigned long long)vd.macro_elems * (unsigned long long)vd.macros_x * (unsigned long long)vd.macros_y, z0, nzc);
  return hipGetLastError();
}

// layout constants the host needs to size the allocation
void volume_layout(int voxel_type, int nx, int ny, int nz, VolumeDesc& vd)
{
  unsigned mcx = 30, mv = 0;
  switch (voxel_type) {
  case VOX_U8: mcx = BrickMap<VOX_U8>::MCX; mv = BrickMap<VOX_U8>::MV; break;
  case VOX_I8: mcx = BrickMap<VOX_I8>::MCX; mv = BrickMap<VOX_I8>::MV; break;
  case VOX_U16: mcx = BrickMap<VOX_U16>::MCX; mv = BrickMap<VOX_U16>::MV; break;
  case VOX_I16: mcx = BrickMap<VOX_I16>::MCX; mv = BrickMap<VOX_I16>::MV; break;
  default: mcx = BrickMap<VOX_F32>::MCX; mv = BrickMap<VOX_F32>::MV; break;
  }
  vd.macros_x = (nx + (int)mcx - 1) / (int)mcx;
  vd.macros_y = (ny + 31) / 32;
  vd.macros_z = (nz + 31) / 32;
  vd.macro_elems = mv;
  vd.bytes = (unsigned long long)vd.macros_x * vd.macros_y * vd.macros_z * mv * voxel_size(voxel_type);
}

hipError_t launch_relayout(const void* src, int vt, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  switch (vt) {
  case 100: return relayout_t<unsigned char, unsigned char, VOX_U8>(src, dst, vd, z0, nzc, stream);
  case 101: return relayout_t<signed char, signed char, VOX_I8>(src, dst, vd, z0, nzc, stream);
  case 200: return relayout_t<unsigned short, unsigned short, VOX_U16>(src, dst, vd, z0, nzc, stream);
  case 201: return relayout_t<short, short, VOX_I16>(src, dst, vd, z0, nzc, stream);
  case 300: return relayout_t<unsigned int, float, VOX_F32>(src, dst, vd, z0, nzc, stream);
  case 301: return relayout_t<int, float, VOX_F32>(src, dst, vd, z0, nzc, stream);
  case 400: return relayout_t<float, float, VOX_F32>(src, dst, vd, z0, nzc, stream);
  case 500: return relayout_t<double, float, VOX_F32>(src, dst, vd, z0, nzc, stream);
  default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// macrocells (reference ovr/devices/optix7/accel/sp_singlemc.cu): 16^3-voxel cells with a value range and, per transfer
// function, the largest opacity any sample inside can get.  The reference builds both grids but only its path tracer uses
// them; here the ray marcher and the shadow march skip the voxel fetch of samples whose cell has majorant 0 - such a
// sample's opacity is exactly 0, so frames are bit-identical with and without skipping.
// ------------------------------------------------------------------------------------------------------------------
// value_range_kernel, sp_singlemc.cu:10-54: one WAVE per macrocell (the reference uses one thread), wave min/max reduction
template <int VT>
__global__ __launch_bounds__(256) void macrocell_range_kernel(const void* __restrict__ vol, VolumeDesc vd, int mcx, int mcy, int mcz, float2* __restrict__ out)
{
  typedef BrickMap<VT> M;
  typedef typename Vox<VT>::T T;
  const int lane = threadIdx.x & 63;
  const unsigned long long cell = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= (unsigned long long)mcx * mcy * mcz) return;
  const int cx = (int)(cell % mcx), cy = (int)((cell / mcx) % mcy), cz = (int)(cell / ((unsigned long long)mcx * mcy));
  const int W = 16;
  const int bx = max(cx * W - 1, 0), by = max(cy * W - 1, 0), bz = max(cz * W - 1, 0);
  const int ex = min(bx + W + 1, vd.nx), ey = min(by + W + 1, vd.ny), ez = min(bz + W + 1, vd.nz);
  const int dx = ex - bx, dy = ey - by, dz = ez - bz;
  const unsigned macro_y = vd.macro_elems * (unsigned)vd.macros_x;
  const unsigned long long macro_z = (unsigned long long)vd.macro_elems * vd.macros_x * vd.macros_y;
  const T* base = static_cast<const T*>(vol);
  float lo = INFINITY, hi = -INFINITY; // range1f() is empty
  for (int i = lane; i < dx * dy * dz; i += 64) {
    const int x = bx + i % dx, y = by + (i / dx) % dy, z = bz + i / (dx * dy);
    const unsigned long long off = (unsigned long long)(M::X((unsigned)x) + M::Y((unsigned)y, macro_y)) + M::Zlo((unsigned)z) + (unsigned long long)((unsigned)z >> 5) * macro_z;
    float f = (float)base[off];
    if (VT == VOX_U8) f = f / 255.f;                            // what the normalized texture read returns (array.cpp:304-306)
    if (VT == VOX_I8) { f = f / 127.f; f = f < -1.f ? -1.f : f; }
    lo = fminf(lo, f);
    hi = fmaxf(hi, f);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  if (lane == 0) out[cell] = make_float2(lo, hi);
}

// majorant_kernel, sp_singlemc.cu:56-97: the alpha table is staged in LDS exactly as the reference stages it in shared memory
__global__ __launch_bounds__(256) void macrocell_majorant_kernel(const float2* __restrict__ ranges, unsigned int count, const float* __restrict__ alphas, int n_alpha,
                                                                float vr_lo, float vr_hi, float* __restrict__ out)
{
  extern __shared__ float lds_alpha[];
  for (int i = threadIdx.x; i < n_alpha; i += 256) lds_alpha[i] = alphas[i];
  __syncthreads();
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float2 r = ranges[i];
  const float rcp = 1.f / (vr_hi - vr_lo);
  const float lower = (fminf(fmaxf(r.x, vr_lo), vr_hi) - vr_lo) * rcp;
  const float upper = (fminf(fmaxf(r.y, vr_lo), vr_hi) - vr_lo) * rcp;
  const float fl = floorf(fmaf(lower, (float)(n_alpha - 1), 0.5f)) - 1.f;
  const float fu = floorf(fmaf(upper, (float)(n_alpha - 1), 0.5f)) + 1.f;
  unsigned int il = fl < 0.f ? 0u : (unsigned int)fl; // float -> uint32 saturates in the reference's device code
  unsigned int iu = fu < 0.f ? 0u : (unsigned int)fu;
  il = min(il, (unsigned int)(n_alpha - 1));
  iu = min(iu, (unsigned int)(n_alpha - 1));
  float op = 0.f;
  for (unsigned int k = il; k <= iu; ++k) op = fmaxf(op, lds_alpha[k]);
  out[i] = op;
}

hipError_t launch_macrocell_ranges(const VolumeDesc& vd, float* out_minmax, hipStream_t stream)
{
  const int mcx = (vd.nx + 15) / 16, mcy = (vd.ny + 15) / 16, mcz = (vd.nz + 15) / 16;
  const unsigned long long cells = (unsigned long long)mcx * mcy * mcz;
  const dim3 grid((unsigned)((cells + 3) / 4)), block(256);
  switch (vd.type) {
  case VOX_U8: hipLaunchKernelGGL(macrocell_range_kernel<VOX_U8>, grid, block, 0, stream, vd.data, vd, mcx, mcy, mcz, (float2*)out_minmax); break;
  case VOX_I8: hipLaunchKernelGGL(macrocell_range_kernel<VOX_I8>, grid, block, 0, stream, vd.data, vd, mcx, mcy, mcz, (float2*)out_minmax); break;
  case VOX_U16: hipLaunchKernelGGL(macrocell_range_kernel<VOX_U16>, grid, block, 0, stream, vd.data, vd, mcx, mcy, mcz, (float2*)out_minmax); break;
  case VOX_I16: hipLaunchKernelGGL(macrocell_range_kernel<VOX_I16>, grid, block, 0, stream, vd.data, vd, mcx, mcy, mcz, (float2*)out_minmax); break;
  default: hipLaunchKernelGGL(macrocell_range_kernel<VOX_F32>, grid, block, 0, stream, vd.data, vd, mcx, mcy, mcz, (float2*)out_minmax); break;
  }
  return hipGetLastError();
}

hipError_t launch_macrocell_majorants(const float* minmax, unsigned int count, const float* alphas, int n_alpha, float vr_lo, float vr_hi, float* out,
                                      hipStream_t stream)
{
  hipLaunchKernelGGL(macrocell_majorant_kernel, dim3((count + 255) / 256), dim3(256), (size_t)n_alpha * sizeof(float), stream, (const float2*)minmax, count, alphas,
                     n_alpha, vr_lo, vr_hi, out);
  return hipGetLastError();
}

// the volume's data range: min / max over all macrocell ranges (every voxel lies in at least one cell); fminf / fmaxf drop
// NaN operands like the reference's std::min / std::max chain does (array.cpp:44-62)
__global__ __launch_bounds__(1024) void minmax_reduce_kernel(const float2* __restrict__ ranges, unsigned long long cells, float* __restrict__ out)
{
  __shared__ float slo[16], shi[16];
  float lo = FLT_MAX, hi = -FLT_MAX; // numeric_limits<float>::max() / lowest()
  for (unsigned long long i = threadIdx.x; i < cells; i += 1024ull) {
    const float2 r = ranges[i];
    lo = fminf(lo, r.x);
    hi = fmaxf(hi, r.y);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    out[0] = lo;
    out[1] = hi;
  }
}
hipError_t launch_minmax_reduce(const float* minmax, unsigned long long cells, float* out, hipStream_t stream)
{
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(1024), 0, stream, (const float2*)minmax, cells, out);
  return hipGetLastError();
}

// occupancy for the per-ray skip interval (skip_interval): a coarse grid of 4^3 macrocells (64^3 voxels) per entry - small
// enough (4 KiB at 1024^3) to stay in L1 while every ray walks it.  An entry is set if any of its macrocells, or any macrocell
// next to one of them (dilation by one macrocell), can hold a sample with opacity > 0.  The dilation is the safety margin of the
// walk: a ray's cell sequence is right to ~1e-4 voxel of position, the nearest non-empty sample is >= 16 voxels inside a set entry.
__global__ __launch_bounds__(256) void macrocell_coarse_kernel(const float* __restrict__ majorant, int mcx, int mcy, int mcz, unsigned char* __restrict__ out)
{
  const int gx = (mcx + 3) / 4, gy = (mcy + 3) / 4, gz = (mcz + 3) / 4;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (unsigned int)(gx * gy * gz)) return;
  const int cx = (int)(i % (unsigned int)gx) * 4, cy = (int)((i / (unsigned int)gx) % (unsigned int)gy) * 4, cz = (int)(i / (unsigned int)(gx * gy)) * 4;
  bool any = false;
  for (int z = max(cz - 1, 0); z <= min(cz + 4, mcz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 4, mcy - 1); ++y)
      for (int x = max(cx - 1, 0); x <= min(cx + 4, mcx - 1); ++x)
        any = any || (majorant[(size_t)x + (size_t)mcx * ((size_t)y + (size_t)mcy * (size_t)z)] > 0.f);
  out[i] = any ? 1 : 0;
}
hipError_t launch_macrocell_coarse(const float* majorant, int nx, int ny, int nz, unsigned char* out, hipStream_t stream)
{
  const int mcx = (nx + 15) / 16, mcy = (ny + 15) / 16, mcz = (nz + 15) / 16;
  const unsigned int cells = (unsigned int)(((mcx + 3) / 4) * ((mcy + 3) / 4) * ((mcz + 3) / 4));
  hipLaunchKernelGGL(macrocell_coarse_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, majorant, mcx, mcy, mcz, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// sparse-sampling mask: noise tile slice staged in LDS, keep test, wave64 ballot + prefix compaction
// (generate_mask.cu:55-96; the reference compacts with thrust::remove, which keeps pixel order - so does this)
// ------------------------------------------------------------------------------------------------------------------
// __expf restated as a fixed sequence of IEEE basic operations, so the integer keep/discard decision is reproducible
// bit for bit (2^n * 2^f with a degree-6 Horner polynomial; relative error 2e-7 near 0, 2e-6 at x = -30: inside __expf's own bound of 2 + |1.16 x| ulp)
__device__ __forceinline__ float exp_det(float x)
{
  if (x < -87.f) return 0.f;
  if (x > 88.f) x = 88.f;
  const float t = x * 1.44269504088896341f;
  const float n = floorf(t + 0.5f);
  const float f = t - n;
  float p = 1.53533031e-4f;
  p = fmaf(p, f, 1.33988696e-3f);
  p = fmaf(p, f, 9.61843120e-3f);
  p = fmaf(p, f, 5.55033022e-2f);
  p = fmaf(p, f, 2.40226504e-1f);
  p = fmaf(p, f, 6.93147182e-1f);
  p = fmaf(p, f, 1.0f);
  return p * __uint_as_float((unsigned int)((int)n + 127) << 23);
}

// The noise tile is stored transposed, [t][y][x] (the reference's file is [y][x][t], blue_noise.h:95-99), so the slice of one
// frame is a contiguous xy x xy block (16 - 64 KiB) instead of one float every 256 bytes.  A workgroup covers 256 consecutive
// pixels = at most two image rows; the noise rows they need are staged in LDS (lds_noise[2][xy]) with coalesced loads.
__device__ __forceinline__ void stage_noise_rows(const SparseMaskParams& p, float* lds_noise, unsigned int first_pixel)
{
  const int xy = p.noise_xy;
  const float* slice = p.noise + (size_t)(p.frame_index % 64) * xy * xy;
  const int y0 = (int)(first_pixel / (unsigned int)p.width);
  for (int i = threadIdx.x; i < 2 * xy; i += 256) {
    const int r = i / xy, c = i - r * xy;
    lds_noise[i] = slice[(size_t)((y0 + r) % xy) * xy + c];
  }
  __syncthreads();
}

__device__ __forceinline__ bool mask_keep(const SparseMaskParams& p, const float* lds_noise, unsigned int first_pixel, unsigned int i, int& x, int& y)
{
  x = (int)(i % (unsigned int)p.width);
  y = (int)(i / (unsigned int)p.width);
  const int xy = p.noise_xy;
  const int r = y - (int)(first_pixel / (unsigned int)p.width); // 0 or 1 (a third row only if width < 128: read the slice directly)
  const float val = r < 2 ? lds_noise[r * xy + (x % xy)] : p.noise[(size_t)(p.frame_index % 64) * xy * xy + (size_t)(y % xy) * xy + (x % xy)];
  const float aspect = (float)p.width / p.height;
  const float fx = ((float)x / p.width - p.mean_x);
  const float fy = ((float)y / p.height - p.mean_y) / aspect;
  const float pr = (1.0f - p.base_noise) * exp_det(-0.5f * (fx * fx + fy * fy) * p.sigma_rcp2) + p.base_noise;
  return val < pr;
}

__global__ __launch_bounds__(256) void mask_count_kernel(const SparseMaskParams p)
{
  __shared__ unsigned int wave_cnt[4];
  __shared__ float lds_noise[2 * 128];
  const unsigned int n = (unsigned int)p.width * (unsigned int)p.height;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  stage_noise_rows(p, lds_noise, blockIdx.x * 256);
  int x, y;
  const bool keep = (i < n) && mask_keep(p, lds_noise, blockIdx.x * 256, i, x, y);
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (unsigned int)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) p.block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of block_counts in place (single workgroup, serial over chunks of 1024)
__global__ __launch_bounds__(1024) void mask_scan_kernel(unsigned int* counts, int n_blocks, unsigned long long* total)
{
  __shared__ unsigned int wsum[16];
  __shared__ unsigned int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned int v = (i < n_blocks) ? counts[i] : 0u;
    unsigned int s = v; // inclusive scan within the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned int t = __shfl_up(s, off);
      if (lane >= off) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    unsigned int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const unsigned int c = carry;
    if (i < n_blocks) counts[i] = c + wbase + s - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + wbase + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = 2ull * carry;
}

__global__ __launch_bounds__(256) void mask_write_kernel(const SparseMaskParams p)
{
  __shared__ unsigned int wave_cnt[4];
  __shared__ float lds_noise[2 * 128];
  const unsigned int n = (unsigned int)p.width * (unsigned int)p.height;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  stage_noise_rows(p, lds_noise, blockIdx.x * 256);
  int x = 0, y = 0;
  const bool keep = (i < n) && mask_keep(p, lds_noise, blockIdx.x * 256, i, x, y);
  const unsigned long long b = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // prefix of the ballot below this lane = v_mbcnt
  const unsigned int prefix = __builtin_amdgcn_mbcnt_hi((unsigned int)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b, 0u));
  if (lane == 0) wave_cnt[wave] = (unsigned int)__popcll(b);
  __syncthreads();
  unsigned int off = p.block_counts[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += wave_cnt[w];
  if (keep) {
    const size_t o = 2 * ((size_t)off + prefix);
    p.out_xy[o] = x;
    p.out_xy[o + 1] = y;
  }
}

size_t sparse_mask_workspace_elems(int width, int height) { return ((size_t)width * height + 255) / 256 + 1; }

hipError_t launch_sparse_mask(const SparseMaskParams& p, hipStream_t stream)
{
  const int n_blocks = (int)(((size_t)p.width * p.height + 255) / 256);
  hipLaunchKernelGGL(mask_count_kernel, dim3(n_blocks), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(mask_scan_kernel, dim3(1), dim3(1024), 0, stream, p.block_counts, n_blocks, p.count);
  hipLaunchKernelGGL(mask_write_kernel, dim3(n_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// TEA known-answer entry
// ------------------------------------------------------------------------------------------------------------------
__global__ void tea_kernel(uint32_t* v0v1, float* out, long long n)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int v0 = v0v1[2 * i], v1 = v0v1[2 * i + 1];
  tea16(v0, v1);
  v0v1[2 * i] = v0; v0v1[2 * i + 1] = v1;
  out[2 * i] = (float)v0 * OVR_TEA_TOFLOAT;
  out[2 * i + 1] = (float)v1 * OVR_TEA_TOFLOAT;
}
hipError_t launch_tea(uint32_t* v0v1, float* out, int64_t n, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(tea_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, v0v1, out, (long long)n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// tile pack / unpack (payload of the per-frame RCCL gather): slot k = k-th tile owned by `rank` in row-major tile order
// ------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int owned_in_row(int tiles_x, int ty, int rank, int world)
{
  const int first = ((rank - ty) % world + world) % world; // smallest tx with (tx + ty) % world == rank
  return first < tiles_x ? (tiles_x - 1 - first) / world + 1 : 0;
}
int count_owned_tiles(int width, int height, int tw, int th, int rank, int world)
{
  const int tiles_x = (width + tw - 1) / tw, tiles_y = (height + th - 1) / th;
  int n = 0;
  for (int ty = 0; ty < tiles_y; ++ty) n += owned_in_row(tiles_x, ty, rank, world);
  return n;
}

// slot of tile (tx, ty) in its owner's payload.  `world` consecutive tile rows hold every column exactly once per rank, so
// a rank owns tiles_x tiles per full period of rows; only the rows of the last, partial period are summed
__host__ __device__ inline int tile_slot(int tiles_x, int tx, int ty, int rank, int world)
{
  int slot = (ty / world) * tiles_x;
  for (int r = ty - ty % world; r < ty; ++r) slot += owned_in_row(tiles_x, r, rank, world);
  const int first = ((rank - ty) % world + world) % world;
  return slot + (tx - first) / world;
}

// one thread per frame pixel.  PACK: frame -> this rank's payload (pixels of foreign tiles exit).  !PACK: payload -> frame;
// rank >= 0 scatters that rank's payload, rank < 0 scatters ALL ranks' payloads, laid out `rank_stride` float4 apart
template <bool PACK>
__global__ __launch_bounds__(256) void tiles_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int width, int height,
                                                   int tw, int th, int rank, int world, size_t rank_stride)
{
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ix >= width || iy >= height) return;
  const int tx = ix / tw, ty = iy / th;
  const int owner = (tx + ty) % world;
  if (rank >= 0 && owner != rank) return;
  const int tiles_x = (width + tw - 1) / tw;
  const int slot = tile_slot(tiles_x, tx, ty, owner, world);
  const size_t pi = (size_t)slot * tw * th + (size_t)(iy - ty * th) * tw + (size_t)(ix - tx * tw) + (rank < 0 ? (size_t)owner * rank_stride : 0);
  const size_t fi = (size_t)iy * width + ix;
  if (PACK) dst[pi] = src[fi];
  else dst[fi] = src[pi];
}

hipError_t launch_pack_tiles(const float* frame, float* dst, int width, int height, int tw, int th, int rank, int world, hipStream_t stream)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL((tiles_kernel<true>), grid, dim3(256), 0, stream, (const float4*)frame, (float4*)dst, width, height, tw, th, rank, world, (size_t)0);
  return hipGetLastError();
}
hipError_t launch_unpack_tiles(const float* src, float* frame, int width, int height, int tw, int th, int rank, int world, size_t rank_stride_floats,
                               hipStream_t stream)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL((tiles_kernel<false>), grid, dim3(256), 0, stream, (const float4*)src, (float4*)frame, width, height, tw, th, rank, world,
                     rank_stride_floats / 4);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// frame output: image_to_rgba8 of the reference (ovr/common/imageio.cpp:146-181, 4 channels), on the device so that the
// host copy of a displayed / saved frame is 4 bytes per pixel instead of 16
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rgba8_kernel(const float4* __restrict__ rgba, uint32_t* __restrict__ out, int width, int height, int flip)
{
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6); // output position
  if (ix >= width || iy >= height) return;
  const int sy = flip ? height - 1 - iy : iy;
  const float4 v = rgba[(size_t)sy * width + ix];
  // std::clamp(v, 0.f, 1.f) * 255.f, truncated (a NaN, undefined in the reference's cast, becomes 0)
  auto q = [](float x) -> uint32_t { return (uint32_t)(((x < 0.f) ? 0.f : (1.f < x) ? 1.f : x) * 255.f); };
  out[(size_t)iy * width + ix] = q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}
hipError_t launch_rgba8(const float* rgba, uint32_t* out, int width, int height, int flip, hipStream_t stream)
{
  if (width <= 0 || height <= 0) return hipSuccess;
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL(rgba8_kernel, grid, dim3(256), 0, stream, (const float4*)rgba, out, width, height, flip);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// frame output, EXR: the reference's save_image(".exr") hands the flipped RGBA32F frame to tinyexr and asks for HALF pixels
// (ovr/common/imageio.cpp:15-83,268-272).  The conversion on the device follows tinyexr's rule, which is NOT the hardware's
// v_cvt_f16_f32 (round to nearest even): the float mantissa is cut to 10 bits and its bit 12 rounds up (ties away from
// zero), the carry may run into the exponent; results below the half normal range shift the significand incl. the hidden
// bit and round by the last bit shifted out; float denormals -> signed 0, NaN -> quiet NaN 0x200, overflow -> infinity.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int exr_half_bits(float f)
{
  const unsigned int u = __float_as_uint(f);
  const unsigned int sign = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
  unsigned int h = 0u;
  if (e == 255u) h = 0x7c00u | (m ? 0x200u : 0u);
  else if (e != 0u) {
    const int ne = (int)e - 112; // re-biased exponent
    if (ne >= 31) h = 0x7c00u;
    else if (ne > 0) h = (((unsigned int)ne << 10) | (m >> 13)) + ((m >> 12) & 1u);
    else if (ne >= -10) {
      const unsigned int sig = m | 0x800000u;
      h = (sig >> (14 - ne)) + ((sig >> (13 - ne)) & 1u);
    }
  }
  return sign | h;
}
__global__ __launch_bounds__(256) void rgba16f_kernel(const float4* __restrict__ rgba, uint2* __restrict__ out, int width, int height, int flip)
{
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6); // output position
  if (ix >= width || iy >= height) return;
  const int sy = flip ? height - 1 - iy : iy;
  const float4 v = rgba[(size_t)sy * width + ix];
  out[(size_t)iy * width + ix] = make_uint2(exr_half_bits(v.x) | (exr_half_bits(v.y) << 16), exr_half_bits(v.z) | (exr_half_bits(v.w) << 16));
}
hipError_t launch_rgba16f(const float* rgba, uint16_t* out, int width, int height, int flip, hipStream_t stream)
{
  if (width <= 0 || height <= 0) return hipSuccess;
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL(rgba16f_kernel, grid, dim3(256), 0, stream, (const float4*)rgba, (uint2*)out, width, height, flip);
  return hipGetLastError();
}

} // namespace ovrhip
