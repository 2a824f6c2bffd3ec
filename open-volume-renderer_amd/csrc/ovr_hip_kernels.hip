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
// Kernel shape: one lane per pixel, one wave64 per 8x8 pixel tile, four waves (16x16 pixels) per workgroup.  The
// transfer function (colour float4 table + alpha table, 20 KiB at the shipped resolution of 1024) is staged in LDS
// once per workgroup; the volume is read from the yz-tiled row layout described in ovr_hip_kernels.h with one
// 8-byte load per (y,z) row of a trilinear tap.  Built with -ffp-contract=off: every fused multiply-add below is
// explicit so the operation order is the one the CPU oracle (oracle/ovr_oracle.c) restates.
#include "ovr_hip_kernels.h"

#include <float.h>

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

// opacity correction, shaders_raymarching.cu:118-122: 1 - __powf(1 - a, base*dt); __powf == exp2(y * log2(x))
__device__ __forceinline__ float opacity_correction(float a, float adj)
{
  if (!(fabsf(adj - 1.f) < 1e-7f)) {
    const float pw = __builtin_amdgcn_exp2f(adj * __builtin_amdgcn_logf(1.f - a));
    a = clamp01(1.f - pw);
  }
  return a;
}

// ------------------------------------------------------------------------------------------------------------------
// voxel access: one load returns the two x-neighbours of a row
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned short u16x2_u __attribute__((ext_vector_type(2), aligned(2)));
typedef short i16x2_u __attribute__((ext_vector_type(2), aligned(2)));
typedef unsigned char u8x2_u __attribute__((ext_vector_type(2), aligned(1)));
typedef signed char i8x2_u __attribute__((ext_vector_type(2), aligned(1)));

template <int VT> struct Vox;
template <> struct Vox<VOX_F32> {
  static __device__ __forceinline__ void pair(const void* base, unsigned long long off, float& a, float& b)
  {
    const f32x2_u v = *reinterpret_cast<const f32x2_u*>(static_cast<const float*>(base) + off);
    a = v.x; b = v.y;
  }
  static constexpr bool kScale = false;
  static constexpr bool kClamp = false;
};
template <> struct Vox<VOX_U16> {
  static __device__ __forceinline__ void pair(const void* base, unsigned long long off, float& a, float& b)
  {
    const u16x2_u v = *reinterpret_cast<const u16x2_u*>(static_cast<const unsigned short*>(base) + off);
    a = (float)v.x; b = (float)v.y;
  }
  static constexpr bool kScale = false; // u16 is sampled as RAW float (array.cpp:335-338)
  static constexpr bool kClamp = false;
};
template <> struct Vox<VOX_I16> {
  static __device__ __forceinline__ void pair(const void* base, unsigned long long off, float& a, float& b)
  {
    const i16x2_u v = *reinterpret_cast<const i16x2_u*>(static_cast<const short*>(base) + off);
    a = (float)v.x; b = (float)v.y;
  }
  static constexpr bool kScale = false;
  static constexpr bool kClamp = false;
};
template <> struct Vox<VOX_U8> {
  static __device__ __forceinline__ void pair(const void* base, unsigned long long off, float& a, float& b)
  {
    const u8x2_u v = *reinterpret_cast<const u8x2_u*>(static_cast<const unsigned char*>(base) + off);
    a = (float)v.x; b = (float)v.y;
  }
  static constexpr bool kScale = true; // normalized read: v / 255 (array.cpp:304-306)
  static constexpr bool kClamp = false;
};
template <> struct Vox<VOX_I8> {
  static __device__ __forceinline__ void pair(const void* base, unsigned long long off, float& a, float& b)
  {
    const i8x2_u v = *reinterpret_cast<const i8x2_u*>(static_cast<const signed char*>(base) + off);
    a = (float)v.x; b = (float)v.y;
  }
  static constexpr bool kScale = true; // max(v / 127, -1)
  static constexpr bool kClamp = true;
};

struct VolConsts {
  const void* data;
  int nx1, ny1, nz1; // n - 1
  unsigned int row_stride;
  int tile_row_z;    // tiles_y * 64
  float fx1, fy1, fz1;
  f3 cs, cb;
  float vscale, vmin;
};

__device__ __forceinline__ void axis_tap(float po, float cs, float cb, float fn1, int n1, int& i0, int& i1, float& f)
{
  const float p = clamp01(po);                       // sample_volume_object_space clamps p to [0,1]
  float x = fmaf(p, cs, cb);                         // cell-centred: p*N - 0.5
  x = fminf(fmaxf(x, 0.f), fn1);                     // == clamp-to-edge addressing (row pad replicates the last voxel)
  const float fl = floorf(x);
  f = x - fl;
  i0 = (int)fl;
  i1 = min(i0 + 1, n1);
}
__device__ __forceinline__ int row_y(int y) { return ((y >> 3) << 6) + (y & 7); }
__device__ __forceinline__ int row_z(int z, int tile_row_z) { return (z >> 3) * tile_row_z + ((z & 7) << 3); }

// trilinear tap at object-space p (shaders_common.h:186-193); returns what tex3D<float> returns
template <int VT>
__device__ __forceinline__ float sample_volume(const VolConsts& vc, f3 p)
{
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  axis_tap(p.x, vc.cs.x, vc.cb.x, vc.fx1, vc.nx1, x0, x1, fx);
  axis_tap(p.y, vc.cs.y, vc.cb.y, vc.fy1, vc.ny1, y0, y1, fy);
  axis_tap(p.z, vc.cs.z, vc.cb.z, vc.fz1, vc.nz1, z0, z1, fz);
  (void)x1;
  const int ry0 = row_y(y0), ry1 = row_y(y1);
  const int rz0 = row_z(z0, vc.tile_row_z), rz1 = row_z(z1, vc.tile_row_z);
  const unsigned long long o00 = (unsigned long long)(unsigned int)(ry0 + rz0) * vc.row_stride + (unsigned int)x0;
  const unsigned long long o10 = (unsigned long long)(unsigned int)(ry1 + rz0) * vc.row_stride + (unsigned int)x0;
  const unsigned long long o01 = (unsigned long long)(unsigned int)(ry0 + rz1) * vc.row_stride + (unsigned int)x0;
  const unsigned long long o11 = (unsigned long long)(unsigned int)(ry1 + rz1) * vc.row_stride + (unsigned int)x0;
  float a00, b00, a10, b10, a01, b01, a11, b11;
  Vox<VT>::pair(vc.data, o00, a00, b00);
  Vox<VT>::pair(vc.data, o10, a10, b10);
  Vox<VT>::pair(vc.data, o01, a01, b01);
  Vox<VT>::pair(vc.data, o11, a11, b11);
  if (Vox<VT>::kClamp) {
    a00 = fmaxf(a00, vc.vmin); b00 = fmaxf(b00, vc.vmin); a10 = fmaxf(a10, vc.vmin); b10 = fmaxf(b10, vc.vmin);
    a01 = fmaxf(a01, vc.vmin); b01 = fmaxf(b01, vc.vmin); a11 = fmaxf(a11, vc.vmin); b11 = fmaxf(b11, vc.vmin);
  }
  const float c00 = lerpf(a00, b00, fx), c10 = lerpf(a10, b10, fx);
  const float c01 = lerpf(a01, b01, fx), c11 = lerpf(a11, b11, fx);
  const float c0 = lerpf(c00, c10, fy), c1 = lerpf(c01, c11, fy);
  float s = lerpf(c0, c1, fz);
  if (Vox<VT>::kScale) s *= vc.vscale;
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

// raymarching_shadow, shaders_raymarching.cu:44-85 (+ :205-229): alpha-only march toward the light
template <int VT>
__device__ __forceinline__ float march_shadow(const VolConsts& vc, const TfConsts& tf, const MarchConsts& mc, f3 org, unsigned int& n_shadow)
{
  const f3 oo = to_object(mc, org);
  const f3 od = mk3(mc.light.x * mc.inv_scale.x, mc.light.y * mc.inv_scale.y, mc.light.z * mc.inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  float alpha = 0.f;
  if (!intersect_unit_box(t0, t1, oo, od)) return alpha;
  float tx = t0, ty = fminf(t1, t0 + mc.shadow_stride);
  while ((ty > tx) && (alpha < 0.9999f)) {
    const float tm = 0.5f * (tx + ty);
    const f3 pos = mk3(fmaf(tm, mc.light.x, org.x), fmaf(tm, mc.light.y, org.y), fmaf(tm, mc.light.z, org.z));
    const float s = sample_volume<VT>(vc, to_object(mc, pos));
    float a = tf_alpha(tf, tf_coord(tf, s));
    a = opacity_correction(a, mc.base * (ty - tx));
    alpha = fmaf(1.f - alpha, a, alpha);
    ++n_shadow;
    tx = ty;
    ty = fminf(tx + mc.shadow_stride, t1);
  }
  return alpha;
}

// ------------------------------------------------------------------------------------------------------------------
// the ray-march kernel
// ------------------------------------------------------------------------------------------------------------------
constexpr int kBlock = 256;

template <int VT, int SHADE, bool TF_LDS>
__global__ __launch_bounds__(kBlock) void raymarch_kernel(const RayMarchParams P)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  TfConsts tf;
  if (TF_LDS) {
    float4* lc = reinterpret_cast<float4*>(lds_raw);
    float* la = reinterpret_cast<float*>(lds_raw + (size_t)P.n_color * sizeof(float4));
    const float4* gc = reinterpret_cast<const float4*>(P.tf_color);
    for (int i = threadIdx.x; i < P.n_color; i += kBlock) lc[i] = gc[i];
    for (int i = threadIdx.x; i < P.n_alpha; i += kBlock) la[i] = P.tf_alpha[i];
    __syncthreads();
    tf.color = lc;
    tf.alpha = la;
  }
  else {
    tf.color = reinterpret_cast<const float4*>(P.tf_color);
    tf.alpha = P.tf_alpha;
  }
  tf.nc1 = P.n_color - 1; tf.na1 = P.n_alpha - 1;
  tf.fnc1 = (float)tf.nc1; tf.fna1 = (float)tf.na1;
  tf.lower = P.tf_lower; tf.upper = P.tf_upper; tf.scale = P.tf_scale;

  // ---- which pixel does this lane own? (compute_screen_position, shaders_common.h:394-451)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ix, iy;
  bool active;
  if (P.sparse_xy) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    active = (2ull * i) < *P.sparse_count;
    ix = active ? P.sparse_xy[2 * i] : 0;
    iy = active ? P.sparse_xy[2 * i + 1] : 0;
  }
  else {
    const int bx = blockIdx.x, by = blockIdx.y;
    ix = bx * 16 + (wave & 1) * 8 + (lane & 7);
    iy = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    active = ix < P.width && iy < P.height;
  }
  if (P.world > 1 && active) active = ((ix / P.tile_w + iy / P.tile_h) % P.world) == P.rank;

  unsigned int n_rays = 0, n_samples = 0, n_shaded = 0, n_shadow = 0;

  if (active) {
    VolConsts vc;
    vc.data = P.vol.data;
    vc.nx1 = P.vol.nx - 1; vc.ny1 = P.vol.ny - 1; vc.nz1 = P.vol.nz - 1;
    vc.fx1 = (float)vc.nx1; vc.fy1 = (float)vc.ny1; vc.fz1 = (float)vc.nz1;
    vc.row_stride = (unsigned int)P.vol.row_stride;
    vc.tile_row_z = P.vol.tiles_y * 64;
    vc.cs = ld3(P.coord_scale); vc.cb = ld3(P.coord_bias);
    vc.vscale = P.vol.value_scale; vc.vmin = P.vol.value_min_clamp;
    MarchConsts mc;
    mc.inv_scale = ld3(P.inv_scale); mc.wto_p = ld3(P.wto_p); mc.otw_it = ld3(P.otw_it); mc.light = ld3(P.light);
    mc.gstep = ld3(P.grad_step);
    mc.ginv = mk3(1.f / mc.gstep.x, 1.f / mc.gstep.y, 1.f / mc.gstep.z);
    mc.step = P.step; mc.base = P.base; mc.shadow_stride = P.shadow_stride;

    const float rsx = 1.f / (float)P.width, rsy = 1.f / (float)P.height;
    const float scx = ((float)ix + .5f) * rsx, scy = ((float)iy + .5f) * rsy;
    const unsigned int pixel_index = (unsigned int)ix + (unsigned int)iy * (unsigned int)P.width;
    unsigned int v0 = (unsigned int)P.frame_index, v1 = pixel_index; // RandomTEA(frame_index, pixel_index)
    const f3 org = ld3(P.cam_pos), cdir = ld3(P.cam_dir), chor = ld3(P.cam_hor), cver = ld3(P.cam_ver);

    float o_a = 0.f;
    f3 o_c = mk3(0, 0, 0), o_g = mk3(0, 0, 0);
    const int spp = P.spp;
    for (int k = 0; k < spp; ++k) {
      float sx = scx, sy = scy;
      if (spp > 1) {
        tea16(v0, v1);
        sx += ((float)v0 * OVR_TEA_TOFLOAT - 0.5f) * rsx;
        sy += ((float)v1 * OVR_TEA_TOFLOAT - 0.5f) * rsy;
      }
      const float ux = sx - 0.5f, uy = sy - 0.5f;
      const f3 dir = normalize3_exact(mk3(cdir.x + ux * chor.x + uy * cver.x, cdir.y + ux * chor.y + uy * cver.y,
                                          cdir.z + ux * chor.z + uy * cver.z));
      ++n_rays;
      // ---- __intersection__volume: object-space ray, direction not renormalised so t is shared
      const f3 oo = to_object(mc, org);
      const f3 od = mk3(dir.x * mc.inv_scale.x, dir.y * mc.inv_scale.y, dir.z * mc.inv_scale.z);
      float t0 = 0.f, t1 = FLT_MAX;
      float alpha = 0.f;
      f3 color = mk3(0, 0, 0), gradient = mk3(0, 0, 0);
      if (intersect_unit_box(t0, t1, oo, od)) {
        float tx = t0, ty = fminf(t1, t0 + mc.step);
        while ((ty > tx) && (alpha < 0.9999f)) {
          const float tm = 0.5f * (tx + ty);
          const f3 pos = mk3(fmaf(tm, dir.x, org.x), fmaf(tm, dir.y, org.y), fmaf(tm, dir.z, org.z));
          const f3 po = to_object(mc, pos);
          const float s = sample_volume<VT>(vc, po);
          const float v = tf_coord(tf, s);
          float a = tf_alpha(tf, v);
          a = opacity_correction(a, mc.base * (ty - tx));
          ++n_samples;
          // A sample whose corrected opacity is exactly 0 adds exactly 0 to colour, gradient and alpha (its colour
          // passes through clamp01 first, so it is finite): the gradient taps and the shadow march are skipped.
          if (a > 0.f) {
            ++n_shaded;
            f3 rgb = tf_color(tf, v);
            f3 n_c = mk3(0, 0, 0);
            if (SHADE != 0) {
              // compute_volume_gradient_object_space, shaders_common.h:195-215 (one-sided, flipped at the upper bound)
              f3 g;
              {
                const bool fl = (po.x + mc.gstep.x) > 1.f;
                const float h = fl ? -mc.gstep.x : mc.gstep.x;
                g.x = (sample_volume<VT>(vc, mk3(po.x + h, po.y, po.z)) - s) * (fl ? -mc.ginv.x : mc.ginv.x);
              }
              {
                const bool fl = (po.y + mc.gstep.y) > 1.f;
                const float h = fl ? -mc.gstep.y : mc.gstep.y;
                g.y = (sample_volume<VT>(vc, mk3(po.x, po.y + h, po.z)) - s) * (fl ? -mc.ginv.y : mc.ginv.y);
              }
              {
                const bool fl = (po.z + mc.gstep.z) > 1.f;
                const float h = fl ? -mc.gstep.z : mc.gstep.z;
                g.z = (sample_volume<VT>(vc, mk3(po.x, po.y, po.z + h)) - s) * (fl ? -mc.ginv.z : mc.ginv.z);
              }
              const f3 gn = normalize3(g);
              const f3 n_o = mk3(-gn.x, -gn.y, -gn.z);
              const f3 n_w = normalize3(mk3(n_o.x * mc.otw_it.x, n_o.y * mc.otw_it.y, n_o.z * mc.otw_it.z));
              if (P.grad) {
                const float* m = P.wtc_it;
                n_c = normalize3(mk3(fmaf(n_w.x, m[0], fmaf(n_w.y, m[3], n_w.z * m[6])), fmaf(n_w.x, m[1], fmaf(n_w.y, m[4], n_w.z * m[7])),
                                     fmaf(n_w.x, m[2], fmaf(n_w.y, m[5], n_w.z * m[8]))));
              }
              float shadow = 0.f;
              if (SHADE == 2) shadow = march_shadow<VT>(vc, tf, mc, pos, n_shadow);
              const float cosNL = fabsf(dot3(mc.light, n_w));
              const float shade = 0.5f + 0.5f * cosNL * 2.f * (1.f - shadow);
              rgb.x *= shade; rgb.y *= shade; rgb.z *= shade;
            }
            const float tr = 1.f - alpha;
            color.x = fmaf(tr * clamp01(rgb.x), a, color.x);
            color.y = fmaf(tr * clamp01(rgb.y), a, color.y);
            color.z = fmaf(tr * clamp01(rgb.z), a, color.z);
            gradient.x = fmaf(tr * clamp01(n_c.x), a, gradient.x);
            gradient.y = fmaf(tr * clamp01(n_c.y), a, gradient.y);
            gradient.z = fmaf(tr * clamp01(n_c.z), a, gradient.z);
            alpha = fmaf(tr, a, alpha);
          }
          tx = ty;
          ty = fminf(tx + mc.step, t1);
        }
      }
      // render_raymarching / alpha_blend with an always-missing background (shaders_raymarching.cu:260-321)
      o_a += alpha;
      if (alpha > 0.f) {
        o_c.x += color.x / alpha; o_c.y += color.y / alpha; o_c.z += color.z / alpha;
        o_g.x += gradient.x / alpha; o_g.y += gradient.y / alpha; o_g.z += gradient.z / alpha;
      }
    }
    const float rspp = 1.f / (float)spp;
    o_a *= rspp;
    o_c.x *= rspp; o_c.y *= rspp; o_c.z *= rspp;
    o_g.x *= rspp; o_g.y *= rspp; o_g.z *= rspp;

    // accumulation, shaders_raymarching.cu:389-403
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

  // ---- counters: wave reduction, one atomic per wave and counter
  unsigned int n_active = active ? 1u : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_rays += __shfl_down(n_rays, off);
    n_samples += __shfl_down(n_samples, off);
    n_shaded += __shfl_down(n_shaded, off);
    n_shadow += __shfl_down(n_shadow, off);
    n_active += __shfl_down(n_active, off);
  }
  if (lane == 0 && P.counters && n_active) {
    atomicAdd(&P.counters[0], (unsigned long long)n_rays);
    atomicAdd(&P.counters[1], (unsigned long long)n_samples);
    atomicAdd(&P.counters[2], (unsigned long long)n_shaded);
    atomicAdd(&P.counters[3], (unsigned long long)n_shadow);
    atomicAdd(&P.counters[4], (unsigned long long)n_active);
  }
}

size_t raymarch_lds_bytes(int n_color, int n_alpha)
{
  const size_t need = (size_t)n_color * sizeof(float4) + (size_t)n_alpha * sizeof(float);
  return need <= 60 * 1024 ? need : 0;
}

template <int VT, int SHADE>
static hipError_t launch_vs(const RayMarchParams& p, hipStream_t stream)
{
  const size_t lds = raymarch_lds_bytes(p.n_color, p.n_alpha);
  dim3 grid, block(kBlock);
  if (p.sparse_xy) {
    grid = dim3((unsigned)(((size_t)p.width * p.height + kBlock - 1) / kBlock));
  }
  else {
    grid = dim3((unsigned)((p.width + 15) / 16), (unsigned)((p.height + 15) / 16));
  }
  if (lds)
    hipLaunchKernelGGL((raymarch_kernel<VT, SHADE, true>), grid, block, lds, stream, p);
  else
    hipLaunchKernelGGL((raymarch_kernel<VT, SHADE, false>), grid, block, 0, stream, p);
  return hipGetLastError();
}

template <int VT>
static hipError_t launch_v(const RayMarchParams& p, hipStream_t stream)
{
  switch (p.shading) {
  case 0: return launch_vs<VT, 0>(p, stream);
  case 1: return launch_vs<VT, 1>(p, stream);
  default: return launch_vs<VT, 2>(p, stream);
  }
}

hipError_t launch_raymarch(const RayMarchParams& p, hipStream_t stream)
{
  switch (p.vol.type) {
  case VOX_U8: return launch_v<VOX_U8>(p, stream);
  case VOX_I8: return launch_v<VOX_I8>(p, stream);
  case VOX_U16: return launch_v<VOX_U16>(p, stream);
  case VOX_I16: return launch_v<VOX_I16>(p, stream);
  case VOX_F32: return launch_v<VOX_F32>(p, stream);
  default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// volume relayout: linear (x fastest) -> yz-tiled rows with a replicated pad element
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

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void relayout_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int nx, int ny, int row_stride,
                                                      int tiles_y, int z0, int nz_chunk)
{
  // grid: x = ceil(row_stride / 256), y = ny, z = nz_chunk ; src holds slices [z0, z0 + nz_chunk)
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, zl = blockIdx.z;
  if (x >= row_stride || zl >= nz_chunk) return;
  const int z = z0 + zl;
  const int xs = min(x, nx - 1);
  const TI v = src[(size_t)xs + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)zl)];
  const size_t row = (size_t)(((z >> 3) * tiles_y + (y >> 3)) * 64 + (z & 7) * 8 + (y & 7));
  dst[row * (size_t)row_stride + (size_t)x] = Conv<TI, TO>::cv(v);
}

template <typename TI, typename TO>
static hipError_t relayout_t(const void* src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  dim3 grid((unsigned)((vd.row_stride + 255) / 256), (unsigned)vd.ny, (unsigned)nzc);
  hipLaunchKernelGGL((relayout_kernel<TI, TO>), grid, dim3(256), 0, stream, (const TI*)src, (TO*)dst, vd.nx, vd.ny, vd.row_stride,
                     vd.tiles_y, z0, nzc);
  return hipGetLastError();
}

hipError_t launch_relayout(const void* src, int vt, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  switch (vt) {
  case 100: return relayout_t<unsigned char, unsigned char>(src, dst, vd, z0, nzc, stream);
  case 101: return relayout_t<signed char, signed char>(src, dst, vd, z0, nzc, stream);
  case 200: return relayout_t<unsigned short, unsigned short>(src, dst, vd, z0, nzc, stream);
  case 201: return relayout_t<short, short>(src, dst, vd, z0, nzc, stream);
  case 300: return relayout_t<unsigned int, float>(src, dst, vd, z0, nzc, stream);
  case 301: return relayout_t<int, float>(src, dst, vd, z0, nzc, stream);
  case 400: return relayout_t<float, float>(src, dst, vd, z0, nzc, stream);
  case 500: return relayout_t<double, float>(src, dst, vd, z0, nzc, stream);
  default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// sparse-sampling mask: noise tile slice staged in LDS, keep test, wave64 ballot + prefix compaction
// (generate_mask.cu:55-96; the reference compacts with thrust::remove, which keeps pixel order - so does this)
// ------------------------------------------------------------------------------------------------------------------
// __expf restated as a fixed sequence of IEEE basic operations, so the integer keep/discard decision is reproducible
// bit for bit (2^n * 2^f with a degree-6 Horner polynomial; max relative error 2e-7, inside __expf's own bound)
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

__device__ __forceinline__ bool mask_keep(const SparseMaskParams& p, unsigned int i, int& x, int& y)
{
  x = (int)(i % (unsigned int)p.width);
  y = (int)(i / (unsigned int)p.width);
  const int xy = p.noise_xy;
  const float val = p.noise[(size_t)(y % xy) * xy * 64 + (size_t)(x % xy) * 64 + (size_t)(p.frame_index % 64)];
  const float aspect = (float)p.width / p.height;
  const float fx = ((float)x / p.width - p.mean_x);
  const float fy = ((float)y / p.height - p.mean_y) / aspect;
  const float pr = (1.0f - p.base_noise) * exp_det(-0.5f * (fx * fx + fy * fy) * p.sigma_rcp2) + p.base_noise;
  return val < pr;
}

__global__ __launch_bounds__(256) void mask_count_kernel(const SparseMaskParams p)
{
  __shared__ unsigned int wave_cnt[4];
  const unsigned int n = (unsigned int)p.width * (unsigned int)p.height;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  int x, y;
  const bool keep = (i < n) && mask_keep(p, i, x, y);
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
  const unsigned int n = (unsigned int)p.width * (unsigned int)p.height;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  int x = 0, y = 0;
  const bool keep = (i < n) && mask_keep(p, i, x, y);
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

template <bool PACK>
__global__ __launch_bounds__(256) void tiles_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int width, int height,
                                                   int tw, int th, int rank, int world)
{
  // one thread per frame pixel; pixels of foreign tiles exit
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ix >= width || iy >= height) return;
  const int tx = ix / tw, ty = iy / th;
  if ((tx + ty) % world != rank) return;
  const int tiles_x = (width + tw - 1) / tw;
  int slot = 0;
  for (int r = 0; r < ty; ++r) slot += owned_in_row(tiles_x, r, rank, world);
  const int first = ((rank - ty) % world + world) % world;
  slot += (tx - first) / world;
  const size_t pi = (size_t)slot * tw * th + (size_t)(iy - ty * th) * tw + (size_t)(ix - tx * tw);
  const size_t fi = (size_t)iy * width + ix;
  if (PACK) dst[pi] = src[fi];
  else dst[fi] = src[pi];
}

hipError_t launch_pack_tiles(const float* frame, float* dst, int width, int height, int tw, int th, int rank, int world, hipStream_t stream)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL((tiles_kernel<true>), grid, dim3(256), 0, stream, (const float4*)frame, (float4*)dst, width, height, tw, th, rank, world);
  return hipGetLastError();
}
hipError_t launch_unpack_tiles(const float* src, float* frame, int width, int height, int tw, int th, int rank, int world, hipStream_t stream)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  hipLaunchKernelGGL((tiles_kernel<false>), grid, dim3(256), 0, stream, (const float4*)src, (float4*)frame, width, height, tw, th, rank, world);
  return hipGetLastError();
}

} // namespace ovrhip
