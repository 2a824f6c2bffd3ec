/* ovr_oracle.c - CPU oracle for the OVR ray-marching path.  TEST INFRASTRUCTURE ONLY (see ovr_oracle.h).
 *
 * Plain C99 restatement of the reference's in-tree ray marcher, one function per reference function, all fp32.
 * "parity unpinned" for the integration arithmetic (the reference has no tests/goldens and neither of its devices
 * can be built here); host-side pieces are pinned against oracle/_ref (see tests/test_oracle_vs_ref.py).
 *
 * Conventions restated here (and nowhere taken from code outside the reference):
 *  - gdt normalize(v) = (v * 1.f) / sqrt(dot(v,v)), per-component divide      extern/gdt/gdt/math/vec.h:443-448
 *  - gdt clamp(x,lo,hi) = min(max(x,lo),hi); in DEVICE code gdt's min/max are CUDA's fminf/fmaxf, which return the
 *    non-NaN operand (extern/gdt/gdt/gdt.h:118-120), so corrected_value(NaN) = 0 on the device.  C99 fminf/fmaxf
 *    have the same NaN rule, so they are used here.
 *  - xfmPoint(M,p) = madd(p.x, M.vx, madd(p.y, M.vy, madd(p.z, M.vz, M.p)))   extern/gdt/gdt/math/mat/AffineSpace.h:133
 *    (for the diagonal instance transform this is one fma per component)
 *  - CUDA linear texture filtering is restated with full fp32 weights (not the hardware's 8-bit fraction):
 *    x = u*N - 0.5, i = floor(x), f = x - i, both taps clamped to [0, N-1]   (CUDA programming guide, "linear filtering")
 *  - __powf(x,y) is restated as what CUDA documents it to be, exp2f(y * __log2f(x)) (CUDA C programming guide, "Intrinsic
 *    functions": "__powf(x, y) is implemented as exp2f(y * __log2f(x))"), with a float product and libm's exp2f / log2f
 *    standing in for the hardware's ex2.approx / lg2.approx (round 5; until round 4 it was libm's powf - the
 *    only place where the restatement departed from the reference's own structure: 1 - (1 - a)^dt for a <~ 1e-6 is a
 *    difference of two floats next to 1, and powf lands on other 6e-8 steps there than the two-step form does).
 *    ovr_oracle_set_powf_mode(1) / OVR_ORACLE_POWF=libm selects libm's powf again.
 *  - __frcp_rn(x) is restated as 1.f/x (correctly rounded)
 */
#include "ovr_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* Every numeric literal of the reference's integration loop, named, so that tests/test_oracle_vs_ref.py can pin them against the reference's TEXT
 * (tests/golden/ref_literals.json, extracted from the cited lines by tests/golden/make_ref_literals.py; ovr_oracle_literals() hands them out) */
#define LIT_ERT_PRIMARY 0.9999f     /* shaders_raymarching.cu:110  while (... payload.alpha < 0.9999f) */
#define LIT_ERT_SHADOW 0.9999f      /* shaders_raymarching.cu:64 */
#define LIT_SHADOW_STEP_SCALE 10.f  /* shaders_raymarching.cu:221  self.step * 10.f */
#define LIT_MIDPOINT 0.5f           /* shaders_raymarching.cu:67,112  org + 0.5f * (t.x + t.y) * dir */
#define LIT_NEARLY_EQUAL_EPS 1e-7f  /* shaders_common.h:322  nearly_equal(x, y, epsilon = 1e-7f) */
#define LIT_LIGHT_X (-907.108f)     /* params.h:79  light_directional_pos */
#define LIT_LIGHT_Y 2205.875f
#define LIT_LIGHT_Z (-400.0267f)
#define LIT_LIGHT_RGB 2.f           /* shaders_raymarching.cu:138  vec3f light_rgb = vec3f(2.f) */
#define LIT_SHADE_AMBIENT 0.5f      /* shaders_raymarching.cu:157  0.5f + 0.5f * cosNL * light_rgb * (1.f - shadow.alpha) */
#define LIT_SHADE_DIFFUSE 0.5f
#define LIT_TEA_ROUNDS 16           /* random.h:184  tea<16> */
#define LIT_TEA_DELTA 0x9e3779b9u   /* random.h:158-160 */
#define LIT_TEA_K0 0xa341316cu
#define LIT_TEA_K1 0xc8013ea4u
#define LIT_TEA_K2 0xad90777du
#define LIT_TEA_K3 0x7e95761eu
#define LIT_TEA_TOFLOAT 2.3283064365386962890625e-10f /* random.h:185 */

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
/* vec.h:428-433; nvcc contracts a*b+c into fma by default (-fmad=true), restated as an explicit fma chain */
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline v3 v3_cross(v3 a, v3 b) /* vec.h:413-418 */
{
  return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 v3_normalize(v3 v) /* vec.h:443-448 */
{
  const float l = sqrtf(v3_dot(v, v));
  return v3_make((v.x * 1.f) / l, (v.y * 1.f) / l, (v.z * 1.f) / l);
}
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float clamp01(float x) { return clampf(x, 0.f, 1.f); } /* corrected_value, shaders_common.h:96-104 */

/* ------------------------------------------------------------------------------------------------ */
/* RandomTEA - ovr/common/random/random.h:146-188                                                   */
/* ------------------------------------------------------------------------------------------------ */
void ovr_oracle_tea_floats(uint32_t* pv0, uint32_t* pv1, float out[2])
{
  uint32_t v0 = *pv0, v1 = *pv1, sum = 0;
  for (int i = 0; i < LIT_TEA_ROUNDS; i++) {
    sum += LIT_TEA_DELTA;
    v0 += ((v1 << 4) + LIT_TEA_K0) ^ (v1 + sum) ^ ((v1 >> 5) + LIT_TEA_K1);
    v1 += ((v0 << 4) + LIT_TEA_K2) ^ (v0 + sum) ^ ((v0 >> 5) + LIT_TEA_K3);
  }
  *pv0 = v0;
  *pv1 = v1;
  const float tofloat = LIT_TEA_TOFLOAT; /* 1/2^32 */
  out[0] = (float)v0 * tofloat;
  out[1] = (float)v1 * tofloat;
}

/* ------------------------------------------------------------------------------------------------ */
/* camera basis - ovr/devices/optix7/device_impl.cpp:125-144                                        */
/* ------------------------------------------------------------------------------------------------ */
void ovr_oracle_camera_basis(const float from[3], const float at[3], const float up[3], float fovy, int width, int height,
                             float out[12])
{
  const v3 f = v3_make(from[0], from[1], from[2]);
  const v3 a = v3_make(at[0], at[1], at[2]);
  const v3 u = v3_make(up[0], up[1], up[2]);
  const float t = 2.f * tanf(fovy * 0.5f * (float)M_PI / 180.f);
  const float aspect = (float)width / (float)height;
  const v3 dir = v3_normalize(v3_sub(a, f));
  const v3 hor = v3_scale(t * aspect, v3_normalize(v3_cross(dir, u)));
  const v3 c = v3_cross(hor, dir);
  const v3 ver = v3_make(c.x / aspect, c.y / aspect, c.z / aspect);
  out[0] = f.x; out[1] = f.y; out[2] = f.z;
  out[3] = dir.x; out[4] = dir.y; out[5] = dir.z;
  out[6] = hor.x; out[7] = hor.y; out[8] = hor.z;
  out[9] = ver.x; out[10] = ver.y; out[11] = ver.z;
}

/* ------------------------------------------------------------------------------------------------ */
/* intersect_box vs [0,1]^3 - ovr/devices/optix7/shaders_common.h:156-184                           */
/* ------------------------------------------------------------------------------------------------ */
int ovr_oracle_intersect_box(float* pt0, float* pt1, const float org[3], const float dir[3])
{
  float t0 = *pt0, t1 = *pt1;
  float tlo[3], thi[3];
  for (int k = 0; k < 3; ++k) {
    const int is_small = fabsf(dir[k]) < FLT_MIN;
    const float rcp = 1.f / dir[k];
    tlo[k] = is_small ? FLT_MAX : (0.f - org[k]) * rcp;
    thi[k] = is_small ? -FLT_MAX : (1.f - org[k]) * rcp;
  }
  const float n0 = fminf(tlo[0], thi[0]), n1 = fminf(tlo[1], thi[1]), n2 = fminf(tlo[2], thi[2]);
  const float f0 = fmaxf(tlo[0], thi[0]), f1 = fmaxf(tlo[1], thi[1]), f2 = fmaxf(tlo[2], thi[2]);
  t0 = fmaxf(t0, fmaxf(fmaxf(n0, n1), n2));
  t1 = fminf(t1, fminf(fminf(f0, f1), f2));
  *pt0 = t0;
  *pt1 = t1;
  return t1 > t0;
}

/* ------------------------------------------------------------------------------------------------ */
/* integer_normalize - ovr/devices/optix7/array.h:68-106                                            */
/* ------------------------------------------------------------------------------------------------ */
float ovr_oracle_integer_normalize(float value, int type)
{
  switch (type) {
  case OVR_ORACLE_UINT8: return (float)(uint8_t)value / 255.f;
  case OVR_ORACLE_INT8: { float n = (float)(int8_t)value / 127.f; return n < -1.f ? -1.f : n; }
  case OVR_ORACLE_UINT16: return (float)(uint16_t)value / 65535.f;
  case OVR_ORACLE_INT16: { float n = (float)(int16_t)value / 32767.f; return n < -1.f ? -1.f : n; }
  case OVR_ORACLE_UINT32: return (float)(uint32_t)value / (float)UINT32_MAX;
  case OVR_ORACLE_INT32: { float n = (float)(int32_t)value / (float)INT32_MAX; return n < -1.f ? -1.f : n; }
  case OVR_ORACLE_FLOAT: return value;
  case OVR_ORACLE_DOUBLE: return (float)value;
  default: return value;
  }
}

/* The value a texture read returns for one voxel: array.cpp:300-306 (float: element; integer: normalized float) and
 * array.cpp:335-345 (u16 / i16 / f64 are converted to RAW float on the host and sampled as a float texture). */
static inline float voxel_value(const ovr_oracle_scene* s, size_t idx)
{
  switch (s->value_type) {
  case OVR_ORACLE_UINT8: return (float)((const uint8_t*)s->volume)[idx] / 255.f;
  case OVR_ORACLE_INT8: { float n = (float)((const int8_t*)s->volume)[idx] / 127.f; return n < -1.f ? -1.f : n; }
  case OVR_ORACLE_UINT16: return (float)((const uint16_t*)s->volume)[idx];
  case OVR_ORACLE_INT16: return (float)((const int16_t*)s->volume)[idx];
  case OVR_ORACLE_UINT32: return (float)((const uint32_t*)s->volume)[idx] / (float)UINT32_MAX;
  case OVR_ORACLE_INT32: { float n = (float)((const int32_t*)s->volume)[idx] / (float)INT32_MAX; return n < -1.f ? -1.f : n; }
  case OVR_ORACLE_FLOAT: return ((const float*)s->volume)[idx];
  case OVR_ORACLE_DOUBLE: return (float)((const double*)s->volume)[idx];
  default: return 0.f;
  }
}

/* the ValueType the device-side volume ends up with (array.cpp:322-347): u16/i16/f64 become FLOAT */
static inline int device_value_type(int t)
{
  if (t == OVR_ORACLE_UINT16 || t == OVR_ORACLE_INT16 || t == OVR_ORACLE_DOUBLE) return OVR_ORACLE_FLOAT;
  return t;
}

static inline float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }

/* ------------------------------------------------------------------------------------------------ */
/* sample_volume_object_space - shaders_common.h:186-193                                            */
/* ------------------------------------------------------------------------------------------------ */
float ovr_oracle_sample_volume(const ovr_oracle_scene* s, const float pin[3])
{
  int i0[3], i1[3];
  float fr[3];
  for (int k = 0; k < 3; ++k) {
    const int n = s->dims[k];
    const float p = clamp01(pin[k]);
    /* cell-centred: texel k centre at (k+.5)/N  |  vertex-centred: texel k at k/(N-1) */
    const float x = (s->grid_convention == OVR_ORACLE_GRID_VERTEX_CENTRED) ? p * (float)(n - 1) : fmaf(p, (float)n, -0.5f);
    const float fl = floorf(x);
    fr[k] = x - fl;
    int a = (int)fl, b = (int)fl + 1;
    if (a < 0) a = 0;
    if (a > n - 1) a = n - 1;
    if (b < 0) b = 0;
    if (b > n - 1) b = n - 1;
    i0[k] = a;
    i1[k] = b;
  }
  const size_t nx = (size_t)s->dims[0], ny = (size_t)s->dims[1];
#define VOX(ix, iy, iz) voxel_value(s, (size_t)(ix) + nx * ((size_t)(iy) + ny * (size_t)(iz)))
  const float c000 = VOX(i0[0], i0[1], i0[2]), c100 = VOX(i1[0], i0[1], i0[2]);
  const float c010 = VOX(i0[0], i1[1], i0[2]), c110 = VOX(i1[0], i1[1], i0[2]);
  const float c001 = VOX(i0[0], i0[1], i1[2]), c101 = VOX(i1[0], i0[1], i1[2]);
  const float c011 = VOX(i0[0], i1[1], i1[2]), c111 = VOX(i1[0], i1[1], i1[2]);
#undef VOX
  const float c00 = lerpf(c000, c100, fr[0]), c10 = lerpf(c010, c110, fr[0]);
  const float c01 = lerpf(c001, c101, fr[0]), c11 = lerpf(c011, c111, fr[0]);
  const float c0 = lerpf(c00, c10, fr[1]), c1 = lerpf(c01, c11, fr[1]);
  return lerpf(c0, c1, fr[2]);
}

/* ------------------------------------------------------------------------------------------------ */
/* compute_volume_gradient_object_space - shaders_common.h:195-215; stp = 1/dims (shaders_raymarching.cu:103) */
/* ------------------------------------------------------------------------------------------------ */
void ovr_oracle_gradient(const ovr_oracle_scene* s, const float c[3], float v, float out[3])
{
  for (int k = 0; k < 3; ++k) {
    /* one voxel in normalized object coordinates */
    float stp = (s->grid_convention == OVR_ORACLE_GRID_VERTEX_CENTRED) ? 1.f / (float)(s->dims[k] - 1) : 1.f / (float)s->dims[k];
    if (c[k] + stp > 1.f) stp *= -1.f;
    float p[3] = { c[0], c[1], c[2] };
    p[k] = c[k] + stp;
    out[k] = (ovr_oracle_sample_volume(s, p) - v) / stp;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* transfer function                                                                                 */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  float lower, upper, scale; /* volume.cpp:131-145 */
} tfn_range;

/* compute_scalar_range (array.cpp:27-66: max from numeric_limits::lowest(), min from ::max(), std::max / std::min in array
 * order) followed by cuda_scalar_range (array.cpp:92-108: integer types are integer_normalize<float, T>'d, floating types cast
 * to float) on the array the device volume is created from: u16 / i16 / f64 have been converted to float by then
 * (array.cpp:335-345), so their range is the range of the converted floats.  voxel_value() returns exactly those per-voxel
 * values, and both conversions are monotone, so min / max of voxel_value equal the normalized min / max. */
void ovr_oracle_data_range(const ovr_oracle_scene* s, float out[2])
{
  const size_t n = (size_t)s->dims[0] * (size_t)s->dims[1] * (size_t)s->dims[2];
  float lo = FLT_MAX, hi = -FLT_MAX; /* numeric_limits<float>::max() / lowest() */
  for (size_t i = 0; i < n; ++i) {
    const float f = voxel_value(s, i);
    hi = (hi < f) ? f : hi; /* std::max(v, value) */
    lo = (f < lo) ? f : lo; /* std::min(v, value) */
  }
  out[0] = lo;
  out[1] = hi;
}

static tfn_range make_tfn_range(const ovr_oracle_scene* s)
{
  tfn_range r;
  const int dt = device_value_type(s->value_type);
  if (s->tfn_range[1] >= s->tfn_range[0]) { /* volume.cpp:135-142 */
    r.upper = ovr_oracle_integer_normalize(s->tfn_range[1], dt);
    r.lower = ovr_oracle_integer_normalize(s->tfn_range[0], dt);
  }
  else { /* an invalid range (the default (1, -1)) keeps what the volume was loaded with: its data range (array.cpp:297) */
    float dr[2] = { s->data_range[0], s->data_range[1] };
    if (!s->have_data_range) ovr_oracle_data_range(s, dr);
    r.lower = dr[0];
    r.upper = dr[1];
  }
  r.scale = 1.f / (r.upper - r.lower); /* volume.cpp:145 */
  return r;
}

/* array1d_nodal - shaders_common.h:311-319: lookup coordinate (v*(N-1)+.5)/N into a linear-filtered, clamped 1D texture
 * == nodal lerp between entries floor(v*(N-1)) and +1.  Colours: volume.cpp:110-129 builds vec4f(r,g,b,1) from the flat
 * RGB triples and takes alpha = o[2i+1] from the (position, alpha) pairs. */
static inline void tfn_coord(float v, int n, int* i0, int* i1, float* f)
{
  v = clamp01(v);
  const float x = v * (float)(n - 1);
  const float fl = floorf(x);
  int a = (int)fl, b = (int)fl + 1;
  if (a > n - 1) a = n - 1;
  if (b > n - 1) b = n - 1;
  *i0 = a;
  *i1 = b;
  *f = x - fl;
}

static inline float tfn_alpha_at(const ovr_oracle_scene* s, const tfn_range* r, float sample)
{
  const float v = (clampf(sample, r->lower, r->upper) - r->lower) * r->scale; /* shaders_common.h:363 */
  int i0, i1;
  float f;
  tfn_coord(v, s->n_alphas, &i0, &i1, &f);
  return lerpf(s->tfn_alphas[2 * i0 + 1], s->tfn_alphas[2 * i1 + 1], f);
}

static inline void tfn_rgba_at(const ovr_oracle_scene* s, const tfn_range* r, float sample, float rgba[4])
{
  const float v = (clampf(sample, r->lower, r->upper) - r->lower) * r->scale;
  int i0, i1;
  float f;
  tfn_coord(v, s->n_colors, &i0, &i1, &f);
  for (int c = 0; c < 3; ++c) rgba[c] = lerpf(s->tfn_colors[3 * i0 + c], s->tfn_colors[3 * i1 + c], f);
  tfn_coord(v, s->n_alphas, &i0, &i1, &f);
  rgba[3] = lerpf(s->tfn_alphas[2 * i0 + 1], s->tfn_alphas[2 * i1 + 1], f);
}

void ovr_oracle_sample_tfn(const ovr_oracle_scene* s, float sample, float rgba[4])
{
  const tfn_range r = make_tfn_range(s);
  tfn_rgba_at(s, &r, sample, rgba);
}

/* __powf of shaders_raymarching.cu:64-66,118-122.  mode 0 (default): exp2f(y * log2f(x)) - CUDA's documented definition of
 * __powf, the product rounded to float (the kernel's v_exp_f32(y * v_log_f32(x)) has the same structure);
 * mode 1: libm's powf (rounds x^y once; the restatement of rounds 1-4).  -1 = not chosen yet: OVR_ORACLE_POWF=libm|exp2. */
static int g_powf_mode = -1;
int ovr_oracle_set_powf_mode(int mode)
{
  const int old = g_powf_mode;
  g_powf_mode = (mode == 1 || mode == 2) ? mode : 0;
  return old < 0 ? 0 : old;
}
static inline int powf_mode(void)
{
  if (g_powf_mode < 0) {
    const char* e = getenv("OVR_ORACLE_POWF");
    g_powf_mode = (e && strcmp(e, "libm") == 0) ? 1 : (e && strcmp(e, "det") == 0) ? 2 : 0;
  }
  return g_powf_mode;
}
int ovr_oracle_get_powf_mode(void) { return powf_mode(); }

/* mode 2 - a log2 / exp2 pair that is the SAME float arithmetic on every machine: fmaf Horner chains, integer exponent handling, no library call.
 * Neither libm's log2f / exp2f nor the GPU's v_log_f32 / v_exp_f32 nor CUDA's lg2.approx / ex2.approx round the last bit alike, and 1 - (1 - a)^dt sits on a
 * 6e-8 grid next to 1 where that bit decides whether a sample has opacity 0 (is it shaded?) or an alpha reaches 0.9999 (does the ray end one step
 * earlier?).  The HIP library can be built with the same pair (-DOVR_PARITY_EXACT=1: libovr_hip_parity.so, a test instrument, not the product): then every
 * count equals the oracle's EXACTLY (tests/test_parity_exact_gpu.py) - the proof that the transcendental's last bit is the only thing the tolerated
 * differences of the parity tests come from.  ~1 ulp each, like the hardware instructions; coefficients: Chebyshev fits, tests/golden/make_detpow.py. */
static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
float ovr_oracle_det_log2f(float x)
{
  if (!(x > 0.f)) return x == 0.f ? -INFINITY : NAN;      /* 0 -> -inf, negative / NaN -> NaN */
  if (x > FLT_MAX) return x;                               /* +inf */
  uint32_t ix = float_to_bits(x);
  int e = (int)(ix >> 23) - 127;
  if (e == -127) { ix = float_to_bits(x * 8388608.f); e = (int)(ix >> 23) - 127 - 23; } /* subnormal: scale by 2^23 (exact) */
  float m = bits_to_float((ix & 0x007fffffu) | 0x3f800000u); /* [1, 2) */
  if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }           /* [sqrt(1/2), sqrt(2)] */
  const float f = m - 1.f;                                     /* exact */
  float p = -0x1.b8f078p-4f;                                   /* log2(1 + f) / f on [-0.294, 0.415], degree 9 */
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
float ovr_oracle_det_exp2f(float m)
{
  if (m != m) return m;
  if (m >= 128.f) return INFINITY;
  if (m < -126.f) return 0.f;                                  /* results below the normal range are flushed (2^-126 itself is kept) */
  const float n = floorf(m + 0.5f);
  const float r = m - n;                                       /* [-0.5, 0.5], exact */
  float p = 0x1.00c0e4p-16f;                                   /* 2^r, degree 7 */
  p = fmaf(p, r, 0x1.446c7ap-13f);
  p = fmaf(p, r, 0x1.5d8776p-10f);
  p = fmaf(p, r, 0x1.3b29d8p-7f);
  p = fmaf(p, r, 0x1.c6b08ep-5f);
  p = fmaf(p, r, 0x1.ebfbep-3f);
  p = fmaf(p, r, 0x1.62e43p-1f);
  p = fmaf(p, r, 1.f);
  const int in = (int)n;                                       /* [-126, 128] */
  const int h = in / 2;                                        /* two exact power-of-two factors: each in range */
  const float s = p * bits_to_float((uint32_t)(h + 127) << 23) * bits_to_float((uint32_t)(in - h + 127) << 23);
  return s < FLT_MIN ? 0.f : s;                                /* p < 1 can push n = -126 below the normal range: flushed like the rest */
}
float ovr_oracle_det_powf(float x, float y) { return ovr_oracle_det_exp2f(y * ovr_oracle_det_log2f(x)); }

static inline float fast_powf(float x, float y)
{
  const int mode = powf_mode();
  if (mode == 1) return powf(x, y);
  if (mode == 2) return ovr_oracle_det_powf(x, y);
  const float l = log2f(x);
  const float m = y * l;
  return exp2f(m);
}

/* shaders_raymarching.cu:118-122 (and :64-66 in the shadow march) with nearly_equal (shaders_common.h:321-327) and corrected_value */
float ovr_oracle_opacity_correction(float alpha, float base, float dt)
{
  const float adj = base * dt;
  if (!(fabsf(adj - 1.f) < LIT_NEARLY_EQUAL_EPS)) alpha = clamp01(1.f - fast_powf(1.f - alpha, adj));
  return alpha;
}

/* the literals above, in the order of oracle.py::LITERAL_NAMES (doubles: the uint32 TEA constants are exact in a double) */
int ovr_oracle_literals(double* out, int capacity)
{
  const double v[] = { LIT_ERT_PRIMARY, LIT_ERT_SHADOW, LIT_SHADOW_STEP_SCALE, LIT_MIDPOINT, LIT_NEARLY_EQUAL_EPS, LIT_LIGHT_X, LIT_LIGHT_Y, LIT_LIGHT_Z, LIT_LIGHT_RGB,
                       LIT_SHADE_AMBIENT, LIT_SHADE_DIFFUSE, LIT_TEA_ROUNDS, LIT_TEA_DELTA, LIT_TEA_K0, LIT_TEA_K1, LIT_TEA_K2, LIT_TEA_K3, LIT_TEA_TOFLOAT,
                       FLT_MIN /* float_small, math_def.h:57 */, FLT_MAX /* float_large, math_def.h:56 */ };
  const int n = (int)(sizeof(v) / sizeof(v[0]));
  for (int i = 0; i < n && i < capacity; ++i) out[i] = v[i];
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* per-frame constants                                                                               */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  const ovr_oracle_scene* s;
  tfn_range tr;
  v3 inv_scale;  /* wto.l diagonal: 1/(spacing*extent)                     device_impl.cpp:288-296 */
  v3 wto_p;      /* wto.p = -(origin/scale)                                                            */
  v3 scale;      /* otw.l diagonal                                                                      */
  float step, base; /* volume.cpp:172-179 */
  v3 cam_pos, cam_dir, cam_hor, cam_ver;
  float wtc_it[9]; /* inverse-transpose of world_to_camera linear part (row-major rows r0,r1,r2 as used by xfmVector) */
  v3 otw_it;       /* inverse-transpose of otw.l (diagonal) */
  v3 light;        /* normalize(params.h:79) */
} frame_consts;

static void make_frame_consts(const ovr_oracle_scene* s, frame_consts* fc)
{
  fc->s = s;
  fc->tr = make_tfn_range(s);
  float ext[3];
  for (int k = 0; k < 3; ++k)
    ext[k] = (s->grid_convention == OVR_ORACLE_GRID_VERTEX_CENTRED) ? (float)(s->dims[k] - 1) : (float)s->dims[k];
  fc->scale = v3_make(s->grid_spacing[0] * ext[0], s->grid_spacing[1] * ext[1], s->grid_spacing[2] * ext[2]);
  /* rcp(affine) = (il, -(il*p))  extern/gdt/gdt/math/mat/AffineSpace.h:114 ; diagonal inverse = adjoint/det reduces to 1/s
     up to rounding - restated as a correctly rounded reciprocal */
  fc->inv_scale = v3_make(1.f / fc->scale.x, 1.f / fc->scale.y, 1.f / fc->scale.z);
  fc->wto_p = v3_make(-(fc->inv_scale.x * s->grid_origin[0]), -(fc->inv_scale.y * s->grid_origin[1]),
                      -(fc->inv_scale.z * s->grid_origin[2]));
  fc->otw_it = fc->inv_scale; /* inverse().transposed() of a diagonal matrix */
  fc->step = 1.f / s->sampling_rate;
  fc->base = 1.f;
  float b[12];
  ovr_oracle_camera_basis(s->cam_from, s->cam_at, s->cam_up, s->fovy, s->width, s->height, b);
  fc->cam_pos = v3_make(b[0], b[1], b[2]);
  fc->cam_dir = v3_make(b[3], b[4], b[5]);
  fc->cam_hor = v3_make(b[6], b[7], b[8]);
  fc->cam_ver = v3_make(b[9], b[10], b[11]);
  /* get_xfm_world_to_camera (shaders_common.h:276-289): columns vx=(x.x,y.x,z.x), vy=(x.y,y.y,z.y), vz=(x.z,y.z,z.z) with
     x = normalize(horizontal), y = normalize(vertical), z = -normalize(direction).  xfmNormal(wtc, n) multiplies by
     inverse().transposed() (LinearSpace.h:215-224,321), restated with the adjoint/det formulas. */
  const v3 x = v3_normalize(fc->cam_hor), y = v3_normalize(fc->cam_ver);
  const v3 zz = v3_normalize(fc->cam_dir);
  const v3 z = v3_make(-zz.x, -zz.y, -zz.z);
  const v3 vx = v3_make(x.x, y.x, z.x), vy = v3_make(x.y, y.y, z.y), vz = v3_make(x.z, y.z, z.z);
  const float det = v3_dot(vx, v3_cross(vy, vz));
  /* adjoint() = LinearSpace3(cross(vy,vz), cross(vz,vx), cross(vx,vy)).transposed(); inverse = adjoint/det;
     inverse().transposed() therefore has COLUMNS cross(vy,vz)/det, cross(vz,vx)/det, cross(vx,vy)/det */
  const v3 c0 = v3_cross(vy, vz), c1 = v3_cross(vz, vx), c2 = v3_cross(vx, vy);
  const v3 m0 = v3_make(c0.x / det, c0.y / det, c0.z / det);
  const v3 m1 = v3_make(c1.x / det, c1.y / det, c1.z / det);
  const v3 m2 = v3_make(c2.x / det, c2.y / det, c2.z / det);
  fc->wtc_it[0] = m0.x; fc->wtc_it[1] = m0.y; fc->wtc_it[2] = m0.z; /* column vx of the normal matrix */
  fc->wtc_it[3] = m1.x; fc->wtc_it[4] = m1.y; fc->wtc_it[5] = m1.z; /* column vy */
  fc->wtc_it[6] = m2.x; fc->wtc_it[7] = m2.y; fc->wtc_it[8] = m2.z; /* column vz */
  fc->light = v3_normalize(v3_make(LIT_LIGHT_X, LIT_LIGHT_Y, LIT_LIGHT_Z)); /* params.h:79 */
}

/* xfmVector(M, a) = madd(a.x, vx, madd(a.y, vy, a.z*vz))  LinearSpace.h:320 */
static inline v3 xfm_vector_cols(const float m[9], v3 a)
{
  return v3_make(fmaf(a.x, m[0], fmaf(a.y, m[3], a.z * m[6])), fmaf(a.x, m[1], fmaf(a.y, m[4], a.z * m[7])),
                 fmaf(a.x, m[2], fmaf(a.y, m[5], a.z * m[8])));
}

static inline v3 to_object(const frame_consts* fc, v3 p) /* xfmPoint(wto, p), diagonal */
{
  return v3_make(fmaf(p.x, fc->inv_scale.x, fc->wto_p.x), fmaf(p.y, fc->inv_scale.y, fc->wto_p.y),
                 fmaf(p.z, fc->inv_scale.z, fc->wto_p.z));
}

/* test access to the transform restatements (pinned against the reference's gdt math, tests/test_oracle_vs_ref.py):
 * out = { xfmPoint(wto,p), xfmVector(wto,p), xfmNormal(otw,p), normalize(p) } for otw = translate(origin)*scale(scale) */
void ovr_oracle_xfm_probe(const float origin[3], const float scale[3], const float p[3], float out[12])
{
  ovr_oracle_scene s;
  memset(&s, 0, sizeof(s));
  frame_consts fc;
  memset(&fc, 0, sizeof(fc));
  fc.scale = v3_make(scale[0], scale[1], scale[2]);
  fc.inv_scale = v3_make(1.f / fc.scale.x, 1.f / fc.scale.y, 1.f / fc.scale.z);
  fc.wto_p = v3_make(-(fc.inv_scale.x * origin[0]), -(fc.inv_scale.y * origin[1]), -(fc.inv_scale.z * origin[2]));
  fc.otw_it = fc.inv_scale;
  const v3 q = v3_make(p[0], p[1], p[2]);
  const v3 a = to_object(&fc, q);
  const v3 b = v3_make(q.x * fc.inv_scale.x, q.y * fc.inv_scale.y, q.z * fc.inv_scale.z);
  const v3 c = v3_make(q.x * fc.otw_it.x, q.y * fc.otw_it.y, q.z * fc.otw_it.z);
  const v3 d = v3_normalize(q);
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = b.x; out[4] = b.y; out[5] = b.z;
  out[6] = c.x; out[7] = c.y; out[8] = c.z; out[9] = d.x; out[10] = d.y; out[11] = d.z;
}

/* ------------------------------------------------------------------------------------------------ */
/* raymarching_shadow + __closesthit__volume_shadow - shaders_raymarching.cu:44-85,205-229          */
/* ------------------------------------------------------------------------------------------------ */
static float march_shadow(const frame_consts* fc, v3 org, v3 dir, uint64_t* n_samples)
{
  const ovr_oracle_scene* s = fc->s;
  /* optixTrace(pos, light_dir, tmin = 0, tmax = inf) -> __intersection__volume in object space (shaders_common.h:379-392) */
  const v3 oo = to_object(fc, org);
  const v3 od = v3_make(dir.x * fc->inv_scale.x, dir.y * fc->inv_scale.y, dir.z * fc->inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  const float o[3] = { oo.x, oo.y, oo.z }, d[3] = { od.x, od.y, od.z };
  float alpha = 0.f;
  if (!ovr_oracle_intersect_box(&t0, &t1, o, d)) return alpha;
  const float sampling_scale = fc->step * LIT_SHADOW_STEP_SCALE; /* :221 */
  const float stride = sampling_scale * fc->step; /* :64,:83 */
  if (t0 >= t1) return alpha;
  float tx = t0, ty = fminf(t1, t0 + stride);
  while ((ty > tx) && (alpha < LIT_ERT_SHADOW)) {
    const float tm = LIT_MIDPOINT * (tx + ty);
    const v3 pos = v3_make(fmaf(tm, dir.x, org.x), fmaf(tm, dir.y, org.y), fmaf(tm, dir.z, org.z));
    const v3 po = to_object(fc, pos);
    const float p[3] = { po.x, po.y, po.z };
    const float sample = ovr_oracle_sample_volume(s, p);
    float a = tfn_alpha_at(s, &fc->tr, sample);
    a = ovr_oracle_opacity_correction(a, fc->base, ty - tx);
    alpha = fmaf(1.f - alpha, a, alpha);
    ++*n_samples;
    tx = ty;
    ty = fminf(tx + stride, t1);
  }
  return alpha;
}

/* ------------------------------------------------------------------------------------------------ */
/* raymarching + __closesthit__volume_raymarching + render_raymarching                               */
/* shaders_raymarching.cu:87-171,178-203,260-321                                                     */
/* ------------------------------------------------------------------------------------------------ */
static void trace_ray(const frame_consts* fc, v3 org, v3 dir, float out_rgba[4], float out_grad[3], ovr_oracle_counters* cnt)
{
  const ovr_oracle_scene* s = fc->s;
  float alpha = 0.f;
  v3 color = v3_make(0, 0, 0), gradient = v3_make(0, 0, 0);
  cnt->rays++;

  const v3 oo = to_object(fc, org);
  const v3 od = v3_make(dir.x * fc->inv_scale.x, dir.y * fc->inv_scale.y, dir.z * fc->inv_scale.z);
  float t0 = 0.f, t1 = FLT_MAX;
  const float o[3] = { oo.x, oo.y, oo.z }, d[3] = { od.x, od.y, od.z };
  if (ovr_oracle_intersect_box(&t0, &t1, o, d) && t0 < t1) {
    float tx = t0, ty = fminf(t1, t0 + fc->step);
    while ((ty > tx) && (alpha < LIT_ERT_PRIMARY)) {
      const float tm = LIT_MIDPOINT * (tx + ty);
      const v3 pos = v3_make(fmaf(tm, dir.x, org.x), fmaf(tm, dir.y, org.y), fmaf(tm, dir.z, org.z));
      const v3 po = to_object(fc, pos);
      const float p[3] = { po.x, po.y, po.z };
      const float sample = ovr_oracle_sample_volume(s, p);
      float rgba[4];
      tfn_rgba_at(s, &fc->tr, sample, rgba);
      const float a_tf = rgba[3];
      rgba[3] = ovr_oracle_opacity_correction(rgba[3], fc->base, ty - tx);
      cnt->samples++;
      if (rgba[3] > 0.f) cnt->shaded_samples++;
      /* a sample whose table opacity is > 0 but whose CORRECTED opacity lies within four 6e-8 steps of 0: whether it counts as shaded is decided
         by the last bit of the pow (libm here, v_exp / v_log on the GPU, ex2.approx / lg2.approx in the reference) */
      if (a_tf > 0.f && rgba[3] <= 0x1p-22f && rgba[3] != a_tf) cnt->borderline_samples++;

      v3 n_c = v3_make(0, 0, 0);
      /* skip_zero_opacity: a sample with corrected opacity exactly 0 adds fma(tr * clamp01(x), 0, acc) == acc to colour,
         gradient and alpha whatever its shading is (clamp01 maps NaN to 0, the colour table is finite), so its gradient
         taps and shadow march can be left out without changing a bit of the frame */
      if (s->shading != OVR_ORACLE_SHADE_NONE && !(s->skip_zero_opacity && !(rgba[3] > 0.f))) {
        float g[3];
        ovr_oracle_gradient(s, p, sample, g);
        const v3 gn = v3_normalize(v3_make(g[0], g[1], g[2]));
        const v3 n_o = v3_make(-gn.x, -gn.y, -gn.z);
        /* xfmNormal(otw, n_o): otw.l is diagonal, inverse-transpose = diag(1/scale) */
        const v3 n_w = v3_normalize(v3_make(n_o.x * fc->otw_it.x, n_o.y * fc->otw_it.y, n_o.z * fc->otw_it.z));
        n_c = v3_normalize(xfm_vector_cols(fc->wtc_it, n_w));
        float shadow = 0.f;
        if (s->shading == OVR_ORACLE_SHADE_FULL) {
          uint64_t ns = 0;
          shadow = march_shadow(fc, pos, fc->light, &ns);
          cnt->shadow_samples += ns;
          if (rgba[3] > 0.f) cnt->shadow_samples_visible += ns;
        }
        const float cosNL = fabsf(v3_dot(fc->light, n_w));
        const float shade = LIT_SHADE_AMBIENT + LIT_SHADE_DIFFUSE * cosNL * LIT_LIGHT_RGB * (1.f - shadow); /* :156-157, light_rgb = 2 (:138) */
        rgba[0] *= shade;
        rgba[1] *= shade;
        rgba[2] *= shade;
      }
      const float tr = 1.f - alpha;
      color.x = fmaf(tr * clamp01(rgba[0]), rgba[3], color.x);
      color.y = fmaf(tr * clamp01(rgba[1]), rgba[3], color.y);
      color.z = fmaf(tr * clamp01(rgba[2]), rgba[3], color.z);
      gradient.x = fmaf(tr * clamp01(n_c.x), rgba[3], gradient.x);
      gradient.y = fmaf(tr * clamp01(n_c.y), rgba[3], gradient.y);
      gradient.z = fmaf(tr * clamp01(n_c.z), rgba[3], gradient.z);
      alpha = fmaf(tr, rgba[3], alpha);
      tx = ty;
      ty = fminf(tx + fc->step, t1);
    }
  }
  /* render_raymarching: background trace always misses (alpha 0, colour 0): alpha = fa + (1-fa)*0;
     alpha_blend = (fg + (1-fa)*0*0)/alpha if alpha > 0 else 0   (shaders_common.h:329-337) */
  const float a = alpha + (1.f - alpha) * 0.f;
  out_rgba[3] = a;
  if (a > 0.f) {
    out_rgba[0] = (color.x + (1.f - alpha) * 0.f * 0.f) / a;
    out_rgba[1] = (color.y + (1.f - alpha) * 0.f * 0.f) / a;
    out_rgba[2] = (color.z + (1.f - alpha) * 0.f * 0.f) / a;
    out_grad[0] = (gradient.x + (1.f - alpha) * 0.f * 0.f) / a;
    out_grad[1] = (gradient.y + (1.f - alpha) * 0.f * 0.f) / a;
    out_grad[2] = (gradient.z + (1.f - alpha) * 0.f * 0.f) / a;
  }
  else {
    out_rgba[0] = out_rgba[1] = out_rgba[2] = 0.f;
    out_grad[0] = out_grad[1] = out_grad[2] = 0.f;
  }
}

void ovr_oracle_trace_ray(const ovr_oracle_scene* s, const float org[3], const float dir[3], float rgba[4], float grad[3],
                          ovr_oracle_counters* counters)
{
  frame_consts fc;
  make_frame_consts(s, &fc);
  ovr_oracle_counters local;
  memset(&local, 0, sizeof(local));
  trace_ray(&fc, v3_make(org[0], org[1], org[2]), v3_make(dir[0], dir[1], dir[2]), rgba, grad, &local);
  if (counters) *counters = local;
}

/* ------------------------------------------------------------------------------------------------ */
/* __raygen__render_frame - shaders_raymarching.cu:323-413                                           */
/* ------------------------------------------------------------------------------------------------ */
int ovr_oracle_tile_owner(int tx, int ty, int tiles_x, int world)
{
  (void)tiles_x;
  /* diagonal round-robin: neighbouring tiles in x AND y go to different ranks */
  return (tx + ty) % world;
}

/* blue-noise pixel jitter (BASELINE C5 / north_star): tile lookup as blue_noise.h:95-99 does it for the sparse mask,
 * value = tile[(y % XY) * XY * T + (x % XY) * T + (slice % T)], T = 64 */
void ovr_oracle_jitter(const ovr_oracle_scene* s, int ix, int iy, int frame_index, int k, float out[2])
{
  const int xy = s->noise_xy;
  const size_t t = (size_t)((((long long)frame_index - 1) * s->spp + k) % 64);
  const int h = xy / 2;
  out[0] = s->noise_tile[(size_t)(iy % xy) * xy * 64 + (size_t)(ix % xy) * 64 + t];
  out[1] = s->noise_tile[(size_t)((iy + h) % xy) * xy * 64 + (size_t)((ix + h) % xy) * 64 + t];
}

static void render_pixel(const frame_consts* fc, int ix, int iy, int frame_index, int frame_accumulation, float* accum,
                         float* out_rgba, float* out_grad, ovr_oracle_counters* cnt)
{
  const ovr_oracle_scene* s = fc->s;
  const int W = s->width, H = s->height;
  const float rsx = 1.f / (float)W, rsy = 1.f / (float)H; /* shaders_common.h:400, device_impl.cpp:242 */
  const float scx = ((float)ix + .5f) * rsx, scy = ((float)iy + .5f) * rsy;
  const uint32_t pixel_index = (uint32_t)ix + (uint32_t)iy * (uint32_t)W;
  uint32_t v0 = (uint32_t)frame_index, v1 = pixel_index; /* RandomTEA(frame_index, pixel_index) :336 */
  float o_a = 0.f;
  v3 o_c = v3_make(0, 0, 0), o_g = v3_make(0, 0, 0);
  const int spp = s->spp;
  for (int k = 0; k < spp; ++k) {
    float sx = scx, sy = scy;
    if (s->pixel_jitter == 1) { /* blue-noise tile, every sample of every frame (progressive accumulation) */
      float r[2];
      ovr_oracle_jitter(s, ix, iy, frame_index, k, r);
      sx += (r[0] - 0.5f) * rsx;
      sy += (r[1] - 0.5f) * rsy;
    }
    else if (spp > 1) {
      float r[2];
      ovr_oracle_tea_floats(&v0, &v1, r);
      sx += (r[0] - 0.5f) * rsx;
      sy += (r[1] - 0.5f) * rsy;
    }
    const float ux = sx - 0.5f, uy = sy - 0.5f;
    const v3 dir = v3_normalize(v3_make(fc->cam_dir.x + ux * fc->cam_hor.x + uy * fc->cam_ver.x,
                                        fc->cam_dir.y + ux * fc->cam_hor.y + uy * fc->cam_ver.y,
                                        fc->cam_dir.z + ux * fc->cam_hor.z + uy * fc->cam_ver.z));
    float rgba[4], g[3];
    trace_ray(fc, fc->cam_pos, dir, rgba, g, cnt);
    o_a += rgba[3];
    o_c = v3_add(o_c, v3_make(rgba[0], rgba[1], rgba[2]));
    o_g = v3_add(o_g, v3_make(g[0], g[1], g[2]));
  }
  const float rspp = 1.f / (float)spp;
  o_a *= rspp;
  o_c = v3_scale(rspp, o_c);
  o_g = v3_scale(rspp, o_g);
  float* px = out_rgba + 4 * (size_t)pixel_index;
  if (frame_accumulation) {
    float* ac = accum + 4 * (size_t)pixel_index;
    if (frame_index == 1) {
      ac[0] = o_c.x; ac[1] = o_c.y; ac[2] = o_c.z; ac[3] = o_a;
      px[0] = o_c.x; px[1] = o_c.y; px[2] = o_c.z; px[3] = o_a;
    }
    else {
      ac[0] += o_c.x; ac[1] += o_c.y; ac[2] += o_c.z; ac[3] += o_a;
      const float fi = (float)frame_index;
      px[0] = ac[0] / fi; px[1] = ac[1] / fi; px[2] = ac[2] / fi; px[3] = ac[3] / fi;
    }
  }
  else {
    px[0] = o_c.x; px[1] = o_c.y; px[2] = o_c.z; px[3] = o_a;
  }
  if (out_grad) {
    float* pg = out_grad + 3 * (size_t)pixel_index;
    pg[0] = o_g.x; pg[1] = o_g.y; pg[2] = o_g.z;
  }
}

/* One frame = a list of work items - 8x8-pixel blocks of the image (dense) or runs of 64 entries of the sparse pixel list -
 * handed out through one atomic cursor to a persistent pool of threads (created on first use, kept between frames; the
 * calling thread works too).  Pixels are independent, so any split gives the same frame. */
typedef struct {
  const frame_consts* fc;
  int frame_index, frame_accumulation;
  float *accum, *out_rgba, *out_grad;
  const int32_t* sparse_xy;
  int64_t n_sparse;
  long long n_items, next;
  int blocks_x;
} frame_job;

static void add_counters(ovr_oracle_counters* a, const ovr_oracle_counters* b)
{
  a->rays += b->rays;
  a->samples += b->samples;
  a->shaded_samples += b->shaded_samples;
  a->shadow_samples += b->shadow_samples;
  a->shadow_samples_visible += b->shadow_samples_visible;
  a->borderline_samples += b->borderline_samples;
}

static void process_items(frame_job* j, ovr_oracle_counters* cnt)
{
  const ovr_oracle_scene* s = j->fc->s;
  const int tw = s->tile_w > 0 ? s->tile_w : s->width, th = s->tile_h > 0 ? s->tile_h : s->height;
  const int tiles_x = (s->width + tw - 1) / tw;
  const int world = s->world > 0 ? s->world : 1;
  for (;;) {
    const long long it = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
    if (it >= j->n_items) break;
    if (j->sparse_xy) { /* the sparse pixel list, restricted to this rank's tiles when the image is sharded */
      const int64_t e = (it + 1) * 64 < j->n_sparse ? (it + 1) * 64 : j->n_sparse;
      for (int64_t i = it * 64; i < e; ++i) {
        const int ix = j->sparse_xy[2 * i], iy = j->sparse_xy[2 * i + 1];
        if (world > 1 && ovr_oracle_tile_owner(ix / tw, iy / th, tiles_x, world) != s->rank) continue;
        render_pixel(j->fc, ix, iy, j->frame_index, j->frame_accumulation, j->accum, j->out_rgba, j->out_grad, cnt);
      }
      continue;
    }
    const int bx = (int)(it % j->blocks_x) * 8, by = (int)(it / j->blocks_x) * 8;
    for (int iy = by; iy < by + 8 && iy < s->height; ++iy)
      for (int ix = bx; ix < bx + 8 && ix < s->width; ++ix) {
        if (world > 1 && ovr_oracle_tile_owner(ix / tw, iy / th, tiles_x, world) != s->rank) continue;
        render_pixel(j->fc, ix, iy, j->frame_index, j->frame_accumulation, j->accum, j->out_rgba, j->out_grad, cnt);
      }
  }
}

#define OVR_ORACLE_MAX_THREADS 512
static struct {
  pthread_mutex_t frame_mtx; /* one frame at a time */
  pthread_mutex_t mtx;
  pthread_cond_t start, done;
  pthread_t th[OVR_ORACLE_MAX_THREADS];
  int n_workers;            /* pool threads (the caller is one more) */
  unsigned long generation; /* bumped once per frame */
  int pending;              /* workers that have not finished the current generation */
  int quit;
  frame_job* job;
  ovr_oracle_counters total;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { 0 }, 0, 0, 0, 0, NULL, { 0, 0, 0, 0, 0 } };

static void* pool_worker(void* arg)
{
  unsigned long seen = (unsigned long)(size_t)arg; /* the generation at the time the pool was (re)built */
  pthread_mutex_lock(&g_pool.mtx);
  for (;;) {
    while (!g_pool.quit && g_pool.generation == seen) pthread_cond_wait(&g_pool.start, &g_pool.mtx);
    if (g_pool.quit) break;
    seen = g_pool.generation;
    frame_job* j = g_pool.job;
    pthread_mutex_unlock(&g_pool.mtx);
    ovr_oracle_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    process_items(j, &cnt);
    pthread_mutex_lock(&g_pool.mtx);
    add_counters(&g_pool.total, &cnt);
    if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
  }
  pthread_mutex_unlock(&g_pool.mtx);
  return NULL;
}

static void pool_resize(int n_workers) /* called with frame_mtx held */
{
  if (n_workers == g_pool.n_workers) return;
  pthread_mutex_lock(&g_pool.mtx);
  g_pool.quit = 1;
  pthread_cond_broadcast(&g_pool.start);
  pthread_mutex_unlock(&g_pool.mtx);
  for (int t = 0; t < g_pool.n_workers; ++t) pthread_join(g_pool.th[t], NULL);
  g_pool.quit = 0;
  g_pool.n_workers = 0;
  for (int t = 0; t < n_workers; ++t) {
    if (pthread_create(&g_pool.th[t], NULL, pool_worker, (void*)(size_t)g_pool.generation) != 0) break;
    g_pool.n_workers = t + 1;
  }
}

void ovr_oracle_render_frame(const ovr_oracle_scene* s, int frame_index, int frame_accumulation, float* accum_rgba,
                             float* out_rgba, float* out_grad, ovr_oracle_counters* counters, int nthreads)
{
  frame_consts fc;
  ovr_oracle_scene sc = *s;
  if (!sc.have_data_range && !(sc.tfn_range[1] >= sc.tfn_range[0])) { /* once per frame, not once per use */
    ovr_oracle_data_range(&sc, sc.data_range);
    sc.have_data_range = 1;
  }
  s = &sc;
  make_frame_consts(s, &fc);
  if (nthreads <= 0) {
    nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nthreads < 1) nthreads = 1;
  }
  if (nthreads > OVR_ORACLE_MAX_THREADS) nthreads = OVR_ORACLE_MAX_THREADS;
  int32_t* sparse = NULL;
  int64_t n_sparse = 0;
  if (s->sparse_sampling) {
    sparse = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)s->width * (size_t)s->height);
    n_sparse = ovr_oracle_sparse_mask(sparse, frame_index, s->width, s->height, s->focus_center, s->focus_scale, s->base_noise,
                                      s->noise_tile, s->noise_xy) / 2;
  }
  frame_job job;
  job.fc = &fc;
  job.frame_index = frame_index;
  job.frame_accumulation = frame_accumulation;
  job.accum = accum_rgba;
  job.out_rgba = out_rgba;
  job.out_grad = out_grad;
  job.sparse_xy = sparse;
  job.n_sparse = n_sparse;
  job.blocks_x = (s->width + 7) / 8;
  job.n_items = sparse ? (n_sparse + 63) / 64 : (long long)job.blocks_x * ((s->height + 7) / 8);
  job.next = 0;

  pthread_mutex_lock(&g_pool.frame_mtx);
  pool_resize(nthreads - 1);
  pthread_mutex_lock(&g_pool.mtx);
  memset(&g_pool.total, 0, sizeof(g_pool.total));
  g_pool.job = &job;
  g_pool.pending = g_pool.n_workers;
  g_pool.generation++;
  pthread_cond_broadcast(&g_pool.start);
  pthread_mutex_unlock(&g_pool.mtx);
  ovr_oracle_counters mine;
  memset(&mine, 0, sizeof(mine));
  process_items(&job, &mine);
  pthread_mutex_lock(&g_pool.mtx);
  while (g_pool.pending > 0) pthread_cond_wait(&g_pool.done, &g_pool.mtx);
  add_counters(&g_pool.total, &mine);
  const ovr_oracle_counters total = g_pool.total;
  g_pool.job = NULL;
  pthread_mutex_unlock(&g_pool.mtx);
  pthread_mutex_unlock(&g_pool.frame_mtx);
  if (counters) *counters = total;
  free(sparse);
}

/* ------------------------------------------------------------------------------------------------ */
/* image_to_rgba8 (4 channels) - ovr/common/imageio.cpp:146-181                                      */
/* ------------------------------------------------------------------------------------------------ */
static inline float std_clamp01(float v) { return (v < 0.f) ? 0.f : (1.f < v) ? 1.f : v; } /* std::clamp semantics */

void ovr_oracle_rgba8(const float* rgba, int width, int height, int flip_vertical, uint8_t* out)
{
  size_t index = 0;
  for (int jj = 0; jj < height; ++jj) {
    const int j = flip_vertical ? height - 1 - jj : jj;
    for (int i = 0; i < width; ++i) {
      const float* in = rgba + 4 * ((size_t)i + (size_t)j * (size_t)width);
      uint8_t* o = out + 4 * (index++);
      for (int c = 0; c < 4; ++c) o[c] = (uint8_t)(std_clamp01(in[c]) * 255.f);
    }
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* float -> half of the reference's EXR output (see ovr_oracle.h)                                    */
/* ------------------------------------------------------------------------------------------------ */
uint16_t ovr_oracle_float_to_half(float f)
{
  union { float f; uint32_t u; } in;
  in.f = f;
  const uint32_t sign = (in.u >> 16) & 0x8000u, e = (in.u >> 23) & 0xffu, m = in.u & 0x7fffffu;
  uint32_t h = 0;
  if (e == 0) h = 0;                                  /* zero and float denormals */
  else if (e == 255) h = 0x7c00u | (m ? 0x200u : 0u); /* infinity / NaN */
  else {
    const int ne = (int)e - 127 + 15;
    if (ne >= 31) h = 0x7c00u;                        /* overflow */
    else if (ne <= 0) {                               /* below the half normal range */
      if (14 - ne <= 24) {
        const uint32_t sig = m | 0x800000u;
        h = sig >> (14 - ne);
        if ((sig >> (13 - ne)) & 1u) h++;
      }
    }
    else {
      h = ((uint32_t)ne << 10) | (m >> 13);
      if (m & 0x1000u) h++; /* the carry may reach the exponent, and infinity */
    }
  }
  return (uint16_t)(sign | h);
}

float ovr_oracle_half_to_float(uint16_t h)
{
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  union { float f; uint32_t u; } out;
  if (e == 0) {
    out.f = (float)m * 5.9604644775390625e-8f; /* m * 2^-24, exact */
    out.u |= sign;
  }
  else if (e == 31) out.u = sign | 0x7f800000u | (m << 13);
  else out.u = sign | ((e + 112u) << 23) | (m << 13);
  return out.f;
}

/* ------------------------------------------------------------------------------------------------ */
/* sparse-sampling mask - generate_mask.cu:55-96,100-120 ; blue_noise.h:81-102                       */
/* ------------------------------------------------------------------------------------------------ */
/* __expf restated as a fixed sequence of IEEE basic operations (range reduction to 2^n * 2^f, degree-6 polynomial in
 * Horner form with explicit fma) so that the keep/discard decision - an integer result - is bit-reproducible on any
 * IEEE machine; relative error 2e-7 near 0 growing to 2e-6 at x = -30 (the rounding of x*log2e), inside __expf's own bound of 2 + |1.16 x| ulp. */
float ovr_oracle_exp_det(float x)
{
  if (x < -87.f) return 0.f;
  if (x > 88.f) x = 88.f;
  const float t = x * 1.44269504088896341f;
  const float n = floorf(t + 0.5f);
  const float f = t - n; /* [-0.5, 0.5] */
  float p = 1.53533031e-4f;
  p = fmaf(p, f, 1.33988696e-3f);
  p = fmaf(p, f, 9.61843120e-3f);
  p = fmaf(p, f, 5.55033022e-2f);
  p = fmaf(p, f, 2.40226504e-1f);
  p = fmaf(p, f, 6.93147182e-1f);
  p = fmaf(p, f, 1.0f);
  union { uint32_t u; float f; } sc;
  sc.u = (uint32_t)((int)n + 127) << 23;
  return p * sc.f;
}

int64_t ovr_oracle_sparse_mask(int32_t* out_xy, int frame_index, int width, int height, const float mean[2],
                               float sigma, float base_noise, const float* noise, int xy)
{
  const float sigma_rcp2 = 1.f / (sigma * sigma); /* generate_mask.cu:90 */
  const float aspect = (float)width / height;
  int64_t n = 0;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const float val = noise[(size_t)(y % xy) * xy * 64 + (size_t)(x % xy) * 64 + (size_t)(frame_index % 64)];
      const float fx = ((float)x / width - mean[0]);
      const float fy = ((float)y / height - mean[1]) / aspect;
      const float p = (1.0f - base_noise) * ovr_oracle_exp_det(-0.5f * (fx * fx + fy * fy) * sigma_rcp2) + base_noise;
      if (val < p) { /* thrust::remove(-1) keeps the survivors in order */
        out_xy[n++] = x;
        out_xy[n++] = y;
      }
    }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* macrocells - ovr/devices/optix7/accel/sp_singlemc.cu:10-54 and :56-97                             */
/* ------------------------------------------------------------------------------------------------ */
void ovr_oracle_macrocell_value_range(const ovr_oracle_scene* s, float* out)
{
  const int W = 16; /* spatial_partition.h:24 MACROCELL_SIZE */
  const int mx = (s->dims[0] + W - 1) / W, my = (s->dims[1] + W - 1) / W, mz = (s->dims[2] + W - 1) / W;
  const size_t nx = (size_t)s->dims[0], ny = (size_t)s->dims[1];
  for (int cz = 0; cz < mz; ++cz)
    for (int cy = 0; cy < my; ++cy)
      for (int cx = 0; cx < mx; ++cx) {
        int b[3] = { cx * W - 1, cy * W - 1, cz * W - 1 }, e[3];
        for (int k = 0; k < 3; ++k) {
          if (b[k] < 0) b[k] = 0;
          e[k] = b[k] + W + 1;
          if (e[k] > s->dims[k]) e[k] = s->dims[k];
        }
        float lo = INFINITY, hi = -INFINITY;
        for (int iz = b[2]; iz < e[2]; ++iz)
          for (int iy = b[1]; iy < e[1]; ++iy)
            for (int ix = b[0]; ix < e[0]; ++ix) {
              const float f = voxel_value(s, (size_t)ix + nx * ((size_t)iy + ny * (size_t)iz));
              lo = fminf(lo, f);
              hi = fmaxf(hi, f);
            }
        float* o = out + 2 * ((size_t)cx + (size_t)mx * ((size_t)cy + (size_t)my * (size_t)cz));
        o[0] = lo;
        o[1] = hi;
      }
}

void ovr_oracle_macrocell_majorant(const ovr_oracle_scene* s, const float* minmax, int n_cells, float* out)
{
  /* tfn.value_range / range_rcp_norm: volume.cpp:147-153 (clipped against the data range - restated with the TF range
     itself, which is what the clip yields whenever the TF range lies inside the data range) */
  const tfn_range r = make_tfn_range(s);
  const float rcp = 1.f / (r.upper - r.lower);
  const int N = s->n_alphas;
  for (int i = 0; i < n_cells; ++i) {
    const float lower = (clampf(minmax[2 * i], r.lower, r.upper) - r.lower) * rcp;
    const float upper = (clampf(minmax[2 * i + 1], r.lower, r.upper) - r.lower) * rcp;
    /* float -> uint32_t in CUDA device code saturates (cvt.rzi.u32.f32): a negative value becomes 0 */
    const float fl = floorf(fmaf(lower, (float)(N - 1), 0.5f)) - 1.f;
    const float fu = floorf(fmaf(upper, (float)(N - 1), 0.5f)) + 1.f;
    uint32_t il = fl < 0.f ? 0u : (uint32_t)fl;
    uint32_t iu = fu < 0.f ? 0u : (uint32_t)fu;
    if (il > (uint32_t)(N - 1)) il = (uint32_t)(N - 1);
    if (iu > (uint32_t)(N - 1)) iu = (uint32_t)(N - 1);
    float op = 0.f;
    for (uint32_t k = il; k <= iu; ++k) op = fmaxf(op, s->tfn_alphas[2 * k + 1]);
    out[i] = op;
  }
}
