/* ovr_oracle.h - CPU oracle for the OVR ray-marching path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's ray-marching
 * algorithm (VIDILabs/open-volume-renderer, in-tree OptiX7 device).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.  The shipped
 * product path (open-volume-renderer_amd/csrc) never includes, links or calls anything here.
 *
 * PARITY STATUS: "parity unpinned" for the ray integration itself - the reference holds no
 * tests, golden images or known-answer vectors for this path, its OptiX
 * device needs nvcc + optix.h and its OSPRay device needs libospray, neither of which exists
 * in this image.  What IS pinned against real reference code (oracle/_ref, built from the
 * reference sources in place by oracle/build_ref.sh): the 8-bit quantisation ("tonemap"),
 * the scene -> transfer-function flattening, and the plugin boundary (renderbatch --device hip).
 *
 * All reference citations are relative to the reference tree root.
 */
#ifndef OVR_ORACLE_H
#define OVR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ovr/scene.h:32-53 (enum ValueType) - numeric values are part of the boundary */
enum {
  OVR_ORACLE_UINT8 = 100,
  OVR_ORACLE_INT8 = 101,
  OVR_ORACLE_UINT16 = 200,
  OVR_ORACLE_INT16 = 201,
  OVR_ORACLE_UINT32 = 300,
  OVR_ORACLE_INT32 = 301,
  OVR_ORACLE_FLOAT = 400,
  OVR_ORACLE_DOUBLE = 500
};

/* shading modes.  FULL is what the reference's live ray marcher always does
 * (shaders_raymarching.cu:124-158); the other two are the sub-modes BASELINE.json's configs name. */
enum {
  OVR_ORACLE_SHADE_NONE = 0,     /* absorption + emission only: 1 trilinear tap + TF per sample   */
  OVR_ORACLE_SHADE_GRADIENT = 1, /* + forward-difference gradient + |N.L| term, shadow term = 0   */
  OVR_ORACLE_SHADE_FULL = 2      /* + per-sample shadow march (reference behaviour)               */
};

/* grid conventions */
enum {
  OVR_ORACLE_GRID_CELL_CENTRED = 0,  /* in-tree OptiX device: bounds origin + spacing*dims, texel k at (k+.5)/N */
  OVR_ORACLE_GRID_VERTEX_CENTRED = 1 /* OSPRay wrapper convention: bounds origin + spacing*(dims-1), texel k at k/(N-1) */
};

typedef struct ovr_oracle_scene {
  /* volume (ovr/scene.h:228-238) */
  const void* volume;
  int value_type;
  int dims[3];
  float grid_origin[3];
  float grid_spacing[3];
  int grid_convention;
  /* transfer function in the app-side format of MainRenderer::set_transfer_function
   * (ovr/renderer.h:154-161,299-341): colours = flat RGB triples, alphas = flat (position, alpha) pairs */
  const float* tfn_colors;
  int n_colors; /* number of RGB triples */
  const float* tfn_alphas;
  int n_alphas; /* number of (pos, alpha) pairs */
  float tfn_range[2];
  /* camera (ovr/scene.h:201-231) */
  float cam_from[3], cam_at[3], cam_up[3];
  float fovy;
  /* frame (ovr/renderer.h:135-203) */
  int width, height;
  int spp;
  float sampling_rate;
  int shading;
  /* sparse sampling (ovr/renderer.h:163-168,180-183) */
  int sparse_sampling;
  float focus_center[2];
  float focus_scale;
  float base_noise;
  const float* noise_tile; /* [xy][xy][64] floats, layout [y][x][t] (ovr/common/random/blue_noise.h:95-99) */
  int noise_xy;            /* 64 (blue) or 128 (STBN) */
  /* image-plane shard (multi-GPU): only pixels of tiles owned by `rank` are rendered */
  int tile_w, tile_h, rank, world;
  /* data range of the volume as the device texture returns it (array.cpp:27-66,92-108,297): used as the transfer-function
   * range when tfn_range is invalid (hi < lo), volume.cpp:135-142.  Filled by ovr_oracle_data_range(); have_data_range = 0
   * makes the oracle compute it on demand. */
  float data_range[2];
  int have_data_range;
  /* pixel jitter: 0 = RandomTEA, applied iff spp > 1 (the reference, shaders_raymarching.cu:351-357);
   * 1 = blue-noise tile (this repo's progressive mode, BASELINE C5): applied to every sample, see ovr_oracle_jitter() */
  int pixel_jitter;
  /* 0 = shade every sample like the reference; 1 = skip gradient + shadow march of samples whose corrected opacity is
   * exactly 0 (they contribute exactly 0; frames are bit-identical - tests/test_oracle_kat.py) - the work the GPU does */
  int skip_zero_opacity;
} ovr_oracle_scene;

typedef struct ovr_oracle_counters {
  uint64_t rays;            /* primary rays traced (pixels x spp) */
  uint64_t samples;         /* primary marching-loop iterations (the metric's "sample") */
  uint64_t shaded_samples;  /* primary samples whose corrected opacity is > 0 */
  uint64_t shadow_samples;  /* shadow-march iterations the reference performs (one march per primary sample) */
  uint64_t shadow_samples_visible; /* shadow-march iterations belonging to primary samples with opacity > 0 */
  uint64_t borderline_samples; /* primary samples the opacity correction's pow took to within 2^-22 of 0 from a table opacity > 0: whether such a
                                  sample is "shaded" hangs on the pow's last bit - the bound of the shaded-count difference between two pow implementations */
} ovr_oracle_counters;

/* every numeric literal of the reference's integration loop as the restatement uses it (ERT thresholds, shadow step scale, midpoint factor, nearly_equal's
 * epsilon, the light, the shading terms, TEA's rounds / constants / scale, float_small / float_large), in the order of oracle.py::LITERAL_NAMES; returns how many.
 * Pinned against the reference's source text: tests/golden/ref_literals.json (extracted from the cited lines by tests/golden/make_ref_literals.py) */
int ovr_oracle_literals(double* out, int capacity);

/* ovr/common/random/random.h:146-188 - two floats from 16 TEA rounds; state is updated in place */
void ovr_oracle_tea_floats(uint32_t* v0, uint32_t* v1, float out[2]);

/* ovr/devices/optix7/device_impl.cpp:125-144 - camera basis {position, direction, horizontal, vertical} */
void ovr_oracle_camera_basis(const float from[3], const float at[3], const float up[3], float fovy, int width,
                             int height, float out_basis[12]);

/* ovr/devices/optix7/shaders_common.h:156-184 - returns 1 on hit, t0/t1 in-out */
int ovr_oracle_intersect_box(float* t0, float* t1, const float org[3], const float dir[3]);

/* ovr/devices/optix7/array.h:68-106 */
float ovr_oracle_integer_normalize(float value, int value_type);

/* compute_scalar_range + cuda_scalar_range (ovr/devices/optix7/array.cpp:27-66,92-108) applied to the array the device
 * ends up with (u16 / i16 / f64 converted to float first, array.cpp:335-345): out = {lower, upper} as seeded at
 * array.cpp:297, i.e. integer-normalized for 8- and 32-bit integers, raw otherwise */
void ovr_oracle_data_range(const ovr_oracle_scene* s, float out[2]);

/* the two jitter variates of sample k of pixel (ix, iy) in frame frame_index for pixel_jitter == 1:
 * slice t = ((frame_index - 1) * spp + k) % 64 of the noise tile (layout blue_noise.h:95-99),
 * xi0 = tile[iy % xy][ix % xy][t], xi1 = tile[(iy + xy/2) % xy][(ix + xy/2) % xy][t] (toroidal half-tile shift) */
void ovr_oracle_jitter(const ovr_oracle_scene* s, int ix, int iy, int frame_index, int k, float out[2]);

/* float -> half as the reference's EXR output does it (ovr/common/imageio.cpp:15-83 asks tinyexr for HALF pixels;
 * extern/tinyexr/tinyexr.h:889-924 float_to_half_full): the mantissa is cut to 10 bits and bit 12 of the float mantissa
 * rounds it up (nearest, ties AWAY from zero - not IEEE ties-to-even), the carry may run into the exponent (and on to
 * infinity); results below the half normal range shift the significand incl. its hidden bit and round the same way; float
 * denormals become signed zero; NaN becomes a quiet NaN (payload 0x200), overflow becomes infinity.
 * Pinned against the reference's own save_exr -> load_exr round trip (tests/golden/ref_probe.json). */
uint16_t ovr_oracle_float_to_half(float f);
float ovr_oracle_half_to_float(uint16_t h);

/* how __powf (shaders_raymarching.cu:64-66,118-122) is restated: 0 (default) = exp2f(y * log2f(x)), CUDA's documented definition of the
 * intrinsic; 1 = libm's powf (rounds 1-4); 2 = the deterministic pair below.  Process-wide; returns the previous mode.  OVR_ORACLE_POWF=libm|det sets the initial mode. */
int ovr_oracle_set_powf_mode(int mode);
int ovr_oracle_get_powf_mode(void);
/* mode 2: a machine-independent log2 / exp2 pair (fmaf Horner chains, ~1 ulp each - the accuracy class of the hardware instructions), the same float
 * arithmetic the HIP library evaluates when it is built with -DOVR_PARITY_EXACT=1 (libovr_hip_parity.so, a test instrument): with the same pow on both
 * sides every sample count is equal exactly - what is left of the parity tests' tolerances is the last bit of the transcendentals */
float ovr_oracle_det_log2f(float x);
float ovr_oracle_det_exp2f(float m);
float ovr_oracle_det_powf(float x, float y);

/* ovr/devices/optix7/shaders_common.h:186-193 + texture setup array.cpp:300-306: clamp p to [0,1]^3,
 * linear-filtered, clamp-addressed sample at normalized coordinate p (object space) */
float ovr_oracle_sample_volume(const ovr_oracle_scene* s, const float p[3]);

/* shaders_common.h:195-215 */
void ovr_oracle_gradient(const ovr_oracle_scene* s, const float c[3], float v, float out[3]);

/* shaders_common.h:311-319,356-367 with ranges from volume.cpp:131-154: returns rgba (alpha uncorrected) */
void ovr_oracle_sample_tfn(const ovr_oracle_scene* s, float sample, float rgba[4]);

/* shaders_raymarching.cu:118-122 */
float ovr_oracle_opacity_correction(float alpha, float base, float dt);

/* One full frame: shaders_raymarching.cu:323-413 for every pixel (or every sparse sample).
 *   frame_index     1-based frame counter (device_impl.cpp:241)
 *   accum_rgba      W*H*4 floats read+written when frame_accumulation != 0 (may be NULL otherwise)
 *   out_rgba        W*H*4 floats, row 0 = bottom (shaders_common.h:402-406)
 *   out_grad        W*H*3 floats or NULL
 *   nthreads        host threads (rows are interleaved); 0 = all online cores
 * Pixels not rendered (other ranks' tiles, sparse-sampling holes) are left untouched. */
void ovr_oracle_render_frame(const ovr_oracle_scene* s, int frame_index, int frame_accumulation, float* accum_rgba,
                             float* out_rgba, float* out_grad, ovr_oracle_counters* counters, int nthreads);

/* one primary ray (no spp loop, no accumulation) - for KATs: returns premultiplied-then-divided colour as
 * render_raymarching does (shaders_raymarching.cu:260-321) */
void ovr_oracle_trace_ray(const ovr_oracle_scene* s, const float org[3], const float dir[3], float rgba[4], float grad[3],
                          ovr_oracle_counters* counters);

/* ovr/common/imageio.cpp:146-181 (image_to_rgba8, 4 channels) + the vertical flip of save_image (imageio.cpp:265-285) */
void ovr_oracle_rgba8(const float* rgba, int width, int height, int flip_vertical, uint8_t* out);

/* ovr/common/generate_mask.cu:55-96 + ovr/common/random/blue_noise.h:81-102: writes the compacted (x,y) list of the
 * pixels kept by the foveated mask for this frame; returns the number of int32 written (= 2 x pixels) */
int64_t ovr_oracle_sparse_mask(int32_t* out_xy, int frame_index, int width, int height, const float focus_center[2],
                               float focus_scale, float base_noise, const float* noise_tile, int noise_xy);

/* test access: transform restatements, see ovr_oracle.c */
void ovr_oracle_xfm_probe(const float origin[3], const float scale[3], const float p[3], float out[12]);

/* deterministic restatement of __expf used by the mask (see ovr_oracle.c) */
float ovr_oracle_exp_det(float x);

/* image-plane tile owner (this repo's multi-GPU addition): tiles are dealt round-robin over ranks
 * along a row-major tile order offset per tile-row */
int ovr_oracle_tile_owner(int tile_x, int tile_y, int tiles_x, int world);

/* ovr/devices/optix7/accel/sp_singlemc.cu:10-54 - per 16^3 macrocell (min,max) of nearest-voxel reads incl. 1-voxel apron */
void ovr_oracle_macrocell_value_range(const ovr_oracle_scene* s, float* out_minmax /* [mz][my][mx][2] */);

/* sp_singlemc.cu:56-97 - per macrocell max TF opacity over the cell's value range */
void ovr_oracle_macrocell_majorant(const ovr_oracle_scene* s, const float* minmax, int n_cells, float* out_majorant);

#ifdef __cplusplus
}
#endif
#endif
