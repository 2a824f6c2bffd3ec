/* ovr_hip.h - C ABI of the MI355X (gfx950) ray-marching backend for OVR's renderer API.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Every entry point
 * replaces one member of the reference's device interface; `file:line` citations are relative to the
 * reference tree (VIDILabs/open-volume-renderer).  The reference-side binding (a ~180-line
 * `DeviceHIP : ovr::MainRenderer` compiled against the reference's own headers and exported as
 * `ovr_create_renderer__hip`, the symbol ovr/renderer.cpp:55-58 looks up) lives in plugin/device_hip.cpp
 * and is described in INTEGRATION.md.
 *
 * Error model: every function returns 0 on success and a negative OVR_HIP_E* code on failure;
 * ovr_hip_last_error() returns the message.  The reference reports errors as std::runtime_error
 * (ovr/common/cuda/cuda_misc.h:44-100); the C++ binding rethrows with the same text.
 *
 * Threading: like the reference (SURVEY.md 8b "Threading") setters may be called from any thread,
 * commit/render/mapframe/swap from one render thread at a time.
 */
#ifndef OVR_HIP_H
#define OVR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVR_HIP_ABI_VERSION 10

/* error codes */
#define OVR_HIP_OK 0
#define OVR_HIP_EINVAL (-1)   /* bad argument (the reference throws std::runtime_error) */
#define OVR_HIP_EDEVICE (-2)  /* HIP runtime error / no gfx950 device */
#define OVR_HIP_ESTATE (-3)   /* call order violated (e.g. render before a volume was set) */

/* Scalar types: numeric values of ovr::ValueType, ovr/scene.h:32-53 */
#define OVR_HIP_TYPE_UINT8 100
#define OVR_HIP_TYPE_INT8 101
#define OVR_HIP_TYPE_UINT16 200
#define OVR_HIP_TYPE_INT16 201
#define OVR_HIP_TYPE_UINT32 300
#define OVR_HIP_TYPE_INT32 301
#define OVR_HIP_TYPE_FLOAT 400
#define OVR_HIP_TYPE_DOUBLE 500

/* where a buffer lives (mirrors CrossDeviceBuffer::Device, ovr/common/cross_device_buffer.h:22-27) */
#define OVR_HIP_MEM_HOST 0
#define OVR_HIP_MEM_DEVICE 1

/* shading modes.  FULL = what the reference's live marcher always does (shaders_raymarching.cu:124-158) */
#define OVR_HIP_SHADE_NONE 0     /* absorption + emission: value tap + TF only (BASELINE config 2)        */
#define OVR_HIP_SHADE_GRADIENT 1 /* + forward-difference gradient and |N.L| term, no shadow march (config 3) */
#define OVR_HIP_SHADE_FULL 2     /* + per-sample shadow march toward the fixed directional light            */

/* grid convention */
#define OVR_HIP_GRID_CELL_CENTRED 0   /* in-tree OptiX device (default)        */
#define OVR_HIP_GRID_VERTEX_CENTRED 1 /* OSPRay wrapper's structuredRegular     */

typedef struct ovr_hip_renderer ovr_hip_renderer; /* opaque; one per GPU */

/* counters of the most recent render() (the metric's "sample" = one primary marching-loop iteration) */
typedef struct ovr_hip_stats {
  uint64_t rays;            /* primary rays (pixels x spp) launched by this rank                             */
  uint64_t samples;         /* primary marching-loop iterations                                               */
  uint64_t shaded_samples;  /* primary samples with corrected opacity > 0 (gradient/shadow taps executed)    */
  uint64_t shadow_samples;  /* shadow-march iterations executed                                               */
  uint64_t active_pixels;   /* pixels rendered (sparse sampling / tile sharding reduce it below W*H)         */
  double kernel_ms;         /* hipEvent time of the ray-march kernel of the last render()                     */
  double render_ms;         /* wall time of the last render() call, as DeviceOptix7::render measures it       */
  int32_t frame_index;      /* accumulation frame counter after the last render (device_impl.cpp:241)         */
  int32_t pipeline;         /* 1 = shaded in place, 2 = pooled (march -> shade -> composite kernels)          */
  double march_ms;          /* hipEvent time of the primary-march kernel (== kernel_ms when shading in place)  */
  double shade_ms;          /* pooled pipeline: the persistent shading kernel                                  */
  double composite_ms;      /* pooled pipeline: composite + counter reduction                                  */
  uint64_t pool_chunks;     /* pooled pipeline: 2 KiB request chunks used by the frame                         */
  uint64_t skipped_samples; /* empty-space skipping: primary iterations whose voxel fetch was skipped (not in `samples`) */
  uint64_t skipped_shadow_samples; /* same for shadow-march iterations (not in `shadow_samples`)                 */
  int32_t layout;           /* which resident layout of the volume the frame read: 0 general, 1 thin, 2 thin transposed, 3 quad */
  int32_t stale_tiles;      /* 1: this frame was rendered again (request-pool overflow) AFTER ovr_hip_pack_tiles had packed it - discard that payload */
  uint64_t lds_fallback_taps;   /* LDS-staged bricks: taps of live samples that fell outside the staged box (read from L1/L2 instead) */
  uint64_t lds_unstaged_rounds; /* LDS-staged bricks: workgroup rounds whose box exceeded the LDS budget (ordinary path)            */
  uint64_t lds_rounds;          /* LDS-staged bricks: workgroup rounds in total                                                      */
  int32_t skipping_kernels;     /* 1: the frame ran the empty-space-skipping kernels; 0: the plain ones - skipping disabled, or enabled but
                                   suspended because the last probed frame skipped < 10 % of its sample steps (probed again after 32 ... 256
                                   frames and on every transfer-function / volume change; the frames are bit-identical either way)      */
  int32_t tuning;               /* automatic layout / pipeline (ABI v7): 0 = the frame ran what the rules say (camera direction, share of
                                   shaded samples), 1 = it was a probe (a shade-heavy configuration: the other pipeline and the general / quad
                                   layouts are timed, two frames each), 2 = it ran the measured winner (kept until the configuration changes) */
  int32_t replicas_building;    /* ABI v8: replicas of the volume still being built in the background when the frame finished (the frame
                                   read the general layout meanwhile - the same frame bit for bit; see ovr_hip_set_volume_layouts) */
} ovr_hip_stats;

const char* ovr_hip_last_error(void);
int ovr_hip_abi_version(void);

/* replaces create_renderer("hip") / new DeviceOptix7 + Impl::init (ovr/renderer.cpp:42-61, device_impl.cpp:70-100).
 * device_id = HIP device ordinal; the reference hard-codes 0 (device_impl.cpp:371-372). */
int ovr_hip_create(ovr_hip_renderer** out, int device_id);
void ovr_hip_destroy(ovr_hip_renderer* r);

/* ABI v8 - several GPUs behind ONE handle, in one process (north_star: "apps run unmodified ... the 8 GPUs of one node shard
 * the image plane into tiles with a final RCCL gather over xGMI").  The reference's device knows one GPU (device_impl.cpp:371-372) and its
 * apps create one renderer (apps/main_batch.cpp:240-318, apps/main_app.cpp:233-278): the returned handle is used exactly like the one of
 * ovr_hip_create - every setter, commit, render, mapframe, swap - and drives n_devices renderers, one per listed device: the volume is
 * replicated, device k renders the image tiles (tx + ty) % n == k (16 x 16 pixels; OVR_HIP_TILE=WxH, or ovr_hip_set_image_shard(r, 0, 1,
 * w, h)), and at the end of every frame the tiles of devices 1 ... n-1 travel to device_ids[0] - ncclSend / ncclRecv between the
 * communicators of this process (librccl is loaded at run time) when the devices are distinct, peer-to-peer copies otherwise or when RCCL is
 * absent (OVR_HIP_GATHER=rccl|copy forces one) - where one launch per layer scatters them into the leader's framebuffer.  Pixels, TEA seeds and
 * accumulation are per pixel: the frame is the single-device frame bit for bit.  A device may be listed more than once (a rehearsal on one
 * card).  n_devices == 1 is ovr_hip_create.  ovr_hip_get_stats sums the members' counters and reports the slowest member's times;
 * OVR_HIP_MAP_GRAD=0 leaves the gradient layer out of the gather (it is then valid for device 0's tiles only). */
int ovr_hip_create_group(ovr_hip_renderer** out, const int32_t* device_ids, int32_t n_devices);
/* n_devices (1 for an ordinary renderer), how the tiles travel (0 = no group, 1 = peer copies, 2 = RCCL), and the host time the last frame
 * spent after its slowest member had finished (waiting for the shipments + the scatter), milliseconds; any pointer may be NULL */
int ovr_hip_group_info(const ovr_hip_renderer* r, int32_t* n_devices, int32_t* gather_kind, double* gather_ms);
/* diagnostic: the RCCL entry points a device group uses (ncclCommInitAll, ncclGroupStart / End, ncclSend, ncclRecv on a stream), exercised on
 * ONE device - a communicator of one rank sends 256 KiB to itself.  0 = they work; OVR_HIP_ESTATE = librccl.so is not loadable (groups use peer copies) */
int ovr_hip_rccl_selftest(int device_id);
/* ABI v10: host time of the last frame's steps on the calling (leader's) thread, microseconds: [0] launching every member's frame (each follower on
 * its own host thread, the leader on this one), [1] packing + shipping the followers' tiles, [2] every member's frame to its end, [3] waiting for the
 * shipments + the scatter on the leader.  Zeros for an ordinary renderer.  Round 4 drove all members from one thread: [0] was 225 us at 8 members. */
int ovr_hip_group_host_times(const ovr_hip_renderer* r, double out_us[4]);
/* the counters of ONE member's last frame (member 0 = the leader's own tiles) */
int ovr_hip_get_member_stats(const ovr_hip_renderer* r, int32_t member, ovr_hip_stats* out);

/* Use a caller-owned hipStream_t for all device work (e.g. torch's current stream); NULL = private streams,
 * one per framebuffer set like DoubleBufferObject (optix7_common.h:328-414). */
int ovr_hip_set_stream(ovr_hip_renderer* r, void* hip_stream);

/* replaces Impl::buildScene + StructuredRegularVolume::load_from_array3d_scalar + CreateArray3DScalarOptix7
 * (device_impl.cpp:283-302, volume.cpp:181-257, array.cpp:287-351).  `data` = dims[0]*dims[1]*dims[2] scalars,
 * x fastest.  The volume is re-laid out into HBM-resident bricks; `data` is not referenced after the call returns. */
int ovr_hip_set_volume(ovr_hip_renderer* r, const void* data, int mem_kind, int value_type, const int32_t dims[3],
                       const float grid_origin[3], const float grid_spacing[3]);
/* ABI v10: where the last ovr_hip_set_volume spent its time, milliseconds: [0] the whole call (a device group: over all members, which upload side by
 * side), [1] allocation (a FRESH hipMalloc costs 30-60 ms per GiB on this platform - the driver maps and clears the pages; C4's 21.5 GB layout: 0.5-1.2 s
 * in a new process, 0.1 ms when the runtime still holds a freed block of that size), [2] copies into the device (host input: through a 1 GiB staging buffer;
 * a device array on another GPU: peer copies), [3] kernels (re-bricking, macrocell ranges, data range; layouts mode 2: the replicas) */
int ovr_hip_get_upload_times(const ovr_hip_renderer* r, double out_ms[4]);
int ovr_hip_set_grid_convention(ovr_hip_renderer* r, int convention);
/* diagnostic, pure host arithmetic (no device needed): the addressing mode the kernels would take for a volume of these dimensions and
 * type in the given layout (0 general ... 3 quad) with a transfer function of n_colors / n_alphas entries: 0 = 32-bit byte offsets,
 * 1 = 32-bit element offsets, 2 = 64-bit z table in LDS, 3 = computed 64-bit offsets (per-axis tables would not fit in the 160 KiB of LDS
 * next to the transfer function and the request queues: a dimension of tens of thousands of voxels).  < 0: no such layout for the type. */
int ovr_hip_query_addressing_mode(const int32_t dims[3], int value_type, int32_t layout, int32_t n_colors, int32_t n_alphas);
/* extension (MI355X: 288 GB of HBM traded for bandwidth): which layouts of the volume ovr_hip_set_volume keeps resident.
 * Besides the general layout (128-byte 3-D bricks) two "thin" replicas serve views whose rays run along a volume axis - 12 of
 * the reference's 21 shipped scene cameras - where rays are sparser than voxels and a general brick is mostly wasted (C3 front
 * view: 2.18 -> 1.69 ms per frame).  Frames are bit-identical whatever layout is read.  mode 0 = general only, 1 (default) =
 * thin replicas for float / uint16 volumes when all replicas fit in 40 % of the free HBM (x 3 - 4 of the volume's size),
 * 2 = always.  Takes effect at the next ovr_hip_set_volume.
 * Round 3: float volumes get a fourth layout under the same rule, the "quad" replica - every cell stores its 2 x 2 (x, y) voxels as 16
 * contiguous bytes (4 x the volume's size), so a trilinear tap is two 16-byte loads instead of four 8-byte ones.  Frames that shade every
 * sample (most shipped scenes; every frame at the scene files' sampling rate 4) are bound by the gather-instruction rate: -7 ... -16 % there. */
int ovr_hip_set_volume_layouts(ovr_hip_renderer* r, int32_t mode);
/* which resident layout a frame reads: -1 (default) = automatic - from the camera direction (a thin replica within ~21 degrees of an
 * axis, cos >= 0.93) and, for configurations whose shading taps outnumber their primary taps, MEASURED: the frames after the first try the
 * other pipeline and the general / quad layouts, two frames each, and the fastest is kept until the configuration changes
 * (ovr_hip_stats.tuning; OVR_HIP_TUNE=0 disables the measuring).  0 / 1 / 2 / 3 = forced: general, thin, thin transposed, quad (general
 * if that replica is not resident).  Applied at commit; does not reset the accumulation - every layout gives the same frame bit for bit. */
int ovr_hip_set_layout_choice(ovr_hip_renderer* r, int32_t choice);

/* replaces MainRenderer::set_transfer_function -> StructuredRegularVolume::set_transfer_function
 * (ovr/renderer.h:154-161, volume.cpp:110-129): colors = n_colors flat RGB triples, alphas = n_alphas flat
 * (position, alpha) pairs (position ignored, as in the reference), range in raw data units. */
int ovr_hip_set_transfer_function(ovr_hip_renderer* r, const float* colors_rgb, int32_t n_colors,
                                  const float* alphas_pos_alpha, int32_t n_alphas, float range_lo, float range_hi);

/* replaces MainRenderer::set_camera (ovr/renderer.h:140-152); fovy in degrees (scene.h:219 default 60) */
int ovr_hip_set_camera(ovr_hip_renderer* r, const float from[3], const float at[3], const float up[3], float fovy);
/* ovr/renderer.h:135-138 */
int ovr_hip_set_fbsize(ovr_hip_renderer* r, int32_t width, int32_t height);
/* ovr/renderer.h:170-173 */
int ovr_hip_set_sample_per_pixel(ovr_hip_renderer* r, int32_t spp);
/* ovr/renderer.h:200-203 -> StructuredRegularVolume::set_sampling_rate (volume.cpp:156-162) */
int ovr_hip_set_volume_sampling_rate(ovr_hip_renderer* r, float rate);
/* ovr/renderer.h:195-198 */
int ovr_hip_set_frame_accumulation(ovr_hip_renderer* r, int32_t enabled);
/* ovr/renderer.h:180-183 */
int ovr_hip_set_sparse_sampling(ovr_hip_renderer* r, int32_t enabled);
/* ovr/renderer.h:163-168 */
int ovr_hip_set_focus(ovr_hip_renderer* r, float center_x, float center_y, float scale, float base_noise);
/* blue-noise / STBN tile used by the sparse-sampling mask: xy*xy*64 floats, layout [y][x][t]
 * (ovr/common/random/blue_noise.h:44-47,95-99).  The reference embeds the tile at build time (ovr/CMakeLists.txt:67-72). */
int ovr_hip_set_noise_tile(ovr_hip_renderer* r, const float* tile, int32_t xy);
/* extension: select the sub-mode BASELINE.json's configs name; default OVR_HIP_SHADE_FULL (= reference) */
int ovr_hip_set_shading(ovr_hip_renderer* r, int32_t mode);
/* extension: how shaded samples are processed.  0 = automatic (pooled, or in place once a frame shaded >= 50 % of its samples - back to
 * pooled below 35 %: ovr_hip_stats::pipeline says which ran), 1 = in place
 * (the tile's own wave shades its request batches), 2 = pooled (request chunks go through a global pool and are shaded
 * by a separate, load-balanced kernel).  Both produce bit-identical frames. */
int ovr_hip_set_shading_pipeline(ovr_hip_renderer* r, int32_t mode);
/* extension: empty-space skipping with the reference's macrocell grids (16^3 value-range cells +
 * per-TF max-opacity cells, ovr/devices/optix7/accel/sp_singlemc.cu:10-97 - the reference computes them but only its path
 * tracer uses them).  Samples whose cell has majorant 0 have opacity exactly 0: their voxel fetch is skipped, frames stay
 * bit-identical.  Off by default (the reference's ray marcher visits every sample).  While it is enabled the renderer keeps the
 * skipping kernels only where they pay: a frame that skipped < 10 % of its sample steps (a dense transfer function, a camera inside the
 * data - there the skipping kernels cost 20-50 % more than the plain ones) switches to the plain kernels and the skipping ones are
 * probed again later (ovr_hip_stats::skipping_kernels says which ran; OVR_HIP_SKIP_ADAPTIVE=0 keeps them on regardless). */
int ovr_hip_set_empty_space_skipping(ovr_hip_renderer* r, int32_t enabled);
/* extension (BASELINE C5, north_star "blue-noise jitter staged in LDS"): how the sub-pixel position of a sample is drawn.
 * 0 (default) = the reference: RandomTEA(frame_index, pixel_index), applied iff sample_per_pixel > 1
 * (shaders_raymarching.cu:336,351-357).  1 = blue-noise tile (ovr_hip_set_noise_tile; lookup as blue_noise.h:95-99): sample k
 * of frame f uses slice ((f - 1) * spp + k) % 64, xi_x = tile[y % xy][x % xy][slice], xi_y = the slice shifted by half a tile
 * in x and y; applied to EVERY sample, so 64 accumulated 1-spp frames give the 64-spp progressive image. */
#define OVR_HIP_JITTER_TEA 0
#define OVR_HIP_JITTER_BLUE_NOISE 1
int ovr_hip_set_pixel_jitter(ovr_hip_renderer* r, int32_t mode);
/* extension (north_star "volume brick-tiled into LDS"): the unshaded (OVR_HIP_SHADE_NONE), non-skipping march of a float volume
 * stages, once per round of 16 steps, every brick its 8x8-pixel workgroup can touch into LDS with whole-line loads and taps read
 * LDS.  Bit-identical frames.  0 = off (default: 2.3 - 3 x slower than the L1 path on its best case, profiles/r02_notes.md), 1 = on. */
int ovr_hip_set_lds_staging(ovr_hip_renderer* r, int32_t mode);
/* extension (ABI v9): per-phase device times.  1 (default): two more events are recorded between the frame's kernels and
 * ovr_hip_stats::march_ms / shade_ms / composite_ms say what each phase took; 0: those three read 0 and a frame is ~16 us shorter (hipEventRecord
 * costs on both sides of the queue: 8 % of a 0.2 ms frame, 4 % of one GPU's share of an 8-GPU frame).  kernel_ms (first to last event) is measured
 * either way.  Takes effect with the next frame launched; frames are identical.  The plugin switches it off (OVR_HIP_PHASE_TIMING=1 keeps it). */
int ovr_hip_set_phase_timing(ovr_hip_renderer* r, int32_t on);
/* downloads the macrocell grids (for known-answer tests): dims = cells per axis; minmax = 2 floats per cell, majorant = 1 */
int ovr_hip_get_macrocells(ovr_hip_renderer* r, int32_t dims[3], float* minmax_host, float* majorant_host, size_t capacity_cells);
/* extension (multi-GPU): this renderer draws only the image tiles owned by `rank` of `world`;
 * owner(tile_x, tile_y) = (tile_x + tile_y) % world.  world = 1 restores the single-GPU behaviour. */
int ovr_hip_set_image_shard(ovr_hip_renderer* r, int32_t rank, int32_t world, int32_t tile_w, int32_t tile_h);

/* replaces DeviceOptix7::Impl::commit (device_impl.cpp:113-197): applies every queued setter; any change resets
 * the accumulation (frame_index restarts at 1 on the next render). */
int ovr_hip_commit(ovr_hip_renderer* r);

/* replaces DeviceOptix7::render (optix7/device.cpp:35-43, device_impl.cpp:199-269): one frame, blocking until the
 * frame is complete on the device; adds the elapsed milliseconds to the value ovr_hip_render_time_ms() returns. */
int ovr_hip_render(ovr_hip_renderer* r);
/* non-blocking variant: enqueues the frame on the renderer's stream and returns (hipEvent timing around it, overlap with the caller's other work -
 * e.g. the gather of the previous frame).  The first frame after a camera / volume / size / spp / jitter change reads 8 bytes back from the device
 * before it launches - how many 8x8-pixel blocks have a ray that meets the volume's box; the others get no workgroup, their pixels are cleared - a
 * stream synchronisation of ~20 us (a device group's members wait side by side, each on its own host thread).  A frame is NOT a unit for hipGraph
 * capture - it records timed events and its counters are read by the host when it is resolved: with a caller's stream (ovr_hip_set_stream) that is
 * capturing, the call fails with OVR_HIP_ESTATE and leaves the capture intact.  (A frame is five launches, ~10 us of host time that overlap its
 * first kernel: there is nothing for a graph to take) */
int ovr_hip_render_async(ovr_hip_renderer* r);
/* waits for the frame enqueued by render_async and for everything else enqueued on the renderer's stream
 * (ovr_hip_pack_tiles / ovr_hip_unpack_tiles launches included) */
int ovr_hip_sync(ovr_hip_renderer* r);

/* replaces Impl::mapframe (device_impl.cpp:271-281): publishes the CURRENT framebuffer set.
 * mem_kind DEVICE: device pointers (as the reference hands out, CrossDeviceBuffer::DEVICE_CUDA);
 * mem_kind HOST: the frame is copied to pinned host memory owned by the renderer (what the caller's
 * CrossDeviceBuffer::to_cpu() would do, cross_device_buffer.h:130-159) and stays valid until the next mapframe
 * of the same set.  rgba = W*H*4 floats, row 0 = bottom; grad = W*H*3 floats (may be NULL to skip). */
int ovr_hip_mapframe(ovr_hip_renderer* r, int mem_kind, const float** rgba, size_t* rgba_bytes, const float** grad,
                     size_t* grad_bytes);
/* frame output: image_to_rgba8 of the reference (ovr/common/imageio.cpp:146-181: clamp to [0,1], * 255,
 * truncate; `flip_vertical` as renderbatch passes it, apps/main_batch.cpp save_image) applied to the CURRENT framebuffer
 * set on the device - the host copy of a saved or displayed frame is 4 bytes per pixel instead of 16.  The pointer
 * stays valid until the next call of this function or a framebuffer resize. */
int ovr_hip_mapframe_rgba8(ovr_hip_renderer* r, int mem_kind, int flip_vertical, const uint32_t** rgba8, size_t* bytes);

/* frame output, EXR: the float -> half step of the reference's save_image(".exr") (ovr/common/imageio.cpp:15-83,
 * 268-272: the flipped RGBA32F frame goes to tinyexr with requested_pixel_types = HALF; conversion rule
 * extern/tinyexr/tinyexr.h:889-924 - nearest, ties away from zero) applied to the CURRENT framebuffer set on the device:
 * W*H*4 IEEE binary16 bit patterns (R, G, B, A per pixel).  Valid until the next call or a framebuffer resize. */
int ovr_hip_mapframe_rgba16f(ovr_hip_renderer* r, int mem_kind, int flip_vertical, const uint16_t** rgba16f, size_t* bytes);

/* replaces Impl::swap (device_impl.cpp:102-111): waits for the current set's stream, flips to the other set */
int ovr_hip_swap(ovr_hip_renderer* r);

/* MainRenderer::render_time (ovr/renderer.h:87): accumulated milliseconds spent in ovr_hip_render */
double ovr_hip_render_time_ms(const ovr_hip_renderer* r);
int ovr_hip_get_stats(const ovr_hip_renderer* r, ovr_hip_stats* out);

/* what load_from_array3d_scalar leaves behind (volume.cpp:181-191): the data range found by compute_scalar_range /
 * cuda_scalar_range (array.cpp:27-66,92-108,297; integer-normalized for 8- and 32-bit integer volumes, raw otherwise), the
 * transfer-function range in effect (set_value_range, volume.cpp:131-145: a range with hi < lo keeps the previous one, which is
 * the data range after a load) and the bytes the re-laid-out volume occupies in HBM */
typedef struct ovr_hip_volume_info {
  int32_t dims[3];
  int32_t value_type;      /* OVR_HIP_TYPE_* as passed to ovr_hip_set_volume */
  uint64_t resident_bytes;
  float data_lower, data_upper;
  float tf_lower, tf_upper;
} ovr_hip_volume_info;
int ovr_hip_get_volume_info(const ovr_hip_renderer* r, ovr_hip_volume_info* out);

/* multi-GPU helpers.  pack: copies this rank's tiles out of its W*H framebuffer into a compact
 * [n_owned_tiles][tile_h][tile_w][4] device buffer (the RCCL gather payload); unpack: scatters the gathered payload
 * of `src_rank` into a W*H*4 frame on the gathering rank.  Both run on the renderer's stream. */
int ovr_hip_owned_tiles(const ovr_hip_renderer* r, int32_t rank, int32_t* n_tiles);
int ovr_hip_pack_tiles(ovr_hip_renderer* r, float* dst_device, size_t dst_bytes);
int ovr_hip_unpack_tiles(ovr_hip_renderer* r, int32_t src_rank, const float* src_device, size_t src_bytes,
                         float* frame_device, size_t frame_bytes);
/* all ranks' payloads in one launch: rank k's payload starts at src_device + k * rank_stride_bytes (the receive
 * buffer of the gather as one allocation); rank_stride_bytes is a multiple of 16 and >= the largest payload */
int ovr_hip_unpack_all_tiles(ovr_hip_renderer* r, const float* src_device, size_t rank_stride_bytes, size_t src_bytes,
                             float* frame_device, size_t frame_bytes);

/* stand-alone pieces of the path, exposed for known-answer tests through the same ABI (device buffers) */
/* ovr/common/generate_mask.cu:100-120: compacted (x,y) list for this frame; returns the int32 count in *n_out */
int ovr_hip_sparse_mask(ovr_hip_renderer* r, int32_t frame_index, int32_t* out_xy_device, size_t out_bytes,
                        int64_t* n_out);
/* ovr/common/random/random.h:146-188: n (v0,v1) states -> 2n floats, states advanced in place */
int ovr_hip_tea_floats(ovr_hip_renderer* r, uint32_t* v0v1_device, float* out_device, int64_t n);

/* ABI v10 - shaders_raymarching.cu:64-66,118-122: the `__powf(x, y)` of the opacity correction as the kernels evaluate it, n (x, y) pairs -> n floats
 * (device buffers).  which = 0: the pow this library was built with - v_exp_f32(y * v_log_f32(x)), CUDA's documented structure of the intrinsic, in
 * the product; 1: a machine-independent log2 / exp2 pair (fmaf Horner chains; the CPU oracle's mode 2 is the same arithmetic, bit for bit).  A library
 * built with -DOVR_PARITY_EXACT=1 (libovr_hip_parity.so: a test instrument for the parity tests, never the product - ovr_hip_built_for_exact_parity() == 1)
 * marches with that pair: every sample count then equals the oracle's exactly, which pins the transcendental's last bit as the one source of the
 * tolerated count differences. */
int ovr_hip_pow_floats(ovr_hip_renderer* r, const float* x_device, const float* y_device, float* out_device, int64_t n, int32_t which);
int ovr_hip_built_for_exact_parity(void);

#ifdef __cplusplus
}
#endif
#endif /* OVR_HIP_H */
