// ovr_hip_kernels.h - launch interface between the C-ABI host layer (ovr_hip_api.cpp) and the gfx950 kernels.
// Internal to libovr_hip.so; the public boundary is include/ovr_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovrhip {

// device-resident scalar types of the bricked volume (u32/i32/f64 inputs are converted at upload, see relayout)
// VOX_*_T / VOX_*_TT are not further scalar types but further LAYOUTS of a resident f32 / u16 volume ("thin" replicas, below)
// VOX_*_Q are the "quad" layouts of a resident f32 / u16 / u8 volume (below)
enum VoxelType : int { VOX_U8 = 0, VOX_I8 = 1, VOX_U16 = 2, VOX_I16 = 3, VOX_F32 = 4, VOX_F32_T = 5, VOX_F32_TT = 6, VOX_U16_T = 7, VOX_U16_TT = 8, VOX_F32_Q = 9,
                       VOX_U16_Q = 10, VOX_U8_Q = 11 };
enum VolumeLayout : int { LAYOUT_GENERAL = 0, LAYOUT_THIN = 1, LAYOUT_THIN_T = 2, LAYOUT_QUAD = 3 };
constexpr int kLayouts = 4;
constexpr int kBlockCounters = 8; // per-workgroup partial counters of the march (rays, samples, shaded, shadow, active pixels, skipped, skipped shadow, hits through an ignored slab)

// Volume layout in HBM ("x-apron bricks in macro blocks"):
//   voxels are grouped into 128-byte bricks = one L1/L2 line.  A brick covers CX x 4 x (2|4) cells and stores CX+1 voxels
//   along x - its last x-column duplicates the first column of its +x neighbour (or replicates the grid's last voxel) - so
//   the x-pair of a trilinear tap is always two ADJACENT elements of one brick:
//       f32: (3+1)x4x2     u16/i16: (3+1)x4x4     u8/i8: (7+1)x4x4          (memory x 4/3, x 8/7 for 8-bit)
//   Bricks are x-fastest inside macro blocks of (10|4) x 8 x (16|8) bricks (30|28 x 32 x 32 cells), macro blocks x-fastest.
//   The element offset is separable, off(x,y,z) = X(x) + Y(y) + Z(z) (BrickMap in ovr_hip_kernels.hip).
//   One line holds one 3-D brick: the 2x2x2 footprint of a tap touches ~2.4 lines for ANY ray direction (a row-major
//   layout touches 4 and loses all reuse as soon as rays do not run along x: measured 81 % L1 / 52 % L2 miss rate on the
//   oblique bench camera), and a tap is 4 pair loads instead of 8 scalar loads.
//
// View-dependent replicas (288 GB of HBM buy bandwidth): when the rays of a frame run within ~21 degrees of a volume axis (cos >= 0.93) - 12 of
//   the reference's 21 shipped scene cameras do - and are sparser than the voxels, a general brick is mostly wasted: a ray uses
//   a 2 x 2 column of it.  A THIN replica stores 1 cell (+ apron) along the pair axis and 4 x 4 (f32) / 4 x 8 (u16) voxels
//   across: thin across the rays, 4-8 steps deep along them.  VOX_*_T has the pair axis on x (for rays along y or z), VOX_*_TT
//   is the same layout of the volume with x and y exchanged (pair axis = the volume's y; for rays along x).  The tap reads the
//   same 8 voxels and lerps them in the same order from any layout, so frames are bit-identical; the host picks the replica
//   per frame from the camera direction (ovr_hip_api.cpp: enqueue_frame).  Measured: profiles/r02_notes.md.
//
// Quad replica (round 3; f32, u16, u8): every cell (x, y, z) stores the four voxels (x, y), (x+1, y), (x, y+1), (x+1, y+1) of its z slice
//   contiguously (16 / 8 / 4 bytes, clamp-to-edge baked in), 2x2x2 / 4x2x2 / 4x4x2 cells per 128-byte brick, 32^3 cells per macro block -
//   4 x the memory.  A trilinear tap is TWO loads (cell z0 and cell z0 + 1) instead of four pair loads.  The frames in which every
//   sample is shaded - most of the reference's shipped scenes, and every frame at the scene files' sampling rate 4 - are bound by the
//   texture addresser's instruction rate (16+ clocks per 64-lane gather whatever its width; TA / TD busy 84-97 %, VALU 80-100 %,
//   profiles/r03_notes.md): half the gathers and three fewer address adds per tap are what moves them.  Same 8 voxels, same lerp
//   order: frames are bit-identical.
struct VolumeDesc {
  const void* data;
  int type;        // VoxelType (of this replica: the base type, or its _T / _TT variant)
  int nx, ny, nz;
  int macros_x, macros_y, macros_z;
  unsigned int macro_elems;         // stored voxels per macro block
  unsigned long long bytes;         // macros_x * macros_y * macros_z * macro_elems * sizeof(voxel)
  float value_scale; // multiplier turning a filtered raw value into what the reference's texture read returns
  float value_min_clamp; // raw clamp applied per voxel before filtering (i8: -127) - see array.h:83-90
  // per-axis ELEMENT offsets of this layout, built once per volume (launch_axis_tables).  Every table is indexed by i + 1 for a
  // voxel index i that starts at -1 (round 4): the reference's clamp-to-edge texture reads the pair (0, 0) in the half voxel
  // outside the first voxel centre (floor(x) = -1) and (n - 1, n - 1) beyond the last one, so entry 0 of every table addresses
  // what index -1 addresses - a copy of voxel 0 - and the entries past n - 1 repeat the last voxel: off(a, b, z) = axis_ab[a + 1] +
  // axis_ab[(na + 1) + b + 1] + axis_z[z + 1] with (a, b) = (x, y), exchanged in a transposed replica.  The pair axis a has its
  // copies in the DATA (stored position = voxel + 1, see BrickMap), the b and z axes in the tables.  The march / shade workgroups
  // copy the tables into LDS (stage_tables) instead of computing ~3 k entries with 64-bit multiplies each (fewer instructions and
  // 0.5 MB less code; the frame time did not move: the prologue hides behind the other workgroups of the CU - profiles/r02_notes.md §7)
  const unsigned int* axis_ab;       // [na + 1][nb + 2]
  const unsigned long long* axis_z;  // [nz + 2]
};
// entries of the per-axis tables (a = pair axis: lower members -1 ... n - 1; b, z: -1 ... n)
__host__ __device__ inline int axis_a_entries(int na) { return na + 1; }
__host__ __device__ inline int axis_b_entries(int nb) { return nb + 2; }
__host__ __device__ inline int axis_z_entries(int nz) { return nz + 2; }

struct float3_ { float x, y, z; };

// global request pool of the pooled shading pipeline (see ovr_hip_kernels.hip): 2 KiB chunks of 64 requests
// The pool is cut into kPoolSubs sub-pools with a chunk counter each, every counter on a 128-byte line of its own: a workgroup
// of the march reserves from sub-pool (blockIdx.x % kPoolSubs).  One counter for the whole pool was a same-address returning
// atomic per reservation - they serialise in L2: 26 k of them were ALL of the march's time (0.42 ms) on a frame whose rays end at
// their first sample, and half of it with a dense transfer function.  4 sub-pools remove that (march 1.00 -> 0.59 ms, dense TF);
// more bring little (8: 0.57) and slow the shade kernel down - its requests lose their creation order - by 1.7 % (8), 2 % (16),
// 11 % (64) on the sparse-TF headline; profiles/r02_notes.md, profiles/r02_ab/r02_ab_ps*.txt.
#ifndef OVR_POOL_SUBS
#define OVR_POOL_SUBS 4
#endif
constexpr int kPoolSubs = OVR_POOL_SUBS; // a power of two, at most 64
constexpr int kPoolCtrlWords = 32 * (kPoolSubs + 2);
// ctrl words: [0] shade ticket cursor, [1] overflow flag of this generation (a sub-pool ran out), [32 * (s + 1)] chunks reserved
// from sub-pool s (all of these zeroed per generation), [32 * (kPoolSubs + 1)] the most any sub-pool was asked for in any
// generation of the frame (zeroed per frame; > sub_capacity: the host grows the pool and renders the frame again)
struct PoolDesc {
  struct ShadeReq* reqs;        // capacity * 64 requests of 32 bytes (null = pipeline disabled)
  unsigned int capacity;        // chunks = kPoolSubs * sub_capacity
  unsigned int sub_capacity;    // chunks per sub-pool (a multiple of 16); sub-pool s owns chunks [s * sub_capacity, (s + 1) * sub_capacity)
  unsigned int* ctrl;           // kPoolCtrlWords words, see above
  int* chunk_next;              // per chunk: next chunk of the same tile
  unsigned int* chunk_n;        // per chunk: requests in it (64 except a tile's last)
  int* tile_first;              // per tile (wave of the march grid): first chunk or -1
  unsigned int* tile_count;     // per tile: requests pushed
  float4* pix_state;            // per pixel: alpha, first request position, request count
  unsigned int* shade_counters; // per shade workgroup: 2 words (shadow-march iterations fetched, skipped)
  // shade order by light beams (round 5, below): null = the shade kernel meets the runs in creation order
  unsigned int* order;          // capacity / 4 entries: the runs (first chunk / 4) sorted by XCD list, then by beam
  unsigned int* order_key;      // per run slot: its beam index - written by the march when it spills the run's first chunk - or kOrderNoKey
  unsigned int* order_ws;       // kOrderWsWords words: [hist G*G][fill G*G][list_start 32][ticket counters 8 x 32]
  int order_grid;               // G: beams per side of the light-perpendicular grid (32, 64 or 128)
  float order_u[4], order_v[4]; // beam (cu, cv) of a world position p: cu = (int)(p . u.xyz + u.w), clamped to [0, G)
};
// Shade order by LIGHT BEAMS.  The shadow rays of a frame all run along ONE direction; a shaded sample's ray sweeps the column of bricks
// that lies toward the light from it, and every other sample in that column sweeps the same bricks.  In creation order (view-space
// tiles) those samples meet the shade kernel at unrelated times and on unrelated XCDs: the headline's shade kernel fetched 7.7 GB, a
// quarter of its L2 lookups hit.  So the runs (4 chunks = 256 requests of one tile: a compact cluster in space) are counting-sorted by
// the cell of a G x G grid perpendicular to the light that their first request projects into (a beam); beams are dealt to 8 lists, and
// the workgroups of XCD x (HW_REG_XCC_ID) take the runs of list x in order: a beam's bricks are fetched once into ONE L2 and hit there
// by every later ray of the beam (2.7 GB, three quarters hit).  A workgroup whose list is done takes from the other lists (balance).
//   march          lane 0 of a wave that spills the first chunk of a run: key -> order_key[run], hist[key] += 1
//   order kernel   every workgroup scans the histogram in LDS; a thread per run slot: order[scan[key] + fill[key]++] = run
//   shade kernel   zeroes hist / fill for the next generation, then takes tickets from its XCD's list first
// Frames are bit-identical: the order in which requests are shaded never enters a result.
constexpr int kOrderMaxGrid = 128;
constexpr int kOrderMaxKeys = kOrderMaxGrid * kOrderMaxGrid;
constexpr unsigned int kOrderNoKey = 0xffffffffu;
constexpr int kOrderLists = 8;
constexpr int kOrderHist = 0;
constexpr int kOrderFill = kOrderMaxKeys;
constexpr int kOrderListStart = 2 * kOrderMaxKeys;
constexpr int kOrderTickets = kOrderListStart + 32;
constexpr int kOrderWsWords = kOrderTickets + kOrderLists * 32;

struct RayMarchParams {
  // framebuffer (optix7/params.h:56-63)
  float* rgba;   // W*H*4
  float* grad;   // W*H*3 (may be null)
  float* accum;  // W*H*4 (may be null when !accumulate)
  int width, height;
  int frame_index;  // 1-based
  int accumulate;
  int spp;
  int spp_index;        // pooled pipeline: the sample-per-pixel generation this launch renders
  int row_loads;        // 16-bit layouts' aligned 8-byte pair loads (launch_vs): 0 = by the layout's size, 1 = never, 2 = always
  float* spp_sum_rgba;  // pooled pipeline, spp > 1: per-pixel sums over the generations (W*H*4, W*H*3)
  float* spp_sum_grad;
  // camera (params.h:65-70), basis from device_impl.cpp:125-144
  float3_ cam_pos, cam_dir, cam_hor, cam_ver;
  // volume transform: world -> object is diagonal (device_impl.cpp:288-296)
  float3_ inv_scale, wto_p, otw_it;
  float wtc_it[9]; // columns of inverse-transpose(world_to_camera)
  float3_ light;   // normalized params.h:79
  // object [0,1] -> voxel coordinate: x = p * coord_scale + coord_bias, clamped to [0, n-1]
  float3_ coord_scale, coord_bias, grad_step;
  float step, base;           // volume.cpp:172-179
  float shadow_stride;        // 10 * step * step (shaders_raymarching.cu:221,64)
  float long_ray_steps;       // samples along the volume's diagonal (scheduling hint only)
  float tf_lower, tf_upper, tf_scale; // volume.cpp:131-145
  const float* tf_color;      // n_color * 4 (rgb, 1)
  const float* tf_alpha;      // n_alpha
  int n_color, n_alpha;
  int shading;                // OVR_HIP_SHADE_*
  // image-plane shard
  int rank, world, tile_w, tile_h;
  // dense mode: the 8x8-pixel blocks this renderer draws, bx | by << 16, longest rays first (launch_schedule)
  const unsigned int* schedule;
  unsigned int n_schedule;          // workgroups of the march / composite launch: the first n_schedule entries of the list
  unsigned int n_blocks_owned;      // all blocks this renderer owns (>= n_schedule: blocks none of whose rays hits the volume's box are not launched)
  // sparse sampling: compacted (x,y) list + device-side count (2 * pixels), null in dense mode
  const int32_t* sparse_xy;
  const unsigned long long* sparse_count;
  unsigned long long sparse_hint_pixels; // pixels the previous sparse frame rendered (0 = unknown): picks the deep march for small lists
  // pixel jitter: 0 = RandomTEA iff spp > 1 (the reference), 1 = blue-noise tile for every sample (jitter_noise: the noise
  // tile stored [t][y][x], jitter_xy its edge; see jitter_slice in ovr_hip_kernels.hip)
  int jitter_mode, jitter_xy;
  const float* jitter_noise;
  // LDS-staged bricks (raymarch_kernel<.., LDSB>): on / off, and where the brick area starts in the workgroup's dynamic LDS
  int lds_staging;
  // workgroups of the persistent shade kernel (0 = the default, 1024): the host takes 768 when the previous frame had few runs of
  // chunks (sparse transfer functions: a smaller window of concurrently shaded requests keeps more of their bricks in L2)
  int shade_blocks;
  unsigned int lds_brick_offset;
  // counters: [0] rays [1] samples [2] shaded samples [3] shadow samples [4] active pixels [5] skipped samples [6] skipped shadow samples
  unsigned long long* counters;
  // (round 4) the frame's last reduction publishes counters[0..7] and the pool's control words to `publish` (pinned host memory: 8 x u64, then
  // kPoolCtrlWords x u32) and zeroes both on the device for the next frame - no memset in front of a frame, no copy behind it.  reduce_done: the
  // reduction's ticket word (zero between launches).  zero_first: this launch may not rely on the last one having cleaned up (first frame, new pool,
  // a failed launch): memset first, as every frame did until round 4.  publish == null: the old protocol (memsets, the caller copies)
  unsigned long long* publish;
  unsigned int* reduce_done;
  int zero_first;
  const float* majorant;        // per-macrocell max TF opacity: empty-space skipping (null = off)
  const unsigned char* occupancy; // per 4^3 macrocells: majorant > 0 in one of them or next to them (set with majorant)
  const unsigned char* occupancy_fine; // the same per macrocell (primary rays refine their skip interval with it)
  unsigned long long* trace;    // diagnostic (OVR_HIP_TRACE=1): 4 words per wave, null otherwise
  unsigned int* block_counters; // workspace: raymarch_grid_blocks() * kBlockCounters per-workgroup partial sums
  PoolDesc pool;
  VolumeDesc vol;
};

// returns hipSuccess or the launch error; ev = 4 events (start, after march, after shade, end) or null
hipError_t launch_raymarch(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev);
size_t pool_shade_blocks();

// dynamic LDS bytes the ray-march kernel needs for this TF (0 when the TF stays in global memory)
size_t raymarch_lds_bytes(int n_color, int n_alpha);
// addressing mode the march / shade kernels take for a layout (0 / 1: 32-bit offsets, 2: 64-bit z table, 3: computed, no LDS tables)
int volume_addressing_mode(const VolumeDesc& vd, int n_color, int n_alpha);
// number of workgroups launch_raymarch will use (size of the block_counters workspace / kBlockCounters)
size_t raymarch_grid_blocks(const RayMarchParams& p);
// sorts the n owned blocks of src (bx | by << 16, any order) by descending ray length into dst; uses p's camera and box
// (workspace: schedule_workspace_elems(n) words).  Round 4 - `exact` (one sample per pixel, no jitter: a pixel's ray is known): the class of a block
// NONE of whose 64 rays meets the volume's box is 0 - tested with the march's own expressions, bit for bit - and nothing else is: those blocks end
// up last in dst, `info` (2 words the device can write - pinned host memory: the caller synchronises the stream and reads them, no copy) receives { blocks with a hit = how many entries of dst need a march / composite workgroup, active pixels of
// the others }.  The march of a C3 frame used to spend its last 100-150 us dispatching ~25 000 workgroups that found no ray to march (77 % of a
// 1920x1080 frame lies outside the box's silhouette); they are no longer launched, their pixels are cleared by launch_clear_blocks.
size_t schedule_workspace_elems(unsigned int n);
hipError_t launch_schedule(const RayMarchParams& p, const unsigned int* src, unsigned int n, unsigned int* dst, unsigned int* workspace, int exact, unsigned int* info,
                           hipStream_t stream);
// zero the pixels (RGBA, gradient layer, optionally the accumulation buffer) of `n` blocks that are not launched: what their rays' miss would write
hipError_t launch_clear_blocks(const RayMarchParams& p, const unsigned int* blocks, unsigned int n, int clear_accum, hipStream_t stream);

// linear (x fastest) -> bricked layout; src may be any reference ValueType, dst is laid out as vd.type says (the VoxelType
// chosen by device_voxel_type() or one of its replicas).  z0/nz_chunk allow chunked uploads from host staging.
int device_voxel_type(int ovr_value_type);
int replica_voxel_type(int base_voxel_type, int layout); // the VoxelType of a layout of a base type, -1 if that replica does not exist
void volume_layout(int voxel_type, int nx, int ny, int nz, VolumeDesc& vd); // fills type, nx.., macros_*, macro_elems, bytes
size_t voxel_size(int voxel_type);
// The taps add the in-plane offsets X(a) + Y(b) of a layout in 32 bits (the 64-bit addressing modes widen only the z term): one z layer of
// macro blocks must hold fewer than 2^32 elements.  False for a slab-shaped volume of ~8192 x 8192 voxels in the quad layout (131072 elements
// per macro block), ~16384 x 16384 in the others: such a layout is not built (replicas) or refused (the general layout).
inline bool layout_offsets_fit(const VolumeDesc& vd)
{
  return (unsigned long long)vd.macro_elems * (unsigned long long)vd.macros_x * (unsigned long long)vd.macros_y <= 0x100000000ull;
}
// slices [z0, z0 + nz_chunk) of the caller's linear array into the layout vd describes.  The launches for all slices together write EVERY element
// of the allocation (padding rows, cells and layers as zeros - never sampled, but they have to be finite -, by the launch of the last slices), so
// dst needs no memset
hipError_t launch_relayout(const void* src_linear, int ovr_value_type, void* dst, const VolumeDesc& vd, int z0, int nz_chunk,
                           hipStream_t stream);
// a replica (vd: one of the _T / _TT / _Q layouts, dst = its storage) built from the volume's resident GENERAL layout: the replicas are
// permutations of its voxels, and building them from it needs no second copy of the caller's data - they can be built later, in the background
hipError_t launch_rebrick(const VolumeDesc& general, void* dst, const VolumeDesc& vd, hipStream_t stream);
// per-axis offset tables of a layout: bytes of the device buffer, and the kernel that fills it ([z: nz + 1 x u64][a: na x u32]
// [b: nb + 1 x u32]) and points vd.axis_z / vd.axis_ab into it
size_t axis_table_bytes(const VolumeDesc& vd);
hipError_t launch_axis_tables(VolumeDesc& vd, void* d_tables, hipStream_t stream);

// sparse-sampling mask (generate_mask.cu:55-120): writes compacted (x,y) pairs, count (int32 elements) to *count
struct SparseMaskParams {
  const float* noise; int noise_xy;
  int width, height, frame_index;
  float mean_x, mean_y, sigma_rcp2, base_noise;
  int32_t* out_xy;
  unsigned int* block_counts;   // workspace: sparse_mask_workspace_elems()
  unsigned long long* count;    // out: number of int32 written
  // order of the compacted list.  0: row-major pixel order - what thrust::remove leaves in the reference (generate_mask.cu:100-120;
  // ovr_hip_sparse_mask, the parity entry).  1: tile-major - 16x16-pixel tiles row by row, inside a tile its sixteen 4x4 sub-tiles
  // one after the other: the frame path's order, so that the 16 rays of a wave are a 4x4 pixel block wherever the mask is dense
  // (the same kept pixels, the same frame: pixels are independent)
  int tile_major;
};
size_t sparse_mask_workspace_elems(int width, int height);
hipError_t launch_sparse_mask(const SparseMaskParams& p, hipStream_t stream);

hipError_t launch_tea(uint32_t* v0v1, float* out, int64_t n, hipStream_t stream);
hipError_t launch_pow(const float* x, const float* y, float* out, int64_t n, int which, hipStream_t stream);
int built_for_exact_parity();

// macrocells (reference accel/sp_singlemc.cu): (min,max) per 16^3 cell once per volume; majorant per cell on every TF change
hipError_t launch_macrocell_ranges(const VolumeDesc& vd, float* out_minmax, hipStream_t stream);
hipError_t launch_macrocell_majorants(const float* minmax, unsigned int count, const float* alphas, int n_alpha, float vr_lo, float vr_hi, float* out,
                                      hipStream_t stream);

// global (min, max) of the macrocell value ranges = the volume's data range as the device's voxel read returns it
// (compute_scalar_range + cuda_scalar_range, array.cpp:27-66,92-108); out = minmax_reduce_floats() floats on the device: the result in
// out[0 .. 1], behind it the partial ranges of the first of the two launches
size_t minmax_reduce_floats();
hipError_t launch_minmax_reduce(const float* minmax, unsigned long long cells, float* out, hipStream_t stream);

// occupancy grids (one byte per 4^3 macrocells / per macrocell, both dilated by one macrocell) for the march's per-ray skip intervals
hipError_t launch_macrocell_coarse(const float* majorant, int nx, int ny, int nz, unsigned char* out_coarse, unsigned char* out_fine, hipStream_t stream);

// tile pack/unpack for the gather payload (RCCL between processes, peer copies / RCCL inside a device group); channels = 4 (RGBA layer) or 3 (gradient layer)
hipError_t launch_pack_tiles(const float* frame, float* dst, int width, int height, int tile_w, int tile_h, int rank, int world,
                             hipStream_t stream, int channels = 4);
// rank >= 0: scatter that rank's payload; rank < 0: scatter all ranks' payloads except skip_rank's, rank r's at src + r * rank_stride_floats
hipError_t launch_unpack_tiles(const float* src, float* frame, int width, int height, int tile_w, int tile_h, int rank, int world,
                               size_t rank_stride_floats, hipStream_t stream, int channels = 4, int skip_rank = -1);
// image_to_rgba8 (imageio.cpp:146-181): RGBA32F frame -> packed RGBA8, optionally flipped vertically
hipError_t launch_rgba8(const float* rgba, uint32_t* out, int width, int height, int flip, hipStream_t stream);
// RGBA32F frame -> RGBA half (4 x uint16 per pixel) with the float -> half rule of the reference's EXR writer
// (imageio.cpp:15-83 via tinyexr), optionally flipped vertically (save_image passes flip = true, imageio.cpp:271)
hipError_t launch_rgba16f(const float* rgba, uint16_t* out, int width, int height, int flip, hipStream_t stream);
int count_owned_tiles(int width, int height, int tile_w, int tile_h, int rank, int world);

} // namespace ovrhip
