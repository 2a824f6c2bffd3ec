// ovr_hip_kernels.h - launch interface between the C-ABI host layer (ovr_hip_api.cpp) and the gfx950 kernels.
// Internal to libovr_hip.so; the public boundary is include/ovr_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovrhip {

// device-resident scalar types of the bricked volume (u32/i32/f64 inputs are converted at upload, see relayout)
enum VoxelType : int { VOX_U8 = 0, VOX_I8 = 1, VOX_U16 = 2, VOX_I16 = 3, VOX_F32 = 4 };

// Volume layout in HBM ("yz-tiled rows"):
//   element (x, y, z) lives at  row(y, z) * row_stride + x,   x in [0, nx]  (element nx replicates element nx-1)
//   row(y, z) = ((z >> 3) * tiles_y + (y >> 3)) * 64 + (z & 7) * 8 + (y & 7)
// Rows stay contiguous in x so the two x-neighbours of a trilinear tap are ONE 8-byte (f32) load; the 8x8 (y,z) tiling
// keeps the 64 rows a wave's 8x8-pixel footprint touches within one 64-row block (256 KiB for nx = 1024 f32).
struct VolumeDesc {
  const void* data;
  int type;        // VoxelType
  int nx, ny, nz;
  int row_stride;  // elements per row, >= nx + 1, multiple of 64 bytes
  int tiles_y;     // ceil(ny / 8)
  int tiles_z;     // ceil(nz / 8)
  float value_scale; // multiplier turning a filtered raw value into what the reference's texture read returns
  float value_min_clamp; // raw clamp applied per voxel before filtering (i8: -127) - see array.h:83-90
};

struct float3_ { float x, y, z; };

struct RayMarchParams {
  // framebuffer (optix7/params.h:56-63)
  float* rgba;   // W*H*4
  float* grad;   // W*H*3 (may be null)
  float* accum;  // W*H*4 (may be null when !accumulate)
  int width, height;
  int frame_index;  // 1-based
  int accumulate;
  int spp;
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
  float tf_lower, tf_upper, tf_scale; // volume.cpp:131-145
  const float* tf_color;      // n_color * 4 (rgb, 1)
  const float* tf_alpha;      // n_alpha
  int n_color, n_alpha;
  int shading;                // OVR_HIP_SHADE_*
  // image-plane shard
  int rank, world, tile_w, tile_h;
  // sparse sampling: compacted (x,y) list + device-side count (2 * pixels), null in dense mode
  const int32_t* sparse_xy;
  const unsigned long long* sparse_count;
  // counters: [0] rays [1] samples [2] shaded samples [3] shadow samples [4] active pixels
  unsigned long long* counters;
  VolumeDesc vol;
};

// returns hipSuccess or the launch error
hipError_t launch_raymarch(const RayMarchParams& p, hipStream_t stream);

// dynamic LDS bytes the ray-march kernel needs for this TF (0 when the TF stays in global memory)
size_t raymarch_lds_bytes(int n_color, int n_alpha);

// linear (x fastest) -> yz-tiled rows; src may be any reference ValueType, dst is the VoxelType chosen by
// device_voxel_type().  z0/nz_chunk allow chunked uploads from host staging.
int device_voxel_type(int ovr_value_type);
size_t voxel_size(int voxel_type);
hipError_t launch_relayout(const void* src_linear, int ovr_value_type, void* dst, const VolumeDesc& vd, int z0, int nz_chunk,
                           hipStream_t stream);

// sparse-sampling mask (generate_mask.cu:55-120): writes compacted (x,y) pairs, count (int32 elements) to *count
struct SparseMaskParams {
  const float* noise; int noise_xy;
  int width, height, frame_index;
  float mean_x, mean_y, sigma_rcp2, base_noise;
  int32_t* out_xy;
  unsigned int* block_counts;   // workspace: ceil(W*H/256) + 1
  unsigned long long* count;    // out: number of int32 written
};
size_t sparse_mask_workspace_elems(int width, int height);
hipError_t launch_sparse_mask(const SparseMaskParams& p, hipStream_t stream);

hipError_t launch_tea(uint32_t* v0v1, float* out, int64_t n, hipStream_t stream);

// tile pack/unpack for the RCCL gather payload
hipError_t launch_pack_tiles(const float* frame, float* dst, int width, int height, int tile_w, int tile_h, int rank, int world,
                             hipStream_t stream);
hipError_t launch_unpack_tiles(const float* src, float* frame, int width, int height, int tile_w, int tile_h, int rank, int world,
                               hipStream_t stream);
int count_owned_tiles(int width, int height, int tile_w, int tile_h, int rank, int world);

} // namespace ovrhip
