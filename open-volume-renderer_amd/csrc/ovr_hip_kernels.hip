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
// (a 4x4-pixel tile) and a workgroup of four waves an 8x8-pixel block; workgroups run longest rays first (launch_schedule).
// Shaded samples become 32-byte requests that a second, persistent kernel shades from a global pool and a third kernel
// composites in ray order (pooled pipeline), or that the tile's own wave shades (in-place pipeline).  The transfer
// function (colour float4 table + alpha table, 20 KiB at the shipped resolution of 1024), the per-axis brick-offset tables,
#include "ovr_hip_device.h"

namespace ovrhip {
static_assert(kPoolSubs <= 64 && (kPoolSubs & (kPoolSubs - 1)) == 0, "reduce_counters_kernel reduces the sub-pool counters with one wave");
// ------------------------------------------------------------------------------------------------------------------
// pooled pipeline, kernel C: every tile walks its chunks in order, composites and writes its pixels
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void composite_kernel(const RayMarchParams P)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ix, iy;
  const bool active = assign_pixel_quad(P, lane, wave, ix, iy) && (lane & 3) == 0; // one owner lane per ray, as in the march
  const PoolDesc& Q = P.pool;
  if (Q.ctrl[1] != 0u) return; // pool overflow: the host re-renders this frame with a larger pool, nothing may be written
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
      if (Q.ctrl[32 * (kPoolSubs + 1)] > Q.sub_capacity) return; // an earlier generation overflowed the pool: the whole frame is re-rendered
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
                                                             int n_shade_blocks, unsigned long long* counters, unsigned int* pool_ctrl,
                                                             unsigned long long* publish, unsigned int* done)
{
  if (pool_ctrl && blockIdx.x == 0 && threadIdx.x < 64) { // most chunks any sub-pool was asked for in any generation (kPoolSubs == 64: one wave)
    unsigned int v = threadIdx.x < (unsigned int)kPoolSubs ? pool_ctrl[32u * (threadIdx.x + 1u)] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, off));
    if (threadIdx.x == 0) atomicMax(&pool_ctrl[32 * (kPoolSubs + 1)], v);
  }
  __shared__ unsigned long long red[4][kNC];
  unsigned long long acc[kNC] = {};
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
  if (!publish) return; // the old protocol, or an earlier generation of a frame with several samples per pixel: the sums go on accumulating
  // The workgroup that finishes last hands the frame's totals and the pool's control words to the host (pinned memory: visible when the
  // stream has been synchronised) and leaves both zeroed for the next frame: no memset in front of a frame, no copy behind it.
  __shared__ unsigned int ticket;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(done, 1u);
  __syncthreads();
  if (ticket != gridDim.x - 1u) return;
  __threadfence();
  if (threadIdx.x < 8) {
    publish[threadIdx.x] = threadIdx.x < kNC ? __hip_atomic_load(&counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    counters[threadIdx.x] = 0ull;
  }
  unsigned int* pub_ctrl = reinterpret_cast<unsigned int*>(publish + 8);
  if (pool_ctrl)
    for (int i = threadIdx.x; i < kPoolCtrlWords; i += 256) {
      pub_ctrl[i] = __hip_atomic_load(&pool_ctrl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pool_ctrl[i] = 0u;
    }
  if (threadIdx.x == 0) *done = 0u;
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
// length of probe ray k of block e inside the box (k = 0: the block's centre, 1 ... 4: its corners), -1 = it misses
__device__ __forceinline__ float schedule_probe(const RayMarchParams& P, const MarchConsts& mc, unsigned int e, int k)
{
  const f3 c0 = ld3(P.cam_dir), h0 = ld3(P.cam_hor), v0 = ld3(P.cam_ver);
  const f3 oo = to_object(mc, ld3(P.cam_pos));
  const int bx = (int)(e & 0xffffu) * 8, by = (int)(e >> 16) * 8;
  const int ix = min(bx + (k == 0 ? 4 : (k & 1) ? 0 : 7), P.width - 1), iy = min(by + (k == 0 ? 4 : (k & 2) ? 0 : 7), P.height - 1);
  const float ux = ((float)ix + .5f) / (float)P.width - 0.5f, uy = ((float)iy + .5f) / (float)P.height - 0.5f;
  const f3 d = normalize3_exact(mk3(c0.x + ux * h0.x + uy * v0.x, c0.y + ux * h0.y + uy * v0.y, c0.z + ux * h0.z + uy * v0.z));
  float a = 0.f, b = FLT_MAX;
  if (intersect_unit_box(a, b, oo, mk3(d.x * mc.inv_scale.x, d.y * mc.inv_scale.y, d.z * mc.inv_scale.z))) return b - a;
  return -1.f;
}
// the block's class from the longest of its five probe rays: 0 = none meets the volume (last), 1 ... kSchedClasses - 1 by length
__device__ __forceinline__ unsigned int schedule_class_of(const RayMarchParams& P, float longest)
{
  if (!(longest >= 0.f)) return 0u;
  const float rel = longest / (P.long_ray_steps * P.step); // 1 = the volume's diagonal
  return (unsigned int)min(kSchedClasses - 1, 1 + (int)(rel * (float)(kSchedClasses - 2)));
}

// Three launches (a single workgroup needed 0.35 ms for the 32 400 blocks of a 1080p frame - on every camera change, i.e. on every frame of
// an interactive session), nothing in front of them and no copy behind (`tools/camera_move_breakdown.py`):
//   schedule_hist_kernel     ceil(n / 1024) workgroups: one histogram row per workgroup, hist[wg][class], + the active pixels of its class-0 blocks
//   schedule_scatter_kernel  workgroup wg's blocks of class c start at (all blocks of longer classes) + (class-c blocks of the
//                            workgroups before wg); inside the workgroup they keep their list order (stable: the host lists the
//                            blocks supertile by supertile, 4x4 blocks = 32x32 pixels, so blocks that run at the same time are
//                            compact squares of the image and share their bricks in L2 / Infinity Cache).  Workgroup 0 writes `info`
//                            (pinned host memory): how many blocks need a workgroup, the pixels of the others
// (One kernel in which every workgroup counts the whole list itself: 38 us instead of 5 + 10 - the per-wave loop over distinct classes is long
// inside the silhouette, where 64 consecutive entries hold ~20 length classes.)
// (Mapping the 16 blocks of a supertile to ONE XCD - workgroup s runs on XCD s % 8 - was measured slower: C3 march 1.63 vs 1.57 ms.)
//   schedule_classify_kernel one WAVE per block: its length class (five probe rays, one per lane) and - `exact` - whether ANY of its 64 pixel rays meets the box
//                            (pixel_ray_hits_box: the march's own test); cls[i] = class | active pixels << 8, class 0 <=> (exact) no ray hits
__global__ __launch_bounds__(256) void schedule_classify_kernel(const RayMarchParams P, const unsigned int* __restrict__ src, unsigned int n, int exact,
                                                               unsigned int* __restrict__ cls)
{
  VolConsts vc;
  MarchConsts mc;
  setup_consts(P, vc, mc);
  const unsigned int i = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (i >= n) return; // wave-uniform
  const int lane = threadIdx.x & 63;
  const unsigned int e = src[i];
  const int ix = (int)(e & 0xffffu) * 8 + (lane & 7), iy = (int)(e >> 16) * 8 + (lane >> 3);
  bool active = ix < P.width && iy < P.height;
  if (P.world > 1 && active) active = ((ix / P.tile_w + iy / P.tile_h) % P.world) == P.rank; // assign_pixel_quad's ownership test
  const bool hit = exact == 1 && active && pixel_ray_hits_box(P, mc, ix, iy);
  const unsigned long long any = __ballot(hit), act = __ballot(active);
  // exact == 2 (several samples per pixel, or jittered ones: a pixel's rays are only known to lie within half a pixel of its centre): the block
  // is surely empty if the cone of ALL rays through the block widened by 1.5 pixels lies beyond one of its own four side planes from the whole
  // box (frustum culling: 4 planes x 8 box corners, one pair per lane, object space), and no ray in it can have a direction component near 0
  // (the reference's box test ignores the slab of such a component, shaders_common.h:162-172: a ray could then "hit" from outside the silhouette)
  bool culled = false;
  if (exact == 2) {
    const float x0 = ((float)((int)(e & 0xffffu) * 8) - 1.5f) / (float)P.width - 0.5f, x1 = ((float)((int)(e & 0xffffu) * 8 + 8) + 1.5f) / (float)P.width - 0.5f;
    const float y0 = ((float)((int)(e >> 16) * 8) - 1.5f) / (float)P.height - 0.5f, y1 = ((float)((int)(e >> 16) * 8 + 8) + 1.5f) / (float)P.height - 0.5f;
    const f3 cd = ld3(P.cam_dir), ch = ld3(P.cam_hor), cv = ld3(P.cam_ver);
    auto ray = [&](int k) { // corner k of the widened block, counter-clockwise; object-space direction (not normalised)
      const float ux = (k == 1 || k == 2) ? x1 : x0, uy = (k >= 2) ? y1 : y0;
      return mk3((cd.x + ux * ch.x + uy * cv.x) * mc.inv_scale.x, (cd.y + ux * ch.y + uy * cv.y) * mc.inv_scale.y, (cd.z + ux * ch.z + uy * cv.z) * mc.inv_scale.z);
    };
    const f3 oo = to_object(mc, ld3(P.cam_pos));
    const int pl = (lane >> 3) & 3, cn = lane & 7;
    const f3 a = ray(pl), b = ray((pl + 1) & 3), in = ray((pl + 2) & 3);
    const f3 nrm = mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    const float side = dot3(nrm, in);                                          // the sign of the cone's inside
    const f3 v = mk3((float)(cn & 1) - oo.x, (float)((cn >> 1) & 1) - oo.y, (float)(cn >> 2) - oo.z);
    const float d = dot3(nrm, v) * (side >= 0.f ? 1.f : -1.f);
    const float scale = sqrtf(dot3(nrm, nrm) * dot3(v, v));
    const bool outside = lane < 32 && fabsf(side) > 0.f && d < -1e-4f * scale;   // this corner lies beyond this plane, with a margin far above the rounding of the products
    const unsigned int bits = (unsigned int)__ballot(outside);
    bool beyond = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) beyond = beyond || ((bits >> (8 * k)) & 0xffu) == 0xffu;
    // direction components over the widened block: linear in (ux, uy), so the extremes are at the corners
    bool sign_safe = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c00 = k == 0 ? ray(0).x : k == 1 ? ray(0).y : ray(0).z, c10 = k == 0 ? ray(1).x : k == 1 ? ray(1).y : ray(1).z;
      const float c11 = k == 0 ? ray(2).x : k == 1 ? ray(2).y : ray(2).z, c01 = k == 0 ? ray(3).x : k == 1 ? ray(3).y : ray(3).z;
      const float lo = fminf(fminf(c00, c10), fminf(c11, c01)), hi = fmaxf(fmaxf(c00, c10), fmaxf(c11, c01));
      const float len = sqrtf(dot3(ray(0), ray(0)));
      sign_safe = sign_safe && (lo > 1e-6f * len || hi < -1e-6f * len);
    }
    culled = beyond && sign_safe;
  }
  float longest = lane < 5 ? schedule_probe(P, mc, e, lane) : -1.f; // (until round 4 lane 0 traced all five: 58 us per camera change at 1080p)
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) longest = fmaxf(longest, __shfl_xor(longest, off));
  if (lane == 0) {
    unsigned int c = schedule_class_of(P, longest);
    if (exact == 1) c = any != 0ull ? max(c, 1u) : 0u;
    if (exact == 2) c = culled ? 0u : max(c, 1u);
    cls[i] = c | ((unsigned int)__popcll(act) << 8);
  }
}

constexpr int kSchedRow = kSchedClasses + 1; // a workgroup's histogram row: its class counts, then the active pixels of its class-0 blocks
__global__ __launch_bounds__(1024) void schedule_hist_kernel(const unsigned int* __restrict__ cls, unsigned int n, unsigned int* __restrict__ hist)
{
  __shared__ unsigned int count[kSchedRow];
  if (threadIdx.x < kSchedRow) count[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned int i = blockIdx.x * 1024u + threadIdx.x;
  if (i < n) {
    const unsigned int c = cls[i];
    atomicAdd(&count[c & 0xffu], 1u);
    if ((c & 0xffu) == 0u) atomicAdd(&count[kSchedClasses], c >> 8);
  }
  __syncthreads();
  if (threadIdx.x < kSchedRow) hist[blockIdx.x * kSchedRow + threadIdx.x] = count[threadIdx.x];
}

__global__ __launch_bounds__(1024) void schedule_scatter_kernel(const unsigned int* __restrict__ src, const unsigned int* __restrict__ cls_in, unsigned int n,
                                                                const unsigned int* __restrict__ hist, unsigned int* __restrict__ dst, unsigned int* __restrict__ info)
{
  __shared__ unsigned int base[kSchedRow], total[kSchedRow];
  __shared__ unsigned int wave_count[16][kSchedClasses];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per class: blocks in the workgroups before this one, and in all of them (last column: the pixels of the class-0 blocks)
  if (threadIdx.x < kSchedRow) {
    unsigned int before = 0u, all = 0u;
    for (unsigned int w = 0; w < gridDim.x; ++w) {
      const unsigned int h = hist[w * kSchedRow + threadIdx.x];
      before += w < blockIdx.x ? h : 0u;
      all += h;
    }
    base[threadIdx.x] = before;
    total[threadIdx.x] = all;
  }
  for (int k = threadIdx.x; k < 16 * kSchedClasses; k += 1024) (&wave_count[0][0])[k] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) { // descending classes: the longest rays first
    unsigned int at = 0u;
    for (int c = kSchedClasses - 1; c >= 0; --c) { const unsigned int t = total[c]; base[c] += at; at += t; }
    // what the host waits for (pinned memory): the blocks that need a workgroup - everything before class 0 - and the active pixels of the others
    if (info && blockIdx.x == 0) { info[0] = n - total[0]; info[1] = total[kSchedClasses]; }
  }
  const unsigned int i = blockIdx.x * 1024u + threadIdx.x;
  const bool valid = i < n;
  const unsigned int e = valid ? src[i] : 0u;
  const unsigned int cls = valid ? (cls_in[i] & 0xffu) : 0u;
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
  // exclusive prefix over the 16 waves, per class
  if (threadIdx.x < kSchedClasses) {
    unsigned int run = base[threadIdx.x];
    for (int w = 0; w < 16; ++w) { const unsigned int t = wave_count[w][threadIdx.x]; wave_count[w][threadIdx.x] = run; run += t; }
  }
  __syncthreads();
  if (valid) dst[wave_count[wave][cls] + rank] = e;
}

size_t schedule_workspace_elems(unsigned int n) { return (size_t)((n + 1023u) / 1024u) * kSchedRow + (size_t)n; } // histogram rows + one class word per block

hipError_t launch_schedule(const RayMarchParams& p, const unsigned int* src, unsigned int n, unsigned int* dst, unsigned int* workspace, int exact, unsigned int* info,
                           hipStream_t stream)
{
  if (n == 0) { // nothing to sort: the caller does not read info
    return hipSuccess;
  }
  const dim3 grid((n + 1023u) / 1024u);
  unsigned int* hist = workspace;
  unsigned int* cls = workspace + (size_t)grid.x * kSchedRow;
  hipLaunchKernelGGL(schedule_classify_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, p, src, n, exact, cls);
  hipLaunchKernelGGL(schedule_hist_kernel, grid, dim3(1024), 0, stream, cls, n, hist);
  hipLaunchKernelGGL(schedule_scatter_kernel, grid, dim3(1024), 0, stream, src, cls, n, hist, dst, info);
  return hipGetLastError();
}

// the pixels of blocks that are not launched (no ray of theirs meets the box): what the march's miss writes - 0 in every layer
// (shaders_raymarching.cu:237-253 + the accumulation of :389-409: accum += 0, rgba = accum / frame_index = 0)
__global__ __launch_bounds__(256) void clear_blocks_kernel(const RayMarchParams P, const unsigned int* __restrict__ blocks, unsigned int n, int clear_accum)
{
  const unsigned int i = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const unsigned int e = blocks[i];
  const int ix = (int)(e & 0xffffu) * 8 + (lane & 7), iy = (int)(e >> 16) * 8 + (lane >> 3);
  bool active = ix < P.width && iy < P.height;
  if (P.world > 1 && active) active = ((ix / P.tile_w + iy / P.tile_h) % P.world) == P.rank;
  if (!active) return;
  const size_t pi = (size_t)ix + (size_t)iy * (size_t)P.width;
  reinterpret_cast<float4*>(P.rgba)[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (P.grad) { float* g = P.grad + 3 * pi; g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }
  if (clear_accum && P.accum) reinterpret_cast<float4*>(P.accum)[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
}
hipError_t launch_clear_blocks(const RayMarchParams& p, const unsigned int* blocks, unsigned int n, int clear_accum, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(clear_blocks_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, p, blocks, n, clear_accum);
  return hipGetLastError();
}

size_t raymarch_lds_bytes(int n_color, int n_alpha)
{
  // the transfer function always lives in LDS; 0 = does not fit next to the request queues (caller reports an error)
  // (+ 32: both tables carry one more entry, a copy of their last one - stage_tf)
  const size_t need = (size_t)n_color * sizeof(float4) + (size_t)n_alpha * sizeof(float) + 32;
  return need <= 96 * 1024 ? need : 0;
}

int volume_addressing_mode(const VolumeDesc& vd, int n_color, int n_alpha) { return addressing_mode(vd, n_color, n_alpha); }

size_t raymarch_grid_blocks(const RayMarchParams& p)
{
  if (p.sparse_xy) return ((size_t)p.width * p.height + kBlock / 4 - 1) / (kBlock / 4);
  return p.n_schedule;
}

hipError_t launch_composite(const RayMarchParams& q, dim3 grid, hipStream_t stream)
{
  hipLaunchKernelGGL(composite_kernel, grid, dim3(kBlock), 0, stream, q);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// shade order by light beams (PoolDesc): the march has left a key per run and the histogram of the keys; every workgroup scans
// the histogram in LDS (G*G entries, the same result everywhere), then a thread per run slot writes its run to its place
// ------------------------------------------------------------------------------------------------------------------
constexpr int kOrderSlots = 4; // run slots per thread of the order kernel
__global__ __launch_bounds__(256) void shade_order_kernel(const RayMarchParams P)
{
  extern __shared__ __attribute__((aligned(16))) unsigned int s_off[]; // G*G exclusive offsets
  __shared__ unsigned int s_wave[4];
  const PoolDesc& Q = P.pool;
  const int n = Q.order_grid * Q.order_grid; // 1024, 4096 or 16384 keys
  // each wave scans a contiguous quarter, 256 entries (one uint4 per lane) at a time, four loads in flight
  const int it = n / 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* h4 = reinterpret_cast<const uint4*>(Q.order_ws + kOrderHist) + wave * (n / 16);
  uint4* o4 = reinterpret_cast<uint4*>(s_off) + wave * (n / 16);
  unsigned int carry = 0;
  for (int g = 0; g < it; g += 4) {
    uint4 h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (g + k < it) ? h4[(g + k) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (g + k < it) {
        const unsigned int sum = h[k].x + h[k].y + h[k].z + h[k].w;
        unsigned int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned int t = __shfl_up(incl, off);
          if (lane >= off) incl += t;
        }
        const unsigned int ex = carry + incl - sum;
        o4[(g + k) * 64 + lane] = make_uint4(ex, ex + h[k].x, ex + h[k].x + h[k].y, ex + h[k].x + h[k].y + h[k].z);
        carry += __shfl(incl, 63);
      }
  }
  if (lane == 0) s_wave[wave] = carry;
  __syncthreads();
  unsigned int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  if (wave > 0)
    for (int j = lane; j < n / 16; j += 64) {
      uint4 v = o4[j];
      v.x += base; v.y += base; v.z += base; v.w += base;
      o4[j] = v;
    }
  __syncthreads();
  if (blockIdx.x == 0) { // the lists' bounds and their ticket counters
    if (threadIdx.x < (unsigned int)kOrderLists) {
      Q.order_ws[kOrderListStart + threadIdx.x] = s_off[threadIdx.x * (n / kOrderLists)];
      Q.order_ws[kOrderTickets + 32 * threadIdx.x] = 0u;
    }
    if (threadIdx.x == 255) Q.order_ws[kOrderListStart + kOrderLists] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  const unsigned int rps = Q.sub_capacity / (unsigned int)kRun, slots = rps * (unsigned int)kPoolSubs; // run slots per sub-pool / of the pool
  // kOrderSlots run slots per thread, their keys loaded together (the chain key -> fill counter -> order entry is all latency)
  unsigned int key[kOrderSlots], gid[kOrderSlots];
#pragma unroll
  for (int k = 0; k < kOrderSlots; ++k) {
    gid[k] = (blockIdx.x * (unsigned int)kOrderSlots + (unsigned int)k) * 256u + threadIdx.x;
    key[k] = kOrderNoKey;
    if (gid[k] < slots) {
      const unsigned int sub = gid[k] / rps, run = gid[k] - sub * rps;
      if (run < min(Q.ctrl[32u * (sub + 1u)], Q.sub_capacity) / (unsigned int)kRun) key[k] = Q.order_key[gid[k]]; // reserved in this generation
    }
  }
#pragma unroll
  for (int k = 0; k < kOrderSlots; ++k) {
    if (key[k] >= (unsigned int)n) continue; // kOrderNoKey: the unused tail of a reservation
    const unsigned int pos = s_off[key[k]] + atomicAdd(&Q.order_ws[kOrderFill + key[k]], 1u);
    if (pos < slots) Q.order[pos] = gid[k]; // (always, unless a launch was lost between a march and its shade kernel)
  }
}

hipError_t launch_shade_order(const RayMarchParams& q, hipStream_t stream)
{
  const unsigned int slots = q.pool.capacity / (unsigned int)kRun;
  const dim3 grid((slots + 256u * kOrderSlots - 1u) / (256u * kOrderSlots));
  if (grid.x == 0) return hipSuccess;
  const size_t lds = (size_t)q.pool.order_grid * q.pool.order_grid * sizeof(unsigned int);
  if (lds >= 48 * 1024) { // 128 x 128 beams: 64 KiB of offsets + the static words (per launch: the attribute belongs to the current device's copy of the kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(shade_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(shade_order_kernel, grid, dim3(256), lds, stream, q);
  return hipGetLastError();
}

hipError_t launch_reduce_counters(const unsigned int* partials, int n_blocks, const unsigned int* shade_partials, int n_shade_blocks,
                                  unsigned long long* counters, unsigned int* pool_ctrl, unsigned long long* publish, unsigned int* done, hipStream_t stream)
{
  hipLaunchKernelGGL(reduce_counters_kernel, dim3(kReduceBlocks), dim3(256), 0, stream, partials, n_blocks, shade_partials, n_shade_blocks, counters, pool_ctrl,
                     done ? publish : nullptr, done);
  return hipGetLastError();
}

extern template hipError_t launch_v<VOX_U8>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_I8>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_U16>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_I16>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_F32>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_F32_T>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_F32_TT>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_U16_T>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_U16_TT>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_F32_Q>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_U16_Q>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
extern template hipError_t launch_v<VOX_U8_Q>(const RayMarchParams&, hipStream_t, const hipEvent_t*);

size_t pool_shade_blocks() { return kShadeBlocks; }

hipError_t launch_raymarch(const RayMarchParams& p, hipStream_t stream, const hipEvent_t* ev)
{
  // ev (optional): ev[0] before the first kernel, ev[1] after the march, ev[2] after the shade kernel (both may be null: no per-phase times), ev[3] at the end
  if (ev) (void)hipEventRecord(ev[0], stream);
  hipError_t e;
  switch (p.vol.type) {
  case VOX_U8: e = launch_v<VOX_U8>(p, stream, ev); break;
  case VOX_I8: e = launch_v<VOX_I8>(p, stream, ev); break;
  case VOX_U16: e = launch_v<VOX_U16>(p, stream, ev); break;
  case VOX_I16: e = launch_v<VOX_I16>(p, stream, ev); break;
  case VOX_F32: e = launch_v<VOX_F32>(p, stream, ev); break;
  case VOX_F32_T: e = launch_v<VOX_F32_T>(p, stream, ev); break;
  case VOX_F32_TT: e = launch_v<VOX_F32_TT>(p, stream, ev); break;
  case VOX_U16_T: e = launch_v<VOX_U16_T>(p, stream, ev); break;
  case VOX_U16_TT: e = launch_v<VOX_U16_TT>(p, stream, ev); break;
  case VOX_F32_Q: e = launch_v<VOX_F32_Q>(p, stream, ev); break;
  case VOX_U16_Q: e = launch_v<VOX_U16_Q>(p, stream, ev); break;
  case VOX_U8_Q: e = launch_v<VOX_U8_Q>(p, stream, ev); break;
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
  case VOX_U8: case VOX_I8: case VOX_U8_Q: return 1;
  case VOX_U16: case VOX_I16: case VOX_U16_T: case VOX_U16_TT: case VOX_U16_Q: return 2;
  default: return 4;
  }
}
int replica_voxel_type(int base, int layout)
{
  if (layout == LAYOUT_GENERAL) return base;
  if (base == VOX_F32) return layout == LAYOUT_THIN ? VOX_F32_T : layout == LAYOUT_THIN_T ? VOX_F32_TT : layout == LAYOUT_QUAD ? VOX_F32_Q : -1;
  if (base == VOX_U16) return layout == LAYOUT_THIN ? VOX_U16_T : layout == LAYOUT_THIN_T ? VOX_U16_TT : layout == LAYOUT_QUAD ? VOX_U16_Q : -1;
  if (base == VOX_U8) return layout == LAYOUT_QUAD ? VOX_U8_Q : -1;
  return -1;
}

template <typename TI, typename TO> struct Conv { static __device__ __forceinline__ TO cv(TI v) { return (TO)v; } };
template <> struct Conv<unsigned int, float> { // integer_normalize<float, uint32_t>, array.h:68-76
  static __device__ __forceinline__ float cv(unsigned int v) { return (float)v / (float)0xffffffffu; }
};
template <> struct Conv<int, float> { // array.h:78-90
  static __device__ __forceinline__ float cv(int v) { const float n = (float)v / (float)0x7fffffff; return n < -1.f ? -1.f : n; }
};

// where a relayout kernel reads voxel (x, y, z) from: the caller's linear array (x fastest; slices [z0, z0 + nz_chunk) of it), or the
// resident GENERAL layout of the same volume (replicas built in the background, round 4: every replica is a permutation of the general layout's voxels)
template <typename TI> struct SrcLinear {
  typedef TI value_type;
  static constexpr bool kLinear = true;
  const TI* p; int nx, ny, z0;
  __device__ __forceinline__ TI get(size_t x, size_t y, unsigned z) const { return p[x + (size_t)nx * (y + (size_t)ny * (size_t)(z - (unsigned)z0))]; }
};
template <int VTB> struct SrcBricked {
  typedef typename Vox<VTB>::T value_type;
  static constexpr bool kLinear = false;
  const value_type* p; unsigned int macro_y; unsigned long long macro_z;
  __device__ __forceinline__ value_type get(size_t x, size_t y, unsigned z) const
  {
    typedef BrickMap<VTB> M;
    return p[(unsigned long long)(M::X((unsigned)x + 1u) + M::Y((unsigned)y, macro_y)) + M::Zlo(z) + (unsigned long long)(z >> 5) * macro_z];
  }
};

// One thread per brick ROW (the SX stored voxels of one (y, z) line of a brick: 8 or 16 bytes), rows taken in the order the layout stores them:
// a workgroup owns the 32 y x 2^bz z slab of one macro row and sweeps it along the pair axis, macro block by macro block - inside a macro block
// that slab is ONE contiguous piece of the layout ((32 >> by) * mbx bricks, 10 KiB for f32), so consecutive lanes store consecutive vectors and
// every 128-byte line leaves the workgroup complete.  (Until round 4 a thread was one stored voxel and a workgroup one source row: a 16-byte
// piece of every line per workgroup, the rest of the line from 7 other workgroups on other XCDs - 8.8 ms for C3's 10 GB, 0.14 of the HBM peak.)
// The source rows of the slab (64 for f32) are read in step along x: each line of the caller's array is fetched once and used up within the workgroup.
template <typename TI, int N> struct __attribute__((packed, aligned(sizeof(TI)))) RowIn { TI v[N]; };
template <typename TO, int N> struct alignas(sizeof(TO) * N) RowOut { TO v[N]; };

template <typename SRC, typename TO, int VT>
__global__ __launch_bounds__(256) void relayout_kernel(const SRC src, TO* __restrict__ dst, int nx, int ny, int macros_a, unsigned int macro_y,
                                                      unsigned long long macro_z, int z0, int nz_chunk, int nz)
{
  typedef typename SRC::value_type TI;
  // layout axes (a, b, z): a = pair axis = x (y in a transposed replica), b = the other one
  // grid: x = macro rows along b, y = z layers (2^bz slices each) that meet [z0, z0 + nz_chunk) - and, in the launch of the volume's last
  // slices, the padding layers up to the end of the allocation ; src holds slices [z0, z0 + nz_chunk)
  typedef BrickMap<VT> M;
  typedef Vox<VT> V;
  constexpr bool TR = V::kTransposed;
  constexpr unsigned SX = M::SX, RB = 1u << (V::by + V::bz);   // stored voxels per row, rows per brick
  constexpr unsigned ROWS = M::sbz / SX;                        // rows of the slab inside one macro block
  constexpr unsigned LZ = 32u >> V::bz;                         // z layers per macro block
  static_assert(ROWS * SX == M::sbz && ROWS == (32u >> V::by) * V::mbx * RB, "the slab is the layout's z-layer stride");
  const int na = TR ? ny : nx, nb = TR ? nx : ny;
  const unsigned my = blockIdx.x, layer = ((unsigned)z0 >> V::bz) + blockIdx.y;
  const unsigned long long slab = (unsigned long long)my * macro_y + (unsigned long long)(layer & (LZ - 1u)) * M::sbz + (unsigned long long)(layer / LZ) * macro_z;
  const unsigned total = (unsigned)macros_a * ROWS;
  for (unsigned g = threadIdx.x; g < total; g += 256u) {
    const unsigned mx = g / ROWS, i = g - mx * ROWS;
    const unsigned rr = i & (RB - 1u), bi = i / RB;
    const unsigned ybk = bi / V::mbx, bm = bi - ybk * V::mbx;
    const unsigned b = my * 32u + (ybk << V::by) + (rr & ((1u << V::by) - 1u));
    const unsigned z = (layer << V::bz) + (rr >> V::by);
    RowOut<TO, SX>* const row = reinterpret_cast<RowOut<TO, SX>*>(dst + slab + (unsigned long long)mx * M::MV + (unsigned long long)i * SX);
    if (b >= (unsigned)nb || z >= (unsigned)nz) { // a padding row: never sampled, but finite (the allocation needs no memset)
      if (z >= (unsigned)z0) {                    // ... written once: by the launch whose slices the row's layer belongs to (or the last one)
        RowOut<TO, SX> zero;
#pragma unroll
        for (unsigned k = 0; k < SX; ++k) zero.v[k] = (TO)0;
        *row = zero;
      }
      continue;
    }
    if (z < (unsigned)z0 || z >= (unsigned)(z0 + nz_chunk)) continue; // another launch's slices
    // the row's stored positions a0 ... a0 + SX - 1 = voxels a0 - 1 ... (clamp addressing baked into the data: position 0 is a copy of voxel 0,
    // positions beyond the grid replicate the last voxel)
    const int v0 = (int)((mx * V::mbx + bm) * V::cx) - 1;
    TI in[SX];
    bool done = false;
    if constexpr (SRC::kLinear && !TR) {
      if (v0 >= 0 && v0 + (int)SX <= nx) { // the whole row lies inside the grid: one (unaligned) vector load
        const RowIn<TI, SX> w = *reinterpret_cast<const RowIn<TI, SX>*>(src.p + ((size_t)v0 + (size_t)nx * ((size_t)b + (size_t)ny * (size_t)(z - (unsigned)z0))));
#pragma unroll
        for (unsigned k = 0; k < SX; ++k) in[k] = w.v[k];
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (unsigned k = 0; k < SX; ++k) {
        const size_t as = (size_t)min(max(v0 + (int)k, 0), na - 1);
        in[k] = TR ? src.get((size_t)b, as, z) : src.get(as, (size_t)b, z);
      }
    }
    RowOut<TO, SX> out;
#pragma unroll
    for (unsigned k = 0; k < SX; ++k) out.v[k] = Conv<TI, TO>::cv(in[k]);
    *row = out;
  }
}

// quad replicas: one thread per cell writes the cell's 2 x 2 (x, y) voxels of its z slice as one 4-vector, neighbours beyond the
// grid replaced by the last voxel (clamp-to-edge addressing).  Cells in storage order, like relayout_kernel's rows: a workgroup owns the
// 32 cells (y) x 2^lz slices slab of one macro row - inside a macro block one contiguous 32 KiB piece - and sweeps it along x, so every
// 128-byte line is written whole (the thread-per-cell-along-x form of round 3 built C3's 17 GB replica in 40 ms), padding cells as zeros.
template <typename SRC, int VT>
__global__ __launch_bounds__(256) void relayout_quad_kernel(const SRC src, typename Vox<VT>::T* __restrict__ dst, int nx, int ny, int macros_x, unsigned int macro_y,
                                                           unsigned long long macro_z, int z0, int nz_chunk, int nz)
{
  typedef typename SRC::value_type TI;
  typedef BrickMap<VT> M;
  typedef Vox<VT> V;
  typedef typename V::T TO;
  typedef typename V::Q Q;
  constexpr unsigned CB = 1u << (V::lx + V::ly + V::lz);          // cells per brick
  constexpr unsigned LAYER = M::bx_ * M::by_ * M::BV;             // elements of one z layer (2^lz slices) of a macro block
  constexpr unsigned CELLS = M::bx_ * M::by_ * CB;                // ... and its cells
  constexpr unsigned LZ = 32u >> V::lz;
  static_assert(CB * 4u == M::BV && CELLS * 4u == LAYER, "a brick is CB cells of 4 elements");
  // grid: x = macro rows along y, y = z layers that meet [z0, z0 + nz_chunk) (+ the padding layers, in the launch of the last slices)
  const unsigned my = blockIdx.x, layer = ((unsigned)z0 >> V::lz) + blockIdx.y;
  const unsigned long long slab = (unsigned long long)my * macro_y + (unsigned long long)(layer & (LZ - 1u)) * LAYER + (unsigned long long)(layer / LZ) * macro_z;
  const unsigned total = (unsigned)macros_x * CELLS;
  for (unsigned g = threadIdx.x; g < total; g += 256u) {
    const unsigned mx = g / CELLS, i = g - mx * CELLS;
    const unsigned j = i & (CB - 1u), brick = i / CB;
    // cell (u, v) = (x + 1, y + 1), x in [-1, nx - 1], y in [-1, ny - 1]
    const int u = (int)(mx * 32u + ((brick & (M::bx_ - 1u)) << V::lx) + (j & ((1u << V::lx) - 1u)));
    const int v = (int)(my * 32u + ((brick / M::bx_) << V::ly) + ((j >> V::lx) & ((1u << V::ly) - 1u)));
    const unsigned z = (layer << V::lz) + (j >> (V::lx + V::ly));
    Q* const cell = reinterpret_cast<Q*>(dst + slab + (unsigned long long)mx * M::MV + (unsigned long long)i * 4u);
    if (u > nx || v > ny || z >= (unsigned)nz) { // a padding cell: never sampled, but finite
      if (z >= (unsigned)z0) { Q zero; zero.x = zero.y = zero.z = zero.w = (TO)0; *cell = zero; }
      continue;
    }
    if (z < (unsigned)z0 || z >= (unsigned)(z0 + nz_chunk)) continue; // another launch's slices
    const int x = max(u - 1, 0), y = max(v - 1, 0);
    const int x1 = min(u, nx - 1), y1 = min(v, ny - 1);
    Q q;
    q.x = Conv<TI, TO>::cv(src.get((size_t)x, (size_t)y, z)); q.y = Conv<TI, TO>::cv(src.get((size_t)x1, (size_t)y, z));
    q.z = Conv<TI, TO>::cv(src.get((size_t)x, (size_t)y1, z)); q.w = Conv<TI, TO>::cv(src.get((size_t)x1, (size_t)y1, z));
    *cell = q;
  }
}
template <typename SRC, int VT>
static hipError_t relayout_quad_s(const SRC& src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  if (nzc <= 0) return hipSuccess;
  const unsigned last = z0 + nzc >= vd.nz ? (unsigned)vd.macros_z * (32u >> Vox<VT>::lz) - 1u : (unsigned)(z0 + nzc - 1) >> Vox<VT>::lz;
  dim3 grid((unsigned)vd.macros_y, last - ((unsigned)z0 >> Vox<VT>::lz) + 1u);
  hipLaunchKernelGGL((relayout_quad_kernel<SRC, VT>), grid, dim3(256), 0, stream, src, (typename Vox<VT>::T*)dst, vd.nx, vd.ny, vd.macros_x, vd.macro_elems * (unsigned)vd.macros_x,
                     (unsigned long long)vd.macro_elems * (unsigned long long)vd.macros_x * (unsigned long long)vd.macros_y, z0, nzc, vd.nz);
  return hipGetLastError();
}
template <typename TI, int VT>
static hipError_t relayout_quad_t(const void* src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  return relayout_quad_s<SrcLinear<TI>, VT>(SrcLinear<TI>{ (const TI*)src, vd.nx, vd.ny, z0 }, dst, vd, z0, nzc, stream);
}

template <typename SRC, typename TO, int VT>
static hipError_t relayout_s(const SRC& src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  if (nzc <= 0) return hipSuccess;
  // the launch of the volume's last slices also writes the padding layers behind them
  const unsigned last = z0 + nzc >= vd.nz ? (unsigned)vd.macros_z * (32u >> Vox<VT>::bz) - 1u : (unsigned)(z0 + nzc - 1) >> Vox<VT>::bz;
  const unsigned layers = last - ((unsigned)z0 >> Vox<VT>::bz) + 1u;
  dim3 grid((unsigned)vd.macros_y, layers);
  hipLaunchKernelGGL((relayout_kernel<SRC, TO, VT>), grid, dim3(256), 0, stream, src, (TO*)dst, vd.nx, vd.ny, vd.macros_x,
                     vd.macro_elems * (unsigned)vd.macros_x, (unsigned long long)vd.macro_elems * (unsigned long long)vd.macros_x * (unsigned long long)vd.macros_y, z0, nzc, vd.nz);
  return hipGetLastError();
}
template <typename TI, typename TO, int VT>
static hipError_t relayout_t(const void* src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  return relayout_s<SrcLinear<TI>, TO, VT>(SrcLinear<TI>{ (const TI*)src, vd.nx, vd.ny, z0 }, dst, vd, z0, nzc, stream);
}

// layout constants the host needs to size the allocation (vd.type, nx, ny, nz and the macro-block geometry of that layout)
template <int VT>
static void volume_layout_t(int nx, int ny, int nz, VolumeDesc& vd)
{
  typedef BrickMap<VT> M;
  const int na = Vox<VT>::kTransposed ? ny : nx, nb = Vox<VT>::kTransposed ? nx : ny;
  // the pair axis holds stored positions 0 ... na + 1 (voxel + 1; lower members 0 ... na); a quad replica has cells 0 ... n on both axes
  vd.macros_x = (na + 1 + (int)M::MCX - 1) / (int)M::MCX;
  vd.macros_y = ((Vox<VT>::kQuad ? nb + 1 : nb) + 31) / 32;
  vd.macros_z = (nz + 31) / 32;
  vd.macro_elems = M::MV;
  vd.bytes = (unsigned long long)vd.macros_x * vd.macros_y * vd.macros_z * M::MV * sizeof(typename Vox<VT>::T);
}
void volume_layout(int voxel_type, int nx, int ny, int nz, VolumeDesc& vd)
{
  vd.type = voxel_type;
  vd.nx = nx; vd.ny = ny; vd.nz = nz;
  switch (voxel_type) {
  case VOX_U8: volume_layout_t<VOX_U8>(nx, ny, nz, vd); break;
  case VOX_I8: volume_layout_t<VOX_I8>(nx, ny, nz, vd); break;
  case VOX_U16: volume_layout_t<VOX_U16>(nx, ny, nz, vd); break;
  case VOX_I16: volume_layout_t<VOX_I16>(nx, ny, nz, vd); break;
  case VOX_F32_T: volume_layout_t<VOX_F32_T>(nx, ny, nz, vd); break;
  case VOX_F32_TT: volume_layout_t<VOX_F32_TT>(nx, ny, nz, vd); break;
  case VOX_U16_T: volume_layout_t<VOX_U16_T>(nx, ny, nz, vd); break;
  case VOX_U16_TT: volume_layout_t<VOX_U16_TT>(nx, ny, nz, vd); break;
  case VOX_F32_Q: volume_layout_t<VOX_F32_Q>(nx, ny, nz, vd); break;
  case VOX_U16_Q: volume_layout_t<VOX_U16_Q>(nx, ny, nz, vd); break;
  case VOX_U8_Q: volume_layout_t<VOX_U8_Q>(nx, ny, nz, vd); break;
  default: volume_layout_t<VOX_F32>(nx, ny, nz, vd); break;
  }
}

template <typename TI>
static hipError_t relayout_f32(const void* src, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  switch (vd.type) {
  case VOX_F32: return relayout_t<TI, float, VOX_F32>(src, dst, vd, z0, nzc, stream);
  case VOX_F32_T: return relayout_t<TI, float, VOX_F32_T>(src, dst, vd, z0, nzc, stream);
  case VOX_F32_TT: return relayout_t<TI, float, VOX_F32_TT>(src, dst, vd, z0, nzc, stream);
  case VOX_F32_Q: return relayout_quad_t<TI, VOX_F32_Q>(src, dst, vd, z0, nzc, stream);
  default: return hipErrorInvalidValue;
  }
}

hipError_t launch_relayout(const void* src, int vt, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  switch (vt) {
  case 100:
    if (vd.type == VOX_U8_Q) return relayout_quad_t<unsigned char, VOX_U8_Q>(src, dst, vd, z0, nzc, stream);
    return relayout_t<unsigned char, unsigned char, VOX_U8>(src, dst, vd, z0, nzc, stream);
  case 101: return relayout_t<signed char, signed char, VOX_I8>(src, dst, vd, z0, nzc, stream);
  case 200:
    switch (vd.type) {
    case VOX_U16: return relayout_t<unsigned short, unsigned short, VOX_U16>(src, dst, vd, z0, nzc, stream);
    case VOX_U16_T: return relayout_t<unsigned short, unsigned short, VOX_U16_T>(src, dst, vd, z0, nzc, stream);
    case VOX_U16_TT: return relayout_t<unsigned short, unsigned short, VOX_U16_TT>(src, dst, vd, z0, nzc, stream);
    case VOX_U16_Q: return relayout_quad_t<unsigned short, VOX_U16_Q>(src, dst, vd, z0, nzc, stream);
    default: return hipErrorInvalidValue;
    }
  case 201: return relayout_t<short, short, VOX_I16>(src, dst, vd, z0, nzc, stream);
  case 300: return relayout_f32<unsigned int>(src, dst, vd, z0, nzc, stream);
  case 301: return relayout_f32<int>(src, dst, vd, z0, nzc, stream);
  case 400: return relayout_f32<float>(src, dst, vd, z0, nzc, stream);
  case 500: return relayout_f32<double>(src, dst, vd, z0, nzc, stream);
  default: return hipErrorInvalidValue;
  }
}

// a replica from the resident general layout (the caller's array is gone by then: ovr_hip_set_volume does not keep it)
template <int VTB>
static hipError_t rebrick_b(const VolumeDesc& g, void* dst, const VolumeDesc& vd, int z0, int nzc, hipStream_t stream)
{
  typedef typename Vox<VTB>::T T;
  const SrcBricked<VTB> src{ (const T*)g.data, g.macro_elems * (unsigned)g.macros_x, (unsigned long long)g.macro_elems * (unsigned long long)g.macros_x * (unsigned long long)g.macros_y };
  constexpr int T1 = VTB == VOX_F32 ? VOX_F32_T : VOX_U16_T, T2 = VTB == VOX_F32 ? VOX_F32_TT : VOX_U16_TT;
  constexpr int TQ = VTB == VOX_F32 ? VOX_F32_Q : VTB == VOX_U16 ? VOX_U16_Q : VOX_U8_Q;
  if (vd.type == TQ) return relayout_quad_s<SrcBricked<VTB>, TQ>(src, dst, vd, z0, nzc, stream);
  if constexpr (VTB != VOX_U8) {
    if (vd.type == T1) return relayout_s<SrcBricked<VTB>, T, T1>(src, dst, vd, z0, nzc, stream);
    if (vd.type == T2) return relayout_s<SrcBricked<VTB>, T, T2>(src, dst, vd, z0, nzc, stream);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_rebrick(const VolumeDesc& general, void* dst, const VolumeDesc& vd, hipStream_t stream)
{
  for (int z0 = 0; z0 < vd.nz; z0 += 32768) { // grid.z limit
    const int nzc = std::min(32768, vd.nz - z0);
    hipError_t e;
    switch (general.type) {
    case VOX_F32: e = rebrick_b<VOX_F32>(general, dst, vd, z0, nzc, stream); break;
    case VOX_U16: e = rebrick_b<VOX_U16>(general, dst, vd, z0, nzc, stream); break;
    case VOX_U8: e = rebrick_b<VOX_U8>(general, dst, vd, z0, nzc, stream); break;
    default: e = hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// per-axis offset tables of a layout (VolumeDesc::axis_ab / axis_z), once per volume
template <int VT>
__global__ __launch_bounds__(256) void axis_tables_kernel(VolumeDesc vd, unsigned int* __restrict__ ab, unsigned long long* __restrict__ tz)
{
  typedef BrickMap<VT> M;
  const int na = Vox<VT>::kTransposed ? vd.ny : vd.nx, nb = Vox<VT>::kTransposed ? vd.nx : vd.ny;
  const unsigned int macro_y = vd.macro_elems * (unsigned int)vd.macros_x;
  const unsigned long long macro_z = (unsigned long long)vd.macro_elems * (unsigned long long)vd.macros_x * (unsigned long long)vd.macros_y;
  // entry i of a table belongs to voxel index i - 1 (VolumeDesc): a = -1 ... na - 1 (lower member of the pair; stored position i),
  // b and z = -1 ... n with the indices clamped into the grid; a quad replica's cells are (x + 1, y + 1), its y entry i is cell i
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i < axis_a_entries(na)) ab[i] = M::X((unsigned)i);
  if (i < axis_b_entries(nb)) ab[axis_a_entries(na) + i] = M::Y(Vox<VT>::kQuad ? (unsigned)min(i, nb) : (unsigned)min(max(i - 1, 0), nb - 1), macro_y);
  if (i < axis_z_entries(vd.nz)) {
    const unsigned z = (unsigned)min(max(i - 1, 0), vd.nz - 1);
    tz[i] = (unsigned long long)M::Zlo(z) + (unsigned long long)(z >> 5) * macro_z;
  }
}
size_t axis_table_bytes(const VolumeDesc& vd)
{
  return (size_t)axis_z_entries(vd.nz) * sizeof(unsigned long long) + (size_t)(vd.nx + vd.ny + 3) * sizeof(unsigned int);
}
template <int VT>
static hipError_t axis_tables_t(const VolumeDesc& vd, unsigned int* ab, unsigned long long* tz, hipStream_t stream)
{
  const int n = std::max(std::max(vd.nx, vd.ny), vd.nz) + 2;
  hipLaunchKernelGGL(axis_tables_kernel<VT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, vd, ab, tz);
  return hipGetLastError();
}
hipError_t launch_axis_tables(VolumeDesc& vd, void* d_tables, hipStream_t stream)
{
  unsigned long long* tz = static_cast<unsigned long long*>(d_tables);
  unsigned int* ab = reinterpret_cast<unsigned int*>(tz + axis_z_entries(vd.nz));
  vd.axis_z = tz;
  vd.axis_ab = ab;
  switch (vd.type) {
  case VOX_U8: return axis_tables_t<VOX_U8>(vd, ab, tz, stream);
  case VOX_I8: return axis_tables_t<VOX_I8>(vd, ab, tz, stream);
  case VOX_U16: return axis_tables_t<VOX_U16>(vd, ab, tz, stream);
  case VOX_I16: return axis_tables_t<VOX_I16>(vd, ab, tz, stream);
  case VOX_F32: return axis_tables_t<VOX_F32>(vd, ab, tz, stream);
  case VOX_F32_T: return axis_tables_t<VOX_F32_T>(vd, ab, tz, stream);
  case VOX_F32_TT: return axis_tables_t<VOX_F32_TT>(vd, ab, tz, stream);
  case VOX_U16_T: return axis_tables_t<VOX_U16_T>(vd, ab, tz, stream);
  case VOX_U16_TT: return axis_tables_t<VOX_U16_TT>(vd, ab, tz, stream);
  case VOX_F32_Q: return axis_tables_t<VOX_F32_Q>(vd, ab, tz, stream);
  case VOX_U16_Q: return axis_tables_t<VOX_U16_Q>(vd, ab, tz, stream);
  case VOX_U8_Q: return axis_tables_t<VOX_U8_Q>(vd, ab, tz, stream);
  default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// macrocells (reference ovr/devices/optix7/accel/sp_singlemc.cu): 16^3-voxel cells with a value range and, per transfer
// function, the largest opacity any sample inside can get.  The reference builds both grids but only its path tracer uses
// them; here the ray marcher and the shadow march skip the voxel fetch of samples whose cell has majorant 0 - such a
// sample's opacity is exactly 0, so frames are bit-identical with and without skipping.
// ------------------------------------------------------------------------------------------------------------------
// value_range_kernel, sp_singlemc.cu:10-54: one WAVE per macrocell (the reference uses one thread), wave min/max reduction.
// A lane owns (y, z) ROWS of the cell's 18^3 box - rows in storage order, so the four lanes of a quad read 64 contiguous bytes - and walks each
// along x brick by brick: one 8- or 16-byte load per brick row (SX stored voxels), the voxels outside the box masked.  (Until round 4: one voxel per
// lane and step, each with its own offset arithmetic and three runtime divisions - 2.8 ms for C3, 20.6 ms for C4, bound by the gather-instruction rate.)
template <int VT>
__global__ __launch_bounds__(256) void macrocell_range_kernel(const void* __restrict__ vol, VolumeDesc vd, int mcx, int mcy, int mcz, float2* __restrict__ out)
{
  typedef BrickMap<VT> M;
  typedef Vox<VT> V;
  typedef typename V::T T;
  constexpr unsigned SX = M::SX, HY = 1u << V::by, HZ = 1u << V::bz;   // a brick: SX x HY x HZ stored voxels
  constexpr int W = 16;
  // brick-aligned spans that hold W + 2 rows at any alignment
  constexpr unsigned AY = (W + 2 + 2 * (HY - 1) + HY - 1) / HY, AZ = (W + 2 + 2 * (HZ - 1) + HZ - 1) / HZ; // in bricks
  const int lane = threadIdx.x & 63;
  const unsigned long long cell = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= (unsigned long long)mcx * mcy * mcz) return;
  const int cx = (int)(cell % mcx), cy = (int)((cell / mcx) % mcy), cz = (int)(cell / ((unsigned long long)mcx * mcy));
  const int bx = max(cx * W - 1, 0), by = max(cy * W - 1, 0), bz = max(cz * W - 1, 0);
  const int ex = min(bx + W + 1, vd.nx), ey = min(by + W + 1, vd.ny), ez = min(bz + W + 1, vd.nz);
  const unsigned macro_y = vd.macro_elems * (unsigned)vd.macros_x;
  const unsigned long long macro_z = (unsigned long long)vd.macro_elems * vd.macros_x * vd.macros_y;
  const T* base = static_cast<const T*>(vol);
  // voxel x is stored at position x + 1: brick (x + 1) / cx, whose row holds the voxels brick * cx - 1 ... brick * cx + cx - 1
  const unsigned b0 = M::div_cx((unsigned)bx + 1u), nbx = M::div_cx((unsigned)ex) - b0 + 1u; // the bricks along x that hold voxels bx ... ex - 1
  const unsigned ya = (unsigned)by & ~(HY - 1u), za = (unsigned)bz & ~(HZ - 1u);
  float lo = INFINITY, hi = -INFINITY; // range1f() is empty
  for (unsigned t = (unsigned)lane; t < AY * AZ * HY * HZ; t += 64u) {
    const unsigned q = t / (HY * HZ);
    const unsigned y = ya + (q % AY) * HY + (t & (HY - 1u)), z = za + (q / AY) * HZ + ((t / HY) & (HZ - 1u));
    if ((int)y < by || (int)y >= ey || (int)z < bz || (int)z >= ez) continue;
    const unsigned long long row = (unsigned long long)M::Y(y, macro_y) + M::Zlo(z) + (unsigned long long)(z >> 5) * macro_z;
    for (unsigned qx = 0; qx < nbx; ++qx) {
      const unsigned br = b0 + qx;
      const unsigned m = M::div_mbx(br), bm = br - m * V::mbx;
      const RowOut<T, SX> w = *reinterpret_cast<const RowOut<T, SX>*>(base + (row + (bm * M::BV + m * M::MV)));
      const int v0 = (int)(br * V::cx) - 1; // the voxel of the row's first element
#pragma unroll
      for (unsigned k = 0; k < SX; ++k) {
        float f = (float)w.v[k];
        if (VT == VOX_U8) f = f / 255.f;                            // what the normalized texture read returns (array.cpp:304-306)
        if (VT == VOX_I8) { f = f / 127.f; f = f < -1.f ? -1.f : f; }
        const bool in = v0 + (int)k >= bx && v0 + (int)k < ex;
        lo = in ? fminf(lo, f) : lo;
        hi = in ? fmaxf(hi, f) : hi;
      }
    }
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
// two launches: kMinmaxBlocks workgroups reduce a slice each into out[2 + 2 b ...], one workgroup reduces those (one workgroup over all cells
// took 0.94 ms for C4's 2 M cells)
constexpr int kMinmaxBlocks = 256;
__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float2* __restrict__ ranges, unsigned long long cells, float* __restrict__ out, int final_pass)
{
  __shared__ float slo[4], shi[4];
  float lo = FLT_MAX, hi = -FLT_MAX; // numeric_limits<float>::max() / lowest()
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < cells; i += stride) {
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
    for (int w = 1; w < 4; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    float* o = final_pass ? out : out + 2 + 2 * blockIdx.x;
    o[0] = lo;
    o[1] = hi;
  }
}
size_t minmax_reduce_floats() { return 2 + 2 * (size_t)kMinmaxBlocks; }
hipError_t launch_minmax_reduce(const float* minmax, unsigned long long cells, float* out, hipStream_t stream)
{
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(kMinmaxBlocks), dim3(256), 0, stream, (const float2*)minmax, cells, out, 0);
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(256), 0, stream, (const float2*)(out + 2), (unsigned long long)kMinmaxBlocks, out, 1);
  return hipGetLastError();
}

// occupancy for the per-ray skip interval (skip_interval), two levels: a coarse grid of 4^3 macrocells (64^3 voxels) per entry -
// small enough (4 KiB at 1024^3) to stay in L1 while every ray walks it - and a fine grid of one entry per macrocell, walked only
// inside the coarse interval.  An entry is set if any of its macrocells, or any macrocell next to one of them (dilation by one
// macrocell), can hold a sample with opacity > 0.  The dilation is the safety margin of the walk: a ray's cell sequence is right
// to ~1e-4 voxel of position, the nearest non-empty sample is >= 16 voxels inside a set entry.   S = macrocells per entry and axis
__global__ __launch_bounds__(256) void macrocell_coarse_kernel(const float* __restrict__ majorant, int mcx, int mcy, int mcz, int S, unsigned char* __restrict__ out)
{
  const int gx = (mcx + S - 1) / S, gy = (mcy + S - 1) / S, gz = (mcz + S - 1) / S;
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (unsigned int)(gx * gy * gz)) return;
  const int cx = (int)(i % (unsigned int)gx) * S, cy = (int)((i / (unsigned int)gx) % (unsigned int)gy) * S, cz = (int)(i / (unsigned int)(gx * gy)) * S;
  bool any = false;
  for (int z = max(cz - 1, 0); z <= min(cz + S, mcz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + S, mcy - 1); ++y)
      for (int x = max(cx - 1, 0); x <= min(cx + S, mcx - 1); ++x)
        any = any || (majorant[(size_t)x + (size_t)mcx * ((size_t)y + (size_t)mcy * (size_t)z)] > 0.f);
  out[i] = any ? 1 : 0;
}
// the coarse grid from the fine one: an entry's dilated 6^3 neighbourhood is the union of the dilated neighbourhoods of its 4^3
// macrocells, so it is the OR of 64 fine entries - one wave per coarse entry, one fine entry per lane (the direct form, one THREAD
// per coarse entry reading 216 majorants, took 0.4 ms at 1024^3: on every transfer-function edit)
__global__ __launch_bounds__(256) void macrocell_coarse_from_fine_kernel(const unsigned char* __restrict__ fine, int mcx, int mcy, int mcz, unsigned char* __restrict__ out)
{
  const int gx = (mcx + 3) / 4, gy = (mcy + 3) / 4, gz = (mcz + 3) / 4;
  const unsigned int i = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (i >= (unsigned int)(gx * gy * gz)) return; // wave-uniform
  const int lane = threadIdx.x & 63;
  const int x = (int)(i % (unsigned int)gx) * 4 + (lane & 3), y = (int)((i / (unsigned int)gx) % (unsigned int)gy) * 4 + ((lane >> 2) & 3),
            z = (int)(i / (unsigned int)(gx * gy)) * 4 + (lane >> 4);
  const bool set = x < mcx && y < mcy && z < mcz && fine[(size_t)x + (size_t)mcx * ((size_t)y + (size_t)mcy * (size_t)z)] != 0;
  const unsigned long long any = __ballot(set);
  if (lane == 0) out[i] = any != 0ull ? 1 : 0;
}
hipError_t launch_macrocell_coarse(const float* majorant, int nx, int ny, int nz, unsigned char* out_coarse, unsigned char* out_fine, hipStream_t stream)
{
  const int mcx = (nx + 15) / 16, mcy = (ny + 15) / 16, mcz = (nz + 15) / 16;
  const unsigned int coarse = (unsigned int)(((mcx + 3) / 4) * ((mcy + 3) / 4) * ((mcz + 3) / 4)), fine = (unsigned int)(mcx * mcy * mcz);
  hipLaunchKernelGGL(macrocell_coarse_kernel, dim3((fine + 255) / 256), dim3(256), 0, stream, majorant, mcx, mcy, mcz, 1, out_fine);
  hipLaunchKernelGGL(macrocell_coarse_from_fine_kernel, dim3((coarse + 3) / 4), dim3(256), 0, stream, out_fine, mcx, mcy, mcz, out_coarse);
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

// enumeration index -> pixel.  Row-major: i = x + y * W.  Tile-major (p.tile_major): 256 consecutive indices are one 16x16-pixel tile in
// Morton order (16 consecutive indices are a 4x4 block, 64 an 8x8 block); pixels beyond the image edge are never kept
__device__ __forceinline__ bool mask_pixel(const SparseMaskParams& p, unsigned int i, int& x, int& y)
{
  if (!p.tile_major) {
    x = (int)(i % (unsigned int)p.width);
    y = (int)(i / (unsigned int)p.width);
    return i < (unsigned int)p.width * (unsigned int)p.height;
  }
  const unsigned int tiles_x = ((unsigned int)p.width + 15u) / 16u;
  const unsigned int tile = i >> 8, in = i & 255u;
#ifndef OVR_MASK_MORTON
#define OVR_MASK_MORTON 1
#endif
  unsigned int lx, ly;
  if (OVR_MASK_MORTON) {
    // Morton (Z) order inside the tile: wherever the mask keeps only part of the pixels, 16 consecutive KEPT pixels - a wave's rays - are
    // still a compact patch (a 4x4 block where everything is kept, ~7x7 pixels at 30 %) instead of a 4-pixel-high strip of sub-tiles;
    // rays that are neighbours share bricks, and the sparse frame is bound by the lines its rays do not share (profiles/r03_notes.md)
    lx = (in & 1u) | ((in >> 1) & 2u) | ((in >> 2) & 4u) | ((in >> 3) & 8u);
    ly = ((in >> 1) & 1u) | ((in >> 2) & 2u) | ((in >> 3) & 4u) | ((in >> 4) & 8u);
  }
  else {
    const unsigned int st = in >> 4, px = in & 15u;
    lx = (st & 3u) * 4u + (px & 3u);
    ly = (st >> 2) * 4u + (px >> 2);
  }
  x = (int)((tile % tiles_x) * 16u + lx);
  y = (int)((tile / tiles_x) * 16u + ly);
  return x < p.width && y < p.height;
}

__device__ __forceinline__ bool mask_keep(const SparseMaskParams& p, const float* lds_noise, unsigned int first_pixel, unsigned int i, int& x, int& y)
{
  if (!mask_pixel(p, i, x, y)) return false;
  const int xy = p.noise_xy;
  float val;
  if (p.tile_major) val = p.noise[(size_t)(p.frame_index % 64) * xy * xy + (size_t)(y % xy) * xy + (x % xy)]; // a 16x16 patch of the slice: cached
  else {
    const int r = y - (int)(first_pixel / (unsigned int)p.width); // 0 or 1 (a third row only if width < 128: read the slice directly)
    val = r < 2 ? lds_noise[r * xy + (x % xy)] : p.noise[(size_t)(p.frame_index % 64) * xy * xy + (size_t)(y % xy) * xy + (x % xy)];
  }
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
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (!p.tile_major) stage_noise_rows(p, lds_noise, blockIdx.x * 256);
  int x, y;
  const bool keep = mask_keep(p, lds_noise, blockIdx.x * 256, i, x, y);
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
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (!p.tile_major) stage_noise_rows(p, lds_noise, blockIdx.x * 256);
  int x = 0, y = 0;
  const bool keep = mask_keep(p, lds_noise, blockIdx.x * 256, i, x, y);
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

static int mask_blocks(const SparseMaskParams& p)
{
  if (p.tile_major) return ((p.width + 15) / 16) * ((p.height + 15) / 16);
  return (int)(((size_t)p.width * p.height + 255) / 256);
}
// per-block counts of either enumeration order (+ 1): the tile-major order has at least as many blocks as the row-major one
size_t sparse_mask_workspace_elems(int width, int height) { return (size_t)((width + 15) / 16) * ((height + 15) / 16) + ((size_t)width * height + 255) / 256 + 1; }

hipError_t launch_sparse_mask(const SparseMaskParams& p, hipStream_t stream)
{
  const int n_blocks = mask_blocks(p);
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

// known-answer entry of the pow the kernels evaluate for __powf (shaders_raymarching.cu:64-66,118-122): which = 0 the one this library was BUILT with
// (v_exp_f32(y * v_log_f32(x)), or the deterministic pair under -DOVR_PARITY_EXACT=1), 1 = the deterministic pair whatever the build
__global__ void pow_kernel(const float* x, const float* y, float* out, long long n, int which)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i], yv = y[i];
  out[i] = (which == 1 || OVR_PARITY_EXACT) ? det_powf(xv, yv) : __builtin_amdgcn_exp2f(yv * __builtin_amdgcn_logf(xv));
}
hipError_t launch_pow(const float* x, const float* y, float* out, int64_t n, int which, hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, out, (long long)n, which);
  return hipGetLastError();
}
int built_for_exact_parity() { return OVR_PARITY_EXACT; }

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
// rank >= 0 scatters that rank's payload, rank < 0 scatters ALL ranks' payloads, laid out `rank_stride` pixels apart, except
// `skip_rank`'s (the in-process device group: the gathering device rendered its own tiles in place).  C = floats per pixel: 4 = the
// RGBA layer, 3 = the gradient layer (device_impl.cpp:271-281 maps both)
template <bool PACK, int C>
__global__ __launch_bounds__(256) void tiles_kernel(const float* __restrict__ src, float* __restrict__ dst, int width, int height,
                                                   int tw, int th, int rank, int world, size_t rank_stride, int skip_rank)
{
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ix >= width || iy >= height) return;
  const int tx = ix / tw, ty = iy / th;
  const int owner = (tx + ty) % world;
  if (rank >= 0 ? owner != rank : owner == skip_rank) return;
  const int tiles_x = (width + tw - 1) / tw;
  const int slot = tile_slot(tiles_x, tx, ty, owner, world);
  const size_t pi = (size_t)slot * tw * th + (size_t)(iy - ty * th) * tw + (size_t)(ix - tx * tw) + (rank < 0 ? (size_t)owner * rank_stride : 0);
  const size_t fi = (size_t)iy * width + ix;
  const size_t si = PACK ? fi : pi, di = PACK ? pi : fi;
  if (C == 4) reinterpret_cast<float4*>(dst)[di] = reinterpret_cast<const float4*>(src)[si];
  else {
#pragma unroll
    for (int c = 0; c < C; ++c) dst[di * C + c] = src[si * C + c];
  }
}

hipError_t launch_pack_tiles(const float* frame, float* dst, int width, int height, int tw, int th, int rank, int world, hipStream_t stream, int channels)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  if (channels == 3) hipLaunchKernelGGL((tiles_kernel<true, 3>), grid, dim3(256), 0, stream, frame, dst, width, height, tw, th, rank, world, (size_t)0, -1);
  else hipLaunchKernelGGL((tiles_kernel<true, 4>), grid, dim3(256), 0, stream, frame, dst, width, height, tw, th, rank, world, (size_t)0, -1);
  return hipGetLastError();
}
hipError_t launch_unpack_tiles(const float* src, float* frame, int width, int height, int tw, int th, int rank, int world, size_t rank_stride_floats,
                               hipStream_t stream, int channels, int skip_rank)
{
  dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4));
  if (channels == 3) hipLaunchKernelGGL((tiles_kernel<false, 3>), grid, dim3(256), 0, stream, src, frame, width, height, tw, th, rank, world, rank_stride_floats / 3, skip_rank);
  else hipLaunchKernelGGL((tiles_kernel<false, 4>), grid, dim3(256), 0, stream, src, frame, width, height, tw, th, rank, world, rank_stride_floats / 4, skip_rank);
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
