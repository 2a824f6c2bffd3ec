// ovr_hip_api.cpp - C ABI (include/ovr_hip.h) of the MI355X ray-marching backend: host-side state machine.
//
// Mirrors the host half of the reference's GPU device (citations relative to the reference tree):
//   queued setters + commit diffing ... ovr/renderer.h:135-285, ovr/devices/optix7/device_impl.cpp:113-197
//   render / frame_index / reset ...... device_impl.cpp:199-269
//   double-buffered framebuffer ....... ovr/devices/optix7/optix7_common.h:328-414, device_impl.cpp:102-111,271-281
//   volume + TF upload ................ volume.cpp:110-179, array.cpp:287-351
// No CPU fallback exists: every entry point needs a HIP device and fails with OVR_HIP_EDEVICE otherwise.
#include "../../include/ovr_hip.h"
#include "ovr_hip_kernels.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace ovrhip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                                  \
  do {                                                                                                                 \
    hipError_t e__ = (expr);                                                                                           \
    if (e__ != hipSuccess)                                                                                             \
      return fail(OVR_HIP_EDEVICE, std::string("[hip] ") + #expr + " failed: " + hipGetErrorString(e__));            \
  } while (0)

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(V3 a, V3 b) { return std::fmaf(a.x, b.x, std::fmaf(a.y, b.y, a.z * b.z)); }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline V3 normalize(V3 v) { const float l = std::sqrt(dot(v, v)); return { (v.x * 1.f) / l, (v.y * 1.f) / l, (v.z * 1.f) / l }; }
inline V3 scale(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }

// array.h:68-106
float integer_normalize(float value, int type)
{
  switch (type) {
  case OVR_HIP_TYPE_UINT8: return (float)(uint8_t)value / 255.f;
  case OVR_HIP_TYPE_INT8: { const float n = (float)(int8_t)value / 127.f; return n < -1.f ? -1.f : n; }
  case OVR_HIP_TYPE_UINT16: return (float)(uint16_t)value / 65535.f;
  case OVR_HIP_TYPE_INT16: { const float n = (float)(int16_t)value / 32767.f; return n < -1.f ? -1.f : n; }
  case OVR_HIP_TYPE_UINT32: return (float)(uint32_t)value / (float)UINT32_MAX;
  case OVR_HIP_TYPE_INT32: { const float n = (float)(int32_t)value / (float)INT32_MAX; return n < -1.f ? -1.f : n; }
  default: return value;
  }
}
// the ValueType the reference's device volume ends up with (array.cpp:322-347): u16 / i16 / f64 are sampled as RAW float
int device_value_type(int t)
{
  if (t == OVR_HIP_TYPE_UINT16 || t == OVR_HIP_TYPE_INT16 || t == OVR_HIP_TYPE_DOUBLE) return OVR_HIP_TYPE_FLOAT;
  return t;
}
size_t value_type_size(int t)
{
  switch (t) {
  case OVR_HIP_TYPE_UINT8: case OVR_HIP_TYPE_INT8: return 1;
  case OVR_HIP_TYPE_UINT16: case OVR_HIP_TYPE_INT16: return 2;
  case OVR_HIP_TYPE_UINT32: case OVR_HIP_TYPE_INT32: case OVR_HIP_TYPE_FLOAT: return 4;
  case OVR_HIP_TYPE_DOUBLE: return 8;
  default: return 0;
  }
}

template <typename T> struct Queued { // vidi::TransactionalValue (ovr/common/vidi_transactional_value.h:26-168), minus the template
  T queued{}, current{};
  bool dirty = false;
  void set(const T& v) { queued = v; dirty = true; }
  bool update() { if (!dirty) return false; current = queued; dirty = false; return true; }
};

struct CameraP { float from[3] = { 0, 0, 0 }, at[3] = { 0, 0, -1 }, up[3] = { 0, 1, 0 }; float fovy = 60.f; };
struct FocusP { float cx = 0.5f, cy = 0.5f, scale = 0.2f, base_noise = 0.1f; }; // params.h:82-84
struct TfnP { std::vector<float> colors, alphas; float lo = 1, hi = -1; };
struct ShardP { int rank = 0, world = 1, tw = 64, th = 64; };
struct Size2 { int w = 0, h = 0; };

} // namespace

// the frame words on the device (ovr_hip_renderer::d_counters): 8 counters, the pool's control words from byte 128, the reduction's ticket word behind them
struct GroupWorker; // one host thread per follower of a device group (below)

constexpr size_t kFrameWordsBytes = 128 + (size_t)ovrhip::kPoolCtrlWords * sizeof(unsigned int) + 128;

struct ovr_hip_renderer {
  int device = 0;
  std::mutex mtx; // protects the queued values (setters may come from any thread)
  hipStream_t own_stream[2] = { nullptr, nullptr };
  hipStream_t user_stream = nullptr;
  bool use_user_stream = false;
  hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };

  Queued<Size2> fbsize;
  Queued<CameraP> camera;
  Queued<TfnP> tfn;
  Queued<FocusP> focus;
  Queued<int> spp, sparse, accumulate, shading, grid_convention, pipeline, skipping, jitter, lds_staging;
  Queued<float> rate;
  Queued<ShardP> shard;

  // volume: d_volume / vd = the general layout (always resident); replica[1..2] = the thin layouts, [3] = the quad layout (ovr_hip_kernels.h), if built
  void* d_volume = nullptr;
  std::atomic<size_t> volume_bytes{ 0 }; // all resident replicas (a builder thread adds its replica's)
  VolumeDesc vd{};
  void* d_replica[kLayouts] = {};  // [3] = the quad replica (f32 volumes)
  void* d_axis[kLayouts] = {}; // per layout: its per-axis offset tables (VolumeDesc::axis_ab / axis_z)
  VolumeDesc vd_replica[kLayouts] = {};
  // Replicas are built in the background (round 4): ovr_hip_set_volume uploads the general layout only and PLANS the others (which exist
  // for the type, fit the 40 % rule and the 32-bit in-plane offsets); a replica is allocated and re-bricked from the general layout on
  // build_stream the first time the layout rule, the tuner or a forced choice asks for it.  Every layout gives the same frame bit for bit,
  // so frames keep rendering from the general layout until the build has finished (a forced choice and the tuner's probes wait for it on the
  // device).  C3: 40 GB and four relayout passes at every ovr_hip_set_volume became 5.9 GB and one.
  //   0 = no such replica, 1 = planned (vd_replica[k] is its geometry, nothing allocated), 4 = a builder thread is allocating it and
  //   enqueueing its construction (hipMalloc of C3's 8.8 GB thin replica takes 0.25 s of HOST time: not on the render thread),
  //   2 = being built on the device (build_ev[k] pending), 3 = resident.  The builder publishes vd_replica[k] / d_replica[k] / d_axis[k]
  //   before it stores 2 (release); the render thread reads them only after it has seen 2 or 3 (acquire).
  std::atomic<int> replica_state[kLayouts] = {};
  std::thread builder[kLayouts];
  hipStream_t build_stream = nullptr;
  hipEvent_t build_ev[kLayouts] = {};
  Queued<int> layouts;      // ovr_hip_set_volume_layouts: which replicas the next ovr_hip_set_volume plans (2: builds at once)
  Queued<int> layout_choice; // -1 = automatic (camera direction / the previous frame's work), 0..3 = forced (falls back to general if not resident)
  // Automatic choice of the volume layout and of the shading pipeline, second stage (round 3).  The rules of round 2 - thin replicas for
  // axis views, in place once half the samples are shaded - were fitted on frames bound by HBM bytes.  Frames that shade (nearly) every
  // sample - most of the reference's shipped scenes, every frame at the scene files' sampling rate 4 - are bound by the gather rate and
  // the instruction stream instead, and what is fastest there depends on the volume's size, the view and the sampling rate in ways no
  // rule captured (profiles/r03_notes.md: quad replica -7 ... -16 % on the big scenes, +25 % on C3 dense at rate 1; pooled 16 ms vs in place
  // 24 ms on C3 front / dense at rate 4, the opposite at rate 1).  Every layout and both pipelines give the same frame bit for bit, so the
  // renderer measures: the first frame of a configuration runs what the rules say; if its shading taps outnumber its primary taps, the
  // following frames try the alternatives - first the other pipeline, then the other candidate layouts (general, quad) under the better
  // pipeline - two frames each, the second one timed (the first one pages the replica in and sizes the request pool), and the fastest
  // (hipEvent kernel time) is kept until the configuration changes.  OVR_HIP_TUNE=0 keeps the rules alone.
  struct TuneCand { int layout, pipeline, frames; float ms; };
  int tune_state = 0;          // 0 = the next frame is the first of a configuration, 1 = probing, 2 = decided
  int tune_phase = 0;          // probing: 0 = pipelines, 1 = layouts
  TuneCand tune_cand[6] = {};
  int tune_n = 0, tune_cur = 0;
  int tune_layout = -1, tune_pipeline = 0; // decided (-1 / 0 = leave it to the rules)
  int tune_rule_choice = -1;   // what the layout rule said for the camera the measurement was made with: a measured LAYOUT is only kept
                               // across camera moves while the rule still says the same (the thin replicas are view-dependent)
  int tune_frame = -1;         // candidate index of the frame in flight, -1 = not a tuned frame
  bool tune_on = true;
  bool shade_order_on = true; // OVR_HIP_SHADE_ORDER, read when the renderer is created (shade_order_params)
  float shade_beam = 32.f;    // OVR_HIP_SHADE_BEAM: voxels across a light beam
  // A camera that moves on every frame (the interactive app) never sits still long enough to be measured: when only the camera changed, a
  // measured decision is kept (the regime - transfer function, sampling rate, volume - is what decides, not the view), dropped as soon as a
  // frame is no longer shade-heavy, and measured again once the configuration has been static for tune_recheck frames.
  int tune_recheck = 0;
  int value_type = 0;
  float origin[3] = { 0, 0, 0 }, spacing[3] = { 1, 1, 1 };
  bool have_volume = false;

  // macrocells (empty-space skipping)
  float* d_mc_minmax = nullptr;
  float* d_mc_majorant = nullptr;
  unsigned char* d_mc_occupancy = nullptr; // coarse, dilated occupancy (skip intervals of the march)
  unsigned char* d_mc_fine = nullptr;      // per-macrocell dilated occupancy (the primary rays' refinement of those intervals)
  // Empty-space skipping pays only where there is empty space: with nothing to skip its kernels cost 20-50 % more than the plain ones
  // (dense transfer functions, a camera inside the data).  While skipping is enabled the renderer therefore watches what it skips: a
  // frame that skipped < 10 % of its sample steps switches to the plain kernels (the frames are bit-identical either way) and the
  // skipping kernels are probed again after 32, 64, ... 256 frames, or at once when the transfer function or the volume changes.
  // OVR_HIP_SKIP_ADAPTIVE=0 keeps the skipping kernels whatever they skip (measurements).
  bool skip_adaptive = true, skip_active = true, frame_used_skip = false;
  // Shading pipeline in automatic mode (ovr_hip_set_shading_pipeline(0)): pooling the shading requests pays when few samples are
  // shaded and they sit in few tiles (sparse transfer functions: 2.7 vs 3.4 ms on C3); when most samples are shaded every tile has
  // the same work and the requests' round trip through HBM (32 B written, read, written and read again per sample) only costs -
  // dense transfer functions 1.57 vs 1.99 ms, the shipped mechhand scene 3.9 vs 5.4 ms.  The renderer follows the previous frame:
  // >= 50 % of its samples shaded -> in place, < 35 % -> pooled (both give the same frame bit for bit).
  bool auto_inplace = false;
  int skip_reprobe_in = 0, skip_backoff = 32;
  size_t mc_cells = 0;
  bool mc_ranges_valid = false, mc_majorant_valid = false;
  float data_lower = 0.f, data_upper = 0.f; // the volume's data range as the voxel read returns it (array.cpp:297)
  float* d_data_range = nullptr;

  // transfer function
  float* d_tf_color = nullptr;
  float* d_tf_alpha = nullptr;   // inside d_tf_color's allocation: the colours (n_color x 4 floats), then the alphas
  float* h_tf = nullptr;         // pinned staging of both tables
  hipEvent_t ev_tf = nullptr;    // behind the last table copy
  bool tf_copy_pending = false;  // a frame has not yet been ordered behind ev_tf
  int n_color = 0, n_alpha = 0;
  bool have_tfn = false;

  // framebuffer sets
  float* d_rgba[2] = { nullptr, nullptr };
  float* d_grad[2] = { nullptr, nullptr };
  float* h_rgba[2] = { nullptr, nullptr };
  float* h_grad[2] = { nullptr, nullptr };
  // the pixel rectangle each host mirror was last refreshed in ({x0, y0, x1, y1}, empty at first): mapframe(HOST) copies only the
  // rectangle the volume's box projects into - every pixel outside it is exactly 0 on the device and stays 0 in the mirror
  int h_rgba_rect[2][4] = {}, h_grad_rect[2][4] = {};
  int d_rect[2][4] = {}; // per framebuffer set: the rectangle of the camera its last frame was rendered with (empty: never rendered, all zeros)
  float* d_accum = nullptr;
  uint32_t* d_rgba8 = nullptr; // mapframe_rgba8: device and pinned host copy of the 8-bit frame
  uint32_t* h_rgba8 = nullptr;
  uint16_t* d_rgba16f = nullptr; // mapframe_rgba16f: the half frame of the EXR writer
  uint16_t* h_rgba16f = nullptr;
  float* d_spp_rgba = nullptr; // pooled pipeline, spp > 1: sums over the sample-per-pixel generations
  float* d_spp_grad = nullptr;
  size_t fb_pixels = 0;
  int cur = 0;
  bool fb_reset = true;
  int frame_index = 0;
  bool camera_dirty = true;

  // sparse sampling
  float* d_noise = nullptr;
  int noise_xy = 0;
  int32_t* d_sparse_xy = nullptr;
  unsigned int* d_block_counts = nullptr;
  unsigned long long* d_sparse_count = nullptr;
  size_t sparse_pixels = 0;
  unsigned long long sparse_prev_pixels = 0; // pixels of the previous sparse frame (RayMarchParams::sparse_hint_pixels)

  // counters
  unsigned long long* d_counters = nullptr;  // start of the frame words: [0, 64) the counters, [128, 128 + ctrl) the pool's control words, then the reduction's ticket
  std::atomic<bool> phase_timing{ true };    // ovr_hip_set_phase_timing (any thread)
  bool frame_phase_timed = true;             // ... as the frame in flight was launched
  bool frame_words_dirty = true;             // the next launch may not rely on the last frame's reduction having zeroed them (RayMarchParams::zero_first)
  unsigned int* d_block_counters = nullptr; // per-workgroup partial counters
  unsigned long long* d_trace = nullptr;    // OVR_HIP_TRACE=1 diagnostic buffer
  size_t trace_words = 0;
  unsigned long long* h_counters = nullptr; // pinned
  bool outside_hits = false;   // a frame of the running accumulation had a ray that hit the box from outside its silhouette (see finish_frame)
  int frame_set = 0;           // the framebuffer set the frame in flight renders into

  // launch order of the march's 8x8-pixel workgroups (dense mode): owned blocks, longest rays first
  unsigned int* d_sched_src = nullptr; // owned blocks in image order (host-built: framebuffer size and shard)
  unsigned int* d_sched = nullptr;     // sorted on the device whenever the camera or the volume's box changes
  unsigned int n_sched = 0;
  bool sched_list_dirty = true, sched_dirty = true;
  // Blocks none of whose rays meets the volume's box are not launched (round 4; launch_schedule's `exact` classification puts them last in the
  // list): n_work = the entries that get a march / composite workgroup, empty_pixels = the active pixels of the others (they count as rays and
  // rendered pixels like before).  Their pixels are zeroed by launch_clear_blocks - once per framebuffer set after anything changed (clear_gen),
  // and on the first frame of an accumulation (the accumulation buffer's turn).
  unsigned int* d_sched_info = nullptr;
  unsigned int n_work = 0, empty_pixels = 0, frame_empty_pixels = 0;
  unsigned long long frame_empty_rays = 0;
  unsigned int clear_gen = 1, set_clear_gen[2] = { 0, 0 };
  int sched_exact = 0;

  // request pool of the pooled shading pipeline
  PoolDesc pool{};
  size_t pool_tiles = 0;
  unsigned int* h_ctrl = nullptr; // pinned copy of pool.ctrl

  RayMarchParams P{};
  ovr_hip_stats stats{};
  double render_time_ms = 0.0;
  bool async_pending = false;
  // pipelined gather (ovr_hip_pack_tiles before the frame is known to be complete): allowed while the request pool has proven
  // roomy for this configuration - the last frame since the last commit that changed anything used at most half of it
  bool pool_roomy = false;
  bool packed_early = false; // the pending frame's tiles were packed without waiting for it

  // ---- device group (ovr_hip_create_group): one process drives several GPUs, each member renders the image tiles of rank group_rank of
  // group_size (ovr_hip_set_image_shard semantics, volume replicated) and the leader - members[0], the handle the caller holds - gathers the
  // followers' tiles into its own framebuffer at the end of every frame.  A follower is an ordinary renderer whose `leader` is set.
  std::vector<ovr_hip_renderer*> members; // leader only: every member, itself first
  // A setter on a group queues its value on every member; ovr_hip_commit applies the queued values member by member.  Both hold this mutex of the
  // leader, so a commit on the render thread never falls between the members of one setter call on the GUI thread (a frame whose tiles come from
  // two cameras); the reference's TransactionalValue gives the same guarantee per value (vidi_transactional_value.h:75-103)
  std::mutex group_mtx;
  ovr_hip_renderer* leader = nullptr;     // follower only
  int group_rank = 0;
  int gather_kind = 0;                    // leader: 0 = no group, 1 = peer copies, 2 = RCCL send / recv
  hipStream_t comm_stream = nullptr;      // every member: where its payload travels (the render stream goes on with the next frame)
  hipEvent_t ev_packed = nullptr, ev_shipped = nullptr; // on the member's own device: its tiles are packed / have left on its comm_stream (peer copies)
  hipEvent_t ev_received = nullptr;       // follower, created on the LEADER's device: its tiles have arrived on the leader's comm_stream (RCCL) - an
                                          // event is recorded on a stream of its own device only
  bool shipped_by_rccl = false;           // which of the two events the leader's scatter waits for
  float* d_payload[2] = { nullptr, nullptr };  // follower: its packed tiles, [0] RGBA, [1] gradient layer
  float* d_gather[2] = { nullptr, nullptr };   // leader: every member's payload, group_stride[] floats apart
  size_t payload_floats[2] = { 0, 0 }, group_stride[2] = { 0, 0 };
  int group_fb[5] = { 0, 0, 0, 0, 0 };    // what the gather buffers were sized for: W, H, tile_w, tile_h, size
  bool group_grad = true;                 // the gradient layer travels too (OVR_HIP_MAP_GRAD=0: RGBA only, like the plugin's mapframe)
  void* rccl_comm = nullptr;              // every member of an RCCL group: its communicator
  ovr_hip_stats own_stats{};              // leader: its own frame's counters (stats holds the group's sums)
  double group_gather_ms = 0.0;           // leader: host time of the last frame's gather tail (after the slowest member's frame: wait for the shipments + scatter)
  // Round 5: every follower has a host thread of its own (device set once, persistent): the leader posts commit / render_async / ship / finish to all of
  // them and waits on their counters, so the members' frames are enqueued side by side instead of one after the other (one thread driving 8 members
  // spent 225 us per frame on it, beside 485 us of device time per member).
  double upload_ms[4] = { 0, 0, 0, 0 };   // the last ovr_hip_set_volume of THIS renderer: total, allocation, copies into the device, kernels (ovr_hip_get_upload_times)
  double group_upload_ms = 0.0;           // leader: wall time of the last upload over all members
  GroupWorker* worker = nullptr;          // follower only
  hipEvent_t ev_gathered = nullptr;       // leader, RCCL: behind the ONE grouped receive of a frame on its comm_stream
  // A member whose commit or volume upload failed leaves the group in mixed state: the group refuses to render until a commit has
  // succeeded on every member AND they agree on what was committed / until a volume upload has succeeded on all of them
  bool group_broken = false;
  std::string group_broken_why;
  double group_host_us[4] = { 0, 0, 0, 0 }; // leader: host time of the last frame's steps (enqueue, ship, finish, scatter), microseconds

  hipStream_t stream() const { return use_user_stream ? user_stream : own_stream[cur]; }
};

namespace {

int set_device(ovr_hip_renderer* r) { HIP_TRY(hipSetDevice(r->device)); return 0; }

int free_framebuffers(ovr_hip_renderer* r)
{
  for (int i = 0; i < 2; ++i) {
    if (r->d_rgba[i]) HIP_TRY(hipFree(r->d_rgba[i]));
    if (r->d_grad[i]) HIP_TRY(hipFree(r->d_grad[i]));
    if (r->h_rgba[i]) HIP_TRY(hipHostFree(r->h_rgba[i]));
    if (r->h_grad[i]) HIP_TRY(hipHostFree(r->h_grad[i]));
    r->d_rgba[i] = r->d_grad[i] = r->h_rgba[i] = r->h_grad[i] = nullptr;
  }
  if (r->d_accum) HIP_TRY(hipFree(r->d_accum));
  r->d_accum = nullptr;
  if (r->d_rgba8) HIP_TRY(hipFree(r->d_rgba8));
  if (r->h_rgba8) HIP_TRY(hipHostFree(r->h_rgba8));
  r->d_rgba8 = nullptr; r->h_rgba8 = nullptr;
  if (r->d_rgba16f) HIP_TRY(hipFree(r->d_rgba16f));
  if (r->h_rgba16f) HIP_TRY(hipHostFree(r->h_rgba16f));
  r->d_rgba16f = nullptr; r->h_rgba16f = nullptr;
  if (r->d_spp_rgba) HIP_TRY(hipFree(r->d_spp_rgba));
  if (r->d_spp_grad) HIP_TRY(hipFree(r->d_spp_grad));
  r->d_spp_rgba = r->d_spp_grad = nullptr;
  if (r->d_block_counters) HIP_TRY(hipFree(r->d_block_counters));
  r->d_block_counters = nullptr;
  if (r->d_sched_src) HIP_TRY(hipFree(r->d_sched_src));
  if (r->d_sched) HIP_TRY(hipFree(r->d_sched));
  r->d_sched_src = r->d_sched = nullptr;
  r->n_sched = 0;
  r->sched_list_dirty = r->sched_dirty = true;
  r->clear_gen++;
  if (r->pool.tile_first) HIP_TRY(hipFree(r->pool.tile_first));
  if (r->pool.tile_count) HIP_TRY(hipFree(r->pool.tile_count));
  if (r->pool.pix_state) HIP_TRY(hipFree(r->pool.pix_state));
  r->pool.tile_first = nullptr; r->pool.tile_count = nullptr; r->pool.pix_state = nullptr;
  if (r->d_sparse_xy) HIP_TRY(hipFree(r->d_sparse_xy));
  if (r->d_block_counts) HIP_TRY(hipFree(r->d_block_counts));
  r->d_sparse_xy = nullptr;
  r->d_block_counts = nullptr;
  r->sparse_pixels = 0;
  r->fb_pixels = 0;
  return 0;
}

int resize_framebuffers(ovr_hip_renderer* r, int w, int h)
{
  HIP_TRY(hipDeviceSynchronize()); // device_impl.cpp:117 stops all async rendering first
  if (int e = free_framebuffers(r)) return e;
  const size_t n = (size_t)w * (size_t)h;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipMalloc((void**)&r->d_rgba[i], n * 4 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&r->d_grad[i], n * 3 * sizeof(float)));
    HIP_TRY(hipMemset(r->d_rgba[i], 0, n * 4 * sizeof(float)));
    HIP_TRY(hipMemset(r->d_grad[i], 0, n * 3 * sizeof(float)));
  }
  HIP_TRY(hipMalloc((void**)&r->d_accum, n * 4 * sizeof(float)));
  HIP_TRY(hipMemset(r->d_accum, 0, n * 4 * sizeof(float)));
  // workgroups of the march: 8x8 pixels (4 waves x 16 rays) in dense mode, 64 list entries in sparse mode
  const size_t nblk = std::max<size_t>((size_t)((w + 7) / 8) * (size_t)((h + 7) / 8), (n + 63) / 64) + 1;
  HIP_TRY(hipMalloc((void**)&r->d_block_counters, nblk * kBlockCounters * sizeof(unsigned int)));
  HIP_TRY(hipMalloc((void**)&r->pool.tile_first, nblk * 4 * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&r->pool.tile_count, nblk * 4 * sizeof(unsigned int)));
  HIP_TRY(hipMalloc((void**)&r->pool.pix_state, std::max<size_t>(n, 1) * sizeof(float4)));
  r->pool_tiles = nblk * 4;
  if (r->d_trace) { HIP_TRY(hipFree(r->d_trace)); r->d_trace = nullptr; }
  if (const char* tr = getenv("OVR_HIP_TRACE")) {
    if (tr[0] == '1') {
      r->trace_words = nblk * 4 * 4;
      HIP_TRY(hipMalloc((void**)&r->d_trace, r->trace_words * sizeof(unsigned long long)));
    }
  }
  r->fb_pixels = n;
  for (int i = 0; i < 2; ++i)
    for (int k = 0; k < 4; ++k) r->d_rect[i][k] = r->h_rgba_rect[i][k] = r->h_grad_rect[i][k] = 0;
  return 0;
}

// Shade order by light beams (PoolDesc): beams of about OVR_HIP_SHADE_BEAM voxels (default 32) across - the bricks one beam's shadow rays sweep have
// to stay in one XCD's 4 MB L2 while its runs are shaded (measured on C3: 16 / 32 / 64 voxels -> shade 0.958 / 0.937 / 0.998 ms; creation order 1.094);
// OVR_HIP_SHADE_ORDER=0: creation order (rounds 1-4).  Called once RayMarchParams holds the frame's light and volume transform.
void shade_order_params(ovr_hip_renderer* r, PoolDesc& pd)
{
  const float beam = r->shade_beam;
  pd.order_grid = 0;
  if (!r->shade_order_on || !pd.order || !pd.order_key || !pd.order_ws) { pd.order = nullptr; return; }
  const RayMarchParams& P = r->P;
  const double diag = std::sqrt((double)r->vd.nx * r->vd.nx + (double)r->vd.ny * r->vd.ny + (double)r->vd.nz * r->vd.nz);
  const int G = diag <= 32.0 * beam ? 32 : diag <= 64.0 * beam ? 64 : 128;
  // two unit vectors perpendicular to the light: U = L x e_k (k = L's smallest component), V = L x U
  const double L[3] = { P.light.x, P.light.y, P.light.z };
  const int k = std::fabs(L[0]) <= std::fabs(L[1]) && std::fabs(L[0]) <= std::fabs(L[2]) ? 0 : std::fabs(L[1]) <= std::fabs(L[2]) ? 1 : 2;
  double U[3] = { 0, 0, 0 }, V[3];
  U[(k + 1) % 3] = L[(k + 2) % 3]; U[(k + 2) % 3] = -L[(k + 1) % 3];
  const double un = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
  if (!(un > 0.0)) { pd.order = nullptr; return; }
  for (double& u : U) u /= un;
  V[0] = L[1] * U[2] - L[2] * U[1]; V[1] = L[2] * U[0] - L[0] * U[2]; V[2] = L[0] * U[1] - L[1] * U[0];
  // the volume's box in world space (object = world * inv_scale + wto_p): centre and half diagonal
  const double sx = 1.0 / P.inv_scale.x, sy = 1.0 / P.inv_scale.y, sz = 1.0 / P.inv_scale.z;
  const double c[3] = { (0.5 - P.wto_p.x) * sx, (0.5 - P.wto_p.y) * sy, (0.5 - P.wto_p.z) * sz };
  const double R = 0.5 * std::sqrt(sx * sx + sy * sy + sz * sz), s = (double)G / (2.0 * R);
  if (!std::isfinite(s) || !(s > 0.0)) { pd.order = nullptr; return; }
  for (int i = 0; i < 3; ++i) { pd.order_u[i] = (float)(U[i] * s); pd.order_v[i] = (float)(V[i] * s); }
  pd.order_u[3] = (float)((R - (c[0] * U[0] + c[1] * U[1] + c[2] * U[2])) * s);
  pd.order_v[3] = (float)((R - (c[0] * V[0] + c[1] * V[1] + c[2] * V[2])) * s);
  pd.order_grid = G;
}

int ensure_pool(ovr_hip_renderer* r, size_t chunks)
{
  // kPoolSubs sub-pools of equal size, each a multiple of 16 chunks (the largest reservation)
  const size_t sub = ((chunks + kPoolSubs - 1) / kPoolSubs + 15) / 16 * 16;
  chunks = sub * kPoolSubs;
  if (r->pool.reqs && r->pool.capacity >= chunks) return 0;
  HIP_TRY(hipDeviceSynchronize());
  if (r->pool.reqs) HIP_TRY(hipFree(r->pool.reqs));
  if (r->pool.chunk_next) HIP_TRY(hipFree(r->pool.chunk_next));
  if (r->pool.chunk_n) HIP_TRY(hipFree(r->pool.chunk_n));
  if (r->pool.order) HIP_TRY(hipFree(r->pool.order));
  if (r->pool.order_key) HIP_TRY(hipFree(r->pool.order_key));
  r->pool.reqs = nullptr; r->pool.chunk_next = nullptr; r->pool.chunk_n = nullptr; r->pool.order = nullptr; r->pool.order_key = nullptr; r->pool.capacity = 0; r->pool.sub_capacity = 0;
  HIP_TRY(hipMalloc((void**)&r->pool.reqs, chunks * 64 * 32));
  HIP_TRY(hipMalloc((void**)&r->pool.chunk_next, chunks * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&r->pool.chunk_n, chunks * sizeof(unsigned int)));
  HIP_TRY(hipMalloc((void**)&r->pool.order, (chunks / 4 + 1) * sizeof(unsigned int))); // one entry per run of 4 chunks (shade order by light beams)
  HIP_TRY(hipMalloc((void**)&r->pool.order_key, (chunks / 4 + 1) * sizeof(unsigned int)));
  HIP_TRY(hipMemset(r->pool.order_key, 0xff, (chunks / 4 + 1) * sizeof(unsigned int)));
  if (!r->pool.order_ws) {
    HIP_TRY(hipMalloc((void**)&r->pool.order_ws, (size_t)kOrderWsWords * sizeof(unsigned int)));
    HIP_TRY(hipMemset(r->pool.order_ws, 0, (size_t)kOrderWsWords * sizeof(unsigned int))); // the shade kernel leaves the histogram zeroed for the next generation
  }
  r->pool.capacity = (unsigned int)chunks;
  r->pool.sub_capacity = (unsigned int)sub;
  return 0;
}

// device_impl.cpp:125-144
void update_camera(ovr_hip_renderer* r)
{
  const CameraP& c = r->camera.current;
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  const float t = 2.f * std::tan(c.fovy * 0.5f * (float)M_PI / 180.f);
  const float aspect = (float)W / (float)H;
  const V3 from = { c.from[0], c.from[1], c.from[2] }, at = { c.at[0], c.at[1], c.at[2] }, up = { c.up[0], c.up[1], c.up[2] };
  const V3 dir = normalize(sub(at, from));
  const V3 hor = scale(t * aspect, normalize(cross(dir, up)));
  const V3 cr = cross(hor, dir);
  const V3 ver = { cr.x / aspect, cr.y / aspect, cr.z / aspect };
  RayMarchParams& P = r->P;
  P.cam_pos = { from.x, from.y, from.z };
  P.cam_dir = { dir.x, dir.y, dir.z };
  P.cam_hor = { hor.x, hor.y, hor.z };
  P.cam_ver = { ver.x, ver.y, ver.z };
  // inverse-transpose of get_xfm_world_to_camera().l (shaders_common.h:276-289; LinearSpace.h:215-224,321)
  const V3 x = normalize(hor), y = normalize(ver), zz = normalize(dir);
  const V3 z = { -zz.x, -zz.y, -zz.z };
  const V3 vx = { x.x, y.x, z.x }, vy = { x.y, y.y, z.y }, vz = { x.z, y.z, z.z };
  const float det = dot(vx, cross(vy, vz));
  const V3 c0 = cross(vy, vz), c1 = cross(vz, vx), c2 = cross(vx, vy);
  const float m[9] = { c0.x / det, c0.y / det, c0.z / det, c1.x / det, c1.y / det, c1.z / det, c2.x / det, c2.y / det, c2.z / det };
  std::memcpy(P.wtc_it, m, sizeof(m));
}

// volume.cpp:131-145 + device_impl.cpp:288-296 + volume.cpp:172-179
void update_volume_params(ovr_hip_renderer* r)
{
  RayMarchParams& P = r->P;
  const bool vertex = r->grid_convention.current == OVR_HIP_GRID_VERTEX_CENTRED;
  const int n[3] = { r->vd.nx, r->vd.ny, r->vd.nz };
  float inv[3], wp[3], cs[3], cb[3], gs[3];
  for (int k = 0; k < 3; ++k) {
    const float ext = vertex ? (float)(n[k] - 1) : (float)n[k];
    const float sc = r->spacing[k] * ext;
    inv[k] = 1.f / sc;
    wp[k] = -(inv[k] * r->origin[k]);
    cs[k] = vertex ? (float)(n[k] - 1) : (float)n[k];
    cb[k] = vertex ? 0.f : -0.5f;
    gs[k] = vertex ? 1.f / (float)(n[k] - 1) : 1.f / (float)n[k];
  }
  P.inv_scale = { inv[0], inv[1], inv[2] };
  P.wto_p = { wp[0], wp[1], wp[2] };
  P.otw_it = P.inv_scale;
  P.coord_scale = { cs[0], cs[1], cs[2] };
  P.coord_bias = { cb[0], cb[1], cb[2] };
  P.grad_step = { gs[0], gs[1], gs[2] };
  const V3 L = normalize({ -907.108f, 2205.875f, -400.0267f }); // params.h:79
  P.light = { L.x, L.y, L.z };
  P.vol = r->vd;
  P.vol.data = r->d_volume;
}

// StructuredRegularVolume::set_value_range (volume.cpp:131-145): a valid range replaces the one in effect, an invalid one
// (hi < lo, the default (1, -1)) keeps it - after a volume load that is the data range (array.cpp:297, volume.cpp:187-191)
void update_tfn_range(ovr_hip_renderer* r)
{
  RayMarchParams& P = r->P;
  const int dt = device_value_type(r->value_type);
  const TfnP& t = r->tfn.current;
  if (t.hi >= t.lo) { // volume.cpp:135
    P.tf_upper = integer_normalize(t.hi, dt);
    P.tf_lower = integer_normalize(t.lo, dt);
  }
  // a degenerate range (upper == lower: a constant volume under the data-range fallback) gives the reference scale = inf and the
  // coordinate clamp01((lower - lower) * inf) = clamp01(NaN) = 0 for every sample; scale = 0 yields the same 0 without the NaN, which
  // the shadow march's transfer-function lookup (tf_alpha2: no clamp) relies on
  P.tf_scale = P.tf_upper == P.tf_lower ? 0.f : 1.f / (P.tf_upper - P.tf_lower);
}

int upload_tfn(ovr_hip_renderer* r)
{
  const TfnP& t = r->tfn.current;
  const int nc = (int)(t.colors.size() / 3), na = (int)(t.alphas.size() / 2);
  if (nc == 0 || na == 0) return 0; // volume.cpp:125: nothing happens for an empty TF
  // One pinned staging buffer, one device buffer (colours, then alphas), ONE stream-ordered copy on the render stream: the frames that read the
  // tables are enqueued behind it, the frame before it has been resolved by the commit.  (Until round 4: hipDeviceSynchronize and two
  // synchronous copies from pageable vectors - 60 of the 105 us a transfer-function edit cost beside its frame, `tools/tf_edit_time.py`.)
  const size_t words = (size_t)nc * 4 + (size_t)na;
  if (r->n_color != nc || r->n_alpha != na) {
    HIP_TRY(hipDeviceSynchronize()); // nothing may still read (or be copying into) the old tables
    if (r->d_tf_color) HIP_TRY(hipFree(r->d_tf_color));
    if (r->h_tf) HIP_TRY(hipHostFree(r->h_tf));
    r->d_tf_color = r->d_tf_alpha = nullptr; r->h_tf = nullptr;
    r->n_color = r->n_alpha = 0;
    HIP_TRY(hipMalloc((void**)&r->d_tf_color, words * sizeof(float)));
    HIP_TRY(hipHostMalloc((void**)&r->h_tf, words * sizeof(float), hipHostMallocDefault));
    r->d_tf_alpha = r->d_tf_color + (size_t)nc * 4;
    r->n_color = nc;
    r->n_alpha = na;
  }
  // (a copy of an earlier commit that has not run yet would read this buffer while it is rewritten - and be overwritten by this commit's copy, which
  // is enqueued behind it, before any frame reads the tables)
  float* c4 = r->h_tf;
  float* a = r->h_tf + (size_t)nc * 4;
  for (int i = 0; i < nc; ++i) { // volume.cpp:112-118
    c4[4 * i + 0] = t.colors[3 * i + 0];
    c4[4 * i + 1] = t.colors[3 * i + 1];
    c4[4 * i + 2] = t.colors[3 * i + 2];
    c4[4 * i + 3] = 1.f;
  }
  for (int i = 0; i < na; ++i) a[i] = t.alphas[2 * i + 1]; // volume.cpp:120-123
  HIP_TRY(hipMemcpyAsync(r->d_tf_color, r->h_tf, words * sizeof(float), hipMemcpyHostToDevice, r->stream()));
  // the next frame may run on the other framebuffer set's stream (a swap in between: renderapp's order): it waits for this event
  if (!r->ev_tf) HIP_TRY(hipEventCreateWithFlags(&r->ev_tf, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(r->ev_tf, r->stream()));
  r->tf_copy_pending = true;
  r->have_tfn = true;
  return 0;
}

int ensure_sparse_buffers(ovr_hip_renderer* r)
{
  const size_t n = r->fb_pixels;
  if (r->sparse_pixels == n && r->d_sparse_xy) return 0;
  if (r->d_sparse_xy) HIP_TRY(hipFree(r->d_sparse_xy));
  if (r->d_block_counts) HIP_TRY(hipFree(r->d_block_counts));
  HIP_TRY(hipMalloc((void**)&r->d_sparse_xy, n * 2 * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&r->d_block_counts, sparse_mask_workspace_elems(r->fbsize.current.w, r->fbsize.current.h) * sizeof(unsigned int)));
  r->sparse_pixels = n;
  return 0;
}

SparseMaskParams make_mask_params(ovr_hip_renderer* r, int frame_index, int32_t* out_xy)
{
  SparseMaskParams m{};
  const FocusP& f = r->focus.current;
  m.noise = r->d_noise;
  m.noise_xy = r->noise_xy;
  m.width = r->fbsize.current.w;
  m.height = r->fbsize.current.h;
  m.frame_index = frame_index;
  m.mean_x = f.cx;
  m.mean_y = f.cy;
  m.sigma_rcp2 = 1.f / (f.scale * f.scale); // generate_mask.cu:90
  m.base_noise = f.base_noise;
  m.out_xy = out_xy;
  m.block_counts = r->d_block_counts;
  m.count = r->d_sparse_count;
  return m;
}

int launch_frame(ovr_hip_renderer* r);
int finish_frame(ovr_hip_renderer* r); // resolves the frame in flight: one renderer, or a device group incl. its gather
int finish_frame_one(ovr_hip_renderer* r);

// Screen-space bounding rectangle (pixels, aligned outwards to 8, two pixels of margin for the sub-pixel jitter) of the volume's box
// under the committed camera: a ray outside it misses the box and its pixel is exactly 0 in every layer (background 0, alpha 0,
// shaders_raymarching.cu:260-321) - in any mode: accumulation adds 0, sparse sampling and image shards leave it cleared.  The whole
// frame when a corner of the box lies at or behind the camera plane.
void nonzero_rect(const ovr_hip_renderer* r, int rect[4])
{
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  rect[0] = 0; rect[1] = 0; rect[2] = W; rect[3] = H;
  if (!r->have_volume || W <= 0 || H <= 0) return;
  const RayMarchParams& P = r->P;
  const double d[3] = { P.cam_dir.x, P.cam_dir.y, P.cam_dir.z }, h[3] = { P.cam_hor.x, P.cam_hor.y, P.cam_hor.z }, v[3] = { P.cam_ver.x, P.cam_ver.y, P.cam_ver.z };
  const double hh = h[0] * h[0] + h[1] * h[1] + h[2] * h[2], vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (!(hh > 0.0) || !(vv > 0.0)) return;
  const double inv[3] = { P.inv_scale.x, P.inv_scale.y, P.inv_scale.z }, wp[3] = { P.wto_p.x, P.wto_p.y, P.wto_p.z }, from[3] = { P.cam_pos.x, P.cam_pos.y, P.cam_pos.z };
  double sx0 = 1e30, sx1 = -1e30, sy0 = 1e30, sy1 = -1e30;
  for (int c = 0; c < 8; ++c) {
    double p[3];
    for (int k = 0; k < 3; ++k) p[k] = (((c >> k) & 1 ? 1.0 : 0.0) - wp[k]) / inv[k] - from[k]; // object corner -> world, relative to the eye
    const double a = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
    if (!(a > 1e-6 * (std::fabs(p[0]) + std::fabs(p[1]) + std::fabs(p[2]) + 1e-30))) return; // at or behind the camera plane: whole frame
    const double sx = 0.5 + (p[0] * h[0] + p[1] * h[1] + p[2] * h[2]) / (hh * a), sy = 0.5 + (p[0] * v[0] + p[1] * v[1] + p[2] * v[2]) / (vv * a);
    sx0 = std::min(sx0, sx); sx1 = std::max(sx1, sx); sy0 = std::min(sy0, sy); sy1 = std::max(sy1, sy);
  }
  auto lo = [](double s, int n) { const double q = std::floor(s * n) - 2.0; return (int)std::min<double>(std::max<double>(q, 0.0), n) & ~7; };
  auto hi = [](double s, int n) { const double q = std::ceil(s * n) + 2.0; return std::min(((int)std::min<double>(std::max<double>(q, 0.0), n) + 7) & ~7, n); };
  rect[0] = lo(sx0, W); rect[2] = hi(sx1, W); rect[1] = lo(sy0, H); rect[3] = hi(sy1, H);
  if (rect[2] <= rect[0] || rect[3] <= rect[1]) { rect[0] = rect[1] = rect[2] = rect[3] = 0; } // the box is off screen: nothing to refresh
}

// device layer -> pinned host mirror, only where a pixel can differ from 0 now or could when the mirror was last refreshed
int refresh_mirror(ovr_hip_renderer* r, float* host, const float* dev, int channels, const int dev_rect[4], int last[4], hipStream_t st)
{
  const int W = r->fbsize.current.w;
  // what the DEVICE set holds is the frame of the camera it was last rendered with - renderapp commits the next camera before it maps
  // the previous frame (main_app.cpp:244-263) - so the rectangle is the one recorded at render time, not the committed camera's
  int now[4] = { dev_rect[0], dev_rect[1], dev_rect[2], dev_rect[3] };
  static const bool whole = getenv("OVR_HIP_MAP_WHOLE_FRAME") != nullptr; // measurements: the uncropped copy
  if (whole) { now[0] = 0; now[1] = 0; now[2] = W; now[3] = r->fbsize.current.h; }
  int u[4] = { now[0], now[1], now[2], now[3] };
  if (last[2] > last[0] && last[3] > last[1]) {
    if (u[2] <= u[0] || u[3] <= u[1]) { u[0] = last[0]; u[1] = last[1]; u[2] = last[2]; u[3] = last[3]; }
    else { u[0] = std::min(u[0], last[0]); u[1] = std::min(u[1], last[1]); u[2] = std::max(u[2], last[2]); u[3] = std::max(u[3], last[3]); }
  }
  if (u[2] > u[0] && u[3] > u[1]) {
    const size_t pitch = (size_t)W * channels * sizeof(float), off = ((size_t)u[1] * W + u[0]) * channels;
    HIP_TRY(hipMemcpy2DAsync(host + off, pitch, dev + off, pitch, (size_t)(u[2] - u[0]) * channels * sizeof(float), (size_t)(u[3] - u[1]), hipMemcpyDeviceToHost, st));
  }
  for (int k = 0; k < 4; ++k) last[k] = now[k];
  return 0;
}

// the macrocell value ranges (sp.compute_value_range at load, volume.cpp:234-237): once per volume
int update_macrocell_ranges(ovr_hip_renderer* r, hipStream_t st)
{
  const size_t cells = (size_t)((r->vd.nx + 15) / 16) * ((r->vd.ny + 15) / 16) * ((r->vd.nz + 15) / 16);
  if (cells != r->mc_cells || !r->d_mc_minmax) {
    HIP_TRY(hipDeviceSynchronize());
    if (r->d_mc_minmax) HIP_TRY(hipFree(r->d_mc_minmax));
    if (r->d_mc_majorant) HIP_TRY(hipFree(r->d_mc_majorant));
    if (r->d_mc_occupancy) HIP_TRY(hipFree(r->d_mc_occupancy));
    if (r->d_mc_fine) HIP_TRY(hipFree(r->d_mc_fine));
    r->d_mc_fine = nullptr;
    r->d_mc_minmax = r->d_mc_majorant = nullptr;
    r->d_mc_occupancy = nullptr;
    r->mc_cells = 0;
    HIP_TRY(hipMalloc((void**)&r->d_mc_minmax, cells * 2 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&r->d_mc_majorant, cells * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&r->d_mc_occupancy, cells));
    HIP_TRY(hipMalloc((void**)&r->d_mc_fine, cells));
    r->mc_cells = cells;
    r->mc_ranges_valid = r->mc_majorant_valid = false;
  }
  if (!r->mc_ranges_valid) {
    VolumeDesc vd = r->vd;
    vd.data = r->d_volume;
    HIP_TRY(launch_macrocell_ranges(vd, r->d_mc_minmax, st));
    r->mc_ranges_valid = true;
    r->mc_majorant_valid = false;
  }
  return 0;
}

// (re)build the majorant grids when empty-space skipping is on: per TF / range change
int update_macrocells(ovr_hip_renderer* r, hipStream_t st)
{
  if (int e = update_macrocell_ranges(r, st)) return e;
  const size_t cells = r->mc_cells;
  if (!r->mc_majorant_valid) {
    // tfn.value_range / range_rcp_norm of the reference (volume.cpp:147-153) - the TF range, normalised like the data
    HIP_TRY(launch_macrocell_majorants(r->d_mc_minmax, (unsigned int)cells, r->d_tf_alpha, r->n_alpha, r->P.tf_lower, r->P.tf_upper, r->d_mc_majorant, st));
    HIP_TRY(launch_macrocell_coarse(r->d_mc_majorant, r->vd.nx, r->vd.ny, r->vd.nz, r->d_mc_occupancy, r->d_mc_fine, st));
    r->mc_majorant_valid = true;
  }
  return 0;
}

// the 8x8-pixel blocks of the image this renderer draws (all of them, or those holding a pixel of one of its tiles),
// in image order; the device sorts them by ray length (launch_schedule) before the next frame
int build_schedule_list(ovr_hip_renderer* r)
{
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  const ShardP& s = r->shard.current;
  const int bw = (W + 7) / 8, bh = (H + 7) / 8;
  if (bw > 0xffff || bh > 0xffff) return fail(OVR_HIP_EINVAL, "[hip] framebuffer larger than 524280 pixels on a side");
  std::vector<unsigned int> list;
  list.reserve((size_t)bw * bh / (size_t)std::max(1, s.world) + 64);
  // listed supertile by supertile (4x4 blocks = 32x32 pixels, row-major inside): 16 consecutive entries are a compact square,
  // which the schedule kernel keeps together on one XCD
  const int sw = (bw + 3) / 4, sh = (bh + 3) / 4;
  // order of the supertiles: row-major (0, default) or along a Z curve (OVR_HIP_SCHED_ORDER=morton, round 5 experiment for C4: the blocks of one
  // ray-length class form a ring around the volume's silhouette - in row-major order consecutive entries of a class alternate between the ring's left and
  // right side, on the Z curve they are 2-D patches: do workgroups that run at the same time then share DRAM pages?)
  static const bool morton = getenv("OVR_HIP_SCHED_ORDER") && std::string(getenv("OVR_HIP_SCHED_ORDER")) == "morton";
  std::vector<std::pair<unsigned int, unsigned int>> order; // (key, sx | sy << 16)
  order.reserve((size_t)sw * sh);
  auto spread = [](unsigned int v) { v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; return (v | (v << 1)) & 0x55555555u; };
  for (int sy = 0; sy < sh; ++sy)
    for (int sx = 0; sx < sw; ++sx) order.push_back({ morton ? (spread((unsigned int)sx) | (spread((unsigned int)sy) << 1)) : (unsigned int)(sy * sw + sx), (unsigned int)sx | ((unsigned int)sy << 16) });
  if (morton) std::sort(order.begin(), order.end());
  for (const auto& o : order) {
      const int sx = (int)(o.second & 0xffffu), sy = (int)(o.second >> 16);
      for (int k = 0; k < 16; ++k) {
        const int bx = sx * 4 + (k & 3), by = sy * 4 + (k >> 2);
        if (bx >= bw || by >= bh) continue;
        bool mine = s.world <= 1;
        if (!mine) {
          const int tx0 = (bx * 8) / s.tw, tx1 = std::min(bx * 8 + 7, W - 1) / s.tw;
          const int ty0 = (by * 8) / s.th, ty1 = std::min(by * 8 + 7, H - 1) / s.th;
          for (int ty = ty0; ty <= ty1 && !mine; ++ty)
            for (int tx = tx0; tx <= tx1 && !mine; ++tx) mine = ((tx + ty) % s.world) == s.rank;
        }
        if (mine) list.push_back((unsigned int)bx | ((unsigned int)by << 16));
      }
  }
  if (r->d_sched_src) HIP_TRY(hipFree(r->d_sched_src));
  if (r->d_sched) HIP_TRY(hipFree(r->d_sched));
  r->d_sched_src = r->d_sched = nullptr;
  r->n_sched = (unsigned int)list.size();
  if (!list.empty()) {
    HIP_TRY(hipMalloc((void**)&r->d_sched_src, list.size() * sizeof(unsigned int)));
    HIP_TRY(hipMalloc((void**)&r->d_sched, (list.size() + schedule_workspace_elems((unsigned int)list.size())) * sizeof(unsigned int))); // + the sort's histograms
    HIP_TRY(hipMemcpy(r->d_sched_src, list.data(), list.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
  }
  r->sched_list_dirty = false;
  r->sched_dirty = true;
  return 0;
}

// ---- replicas built in the background (see replica_state)
void join_builders(ovr_hip_renderer* r)
{
  for (int k = 0; k < kLayouts; ++k)
    if (r->builder[k].joinable()) r->builder[k].join();
}

void drop_replica(ovr_hip_renderer* r, int k)
{
  if (k == LAYOUT_GENERAL) return;
  if (r->builder[k].joinable()) r->builder[k].join();
  if (r->d_replica[k]) { (void)hipFree(r->d_replica[k]); r->volume_bytes -= (size_t)r->vd_replica[k].bytes; }
  if (r->d_axis[k]) (void)hipFree(r->d_axis[k]);
  r->d_replica[k] = nullptr; r->d_axis[k] = nullptr;
  r->replica_state[k] = 0;
}

// allocate replica k and enqueue its construction from the general layout on build_stream (host work: on the calling thread); on an
// allocation failure the replica is given up (an optimisation: the general layout renders the same frames) and `err` says why.
// Expects replica_state[k] == 4 (claimed); leaves 2 (enqueued) or 0 (given up).
void build_replica_host(ovr_hip_renderer* r, int k, std::string* err)
{
  (void)hipSetDevice(r->device);
  VolumeDesc t = r->vd_replica[k];
  void *d_data = nullptr, *d_tab = nullptr;
  hipError_t e = hipMalloc(&d_data, (size_t)t.bytes + 64); // + slack: the pair load of the very last element
  if (e == hipSuccess) e = hipMalloc(&d_tab, axis_table_bytes(t));
  if (e == hipSuccess) {
    t.data = d_data;
    // padding voxels are never sampled, but must be finite: the build writes them itself (launch_relayout), only the slack is set here
    if (getenv("OVR_HIP_POISON_ALLOC") && atoi(getenv("OVR_HIP_POISON_ALLOC")) != 0) e = hipMemsetAsync(d_data, 0xff, (size_t)t.bytes, r->build_stream);
    if (e == hipSuccess) e = hipMemsetAsync((char*)d_data + (size_t)t.bytes, 0, 64, r->build_stream);
  }
  if (e == hipSuccess) e = launch_axis_tables(t, d_tab, r->build_stream);
  if (e == hipSuccess) {
    VolumeDesc g = r->vd_replica[LAYOUT_GENERAL];
    g.data = r->d_volume;
    e = launch_rebrick(g, d_data, t, r->build_stream);
  }
  if (e == hipSuccess) e = hipEventRecord(r->build_ev[k], r->build_stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(r->build_stream);
    if (err) *err = hipGetErrorString(e);
    if (d_data) (void)hipFree(d_data);
    if (d_tab) (void)hipFree(d_tab);
    r->replica_state[k].store(0, std::memory_order_release);
    return;
  }
  r->vd_replica[k] = t;
  r->d_replica[k] = d_data;
  r->d_axis[k] = d_tab;
  r->volume_bytes += (size_t)t.bytes;
  r->replica_state[k].store(2, std::memory_order_release);
}

// a planned replica is wanted: in the background (a builder thread does the host work; frames go on from the general layout) or at once
int start_replica_build(ovr_hip_renderer* r, int k, std::string* err, bool background = true)
{
  int planned = 1;
  if (!r->replica_state[k].compare_exchange_strong(planned, 4)) return 0; // not planned, or already on its way
  if (r->builder[k].joinable()) r->builder[k].join();
  static const bool threads = !(getenv("OVR_HIP_BUILD_THREAD") && atoi(getenv("OVR_HIP_BUILD_THREAD")) == 0); // measurements: 0 = host work on the caller
  bool started = false;
  if (background && threads) {
    try { r->builder[k] = std::thread(build_replica_host, r, k, nullptr); started = true; }
    catch (const std::exception&) { started = false; } // no thread to be had: the caller does the host work
  }
  if (!started) build_replica_host(r, k, err);
  return 0;
}

// the layout a frame on stream `st` reads when it asks for `choice`: the replica if it is resident; if it is still being built, the replica
// when the frame may wait for it (forced choice), else the general layout (same frame bit for bit)
int resolve_layout(ovr_hip_renderer* r, int choice, bool may_wait, hipStream_t st)
{
  if (choice <= LAYOUT_GENERAL || choice >= kLayouts || r->replica_state[choice].load(std::memory_order_acquire) == 0) return LAYOUT_GENERAL;
  if (r->replica_state[choice] == 1) (void)start_replica_build(r, choice, nullptr, !may_wait);
  if (r->replica_state[choice] == 4 && may_wait && r->builder[choice].joinable()) r->builder[choice].join(); // the host part of the build
  int state = r->replica_state[choice].load(std::memory_order_acquire);
  if (state == 2 && hipEventQuery(r->build_ev[choice]) == hipSuccess) { r->replica_state[choice] = 3; state = 3; }
  if (state == 3) return choice;
  if (state == 2 && may_wait && hipStreamWaitEvent(st, r->build_ev[choice], 0) == hipSuccess) return choice;
  (void)hipGetLastError();
  return LAYOUT_GENERAL;
}

// Impl::render up to and including the launch (device_impl.cpp:199-262); no host synchronisation
int enqueue_frame(ovr_hip_renderer* r)
{
  if (!r->have_volume) return fail(OVR_HIP_ESTATE, "[hip] render() called before a volume was set");
  if (!r->have_tfn) return fail(OVR_HIP_ESTATE, "[hip] render() called before a transfer function was set");
  if (raymarch_lds_bytes(r->n_color, r->n_alpha) == 0)
    return fail(OVR_HIP_EINVAL, "[hip] transfer function too large for LDS staging (colour + alpha tables must fit in 96 KiB)");
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  if (W <= 0 || H <= 0) return 0; // device_impl.cpp:216-217
  hipStream_t st = r->stream();
  if (r->use_user_stream) {
    // A frame is not a capturable unit: its first and last events carry timing, the frame after a camera / volume / size change waits for two
    // words from the device, and the host reads the frame's counters when it resolves it - a caller's stream under graph capture is refused, not corrupted
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_render_async: the caller's stream is capturing a graph - frames cannot be captured (they record timed events and hand "
                                  "counters back to the host); enqueue them on the stream outside the capture");
    (void)hipGetLastError();
  }
  if (r->tf_copy_pending) { HIP_TRY(hipStreamWaitEvent(st, r->ev_tf, 0)); r->tf_copy_pending = false; } // the tables' copy may sit on the other set's stream
  RayMarchParams& P = r->P;
  const bool accumulate = r->accumulate.current != 0;
  const bool sparse = r->sparse.current != 0;
  const size_t n = r->fb_pixels;
  if (accumulate) { // device_impl.cpp:226-233
    if (r->fb_reset) {
      for (int i = 0; i < 2; ++i) {
        // (the set this frame renders into needs no memset when the frame writes every pixel of it: dense sampling, the whole image - launched
        // blocks write their pixels, the others are cleared by launch_clear_blocks; 2 x 7 us per camera change of an accumulating session)
        if (i == r->cur && !sparse && r->shard.current.world <= 1) continue;
        // the other set is cleared on ITS stream (whatever touches it next - a frame after a swap, a mapframe - is enqueued there): the memsets
        // run beside this frame's schedule kernels instead of in front of them
        hipStream_t si = (r->use_user_stream || i == r->cur) ? st : r->own_stream[i];
        HIP_TRY(hipMemsetAsync(r->d_rgba[i], 0, n * 4 * sizeof(float), si));
        HIP_TRY(hipMemsetAsync(r->d_grad[i], 0, n * 3 * sizeof(float), si));
      }
      // the reference leaves the accumulation buffer as it is (device_impl.cpp:229-230 are commented out), which is
      // only sound because frame 1 overwrites it; sparse frames do not overwrite every pixel, so it is cleared here
      if (sparse) HIP_TRY(hipMemsetAsync(r->d_accum, 0, n * 4 * sizeof(float), st));
      r->fb_reset = false;
      r->frame_index = 0;
    }
  }
  else if (sparse) { // device_impl.cpp:234-239
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipMemsetAsync(r->d_rgba[i], 0, n * 4 * sizeof(float), st));
      HIP_TRY(hipMemsetAsync(r->d_grad[i], 0, n * 3 * sizeof(float), st));
    }
  }
  r->frame_index++;

  P.rgba = r->d_rgba[r->cur];
  nonzero_rect(r, r->d_rect[r->cur]); // what mapframe(HOST) will have to copy of this set (the committed camera is the one this frame renders)
  if (r->shard.current.world > 1 && r->members.size() <= 1) { // (a group leader's set is complete after every frame: its rectangle holds)
    // an image shard leaves the tiles of the other ranks as they are - whatever an earlier, unsharded frame put there,
    int* q = r->d_rect[r->cur];     // also outside the box's rectangle: such a set is mapped whole (found by tests/fuzz_states.py, seeds 61 / 62)
    q[0] = 0; q[1] = 0; q[2] = W; q[3] = H;
  }
  r->frame_set = r->cur;
  P.grad = r->d_grad[r->cur];
  P.accum = r->d_accum;
  P.width = W;
  P.height = H;
  P.frame_index = r->frame_index;
  P.accumulate = accumulate ? 1 : 0;
  P.spp = r->spp.current;
  P.shading = r->shading.current;
  P.step = 1.f / r->rate.current; // volume.cpp:176
  P.base = 1.f;                   // volume.h:125
  P.shadow_stride = (P.step * 10.f) * P.step; // shaders_raymarching.cu:221 then :64
  {
    const float ex = r->spacing[0] * r->vd.nx, ey = r->spacing[1] * r->vd.ny, ez = r->spacing[2] * r->vd.nz;
    P.long_ray_steps = std::sqrt(ex * ex + ey * ey + ez * ez) / P.step;
  }
  {
    // which replica of the volume this frame reads (ovr_hip_kernels.h "View-dependent replicas"): the central ray's direction in
    // object space; within ~21 degrees of an axis (measured crossover 18-23 degrees, profiles/r02_notes.md) a thin replica
    int choice = r->layout_choice.current;
    if (choice < 0) {
      const float dx = P.cam_dir.x * P.inv_scale.x, dy = P.cam_dir.y * P.inv_scale.y, dz = P.cam_dir.z * P.inv_scale.z;
      const float len = std::sqrt(dx * dx + dy * dy + dz * dz);
      const float ax = std::fabs(dx) / len, ay = std::fabs(dy) / len, az = std::fabs(dz) / len;
      // rays along x need the replica that is thin in y (pair axis y), rays along y the one thin in x; rays along z can use
      // either - the one that is 4 voxels wide in the direction the rays drift to is 4-7 % faster (profiles/r02_notes.md)
      choice = LAYOUT_GENERAL;
      const float kAxis = 0.93f; // ~21 degrees
      if (ax >= kAxis) choice = LAYOUT_THIN_T;
      else if (ay >= kAxis) choice = LAYOUT_THIN;
      else if (az >= kAxis) choice = ax >= ay ? LAYOUT_THIN_T : LAYOUT_THIN;
    }
    // second stage (see tune_state): the layout / pipeline under test, or the measured winner
    r->tune_frame = -1;
    const bool tune_l = r->layout_choice.current < 0, tune_p = r->pipeline.current == 0;
    bool may_wait = !tune_l; // a forced layout is what the frame reads, even if the replica has to be built first
    if (choice < 0 || choice >= kLayouts || r->replica_state[choice] == 0) choice = LAYOUT_GENERAL;
    const int rule_choice = choice;
    if (r->tune_on && (tune_l || tune_p)) {
      if (r->tune_state == 0) { // the rules' candidate
        r->tune_cand[0] = { tune_l ? choice : -1, 0, 0, 0.f };
        r->tune_n = 1; r->tune_cur = 0; r->tune_phase = 0;
        r->tune_layout = -1; r->tune_pipeline = 0;
        r->tune_frame = 0;
        r->tune_rule_choice = rule_choice;
      }
      else if (r->tune_state == 1) {
        const auto& c = r->tune_cand[r->tune_cur];
        // Nothing is probed while ANY replica is being built in the background: the frame runs the rules' choice and counts for nothing.
        // Waiting for the candidate's own replica on the device would stall an interactive frame for the length of the build (24 ms for
        // C3's 17 GB quad replica), and a candidate timed beside a build is timed wrong - the build has the memory system: the rules'
        // candidate measured 4.7 instead of 1.2 ms that way and lost to the quad replica it beats (profiles/r04_notes.md section 4)
        if (tune_l && c.layout > LAYOUT_GENERAL && r->replica_state[c.layout] == 1) (void)start_replica_build(r, c.layout, nullptr);
        bool ready = true;
        for (int k = 1; k < kLayouts; ++k) {
          int state = r->replica_state[k].load(std::memory_order_acquire);
          if (state == 2 && hipEventQuery(r->build_ev[k]) == hipSuccess) { r->replica_state[k] = 3; state = 3; }
          if (state == 2 || state == 4) ready = false;
        }
        (void)hipGetLastError();
        if (ready) {
          if (tune_l && c.layout >= 0) choice = c.layout; // (a forced layout is never overridden by a probe)
          r->tune_pipeline = c.pipeline;
          r->tune_frame = r->tune_cur;
        }
        else r->tune_pipeline = r->tune_cand[0].pipeline; // the pipeline the rules' frame ran (tune_cand[0] was filled in by its finish)
      }
      else if (r->tune_layout >= 0 && tune_l) {
        // a measured layout outlives camera moves only while the layout rule still says what it said when the measurement was made:
        // the thin replicas are view-dependent (an axis view's winner loses 30-60 % at an oblique angle), and general / quad were
        // measured against THAT rule's choice.  The measured pipeline stays (the regime decides it, not the view).
        if (rule_choice != r->tune_rule_choice) r->tune_layout = -1;
        else choice = r->tune_layout; // (probed, hence resident)
      }
    }
    choice = resolve_layout(r, choice, may_wait, st);
    const float vs = P.vol.value_scale, vm = P.vol.value_min_clamp;
    P.vol = r->vd_replica[choice];
    P.vol.value_scale = vs;
    P.vol.value_min_clamp = vm;
    r->stats.layout = choice;
  }
  P.tf_color = r->d_tf_color;
  P.tf_alpha = r->d_tf_alpha;
  P.n_color = r->n_color;
  P.n_alpha = r->n_alpha;
  P.rank = r->shard.current.rank;
  P.world = r->shard.current.world;
  P.tile_w = r->shard.current.tw;
  P.tile_h = r->shard.current.th;
  P.lds_staging = r->lds_staging.current;
  // Shade grid: with at most 64 runs of 4 chunks per workgroup in the previous frame (sparse transfer functions; C3: 36 k runs) 768
  // workgroups shade 4 % faster than 1024 - a smaller window of requests in flight, more of their bricks still in L2 - with many
  // runs (dense transfer functions, 4K frames) 1024 hide more latency (profiles/r02_ab/r02b_ab_shadeblocks.txt)
  P.lds_brick_offset = 0;
  P.jitter_mode = r->jitter.current;
  P.jitter_noise = r->d_noise;
  P.jitter_xy = r->noise_xy;
  if (P.jitter_mode == 1 && !r->d_noise)
    return fail(OVR_HIP_ESTATE, "[hip] blue-noise pixel jitter enabled but no noise tile was set (ovr_hip_set_noise_tile)");
  P.counters = r->d_counters;
  P.majorant = nullptr;
  P.occupancy = nullptr;
  P.occupancy_fine = nullptr;
  bool use_skip = r->skipping.current != 0;
  if (use_skip && r->skip_adaptive) {
    if (!r->mc_majorant_valid) { r->skip_active = true; r->skip_backoff = 32; } // new transfer function / volume: what is empty has changed
    else if (!r->skip_active && --r->skip_reprobe_in <= 0) r->skip_active = true; // probe the skipping kernels again with this frame
    use_skip = r->skip_active;
  }
  r->frame_used_skip = use_skip;
  if (use_skip) {
    if (int e = update_macrocells(r, st)) return e;
    P.majorant = r->d_mc_majorant;
    P.occupancy = r->d_mc_occupancy;
    P.occupancy_fine = r->d_mc_fine;
  }
  // (the shade grid, continued: not with empty-space skipping - that shade kernel is bound by instructions, not by memory, and wants
  // every wave: 0.67 vs 0.77 ms)
  P.shade_blocks = (!use_skip && r->stats.pool_chunks > 0 && r->stats.pool_chunks <= (size_t)64 * 1024 * 4) ? 768 : 1024;
  P.block_counters = r->d_block_counters;
  P.trace = r->d_trace;
  P.sparse_xy = nullptr;
  P.sparse_count = nullptr;
  P.sparse_hint_pixels = 0;
  P.schedule = nullptr;
  P.n_schedule = 0;
  P.n_blocks_owned = 0;
  r->frame_empty_pixels = 0;
  r->frame_empty_rays = 0;
  if (!sparse) {
    if (r->sched_list_dirty)
      if (int e = build_schedule_list(r)) return e;
    static const bool unsorted = getenv("OVR_HIP_SCHED_SORT") && atoi(getenv("OVR_HIP_SCHED_SORT")) == 0; // experiment: image (supertile) order
    static const bool skip_blocks = !(getenv("OVR_HIP_EMPTY_BLOCKS") && atoi(getenv("OVR_HIP_EMPTY_BLOCKS")) == 0); // 0: launch every block (measurements)
    // a pixel's ray is known to the schedule kernels when it has one sample and no jitter: blocks without a hit are found exactly (1); with
    // several samples or jitter its rays lie within half a pixel of the centre: blocks whose widened cone misses the box are found (2)
    const int exact = (!skip_blocks || unsorted) ? 0 : (P.spp == 1 && P.jitter_mode == 0) ? 1 : 2;
    if (r->sched_dirty || exact != r->sched_exact) {
      if (!r->d_sched_info) HIP_TRY(hipHostMalloc((void**)&r->d_sched_info, 2 * sizeof(unsigned int), hipHostMallocDefault)); // pinned: the scatter kernel writes it
      HIP_TRY(launch_schedule(P, r->d_sched_src, r->n_sched, r->d_sched, r->d_sched + r->n_sched, exact, r->d_sched_info, st));
      r->n_work = r->n_sched;
      r->empty_pixels = 0;
      if (exact && r->n_sched > 0) { // how many entries need a workgroup: two words from the device, once per camera / volume / size change
        // (in a device group every member waits on its own thread, side by side - round 5)
        HIP_TRY(hipStreamSynchronize(st));
        r->n_work = std::min(r->d_sched_info[0], r->n_sched);
        r->empty_pixels = r->d_sched_info[1];
      }
      r->sched_dirty = false;
      r->sched_exact = exact;
      r->clear_gen++;
    }
    P.schedule = unsorted ? r->d_sched_src : r->d_sched;
    P.n_schedule = r->n_work;
    P.n_blocks_owned = r->n_sched;
    r->frame_empty_pixels = r->empty_pixels;
    r->frame_empty_rays = (unsigned long long)r->empty_pixels * (unsigned long long)std::max(P.spp, 1); // every sample of a pixel is a ray
    if (r->n_work < r->n_sched) {
      const bool first_accumulated = accumulate && r->frame_index == 1;
      if (r->set_clear_gen[r->cur] != r->clear_gen || first_accumulated) {
        HIP_TRY(launch_clear_blocks(P, r->d_sched + r->n_work, r->n_sched - r->n_work, first_accumulated ? 1 : 0, st));
        r->set_clear_gen[r->cur] = r->clear_gen;
      }
    }
  }
  if (sparse) { // createSparseSamples, device_impl.cpp:304-342
    if (!r->d_noise) return fail(OVR_HIP_ESTATE, "[hip] sparse sampling enabled but no noise tile was set (ovr_hip_set_noise_tile)");
    if (int e = ensure_sparse_buffers(r)) return e;
    SparseMaskParams mp = make_mask_params(r, r->frame_index, r->d_sparse_xy);
    mp.tile_major = 1; // the frame's own list: 4x4-pixel blocks per wave where the mask is dense (the kept pixels are the same)
    HIP_TRY(launch_sparse_mask(mp, st));
    P.sparse_xy = r->d_sparse_xy;
    P.sparse_count = r->d_sparse_count;
    P.sparse_hint_pixels = r->sparse_prev_pixels;
  }
  // ---- shading pipeline: pooled (march -> shade -> composite) when it applies, else in place
  const int pipe = r->pipeline.current;
  bool want_pool = P.shading != 0 && pipe != 1 && !(pipe == 0 && r->auto_inplace);
  if (P.shading != 0 && pipe == 0 && r->tune_on && r->tune_state != 0 && r->tune_pipeline != 0) want_pool = r->tune_pipeline == 2; // measured (tune_state)
  P.pool = PoolDesc{};
  if (want_pool) {
    // first guess: room for 8 shaded samples per pixel; grown after an overflow (finish_frame)
    // (every tile reserves runs of 8 chunks, so add one run per tile)
    size_t guess = std::min<size_t>((std::max<size_t>(n * 8 / 64, 4096) + r->pool_tiles * 4) * 5 / 4, (size_t)1 << 22); // x 1.25: sub-pools fill unevenly
    if (const char* pc = getenv("OVR_HIP_POOL_CHUNKS")) guess = std::max<size_t>(8, (size_t)atoll(pc)); // diagnostic: force the overflow path
    if (int e = ensure_pool(r, std::max<size_t>(guess, r->pool.capacity))) return e;
    if (!r->pool.ctrl) {
      r->pool.ctrl = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(r->d_counters) + 128); // every sub-pool counter on a 128-byte line of its own
      HIP_TRY(hipMalloc((void**)&r->pool.shade_counters, pool_shade_blocks() * 2 * sizeof(unsigned int)));
    }
    P.pool = r->pool;
    shade_order_params(r, P.pool);
    if (P.spp > 1 && !r->d_spp_rgba) {
      HIP_TRY(hipMalloc((void**)&r->d_spp_rgba, std::max<size_t>(n, 1) * 4 * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&r->d_spp_grad, std::max<size_t>(n, 1) * 3 * sizeof(float)));
    }
  }
  P.spp_index = 0;
  P.spp_sum_rgba = r->d_spp_rgba;
  P.spp_sum_grad = r->d_spp_grad;
  r->stats.pipeline = want_pool ? 2 : 1;
  return launch_frame(r);
}

int launch_frame(ovr_hip_renderer* r)
{
  hipStream_t st = r->stream();
  // the frame's last reduction kernel writes the counters and the pool's control words into h_counters / h_ctrl itself and zeroes them on the
  // device for the next frame (RayMarchParams::publish): a frame is its kernels, nothing in front of them and nothing behind
  r->P.publish = r->h_counters;
  r->P.reduce_done = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(r->d_counters) + 128 + (size_t)kPoolCtrlWords * sizeof(unsigned int));
  r->P.zero_first = r->frame_words_dirty ? 1 : 0;
  if (r->frame_words_dirty) HIP_TRY(hipMemsetAsync(r->P.reduce_done, 0, sizeof(unsigned int), st));
  if (r->frame_words_dirty && r->pool.order_ws) HIP_TRY(hipMemsetAsync(r->pool.order_ws, 0, (size_t)kOrderWsWords * sizeof(unsigned int), st)); // (a march without its shade kernel)
  r->frame_words_dirty = true; // until the launch below has been enqueued completely
  // the events between the frame's kernels (per-phase times) cost ~16 us a frame - hipEventRecord is not free on either side of the queue; the
  // first and the last one (kernel_ms: what the layout / pipeline tuner compares) stay
  const bool phases = r->phase_timing.load();
  hipEvent_t evs[4] = { r->ev[0], phases ? r->ev[1] : nullptr, phases ? r->ev[2] : nullptr, r->ev[3] };
  r->frame_phase_timed = phases;
  HIP_TRY(launch_raymarch(r->P, st, evs));
  r->frame_words_dirty = false;
  r->async_pending = true;
  return 0;
}

int finish_frame_one(ovr_hip_renderer* r)
{
  if (!r->async_pending) return 0;
  HIP_TRY(hipStreamSynchronize(r->stream()));
  r->stats.stale_tiles = 0;
  if (r->P.pool.reqs) {
    // pool overflow: the march asked for more chunks than the pool holds; nothing was written to the framebuffer.
    // Grow the pool to what the frame needs (+25 %) and render the same frame again.
    // the most chunks any sub-pool was asked for in any generation of the frame
    auto asked = [&]() { return (size_t)r->h_ctrl[32 * (kPoolSubs + 1)]; };
    if (asked() > r->pool.sub_capacity && r->packed_early) r->stats.stale_tiles = 1; // packed before this re-render: the caller must not use them
    for (int attempt = 0; attempt < 4 && asked() > r->pool.sub_capacity; ++attempt) {
      const size_t need = (asked() + asked() / 4 + 16) * kPoolSubs;
      if (int e = ensure_pool(r, need)) return e;
      r->P.pool = r->pool;
      shade_order_params(r, r->P.pool);
      if (int e = launch_frame(r)) return e;
      HIP_TRY(hipStreamSynchronize(r->stream()));
    }
    if (asked() > r->pool.sub_capacity) return fail(OVR_HIP_EDEVICE, "[hip] request pool overflow persists after re-sizing");
    size_t used = 0; // chunks reserved by the last generation
    for (int k = 0; k < kPoolSubs; ++k) used += r->h_ctrl[32 * (k + 1)];
    r->stats.pool_chunks = used;
    r->pool_roomy = asked() * 2 <= (size_t)r->pool.sub_capacity;
  }
  else {
    // no pool: this frame is never rendered twice - but it proves nothing about the pool: only a pooled frame of the same
    // configuration does (the automatic pipeline may flip the NEXT frame back to pooled without a commit in between, and
    // ovr_hip_pack_tiles packs right behind a frame only when pool_roomy says it cannot overflow)
    r->stats.pool_chunks = 0;
    r->pool_roomy = false;
  }
  float ms = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, r->ev[0], r->ev[3]));
  if (r->frame_phase_timed) {
    HIP_TRY(hipEventElapsedTime(&m1, r->ev[0], r->ev[1]));
    HIP_TRY(hipEventElapsedTime(&m2, r->ev[1], r->ev[2]));
    HIP_TRY(hipEventElapsedTime(&m3, r->ev[2], r->ev[3]));
  }
  r->stats.kernel_ms = ms;
  r->stats.march_ms = m1;
  r->stats.shade_ms = m2;
  r->stats.composite_ms = m3;
  r->stats.rays = r->h_counters[0] + r->frame_empty_rays;   // (blocks without a hit are not launched: their pixels' rays missed)
  r->stats.samples = r->h_counters[1];
  r->stats.shaded_samples = r->h_counters[2];
  r->stats.shadow_samples = r->h_counters[3];
  r->stats.active_pixels = r->h_counters[4] + r->frame_empty_pixels;
  r->stats.skipped_samples = r->h_counters[5];
  r->stats.skipped_shadow_samples = r->h_counters[6];
  // The reference's box test switches a slab off for a ray whose direction component on that axis is below FLT_MIN (shaders_common.h:162-172,
  // restated): such a ray hits the box wherever its origin lies along that axis - from OUTSIDE the box's silhouette if the origin is
  // outside the slab (an axis-aligned camera's centre row / column; any camera now and then, by one pixel on the line where a component
  // changes sign).  The march counts those hits; a frame that has one - or accumulates onto one - is mapped whole, not by its rectangle.
  if (r->accumulate.current == 0 || r->frame_index <= 1) r->outside_hits = false;
  if (r->h_counters[7] > 0) r->outside_hits = true;
  if (r->outside_hits) { int* q = r->d_rect[r->frame_set]; q[0] = 0; q[1] = 0; q[2] = r->fbsize.current.w; q[3] = r->fbsize.current.h; }
  r->stats.lds_fallback_taps = r->stats.lds_unstaged_rounds = r->stats.lds_rounds = 0;
  if (r->P.lds_staging && !r->P.majorant && r->P.shading == 0 && !r->P.sparse_xy && r->P.vol.type == VOX_F32) {
    // the unshaded f32 march ran its LDS-staged variant: the two skip counters carried its diagnostics
    r->stats.lds_fallback_taps = r->h_counters[5];
    r->stats.lds_unstaged_rounds = r->h_counters[6];
    r->stats.lds_rounds = r->h_counters[3];
    r->stats.skipped_samples = r->stats.skipped_shadow_samples = r->stats.shadow_samples = 0;
  }
  r->stats.frame_index = r->frame_index;
  r->sparse_prev_pixels = r->P.sparse_xy ? r->stats.active_pixels : 0;
  r->stats.skipping_kernels = r->frame_used_skip ? 1 : 0;
  r->stats.replicas_building = 0;
  for (int k = 1; k < kLayouts; ++k) {
    int state = r->replica_state[k].load(std::memory_order_acquire);
    if (state == 2 && hipEventQuery(r->build_ev[k]) == hipSuccess) { r->replica_state[k] = 3; state = 3; }
    if (state == 2 || state == 4) r->stats.replicas_building++;
  }
  (void)hipGetLastError(); // hipErrorNotReady of the query is not an error
  r->stats.tuning = r->tune_frame >= 0 ? 1 : (r->tune_on && r->tune_state == 2 && (r->tune_layout >= 0 || r->tune_pipeline != 0)) ? 2 : 0;
  if (r->tune_frame == 0 && r->tune_state == 0) r->stats.tuning = 0; // the first frame of a configuration runs the rules' choice
  if (r->tune_frame >= 0 && r->tune_state < 2) { // measured choice of layout and pipeline (see tune_state)
    static const bool trace = getenv("OVR_HIP_TUNE_TRACE") != nullptr;
    auto& c = r->tune_cand[r->tune_frame];
    c.frames++;
    c.ms = ms;
    if (trace) fprintf(stderr, "[hip] tune: state %d phase %d candidate %d (layout %d pipeline %d) frame %d: layout %d pipeline %d kernel %.3f ms\n", r->tune_state, r->tune_phase,
                       r->tune_frame, c.layout, c.pipeline, c.frames, r->stats.layout, r->stats.pipeline, ms);
    const bool tune_l = r->layout_choice.current < 0, tune_p = r->pipeline.current == 0;
    auto decide = [&]() {
      int best = 0;
      for (int k = 1; k < r->tune_n; ++k)
        if (r->tune_cand[k].frames >= 2 && r->tune_cand[k].ms < r->tune_cand[best].ms) best = k;
      r->tune_layout = tune_l ? r->tune_cand[best].layout : -1;
      r->tune_pipeline = tune_p ? r->tune_cand[best].pipeline : 0;
      r->tune_state = 2;
      if (trace) fprintf(stderr, "[hip] tune: decided layout %d pipeline %d (%.3f ms)\n", r->tune_layout, r->tune_pipeline, r->tune_cand[best].ms);
    };
    // the layouts worth a try besides the rules' one: the general layout and the quad replica (a thin replica the camera did not ask
    // for loses 30-60 %: profiles/r02_notes.md section 2)
    auto add_layouts = [&](int pipeline) {
      const int base = r->tune_cand[0].layout;
      for (int l : { (int)LAYOUT_GENERAL, (int)LAYOUT_QUAD })
        if (tune_l && l != base && (l == LAYOUT_GENERAL || r->replica_state[l] != 0) && r->tune_n < 6) {
          r->tune_cand[r->tune_n++] = { l, pipeline, 0, 0.f };
          if (r->replica_state[l] == 1) (void)start_replica_build(r, l, nullptr); // built while the candidates before it are timed
        }
      r->tune_phase = 1;
    };
    if (r->tune_state == 0) {
      c.layout = tune_l ? r->stats.layout : -1;
      c.pipeline = r->stats.pipeline;
      // shade-heavy: the gradient (3 per shaded sample) and shadow taps outnumber the primary taps
      const double shade_taps = 3.0 * (double)r->stats.shaded_samples + (double)r->stats.shadow_samples;
      const bool heavy = r->P.shading != 0 && r->stats.samples > 0 && shade_taps >= 3.0 * (double)r->stats.samples;
      if (!heavy) { r->tune_state = 2; r->tune_layout = -1; r->tune_pipeline = 0; } // the rules stay in charge
      else {
        r->tune_state = 1;
        r->tune_cur = 0;          // the rules' candidate runs once more, timed
        r->tune_pipeline = c.pipeline;
        // the other pipeline: pooled is never far off, but shading in place is only worth a frame when most samples are shaded (with
        // few, a handful of waves shade alone for many milliseconds: DESIGN.md section 4)
        const bool try_other = tune_p && r->P.spp == 1 &&
                               (c.pipeline == 1 || (double)r->stats.shaded_samples >= 0.35 * ((double)r->stats.samples + (double)r->stats.skipped_samples));
        if (try_other) r->tune_cand[r->tune_n++] = { c.layout, c.pipeline == 2 ? 1 : 2, 0, 0.f };
        else add_layouts(c.pipeline);
        if (r->tune_n == 1) decide();
      }
    }
    else if (c.frames >= 2) {
      r->tune_cur++;
      if (r->tune_cur >= r->tune_n) {
        if (r->tune_phase == 0) {
          const int pbest = (r->tune_n > 1 && r->tune_cand[1].ms < r->tune_cand[0].ms) ? r->tune_cand[1].pipeline : r->tune_cand[0].pipeline;
          // the two pipelines within a quarter of each other on the rules' layout: the quad replica may order them the other way round (round 5: a 256 x 256 x 226
          // u16 volume, all samples shaded - general layout pooled 1.00 / in place 1.08 ms, quad replica pooled 0.89 / in place 0.77) - it gets both
          const bool close = r->tune_n > 1 && std::min(r->tune_cand[0].ms, r->tune_cand[1].ms) * 1.25f >= std::max(r->tune_cand[0].ms, r->tune_cand[1].ms);
          const int pother = pbest == 2 ? 1 : 2;
          add_layouts(pbest);
          if (close && tune_l && tune_p && r->tune_cand[0].layout != LAYOUT_QUAD && r->replica_state[LAYOUT_QUAD] != 0 && r->tune_n < 6) r->tune_cand[r->tune_n++] = { LAYOUT_QUAD, pother, 0, 0.f };
          if (r->tune_cur >= r->tune_n) decide();
        }
        else decide();
      }
    }
  }
  if (r->tune_on && r->tune_frame < 0 && r->tune_state == 2 && r->tune_recheck > 0) { // a decision kept across camera moves
    const double shade_taps = 3.0 * (double)r->stats.shaded_samples + (double)r->stats.shadow_samples;
    const bool heavy = r->P.shading != 0 && r->stats.samples > 0 && shade_taps >= 3.0 * (double)r->stats.samples;
    if (!heavy) { r->tune_layout = -1; r->tune_pipeline = 0; r->tune_recheck = 0; }   // not that regime any more: back to the rules
    else if (--r->tune_recheck == 0) r->tune_state = 0;                              // static for a while: measure again
  }
  { // automatic shading pipeline of the next frame (see auto_inplace)
    const double steps = (double)r->stats.samples + (double)r->stats.skipped_samples, shaded = (double)r->stats.shaded_samples;
    if (steps > 0.0) {
      // (round 3) ... unless the shadow marches are long - more than 60 iterations per shaded sample, i.e. sampling rates above ~2 (the
      // stride is 10 / rate^2 voxels): then tiles differ again by what their shadow rays cross, and pooling wins by 30-50 % (C3 front /
      // dense at rate 4: 22.7 ms in place, 13.4 pooled; profiles/r03_notes.md section 5)
      const bool long_shadows = (double)r->stats.shadow_samples > 60.0 * std::max(shaded, 1.0);
      if (shaded >= 0.50 * steps && !long_shadows) r->auto_inplace = true;
      else if (shaded < 0.35 * steps || long_shadows) r->auto_inplace = false;
    }
  }
  if (r->frame_used_skip && r->skip_adaptive) { // did skipping pay?  (see skip_active)
    const double skipped = (double)r->stats.skipped_samples + (double)r->stats.skipped_shadow_samples;
    const double all = skipped + (double)r->stats.samples + (double)r->stats.shadow_samples;
    if (all > 0.0 && skipped < 0.10 * all) {
      r->skip_active = false;
      r->skip_reprobe_in = r->skip_backoff;
      r->skip_backoff = std::min(r->skip_backoff * 2, 256);
    }
    else r->skip_backoff = 32;
  }
  if (r->d_trace) {
    if (const char* path = getenv("OVR_HIP_TRACE_FILE")) {
      std::vector<unsigned long long> h(r->trace_words);
      HIP_TRY(hipMemcpy(h.data(), r->d_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
  }
  r->async_pending = false;
  r->packed_early = false;
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Device group: the image-plane shard of SURVEY.md 8e INSIDE one process, behind the same handle - what the reference's unmodified
// apps reach through the plugin (north_star: "apps run unmodified ... the 8 GPUs of one node shard the image plane into
// tiles with a final RCCL gather over xGMI").  Every member renders its own tiles of the frame on its own device (volume replicated,
// TEA seeds by global pixel index: the assembled frame is the single-GPU frame bit for bit); at the end of the frame each follower packs
// its tiles, its comm_stream ships them to the leader's device - ncclSend / ncclRecv over RCCL's communicators of this process
// (ncclCommInitAll) when the devices are distinct, peer-to-peer copies otherwise (OVR_HIP_GATHER=rccl|copy forces one) - and the leader
// scatters all payloads into its framebuffer with one launch per layer.  mapframe / swap / accumulation work on the leader's
// framebuffer as with one device.  The reference has one device (device_impl.cpp:371-372) and nothing to mirror here.
// ------------------------------------------------------------------------------------------------------------------
struct RcclApi { // resolved at run time: libovr_hip.so does not link RCCL, a host without it falls back to peer copies
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const RcclApi& rccl_api()
{
  static const RcclApi api = [] {
    RcclApi a;
    for (const char* name : { "librccl.so.1", "librccl.so" })
      if ((a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!a.lib) return a;
    a.CommInitAll = (int (*)(void**, int, const int*))dlsym(a.lib, "ncclCommInitAll");
    a.CommDestroy = (int (*)(void*))dlsym(a.lib, "ncclCommDestroy");
    a.GroupStart = (int (*)())dlsym(a.lib, "ncclGroupStart");
    a.GroupEnd = (int (*)())dlsym(a.lib, "ncclGroupEnd");
    a.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(a.lib, "ncclSend");
    a.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(a.lib, "ncclRecv");
    a.GetErrorString = (const char* (*)(int))dlsym(a.lib, "ncclGetErrorString");
    a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
    return a;
  }();
  return api;
}
constexpr int kNcclFloat = 7; // ncclFloat32, rccl.h

// (see ovr_hip_renderer::group_mtx; followers are only ever called with their leader's lock held)
struct GroupLock {
  std::unique_lock<std::mutex> lk;
  explicit GroupLock(ovr_hip_renderer* r) { if (r && !r->leader && r->members.size() > 1) lk = std::unique_lock<std::mutex>(r->group_mtx); }
};

#define GROUP_FORWARD(r, call)                                                                                          \
  do {                                                                                                                  \
    for (size_t i_ = 1; i_ < (r)->members.size(); ++i_) {                                                               \
      ovr_hip_renderer* m = (r)->members[i_];                                                                           \
      if (int e_ = (call)) { (void)hipSetDevice((r)->device); return e_; }                                              \
    }                                                                                                                   \
    if ((r)->members.size() > 1) (void)hipSetDevice((r)->device);                                                       \
  } while (0)

// (re)size the payload and gather buffers for the committed framebuffer size and tiles
int group_buffers(ovr_hip_renderer* L)
{
  const int n = (int)L->members.size();
  const ShardP& s0 = L->shard.current;
  const int W = L->fbsize.current.w, H = L->fbsize.current.h;
  if (L->group_fb[0] == W && L->group_fb[1] == H && L->group_fb[2] == s0.tw && L->group_fb[3] == s0.th && L->group_fb[4] == n) return 0;
  size_t most = 0;
  for (int i = 0; i < n; ++i) {
    ovr_hip_renderer* m = L->members[i];
    const size_t px = (size_t)count_owned_tiles(W, H, s0.tw, s0.th, i, n) * s0.tw * s0.th;
    most = std::max(most, px);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->comm_stream));
    for (int c = 0; c < 2; ++c) {
      if (m->d_payload[c]) HIP_TRY(hipFree(m->d_payload[c]));
      m->d_payload[c] = nullptr;
      m->payload_floats[c] = px * (c == 0 ? 4 : 3);
      if (i > 0 && px > 0 && (c == 0 || L->group_grad)) HIP_TRY(hipMalloc((void**)&m->d_payload[c], m->payload_floats[c] * sizeof(float)));
    }
  }
  HIP_TRY(hipSetDevice(L->device));
  for (int c = 0; c < 2; ++c) {
    if (L->d_gather[c]) HIP_TRY(hipFree(L->d_gather[c]));
    L->d_gather[c] = nullptr;
    L->group_stride[c] = ((most * (c == 0 ? 4 : 3) + 11) / 12) * 12; // floats; a multiple of 3 and of 4: whole pixels in either layer
    if (most > 0 && (c == 0 || L->group_grad)) HIP_TRY(hipMalloc((void**)&L->d_gather[c], L->group_stride[c] * (size_t)n * sizeof(float)));
  }
  L->group_fb[0] = W; L->group_fb[1] = H; L->group_fb[2] = s0.tw; L->group_fb[3] = s0.th; L->group_fb[4] = n;
  return 0;
}

// ---- the followers' host threads --------------------------------------------------------------------------------------------------
// One per follower, created with the group, pinned to the member's device once (hipSetDevice is per thread).  The leader's thread posts a
// command to every worker and waits for all of them: the members' commits, frame launches, packs and frame ends run side by side.  A worker
// spins on its mailbox for a short while after a command (a render loop posts the next one within microseconds) and then sleeps on a
// condition variable; results travel back in the mailbox (g_last_error is thread-local).
enum { GW_NONE = 0, GW_COMMIT, GW_RENDER, GW_SHIP, GW_FINISH, GW_SWAP, GW_CALL, GW_QUIT };
} // namespace
struct GroupWorker {
  ovr_hip_renderer* m = nullptr;
  std::thread th;
  std::mutex mx;
  std::condition_variable cv;
  std::atomic<uint64_t> posted{ 0 }, done{ 0 };
  std::atomic<bool> asleep{ false };
  int cmd = GW_NONE;          // written by the leader before `posted` advances, read by the worker after it has seen it
  std::function<int()> fn;    // GW_CALL
  int rc = 0;                 // the command's result ...
  std::string err;            // ... and its message
};
namespace {
int group_pack(ovr_hip_renderer* L, ovr_hip_renderer* m);
int finish_frame_one(ovr_hip_renderer* r);

inline void cpu_relax() { __builtin_ia32_pause(); }

void group_worker_main(GroupWorker* w)
{
  ovr_hip_renderer* m = w->m;
  (void)hipSetDevice(m->device);
  uint64_t seen = 0;
  static const int spin_us = getenv("OVR_HIP_WORKER_SPIN_US") ? atoi(getenv("OVR_HIP_WORKER_SPIN_US")) : 200;
  for (;;) {
    // wait for the next command: spin first (the render loop's next post is microseconds away), then sleep
    const auto t0 = std::chrono::steady_clock::now();
    int polls = 0;
    while (w->posted.load(std::memory_order_acquire) == seen) {
      cpu_relax();
      if ((++polls & 255) == 0 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) {
        std::unique_lock<std::mutex> lk(w->mx);
        w->asleep.store(true);
        w->cv.wait(lk, [&] { return w->posted.load() != seen; });
        w->asleep.store(false);
      }
    }
    seen = w->posted.load(std::memory_order_acquire);
    int rc = 0;
    g_last_error.clear();
    switch (w->cmd) {
    case GW_COMMIT: rc = ovr_hip_commit(m); break;
    case GW_RENDER: rc = ovr_hip_render_async(m); break;
    case GW_SHIP: rc = group_pack(m->leader, m); break;
    case GW_FINISH: rc = finish_frame_one(m); break;
    case GW_SWAP: rc = ovr_hip_swap(m); break;
    case GW_CALL: rc = w->fn ? w->fn() : 0; break;
    case GW_QUIT: w->done.store(seen, std::memory_order_release); return;
    default: break;
    }
    w->rc = rc;
    if (rc) w->err = g_last_error;
    w->done.store(seen, std::memory_order_release);
  }
}

void gw_post(GroupWorker* w, int cmd)
{
  w->cmd = cmd;
  w->posted.fetch_add(1); // (seq_cst: ordered against the worker's `asleep` store - either it sees the post or this thread sees it asleep)
  if (w->asleep.load()) {
    std::lock_guard<std::mutex> lk(w->mx);
    w->cv.notify_one();
  }
}
int gw_wait(GroupWorker* w)
{
  const uint64_t want = w->posted.load();
  int polls = 0;
  while (w->done.load(std::memory_order_acquire) != want) {
    cpu_relax();
    if ((++polls & 4095) == 0) std::this_thread::yield();
  }
  if (w->rc) g_last_error = w->err;
  return w->rc;
}
// the same command on every follower, the leader's own share (if any) meanwhile on this thread; the first error wins, every worker is waited for
template <typename Own> int group_run(ovr_hip_renderer* L, int cmd, Own own)
{
  const size_t n = L->members.size();
  for (size_t i = 1; i < n; ++i) gw_post(L->members[i]->worker, cmd);
  int e = own();
  std::string msg = e ? g_last_error : std::string();
  for (size_t i = 1; i < n; ++i) {
    const int ei = gw_wait(L->members[i]->worker);
    if (ei && !e) { e = ei; msg = g_last_error; }
  }
  if (e) g_last_error = msg;
  return e;
}
int group_run(ovr_hip_renderer* L, int cmd) { return group_run(L, cmd, [] { return 0; }); }
// an arbitrary call per follower (volume upload, noise tile: rare, may allocate)
int group_call(ovr_hip_renderer* L, const std::function<int(ovr_hip_renderer*)>& f)
{
  const size_t n = L->members.size();
  for (size_t i = 1; i < n; ++i) {
    ovr_hip_renderer* m = L->members[i];
    m->worker->fn = [m, &f] { return f(m); };
  }
  return group_run(L, GW_CALL);
}
void group_stop_workers(ovr_hip_renderer* L)
{
  for (size_t i = 1; i < L->members.size(); ++i) {
    GroupWorker* w = L->members[i]->worker;
    if (!w) continue;
    if (w->th.joinable()) { gw_post(w, GW_QUIT); w->th.join(); }
    delete w;
    L->members[i]->worker = nullptr;
  }
}

// follower m (on its own thread): pack the tiles of the frame in flight (or just finished) behind it; with peer copies the payload leaves at once
// on m's comm_stream (ends in m->ev_shipped), with RCCL it waits in m->d_payload for the leader's grouped send / recv (group_rccl)
int group_pack(ovr_hip_renderer* L, ovr_hip_renderer* m)
{
  const int W = L->fbsize.current.w, H = L->fbsize.current.h;
  const ShardP& s = m->shard.current;
  m->shipped_by_rccl = false;
  if (m->payload_floats[0] == 0) { HIP_TRY(hipEventRecord(m->ev_shipped, m->comm_stream)); return 0; }
  // ovr_hip_pack_tiles resolves a frame whose request pool is not yet known to be roomy before it packs (a pool overflow renders the frame again)
  if (int e = ovr_hip_pack_tiles(m, m->d_payload[0], m->payload_floats[0] * sizeof(float))) return e;
  if (L->group_grad) HIP_TRY(launch_pack_tiles(m->d_grad[m->cur], m->d_payload[1], W, H, s.tw, s.th, s.rank, s.world, m->stream(), 3));
  HIP_TRY(hipEventRecord(m->ev_packed, m->stream()));
  HIP_TRY(hipStreamWaitEvent(m->comm_stream, m->ev_packed, 0));
  if (L->gather_kind == 2) return 0;
  for (int c = 0; c < 2; ++c) {
    if (c == 1 && !L->group_grad) break;
    float* dst = L->d_gather[c] + (size_t)m->group_rank * L->group_stride[c];
    if (m->device == L->device) HIP_TRY(hipMemcpyAsync(dst, m->d_payload[c], m->payload_floats[c] * sizeof(float), hipMemcpyDeviceToDevice, m->comm_stream));
    else HIP_TRY(hipMemcpyPeerAsync(dst, L->device, m->d_payload[c], m->device, m->payload_floats[c] * sizeof(float), m->comm_stream));
  }
  HIP_TRY(hipEventRecord(m->ev_shipped, m->comm_stream));
  return 0;
}

// leader thread, RCCL: the payloads of the listed followers travel inside ONE ncclGroupStart / End - n - 1 sends on the followers' communicators and
// comm streams, n - 1 receives on the leader's - and the leader's comm_stream records ONE event behind them (round 4 issued a group per follower: 7
// launch pairs a frame at 8 members, each receive queued behind the slowest member packed before it).  false = RCCL refused: the caller falls back.
bool group_rccl(ovr_hip_renderer* L, const std::vector<ovr_hip_renderer*>& who)
{
  const RcclApi& N = rccl_api();
  if (hipSetDevice(L->device) != hipSuccess) return false;
  // the receives start once every listed payload exists (a receive kernel spinning beside the leader's own frame would only take CUs from it)
  for (ovr_hip_renderer* m : who)
    if (m->payload_floats[0] > 0 && hipStreamWaitEvent(L->comm_stream, m->ev_packed, 0) != hipSuccess) return false;
  int rc = N.GroupStart();
  for (size_t k = 0; k < who.size() && rc == 0; ++k) {
    ovr_hip_renderer* m = who[k];
    if (m->payload_floats[0] == 0) continue;
    for (int c = 0; c < 2 && rc == 0; ++c) {
      if (c == 1 && !L->group_grad) break;
      (void)hipSetDevice(m->device);
      rc = N.Send(m->d_payload[c], m->payload_floats[c], kNcclFloat, 0, m->rccl_comm, m->comm_stream);
      (void)hipSetDevice(L->device);
      if (rc == 0) rc = N.Recv(L->d_gather[c] + (size_t)m->group_rank * L->group_stride[c], m->payload_floats[c], kNcclFloat, m->group_rank, L->rccl_comm, L->comm_stream);
    }
  }
  const int rc2 = N.GroupEnd();
  (void)hipSetDevice(L->device);
  if (rc != 0 || rc2 != 0) {
    fprintf(stderr, "[hip] RCCL send / recv of a device group's tiles failed (%s): the group goes on with peer copies\n", N.GetErrorString ? N.GetErrorString(rc != 0 ? rc : rc2) : "?");
    return false;
  }
  if (hipEventRecord(L->ev_gathered, L->comm_stream) != hipSuccess) return false;
  for (ovr_hip_renderer* m : who) m->shipped_by_rccl = true;
  return true;
}

// packs (workers, side by side) and ships the tiles of the listed followers; `all` = every follower (the common case: one post per worker)
int group_ship(ovr_hip_renderer* L, const std::vector<ovr_hip_renderer*>& who, bool all)
{
  auto pack = [&]() -> int {
    if (all) return group_run(L, GW_SHIP);
    for (ovr_hip_renderer* m : who) gw_post(m->worker, GW_SHIP);
    int e = 0;
    std::string msg;
    for (ovr_hip_renderer* m : who) { const int ei = gw_wait(m->worker); if (ei && !e) { e = ei; msg = g_last_error; } }
    if (e) g_last_error = msg;
    return e;
  };
  if (int e = pack()) return e;
  if (L->gather_kind == 2 && !group_rccl(L, who)) {
    L->gather_kind = 1; // same bytes, same destination, by peer copies from here on
    (void)hipGetLastError();
    if (int e = pack()) return e;
  }
  return 0;
}

int group_finish(ovr_hip_renderer* L)
{
  const int n = (int)L->members.size();
  bool any = L->async_pending;
  for (int i = 1; i < n; ++i) any = any || L->members[i]->async_pending;
  if (!any) return 0;
  const int W = L->fbsize.current.w, H = L->fbsize.current.h;
  if (int e = group_buffers(L)) return e;
  const auto t0 = std::chrono::high_resolution_clock::now();
  // 1. the followers pack behind their frames and ship on their comm streams while everything still renders
  std::vector<ovr_hip_renderer*> followers(L->members.begin() + 1, L->members.end());
  if (int e = group_ship(L, followers, true)) return e;
  const auto t1 = std::chrono::high_resolution_clock::now();
  // 2. every frame to its end (a follower whose request pool overflowed renders again: its early payload is stale)
  if (int e = group_run(L, GW_FINISH, [&] { return finish_frame_one(L); })) return e;
  std::vector<ovr_hip_renderer*> stale;
  for (ovr_hip_renderer* m : followers)
    if (m->stats.stale_tiles) stale.push_back(m);
  if (!stale.empty())
    if (int e = group_ship(L, stale, false)) return e;
  const auto t2 = std::chrono::high_resolution_clock::now();
  // 3. the leader scatters the payloads into the framebuffer set the frame rendered into
  hipStream_t st = L->stream();
  bool waited_rccl = false;
  for (ovr_hip_renderer* m : followers) {
    if (!m->shipped_by_rccl) HIP_TRY(hipStreamWaitEvent(st, m->ev_shipped, 0));
    else if (!waited_rccl) { HIP_TRY(hipStreamWaitEvent(st, L->ev_gathered, 0)); waited_rccl = true; } // (recorded behind the LAST grouped receive: covers a re-shipment too)
  }
  const ShardP& s = L->shard.current;
  if (W > 0 && H > 0 && L->d_gather[0]) {
    HIP_TRY(launch_unpack_tiles(L->d_gather[0], L->d_rgba[L->frame_set], W, H, s.tw, s.th, -1, n, L->group_stride[0], st, 4, 0));
    if (L->group_grad && L->d_gather[1]) HIP_TRY(launch_unpack_tiles(L->d_gather[1], L->d_grad[L->frame_set], W, H, s.tw, s.th, -1, n, L->group_stride[1], st, 3, 0));
  }
  HIP_TRY(hipStreamSynchronize(st));
  const auto t3 = std::chrono::high_resolution_clock::now();
  L->group_gather_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
  L->group_host_us[1] = std::chrono::duration<double, std::micro>(t1 - t0).count();
  L->group_host_us[2] = std::chrono::duration<double, std::micro>(t2 - t1).count();
  L->group_host_us[3] = std::chrono::duration<double, std::micro>(t3 - t2).count();
  // 4. the frame's counters: sums over the members, times of the slowest
  L->own_stats = L->stats;
  for (int i = 1; i < n; ++i) {
    const ovr_hip_stats& a = L->members[i]->stats;
    ovr_hip_stats& t = L->stats;
    t.rays += a.rays; t.samples += a.samples; t.shaded_samples += a.shaded_samples; t.shadow_samples += a.shadow_samples;
    t.active_pixels += a.active_pixels; t.skipped_samples += a.skipped_samples; t.skipped_shadow_samples += a.skipped_shadow_samples;
    t.pool_chunks += a.pool_chunks;
    t.kernel_ms = std::max(t.kernel_ms, a.kernel_ms); t.march_ms = std::max(t.march_ms, a.march_ms);
    t.shade_ms = std::max(t.shade_ms, a.shade_ms); t.composite_ms = std::max(t.composite_ms, a.composite_ms);
    t.replicas_building += a.replicas_building;
    if (a.tuning == 1) t.tuning = 1;
    if (L->members[i]->outside_hits) { // a hit through an ignored slab on any member: the leader's set is mapped whole (see finish_frame_one)
      int* q = L->d_rect[L->frame_set]; q[0] = 0; q[1] = 0; q[2] = W; q[3] = H;
    }
  }
  L->stats.stale_tiles = 0;
  return 0;
}

int finish_frame(ovr_hip_renderer* r)
{
  if (r->members.size() > 1) return group_finish(r);
  return finish_frame_one(r);
}

} // namespace

extern "C" {

const char* ovr_hip_last_error(void) { return g_last_error.c_str(); }
int ovr_hip_abi_version(void) { return OVR_HIP_ABI_VERSION; }

int ovr_hip_create(ovr_hip_renderer** out, int device_id)
{
  if (!out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_create: null output pointer");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(OVR_HIP_EDEVICE, "[hip] no HIP device available - this backend has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(OVR_HIP_EINVAL, "[hip] invalid device ordinal " + std::to_string(device_id));
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(OVR_HIP_EDEVICE, std::string("[hip] device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
  ovr_hip_renderer* r = new ovr_hip_renderer();
  r->device = device_id;
  auto acquire = [&]() -> int {
    HIP_TRY(hipStreamCreate(&r->own_stream[0]));
    HIP_TRY(hipStreamCreate(&r->own_stream[1]));
    for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&r->ev[i]));
    {
      // replicas are built at the highest stream priority: the sooner a replica is resident, the sooner the frames that asked for it get faster
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
      HIP_TRY(hipStreamCreateWithPriority(&r->build_stream, hipStreamNonBlocking, hi));
    }
    for (int k = 0; k < kLayouts; ++k) HIP_TRY(hipEventCreateWithFlags(&r->build_ev[k], hipEventDisableTiming));
    HIP_TRY(hipMalloc((void**)&r->d_counters, kFrameWordsBytes));
    HIP_TRY(hipMemset(r->d_counters, 0, kFrameWordsBytes));
    HIP_TRY(hipMalloc((void**)&r->d_sparse_count, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&r->d_data_range, minmax_reduce_floats() * sizeof(float)));
    // pinned, read by the host after a frame: 8 counters, then the pool's control words (written by the frame's last reduction kernel)
    HIP_TRY(hipHostMalloc((void**)&r->h_counters, 8 * sizeof(unsigned long long) + (size_t)kPoolCtrlWords * sizeof(unsigned int), hipHostMallocDefault));
    r->h_ctrl = reinterpret_cast<unsigned int*>(r->h_counters + 8);
    return 0;
  };
  if (int e = acquire()) { ovr_hip_destroy(r); return e; } // nothing half-built leaks
  std::memset(r->h_counters, 0, 8 * sizeof(unsigned long long) + (size_t)kPoolCtrlWords * sizeof(unsigned int));
  // defaults of the reference's parameter block (params.h:55-99, renderer.h:255-285)
  r->spp.current = r->spp.queued = 1;
  r->sparse.current = r->sparse.queued = 0;
  r->accumulate.current = r->accumulate.queued = 0;
  r->shading.current = r->shading.queued = OVR_HIP_SHADE_FULL;
  r->grid_convention.current = r->grid_convention.queued = OVR_HIP_GRID_CELL_CENTRED;
  r->rate.current = r->rate.queued = 1.f;
  r->layouts.current = r->layouts.queued = 1;
  r->layout_choice.current = r->layout_choice.queued = -1;
  if (const char* f = getenv("OVR_HIP_LAYOUTS")) r->layouts.current = r->layouts.queued = atoi(f); // diagnostic override
  if (const char* f = getenv("OVR_HIP_SKIP_ADAPTIVE")) r->skip_adaptive = atoi(f) != 0;
  if (const char* f = getenv("OVR_HIP_TUNE")) r->tune_on = atoi(f) != 0;
  if (const char* f = getenv("OVR_HIP_SHADE_ORDER")) r->shade_order_on = atoi(f) != 0;
  if (const char* f = getenv("OVR_HIP_ROW_LOADS")) r->P.row_loads = atoi(f) != 0 ? 2 : 1; // (RayMarchParams::row_loads; 0 = by size)
  if (const char* f = getenv("OVR_HIP_SHADE_BEAM")) r->shade_beam = std::max(1.f, (float)atof(f));
  *out = r;
  return 0;
}

int ovr_hip_create_group(ovr_hip_renderer** out, const int32_t* device_ids, int32_t n_devices)
{
  if (!out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_create_group: null output pointer");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_create_group: 1 to 64 device ordinals expected");
  ovr_hip_renderer* L = nullptr;
  if (int e = ovr_hip_create(&L, device_ids[0])) return e;
  if (n_devices == 1) { *out = L; return 0; } // one device: an ordinary renderer
  L->members.push_back(L);
  auto build = [&]() -> int {
    for (int i = 1; i < n_devices; ++i) {
      ovr_hip_renderer* m = nullptr;
      if (int e = ovr_hip_create(&m, device_ids[i])) return e;
      m->leader = L;
      m->group_rank = i;
      L->members.push_back(m);
    }
    int tw = 16, th = 16; // measured work max / mean over 8 ranks 1.07 (1.33 with 64 x 64): DESIGN.md section 5
    if (const char* t = getenv("OVR_HIP_TILE")) { int a = 0, b = 0; if (sscanf(t, "%dx%d", &a, &b) == 2 && a > 0 && b > 0) { tw = a; th = b; } }
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i)
      for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    for (int i = 0; i < n_devices; ++i) {
      ovr_hip_renderer* m = L->members[i];
      HIP_TRY(hipSetDevice(m->device));
      HIP_TRY(hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&m->ev_packed, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&m->ev_shipped, hipEventDisableTiming));
      if (i > 0) {
        HIP_TRY(hipSetDevice(L->device));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_received, hipEventDisableTiming));
        HIP_TRY(hipSetDevice(m->device));
      }
      ShardP s; s.rank = i; s.world = n_devices; s.tw = tw; s.th = th;
      m->shard.current = m->shard.queued = s;
      m->sched_list_dirty = true;
      for (int j = 0; j < n_devices; ++j) // direct xGMI / PCIe peer access where the platform has it (the copies work without, staged by the runtime)
        if (device_ids[j] != m->device) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->device, device_ids[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(device_ids[j], 0); }
      (void)hipGetLastError(); // hipErrorPeerAccessAlreadyEnabled is not an error
    }
    HIP_TRY(hipSetDevice(L->device));
    HIP_TRY(hipEventCreateWithFlags(&L->ev_gathered, hipEventDisableTiming));
    if (const char* g = getenv("OVR_HIP_MAP_GRAD")) L->group_grad = g[0] != '0';
    const char* want = getenv("OVR_HIP_GATHER");
    const bool force_rccl = want && std::string(want) == "rccl", force_copy = want && std::string(want) == "copy";
    if (want && !force_rccl && !force_copy) return fail(OVR_HIP_EINVAL, "[hip] OVR_HIP_GATHER must be 'rccl' or 'copy'");
    L->gather_kind = 1;
    if (!force_copy && distinct && rccl_api().ok) { // one communicator per device of this process (ncclCommInitAll refuses a device listed twice)
      std::vector<void*> comms((size_t)n_devices, nullptr);
      std::vector<int> devs(device_ids, device_ids + n_devices);
      const int rc = rccl_api().CommInitAll(comms.data(), n_devices, devs.data());
      if (rc == 0) {
        for (int i = 0; i < n_devices; ++i) L->members[i]->rccl_comm = comms[(size_t)i];
        L->gather_kind = 2;
      }
      else if (force_rccl) return fail(OVR_HIP_EDEVICE, std::string("[hip] ncclCommInitAll failed: ") + (rccl_api().GetErrorString ? rccl_api().GetErrorString(rc) : "?"));
      HIP_TRY(hipSetDevice(L->device));
    }
    else if (force_rccl) return fail(OVR_HIP_EDEVICE, distinct ? "[hip] OVR_HIP_GATHER=rccl but librccl.so could not be loaded" : "[hip] OVR_HIP_GATHER=rccl needs distinct devices (RCCL refuses a device listed twice)");
    // one host thread per follower (round 5): device set once, persistent - commits, frame launches, packs and frame ends of the members run side by side
    for (int i = 1; i < n_devices; ++i) {
      GroupWorker* w = new GroupWorker();
      w->m = L->members[i];
      L->members[i]->worker = w;
      try { w->th = std::thread(group_worker_main, w); }
      catch (const std::exception& ex) { return fail(OVR_HIP_EDEVICE, std::string("[hip] ovr_hip_create_group: no host thread for a member: ") + ex.what()); }
    }
    // the first multi-GPU run explains itself: which way the tiles travel and why, and what the platform says about every pair
    if (!(getenv("OVR_HIP_QUIET") && atoi(getenv("OVR_HIP_QUIET")) != 0)) {
      std::string why = L->gather_kind == 2 ? "RCCL send / recv (ncclCommInitAll over the listed devices; one ncclGroupStart / End per frame)"
                        : force_copy       ? "peer copies (OVR_HIP_GATHER=copy)"
                        : !distinct        ? "peer copies (a device is listed more than once: RCCL refuses that)"
                        : !rccl_api().ok   ? "peer copies (librccl.so could not be loaded)"
                                           : "peer copies (ncclCommInitAll failed)";
      std::string devs, peers;
      for (int i = 0; i < n_devices; ++i) devs += (i ? "," : "") + std::to_string(device_ids[i]);
      for (int i = 0; i < n_devices; ++i) {
        peers += i ? " | " : "";
        for (int j = 0; j < n_devices; ++j) {
          int can = device_ids[i] == device_ids[j] ? 1 : 0;
          if (!can && hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) != hipSuccess) { (void)hipGetLastError(); can = -1; }
          peers += can < 0 ? "?" : can ? "1" : "0";
        }
      }
      fprintf(stderr, "[hip] device group: devices [%s], image tiles %dx%d, gather to device %d by %s; gradient layer %s; hipDeviceCanAccessPeer rows (member -> member): %s\n",
              devs.c_str(), tw, th, device_ids[0], why.c_str(), L->group_grad ? "travels too" : "stays (OVR_HIP_MAP_GRAD=0)", peers.c_str());
    }
    return 0;
  };
  if (int e = build()) { const std::string msg = g_last_error; ovr_hip_destroy(L); g_last_error = msg; return e; }
  *out = L;
  return 0;
}

// The RCCL entry points a device group uses, exercised on ONE device: a communicator of one rank (ncclCommInitAll), a send to itself matched by
// a receive inside one ncclGroupStart / End on a stream, the payload compared.  A device group's own RCCL branch needs distinct devices; this
// pins what can be pinned on one card - the library resolves, the prototypes and the data-type constant are right, the calls run on a stream.
int ovr_hip_rccl_selftest(int device_id)
{
  const RcclApi& N = rccl_api();
  if (!N.ok) return fail(OVR_HIP_ESTATE, "[hip] librccl.so could not be loaded (a device group then gathers with peer copies)");
  HIP_TRY(hipSetDevice(device_id));
  void* comm = nullptr;
  const int dev = device_id;
  int rc = N.CommInitAll(&comm, 1, &dev);
  if (rc != 0) return fail(OVR_HIP_EDEVICE, std::string("[hip] ncclCommInitAll(1 device) failed: ") + (N.GetErrorString ? N.GetErrorString(rc) : "?"));
  const size_t n = 1 << 16;
  float *src = nullptr, *dst = nullptr;
  hipStream_t st = nullptr;
  std::vector<float> h(n), back(n, -1.f);
  for (size_t i = 0; i < n; ++i) h[i] = (float)i * 0.25f;
  auto run = [&]() -> int {
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&src, n * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&dst, n * sizeof(float)));
    HIP_TRY(hipMemcpy(src, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dst, 0, n * sizeof(float)));
    int a = N.GroupStart();
    if (a == 0) a = N.Send(src, n, kNcclFloat, 0, comm, st);
    if (a == 0) a = N.Recv(dst, n, kNcclFloat, 0, comm, st);
    const int b = N.GroupEnd();
    if (a != 0 || b != 0) return fail(OVR_HIP_EDEVICE, std::string("[hip] RCCL self send / recv failed: ") + (N.GetErrorString ? N.GetErrorString(a != 0 ? a : b) : "?"));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(back.data(), dst, n * sizeof(float), hipMemcpyDeviceToHost));
    if (std::memcmp(back.data(), h.data(), n * sizeof(float)) != 0) return fail(OVR_HIP_EDEVICE, "[hip] RCCL self send / recv delivered other bytes than were sent");
    return 0;
  };
  const int e = run();
  const std::string msg = g_last_error;
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  if (st) (void)hipStreamDestroy(st);
  (void)N.CommDestroy(comm);
  g_last_error = msg;
  return e;
}

int ovr_hip_group_info(const ovr_hip_renderer* r, int32_t* n_devices, int32_t* gather_kind, double* gather_ms)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (n_devices) *n_devices = r->members.size() > 1 ? (int32_t)r->members.size() : 1;
  if (gather_kind) *gather_kind = r->members.size() > 1 ? r->gather_kind : 0;
  if (gather_ms) *gather_ms = r->members.size() > 1 ? r->group_gather_ms : 0.0;
  return 0;
}

int ovr_hip_group_host_times(const ovr_hip_renderer* r, double out_us[4])
{
  if (!r || !out_us) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_group_host_times: null argument");
  for (int i = 0; i < 4; ++i) out_us[i] = r->members.size() > 1 ? r->group_host_us[i] : 0.0;
  return 0;
}

int ovr_hip_get_member_stats(const ovr_hip_renderer* r, int32_t member, ovr_hip_stats* out)
{
  if (!r || !out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_member_stats: null argument");
  const int n = r->members.size() > 1 ? (int)r->members.size() : 1;
  if (member < 0 || member >= n) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_member_stats: no such member");
  if (r->async_pending) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_get_member_stats: a frame is still in flight (call ovr_hip_sync)");
  *out = n > 1 ? (member == 0 ? r->own_stats : r->members[(size_t)member]->stats) : r->stats;
  return 0;
}

void ovr_hip_destroy(ovr_hip_renderer* r)
{
  if (!r) return;
  if (r->members.size() > 1) group_stop_workers(r);
  for (size_t i = 1; i < r->members.size(); ++i) ovr_hip_destroy(r->members[i]); // a group leader takes its followers with it
  r->members.clear();
  join_builders(r);
  (void)hipSetDevice(r->device);
  (void)hipDeviceSynchronize();
  if (r->rccl_comm && rccl_api().ok) (void)rccl_api().CommDestroy(r->rccl_comm);
  for (int c = 0; c < 2; ++c) {
    if (r->d_payload[c]) (void)hipFree(r->d_payload[c]);
    if (r->d_gather[c]) (void)hipFree(r->d_gather[c]);
  }
  if (r->comm_stream) (void)hipStreamDestroy(r->comm_stream);
  if (r->ev_packed) (void)hipEventDestroy(r->ev_packed);
  if (r->ev_shipped) (void)hipEventDestroy(r->ev_shipped);
  if (r->ev_received) (void)hipEventDestroy(r->ev_received);
  if (r->ev_gathered) (void)hipEventDestroy(r->ev_gathered);
  (void)free_framebuffers(r);
  for (int k = 0; k < kLayouts; ++k) {
    if (r->d_replica[k]) (void)hipFree(r->d_replica[k]);
    if (r->d_axis[k]) (void)hipFree(r->d_axis[k]);
  }
  if (r->d_tf_color) (void)hipFree(r->d_tf_color);
  if (r->h_tf) (void)hipHostFree(r->h_tf);
  if (r->ev_tf) (void)hipEventDestroy(r->ev_tf);
  if (r->d_noise) (void)hipFree(r->d_noise);
  if (r->d_mc_minmax) (void)hipFree(r->d_mc_minmax);
  if (r->d_mc_majorant) (void)hipFree(r->d_mc_majorant);
  if (r->d_mc_occupancy) (void)hipFree(r->d_mc_occupancy);
  if (r->d_mc_fine) (void)hipFree(r->d_mc_fine);
  if (r->d_counters) (void)hipFree(r->d_counters);
  if (r->d_sparse_count) (void)hipFree(r->d_sparse_count);
  if (r->d_data_range) (void)hipFree(r->d_data_range);
  if (r->h_counters) (void)hipHostFree(r->h_counters);
  for (int i = 0; i < 4; ++i) if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
  if (r->pool.reqs) (void)hipFree(r->pool.reqs);
  if (r->pool.chunk_next) (void)hipFree(r->pool.chunk_next);
  if (r->pool.chunk_n) (void)hipFree(r->pool.chunk_n);
  if (r->pool.shade_counters) (void)hipFree(r->pool.shade_counters);
  if (r->pool.order) (void)hipFree(r->pool.order);
  if (r->pool.order_key) (void)hipFree(r->pool.order_key);
  if (r->pool.order_ws) (void)hipFree(r->pool.order_ws);

  if (r->d_block_counters) (void)hipFree(r->d_block_counters);
  if (r->d_sched_info) (void)hipHostFree(r->d_sched_info);
  if (r->d_trace) (void)hipFree(r->d_trace);
  for (int i = 0; i < 2; ++i)
    if (r->own_stream[i]) (void)hipStreamDestroy(r->own_stream[i]);
  if (r->build_stream) (void)hipStreamDestroy(r->build_stream);
  for (int k = 0; k < kLayouts; ++k)
    if (r->build_ev[k]) (void)hipEventDestroy(r->build_ev[k]);
  delete r;
}

int ovr_hip_set_stream(ovr_hip_renderer* r, void* s)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (r->members.size() > 1 && s) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_stream: a device group renders on one stream per device");
  if (int e = finish_frame(r)) return e;
  HIP_TRY(hipStreamSynchronize(r->stream())); // table copies and the like enqueued on the stream that is being left
  // ... and what a reset left on the OTHER set's own stream (its memsets: enqueue_frame; a table copy recorded there): the caller's stream is not
  // ordered behind either of them
  for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamSynchronize(r->own_stream[i]));
  r->user_stream = (hipStream_t)s;
  r->use_user_stream = (s != nullptr);
  return 0;
}

} // extern "C"
namespace {
double ms_since(std::chrono::high_resolution_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count(); }

// one renderer's volume upload (a group member's or a single renderer's).  r->upload_ms = { total, allocation, host-to-device / peer copies, kernels }
int set_volume_one(ovr_hip_renderer* r, const void* data, int mem_kind, int value_type, const int32_t dims[3],
                   const float grid_origin[3], const float grid_spacing[3])
{
  const auto t_call = std::chrono::high_resolution_clock::now();
  double t_alloc = 0.0, t_copy = 0.0, t_kern = 0.0;
  if (!r || !data || !dims || !grid_origin || !grid_spacing) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_volume: null argument");
  for (double& v : r->upload_ms) v = 0.0;
  if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_volume: dims must be positive");
  const int vt = device_voxel_type(value_type);
  if (vt < 0) return fail(OVR_HIP_EINVAL, "[Optix7] unexpected volume type ..."); // array.cpp:348, same text
  if (mem_kind != OVR_HIP_MEM_HOST && mem_kind != OVR_HIP_MEM_DEVICE) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_volume: bad mem_kind");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame_one(r)) return e;
  HIP_TRY(hipDeviceSynchronize());

  hipStream_t st_ = r->own_stream[0];
  VolumeDesc vd{};
  volume_layout(vt, dims[0], dims[1], dims[2], vd);
  vd.value_scale = 1.f;
  vd.value_min_clamp = -FLT_MAX;
  if (vt == VOX_U8) vd.value_scale = 1.f / 255.f;
  if (vt == VOX_I8) { vd.value_scale = 1.f / 127.f; vd.value_min_clamp = -127.f; }
  const size_t bytes = (size_t)vd.bytes;
  if (!layout_offsets_fit(vd))
    return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_volume: one z layer of the volume exceeds 2^32 stored voxels (x * y too large for the 32-bit in-plane offsets)");

  auto t_seg = std::chrono::high_resolution_clock::now();
  join_builders(r);                               // a builder thread may still be enqueueing a replica of the OLD volume
  HIP_TRY(hipStreamSynchronize(r->build_stream)); // ... which reads the old general layout
  for (int k = 0; k < kLayouts; ++k) {
    if (r->d_replica[k]) HIP_TRY(hipFree(r->d_replica[k]));
    r->d_replica[k] = nullptr;
    if (r->d_axis[k]) HIP_TRY(hipFree(r->d_axis[k]));
    r->d_axis[k] = nullptr;
    r->vd_replica[k] = VolumeDesc{};
    r->replica_state[k] = 0;
  }
  r->d_volume = nullptr;
  r->have_volume = false; // until the new one is completely resident: a failure below must not leave a half-built volume renderable
  HIP_TRY(hipMalloc(&r->d_replica[0], bytes + 64)); // + slack: the pair load of the very last element
  // (tests: OVR_HIP_POISON_ALLOC=1 fills a fresh layout with 0xff bytes - NaN / the type's extreme - first: a padding element the build left out would show)
  if (getenv("OVR_HIP_POISON_ALLOC") && atoi(getenv("OVR_HIP_POISON_ALLOC")) != 0) HIP_TRY(hipMemsetAsync(r->d_replica[0], 0xff, bytes, st_));
  HIP_TRY(hipMemsetAsync((char*)r->d_replica[0] + bytes, 0, 64, st_)); // the slack; the relayout launches write every element of the layout itself, padding included
  r->d_volume = r->d_replica[0];
  r->volume_bytes = bytes;
  vd.data = r->d_volume;
  r->vd_replica[0] = vd;
  r->replica_state[0] = 3;
  // The other layouts (ovr_hip_kernels.h: thin replicas for views along a volume axis, the quad replica for frames that shade every sample) are
  // only PLANNED here and built in the background when a frame first asks for one (replica_state).  mode 1 (default): planned when the type has
  // them and they fit comfortably (all replicas <= 40 % of the free HBM); mode 2: all of them, built before this call returns (an allocation
  // failure is an error); mode 0: none.  OVR_HIP_QUAD=0 leaves the quad replica out (measurements).
  {
    std::lock_guard<std::mutex> lk(r->mtx);
    (void)r->layouts.update();
  }
  {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    double planned = 0.0;
    auto plan = [&](int k) { VolumeDesc t = vd; t.data = nullptr; t.axis_ab = nullptr; t.axis_z = nullptr; volume_layout(replica_voxel_type(vt, k), vd.nx, vd.ny, vd.nz, t); return t; };
    if (r->layouts.current != 0 && replica_voxel_type(vt, LAYOUT_THIN) >= 0) {
      const VolumeDesc t1 = plan(LAYOUT_THIN), t2 = plan(LAYOUT_THIN_T);
      if (layout_offsets_fit(t1) && layout_offsets_fit(t2) && (r->layouts.current == 2 || (double)(t1.bytes + t2.bytes) <= 0.4 * (double)free_b)) {
        r->vd_replica[LAYOUT_THIN] = t1; r->vd_replica[LAYOUT_THIN_T] = t2;
        r->replica_state[LAYOUT_THIN] = r->replica_state[LAYOUT_THIN_T] = 1;
        planned += (double)(t1.bytes + t2.bytes);
      }
      else if (r->layouts.current == 2) return fail(OVR_HIP_EINVAL, "[hip] volume replicas requested (layouts mode 2) but one z layer of a thin replica exceeds 2^32 elements");
    }
    static const bool want_quad = !(getenv("OVR_HIP_QUAD") && atoi(getenv("OVR_HIP_QUAD")) == 0);
    if (want_quad && r->layouts.current != 0 && replica_voxel_type(vt, LAYOUT_QUAD) >= 0) {
      const VolumeDesc tq = plan(LAYOUT_QUAD);
      if (layout_offsets_fit(tq) && (r->layouts.current == 2 || (double)tq.bytes + planned <= 0.4 * (double)free_b)) {
        r->vd_replica[LAYOUT_QUAD] = tq;
        r->replica_state[LAYOUT_QUAD] = 1;
      }
    }
  }
  HIP_TRY(hipMalloc(&r->d_axis[0], axis_table_bytes(r->vd_replica[0]))); // the layout's per-axis offset tables, staged into LDS by every march / shade workgroup
  // (allocation: a FRESH hipMalloc costs 30-60 ms per GiB on this platform - the driver maps and clears the pages - and microseconds when the runtime
  // still holds a freed block of that size: C4's 21.5 GB layout is 0.5-1.2 s in a new process, 0.1 ms on the next upload; tools/malloc_time.cpp)
  t_alloc += ms_since(t_seg);
  t_seg = std::chrono::high_resolution_clock::now();
  HIP_TRY(launch_axis_tables(r->vd_replica[0], r->d_axis[0], st_));
  vd = r->vd_replica[0];
  auto relayout_all = [&](const void* src, int z0, int nzc) -> hipError_t {
    return launch_relayout(src, value_type, r->d_replica[0], r->vd_replica[0], z0, nzc, st_);
  };

  const size_t in_es = value_type_size(value_type);
  const size_t slice_bytes = (size_t)vd.nx * vd.ny * in_es;
  hipStream_t st = r->own_stream[0];
  int src_device = r->device; // a device array may live on another GPU (a device group's followers): staged through peer copies
  if (mem_kind == OVR_HIP_MEM_DEVICE) {
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, data) == hipSuccess) src_device = attr.device;
    else (void)hipGetLastError();
  }
  if (mem_kind == OVR_HIP_MEM_DEVICE && src_device == r->device) {
    // chunk over z only to keep grid.z within limits
    for (int z0 = 0; z0 < vd.nz; z0 += 32768) {
      const int nzc = std::min(32768, vd.nz - z0);
      HIP_TRY(relayout_all((const char*)data + (size_t)z0 * slice_bytes, z0, nzc));
    }
    HIP_TRY(hipStreamSynchronize(st));
    t_kern += ms_since(t_seg);
  }
  else {
    // host input: staged through a bounded device buffer (<= 1 GiB), slab by slab
    size_t slab = std::max<size_t>(1, ((size_t)1 << 30) / std::max<size_t>(1, slice_bytes));
    slab = std::min<size_t>(slab, (size_t)vd.nz);
    void* d_stage = nullptr;
    t_seg = std::chrono::high_resolution_clock::now();
    HIP_TRY(hipMalloc(&d_stage, slab * slice_bytes));
    t_alloc += ms_since(t_seg);
    for (int z0 = 0; z0 < vd.nz; z0 += (int)slab) {
      const int nzc = (int)std::min<size_t>(slab, (size_t)(vd.nz - z0));
      t_seg = std::chrono::high_resolution_clock::now();
      hipError_t e = mem_kind == OVR_HIP_MEM_DEVICE
                         ? hipMemcpyPeerAsync(d_stage, r->device, (const char*)data + (size_t)z0 * slice_bytes, src_device, (size_t)nzc * slice_bytes, st)
                         : hipMemcpyAsync(d_stage, (const char*)data + (size_t)z0 * slice_bytes, (size_t)nzc * slice_bytes, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st); // (the slab is waited for before its re-bricking starts: the split of the upload time is exact, and
      t_copy += ms_since(t_seg);                          // the next copy could not overlap the kernel anyway - it writes the one staging buffer the kernel reads)
      t_seg = std::chrono::high_resolution_clock::now();
      if (e == hipSuccess) e = relayout_all(d_stage, z0, nzc);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      t_kern += ms_since(t_seg);
      if (e != hipSuccess) { (void)hipFree(d_stage); return fail(OVR_HIP_EDEVICE, std::string("[hip] volume upload failed: ") + hipGetErrorString(e)); }
    }
    HIP_TRY(hipFree(d_stage));
  }
  t_seg = std::chrono::high_resolution_clock::now();
  r->vd = vd;
  if (r->layouts.current == 2) { // all planned replicas now, before the call returns
    for (int k = 1; k < kLayouts; ++k) {
      if (r->replica_state[k] != 1) continue;
      std::string err;
      (void)start_replica_build(r, k, &err, false);
      if (r->replica_state[k] != 2) {
        for (int j = 1; j < kLayouts; ++j) drop_replica(r, j);
        return fail(OVR_HIP_EDEVICE, std::string("[hip] volume replicas requested (layouts mode 2) but their allocation failed: ") + err);
      }
    }
    HIP_TRY(hipStreamSynchronize(r->build_stream));
    for (int k = 1; k < kLayouts; ++k)
      if (r->replica_state[k] == 2) r->replica_state[k] = 3;
  }
  r->value_type = value_type;
  std::memcpy(r->origin, grid_origin, sizeof(r->origin));
  std::memcpy(r->spacing, grid_spacing, sizeof(r->spacing));
  r->have_volume = true;
  r->mc_ranges_valid = r->mc_majorant_valid = false;
  update_volume_params(r);
  // load_from_array3d_scalar (volume.cpp:181-191, 234-237): the macrocell value ranges and, from them, the data range the
  // reference finds with compute_scalar_range (array.cpp:27-66,297) - it is the transfer-function range until a valid one is set
  if (int e = update_macrocell_ranges(r, st)) return e;
  HIP_TRY(launch_minmax_reduce(r->d_mc_minmax, (unsigned long long)r->mc_cells, r->d_data_range, st));
  float dr[2] = { 0.f, 0.f };
  HIP_TRY(hipMemcpyAsync(dr, r->d_data_range, sizeof(dr), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  r->data_lower = dr[0];
  r->data_upper = dr[1];
  r->P.tf_lower = dr[0];
  r->P.tf_upper = dr[1];
  update_tfn_range(r);
  r->sched_dirty = true;
  r->fb_reset = true;
  r->tune_state = 0;
  r->pool_roomy = false; // a pooled frame of THIS volume has to prove the pool (ovr_hip_pack_tiles packs early only then)
  t_kern += ms_since(t_seg); // mode-2 replica builds, macrocell ranges, the data range
  r->upload_ms[0] = ms_since(t_call); r->upload_ms[1] = t_alloc; r->upload_ms[2] = t_copy; r->upload_ms[3] = t_kern;
  return 0;
}
} // namespace
extern "C" {

int ovr_hip_set_volume(ovr_hip_renderer* r, const void* data, int mem_kind, int value_type, const int32_t dims[3],
                       const float grid_origin[3], const float grid_spacing[3])
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_volume: null argument");
  if (r->members.size() <= 1) return set_volume_one(r, data, mem_kind, value_type, dims, grid_origin, grid_spacing);
  // a device group: the volume is replicated - every member uploads on its own thread, side by side (a device array is staged to the other
  // devices by peer copies).  A failure on ANY member leaves no member renderable: the members must never render different volumes
  const auto t0 = std::chrono::high_resolution_clock::now();
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  for (size_t i = 1; i < r->members.size(); ++i) {
    ovr_hip_renderer* m = r->members[i];
    m->worker->fn = [=] { return set_volume_one(m, data, mem_kind, value_type, dims, grid_origin, grid_spacing); };
  }
  const int e = group_run(r, GW_CALL, [&] { return set_volume_one(r, data, mem_kind, value_type, dims, grid_origin, grid_spacing); });
  if (e) {
    const std::string msg = g_last_error;
    for (ovr_hip_renderer* m : r->members) m->have_volume = false;
    return fail(e, msg + " (device group: the upload failed on a member - no member keeps a volume)");
  }
  r->group_upload_ms = ms_since(t0);
  return 0;
}

int ovr_hip_get_upload_times(const ovr_hip_renderer* r, double out_ms[4])
{
  if (!r || !out_ms) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_upload_times: null argument");
  for (int i = 0; i < 4; ++i) out_ms[i] = r->upload_ms[i];
  if (r->members.size() > 1) out_ms[0] = r->group_upload_ms;
  return 0;
}

int ovr_hip_query_addressing_mode(const int32_t dims[3], int value_type, int32_t layout, int32_t n_colors, int32_t n_alphas)
{
  if (!dims || dims[0] < 1 || dims[1] < 1 || dims[2] < 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_query_addressing_mode: bad dims");
  const int vt = device_voxel_type(value_type);
  if (vt < 0) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_query_addressing_mode: unknown value type");
  const int rt = replica_voxel_type(vt, layout);
  if (layout < 0 || layout >= kLayouts || rt < 0) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_query_addressing_mode: the type has no such layout");
  VolumeDesc vd{};
  volume_layout(rt, dims[0], dims[1], dims[2], vd);
  if (!layout_offsets_fit(vd)) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_query_addressing_mode: one z layer of this layout exceeds 2^32 elements - it is not built");
  return volume_addressing_mode(vd, n_colors, n_alphas);
}

int ovr_hip_set_phase_timing(ovr_hip_renderer* r, int32_t on)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  GroupLock gl(r);
  r->phase_timing.store(on != 0); // read when the next frame is launched; a frame in flight keeps what it was launched with
  GROUP_FORWARD(r, ovr_hip_set_phase_timing(m, on));
  return 0;
}

int ovr_hip_set_grid_convention(ovr_hip_renderer* r, int c)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (c != OVR_HIP_GRID_CELL_CENTRED && c != OVR_HIP_GRID_VERTEX_CENTRED) return fail(OVR_HIP_EINVAL, "[hip] unknown grid convention");
  GroupLock gl(r);
  { std::lock_guard<std::mutex> lk(r->mtx); r->grid_convention.set(c); }
  GROUP_FORWARD(r, ovr_hip_set_grid_convention(m, c));
  return 0;
}

int ovr_hip_set_transfer_function(ovr_hip_renderer* r, const float* colors, int32_t n_colors, const float* alphas, int32_t n_alphas,
                                  float lo, float hi)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (n_colors < 0 || n_alphas < 0 || (n_colors > 0 && !colors) || (n_alphas > 0 && !alphas))
    return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_transfer_function: bad arguments");
  GroupLock gl(r);
  {
    std::lock_guard<std::mutex> lk(r->mtx);
    TfnP t;
    t.colors.assign(colors, colors + (size_t)n_colors * 3);
    t.alphas.assign(alphas, alphas + (size_t)n_alphas * 2);
    t.lo = lo;
    t.hi = hi;
    r->tfn.set(t);
  }
  GROUP_FORWARD(r, ovr_hip_set_transfer_function(m, colors, n_colors, alphas, n_alphas, lo, hi));
  return 0;
}

int ovr_hip_set_camera(ovr_hip_renderer* r, const float from[3], const float at[3], const float up[3], float fovy)
{
  if (!r || !from || !at || !up) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_camera: null argument");
  GroupLock gl(r);
  {
    std::lock_guard<std::mutex> lk(r->mtx);
    CameraP c;
    std::memcpy(c.from, from, sizeof(c.from));
    std::memcpy(c.at, at, sizeof(c.at));
    std::memcpy(c.up, up, sizeof(c.up));
    c.fovy = fovy;
    r->camera.set(c);
  }
  GROUP_FORWARD(r, ovr_hip_set_camera(m, from, at, up, fovy));
  return 0;
}

int ovr_hip_set_fbsize(ovr_hip_renderer* r, int32_t w, int32_t h)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (w < 0 || h < 0) return fail(OVR_HIP_EINVAL, "[hip] negative framebuffer size");
  GroupLock gl(r);
  { std::lock_guard<std::mutex> lk(r->mtx); Size2 s; s.w = w; s.h = h; r->fbsize.set(s); }
  GROUP_FORWARD(r, ovr_hip_set_fbsize(m, w, h));
  return 0;
}

#define OVR_SIMPLE_SETTER(name, field, type, check, msg)                                                               \
  int name(ovr_hip_renderer* r, type v)                                                                                \
  {                                                                                                                    \
    if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");                                                        \
    if (!(check)) return fail(OVR_HIP_EINVAL, msg);                                                                    \
    GroupLock gl(r);                                                                                                   \
    { std::lock_guard<std::mutex> lk(r->mtx); r->field.set(v); }                                                       \
    GROUP_FORWARD(r, name(m, v));                                                                                      \
    return 0;                                                                                                          \
  }
OVR_SIMPLE_SETTER(ovr_hip_set_sample_per_pixel, spp, int32_t, v > 0, "'sample_per_pixel' should always be positive")
OVR_SIMPLE_SETTER(ovr_hip_set_volume_sampling_rate, rate, float, v > 0.f, "[hip] sampling rate must be positive")
OVR_SIMPLE_SETTER(ovr_hip_set_frame_accumulation, accumulate, int32_t, true, "")
OVR_SIMPLE_SETTER(ovr_hip_set_sparse_sampling, sparse, int32_t, true, "")
OVR_SIMPLE_SETTER(ovr_hip_set_shading, shading, int32_t, v >= 0 && v <= 2, "[hip] unknown shading mode")
OVR_SIMPLE_SETTER(ovr_hip_set_shading_pipeline, pipeline, int32_t, v >= 0 && v <= 2, "[hip] unknown shading pipeline")
OVR_SIMPLE_SETTER(ovr_hip_set_empty_space_skipping, skipping, int32_t, true, "")
OVR_SIMPLE_SETTER(ovr_hip_set_volume_layouts, layouts, int32_t, v >= 0 && v <= 2, "[hip] unknown volume-layout mode")
OVR_SIMPLE_SETTER(ovr_hip_set_layout_choice, layout_choice, int32_t, v >= -1 && v < kLayouts, "[hip] unknown layout choice")
OVR_SIMPLE_SETTER(ovr_hip_set_lds_staging, lds_staging, int32_t, v == 0 || v == 1, "[hip] unknown LDS-staging mode")
OVR_SIMPLE_SETTER(ovr_hip_set_pixel_jitter, jitter, int32_t, v == 0 || v == 1, "[hip] unknown pixel-jitter mode")

int ovr_hip_set_focus(ovr_hip_renderer* r, float cx, float cy, float scale, float base_noise)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  GroupLock gl(r);
  { std::lock_guard<std::mutex> lk(r->mtx); FocusP f; f.cx = cx; f.cy = cy; f.scale = scale; f.base_noise = base_noise; r->focus.set(f); }
  GROUP_FORWARD(r, ovr_hip_set_focus(m, cx, cy, scale, base_noise));
  return 0;
}

int ovr_hip_set_noise_tile(ovr_hip_renderer* r, const float* tile, int32_t xy)
{
  if (!r || !tile || xy <= 0) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_noise_tile: bad arguments");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  HIP_TRY(hipDeviceSynchronize());
  if (r->d_noise) HIP_TRY(hipFree(r->d_noise));
  if (xy > 128) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_noise_tile: tiles larger than 128 x 128 are not supported");
  const size_t bytes = (size_t)xy * xy * 64 * sizeof(float);
  // stored transposed, [t][y][x], so that one frame's slice is contiguous (the reference's layout is [y][x][t])
  std::vector<float> tr((size_t)xy * xy * 64);
  for (int y = 0; y < xy; ++y)
    for (int x = 0; x < xy; ++x)
      for (int t = 0; t < 64; ++t) tr[((size_t)t * xy + y) * xy + x] = tile[((size_t)y * xy + x) * 64 + t];
  HIP_TRY(hipMalloc((void**)&r->d_noise, bytes));
  HIP_TRY(hipMemcpy(r->d_noise, tr.data(), bytes, hipMemcpyHostToDevice));
  r->noise_xy = xy;
  r->fb_reset = true;
  if (r->members.size() > 1)
    if (int e = group_call(r, [=](ovr_hip_renderer* m) { return ovr_hip_set_noise_tile(m, tile, xy); })) return e;
  return 0;
}

int ovr_hip_set_image_shard(ovr_hip_renderer* r, int32_t rank, int32_t world, int32_t tw, int32_t th)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (world < 1 || rank < 0 || rank >= world || tw < 1 || th < 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_image_shard: bad arguments");
  GroupLock gl(r);
  if (r->members.size() > 1) { // a device group shards the image among its members itself: only the tile size is the caller's
    if (world != 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_set_image_shard: a device group cannot be one rank of a larger shard (use world = 1 to set its tile size)");
    for (size_t i = 0; i < r->members.size(); ++i) {
      ovr_hip_renderer* m = r->members[i];
      std::lock_guard<std::mutex> lk(m->mtx);
      ShardP s; s.rank = (int)i; s.world = (int)r->members.size(); s.tw = tw; s.th = th;
      m->shard.set(s);
    }
    return 0;
  }
  std::lock_guard<std::mutex> lk(r->mtx);
  ShardP s; s.rank = rank; s.world = world; s.tw = tw; s.th = th;
  r->shard.set(s);
  return 0;
}

} // extern "C"
namespace {
int commit_one(ovr_hip_renderer* r);
// what every member of a group must agree on after a commit (each applied its own copy of the queued values)
bool same_committed_state(const ovr_hip_renderer* a, const ovr_hip_renderer* b)
{
  return a->fbsize.current.w == b->fbsize.current.w && a->fbsize.current.h == b->fbsize.current.h && std::memcmp(&a->camera.current, &b->camera.current, sizeof(CameraP)) == 0
         && a->spp.current == b->spp.current && a->sparse.current == b->sparse.current && a->accumulate.current == b->accumulate.current && a->shading.current == b->shading.current
         && a->rate.current == b->rate.current && a->jitter.current == b->jitter.current && a->grid_convention.current == b->grid_convention.current
         && a->tfn.current.lo == b->tfn.current.lo && a->tfn.current.hi == b->tfn.current.hi && a->tfn.current.colors == b->tfn.current.colors && a->tfn.current.alphas == b->tfn.current.alphas
         && std::memcmp(&a->focus.current, &b->focus.current, sizeof(FocusP)) == 0 && a->shard.current.world == b->shard.current.world && a->shard.current.tw == b->shard.current.tw
         && a->shard.current.th == b->shard.current.th && a->have_tfn == b->have_tfn;
}
} // namespace
extern "C" {

int ovr_hip_commit(ovr_hip_renderer* r)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  if (r->members.size() <= 1) return commit_one(r);
  GroupLock gl(r); // every member applies the same queued values: no setter call of another thread is split between them
  // every member commits on its own thread, the leader on this one.  A failure on any member (a framebuffer that does not fit ...) leaves the members
  // on different states: the group then refuses to render until a later commit has succeeded everywhere and the members agree again
  const int e = group_run(r, GW_COMMIT, [&] { return commit_one(r); });
  bool same = true;
  for (size_t i = 1; i < r->members.size() && same; ++i) same = same_committed_state(r, r->members[i]);
  if (e || !same) {
    const std::string msg = e ? g_last_error : std::string("[hip] the members of the device group disagree on the committed state");
    r->group_broken = true;
    r->group_broken_why = msg;
    return fail(e ? e : OVR_HIP_ESTATE, msg);
  }
  r->group_broken = false;
  return 0;
}

} // extern "C"
namespace {
int commit_one(ovr_hip_renderer* r)
{
  std::lock_guard<std::mutex> lk(r->mtx);
  // test hook (tests/test_round5_gpu.py): OVR_HIP_TEST_FAIL_MEMBER=k makes member k of a device group fail the commit of a framebuffer 4242 pixels wide -
  // the way a member that cannot allocate its framebuffer would - so that the group's refusal to render in mixed state can be exercised on one card
  if (r->fbsize.dirty && r->fbsize.queued.w == 4242)
    if (const char* t = getenv("OVR_HIP_TEST_FAIL_MEMBER"))
      if ((r->leader ? r->group_rank : 0) == atoi(t) && (r->leader || r->members.size() > 1)) {
        r->fbsize.dirty = false; // (the value is consumed, like a real failure half-way through)
        return fail(OVR_HIP_EDEVICE, "[hip] commit failed on member " + std::string(t) + " (OVR_HIP_TEST_FAIL_MEMBER)");
      }
  const bool reset_pending = r->fb_reset; // (without accumulation the flag is never consumed)
  r->fb_reset = false;
  bool fb_size_updated = false, camera_changed = false;
  if (r->fbsize.update()) { // device_impl.cpp:116-122
    if (int e = resize_framebuffers(r, r->fbsize.current.w, r->fbsize.current.h)) return e;
    fb_size_updated = true;
    r->fb_reset = true;
  }
  if (r->camera.update() || fb_size_updated || r->camera_dirty) { // :125-144
    if (r->fbsize.current.w > 0 && r->fbsize.current.h > 0) {
      update_camera(r);
      r->camera_dirty = false;
    }
    r->sched_dirty = true;
    r->fb_reset = true;
    camera_changed = !fb_size_updated;
  }
  const bool only_camera_so_far = camera_changed;
  if (camera_changed) r->fb_reset = false; // (restored below: the flag doubles as "something besides the camera changed")
  if (r->tfn.update()) { // :146-153
    if (int e = upload_tfn(r)) return e;
    update_tfn_range(r);
    r->mc_majorant_valid = false;
    r->fb_reset = true;
  }
  if (r->grid_convention.update()) {
    if (r->have_volume) update_volume_params(r);
    r->sched_dirty = true;
    r->fb_reset = true;
  }
  if (r->focus.update()) r->fb_reset = true;      // :155-168
  if (r->spp.update()) { r->fb_reset = true; r->sched_dirty = true; }        // :170-173 (one sample per pixel: the schedule knows every ray)
  if (r->sparse.update()) r->fb_reset = true;     // :180-183
  if (r->accumulate.update()) r->fb_reset = true; // :185-188
  if (r->rate.update()) r->fb_reset = true;       // :190-196
  if (r->shading.update()) r->fb_reset = true;
  if (r->jitter.update()) { r->fb_reset = true; r->sched_dirty = true; }
  (void)r->lds_staging.update(); // same frame either way
  // every layout and both pipelines give the same frame: no accumulation reset - but what was measured under the old setting is void
  // (a probe must not override a layout forced meanwhile; forced -> automatic has to measure again)
  bool tune_void = false;
  { const int before = r->layout_choice.current; if (r->layout_choice.update() && r->layout_choice.current != before) tune_void = true; }
  { const int before = r->pipeline.current; if (r->pipeline.update() && r->pipeline.current != before) tune_void = true; }
  if (r->skipping.update()) { r->skip_active = true; r->skip_backoff = 32; } // skipping does not change the frame either
  if (r->shard.update()) {
    r->sched_list_dirty = true;
    r->fb_reset = true;
  }
  const bool other_changed = r->fb_reset;
  if (other_changed || only_camera_so_far) {
    r->clear_gen++;        // ... and so is what the pixels of the blocks that are not launched hold
    r->pool_roomy = false; // something changed: the next frame's request count is unknown
    const bool keep = r->tune_on && !other_changed && r->tune_state == 2 && (r->tune_layout >= 0 || r->tune_pipeline != 0);
    if (keep) r->tune_recheck = 12; // only the camera moved: the measured layout / pipeline stay (see tune_recheck)
    else { r->tune_state = 0; r->tune_recheck = 0; } // ... and so are the fastest layout and pipeline
  }
  if (tune_void) { r->tune_state = 0; r->tune_recheck = 0; }
  r->fb_reset = other_changed || only_camera_so_far || reset_pending;
  return 0;
}
} // namespace
extern "C" {

int ovr_hip_render_async(ovr_hip_renderer* r)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  if (r->members.size() <= 1) return enqueue_frame(r);
  if (r->group_broken) return fail(OVR_HIP_ESTATE, "[hip] the device group is not renderable since a member failed: " + r->group_broken_why);
  // every device starts its tiles - each member's frame is enqueued by its own thread, the leader's by this one; the gather happens when the
  // frame is resolved (group_finish)
  const auto t0 = std::chrono::high_resolution_clock::now();
  const int e = group_run(r, GW_RENDER, [&] { return enqueue_frame(r); });
  r->group_host_us[0] = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  return e;
}

int ovr_hip_sync(ovr_hip_renderer* r)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  HIP_TRY(hipStreamSynchronize(r->stream())); // also covers pack / unpack / mask work enqueued after the frame
  return 0;
}

int ovr_hip_render(ovr_hip_renderer* r)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  const auto t0 = std::chrono::high_resolution_clock::now(); // optix7/device.cpp:37-42
  if (int e = ovr_hip_render_async(r)) return e;
  const auto tm = std::chrono::high_resolution_clock::now();
  if (int e = finish_frame(r)) return e;
  const auto t1 = std::chrono::high_resolution_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  static const bool slow_trace = getenv("OVR_HIP_SLOW_FRAMES") != nullptr; // diagnostic: where a frame that took > 20 ms of host time spent it
  if (slow_trace && ms > 20.0)
    fprintf(stderr, "[hip] slow frame %d: enqueue %.2f ms, wait + finish %.2f ms (kernels %.2f ms)\n", r->frame_index, std::chrono::duration<double, std::milli>(tm - t0).count(),
            std::chrono::duration<double, std::milli>(t1 - tm).count(), r->stats.kernel_ms);
  // diagnostic: OVR_HIP_FRAME_TIMING=n prints, every n frames, the mean host time of a blocking render() split into enqueue / wait, next to the
  // frame's device time between its first and last event - the difference is launch latency, the counter read-back and the wake-up
  static const int timing_every = getenv("OVR_HIP_FRAME_TIMING") ? atoi(getenv("OVR_HIP_FRAME_TIMING")) : 0;
  if (timing_every > 0) {
    static thread_local double acc[3] = { 0, 0, 0 };
    static thread_local int n_acc = 0;
    acc[0] += std::chrono::duration<double, std::milli>(tm - t0).count(); acc[1] += std::chrono::duration<double, std::milli>(t1 - tm).count(); acc[2] += r->stats.kernel_ms;
    if (++n_acc == timing_every) {
      fprintf(stderr, "[hip] %d frames: render() %.4f ms = enqueue %.4f + wait %.4f ; device first-to-last event %.4f ms ; outside the events %.4f ms\n", n_acc,
              (acc[0] + acc[1]) / n_acc, acc[0] / n_acc, acc[1] / n_acc, acc[2] / n_acc, (acc[0] + acc[1] - acc[2]) / n_acc);
      acc[0] = acc[1] = acc[2] = 0; n_acc = 0;
    }
  }
  r->stats.render_ms = ms;
  r->render_time_ms += ms;
  return 0;
}

int ovr_hip_mapframe(ovr_hip_renderer* r, int mem_kind, const float** rgba, size_t* rgba_bytes, const float** grad, size_t* grad_bytes)
{
  if (!r || !rgba || !rgba_bytes) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe: null argument");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  const size_t n = r->fb_pixels;
  const int c = r->cur;
  if (mem_kind == OVR_HIP_MEM_DEVICE) {
    *rgba = r->d_rgba[c];
    *rgba_bytes = n * 4 * sizeof(float);
    if (grad) *grad = r->d_grad[c];
    if (grad_bytes) *grad_bytes = n * 3 * sizeof(float);
    return 0;
  }
  if (mem_kind != OVR_HIP_MEM_HOST) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe: bad mem_kind");
  if (n == 0) { *rgba = nullptr; *rgba_bytes = 0; if (grad) *grad = nullptr; if (grad_bytes) *grad_bytes = 0; return 0; }
  hipStream_t st = r->stream();
  // the mirrors start as zeros and only the rectangle the box projects into is ever copied (refresh_mirror): C3's oblique view moves 22 of
  // the frame's 58 MB, 0.4 instead of 1.04 ms per mapped frame (profiles/r03_notes.md)
  if (!r->h_rgba[c]) {
    HIP_TRY(hipHostMalloc((void**)&r->h_rgba[c], n * 4 * sizeof(float), hipHostMallocDefault));
    std::memset(r->h_rgba[c], 0, n * 4 * sizeof(float));
    for (int k = 0; k < 4; ++k) r->h_rgba_rect[c][k] = 0;
  }
  if (int e = refresh_mirror(r, r->h_rgba[c], r->d_rgba[c], 4, r->d_rect[c], r->h_rgba_rect[c], st)) return e;
  if (grad) {
    if (!r->h_grad[c]) {
      HIP_TRY(hipHostMalloc((void**)&r->h_grad[c], n * 3 * sizeof(float), hipHostMallocDefault));
      std::memset(r->h_grad[c], 0, n * 3 * sizeof(float));
      for (int k = 0; k < 4; ++k) r->h_grad_rect[c][k] = 0;
    }
    if (int e = refresh_mirror(r, r->h_grad[c], r->d_grad[c], 3, r->d_rect[c], r->h_grad_rect[c], st)) return e;
  }
  HIP_TRY(hipStreamSynchronize(st));
  *rgba = r->h_rgba[c];
  *rgba_bytes = n * 4 * sizeof(float);
  if (grad) *grad = r->h_grad[c];
  if (grad_bytes) *grad_bytes = n * 3 * sizeof(float);
  return 0;
}

int ovr_hip_mapframe_rgba8(ovr_hip_renderer* r, int mem_kind, int flip_vertical, const uint32_t** rgba8, size_t* bytes)
{
  if (!r || !rgba8 || !bytes) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe_rgba8: null argument");
  if (mem_kind != OVR_HIP_MEM_DEVICE && mem_kind != OVR_HIP_MEM_HOST) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe_rgba8: bad mem_kind");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  const size_t n = r->fb_pixels;
  *rgba8 = nullptr;
  *bytes = 0;
  if (n == 0) return 0;
  hipStream_t st = r->stream();
  if (!r->d_rgba8) HIP_TRY(hipMalloc((void**)&r->d_rgba8, n * sizeof(uint32_t)));
  HIP_TRY(launch_rgba8(r->d_rgba[r->cur], r->d_rgba8, r->fbsize.current.w, r->fbsize.current.h, flip_vertical ? 1 : 0, st));
  if (mem_kind == OVR_HIP_MEM_HOST) {
    if (!r->h_rgba8) HIP_TRY(hipHostMalloc((void**)&r->h_rgba8, n * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(r->h_rgba8, r->d_rgba8, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  *rgba8 = mem_kind == OVR_HIP_MEM_HOST ? r->h_rgba8 : r->d_rgba8;
  *bytes = n * sizeof(uint32_t);
  return 0;
}

int ovr_hip_mapframe_rgba16f(ovr_hip_renderer* r, int mem_kind, int flip_vertical, const uint16_t** rgba16f, size_t* bytes)
{
  if (!r || !rgba16f || !bytes) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe_rgba16f: null argument");
  if (mem_kind != OVR_HIP_MEM_DEVICE && mem_kind != OVR_HIP_MEM_HOST) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_mapframe_rgba16f: bad mem_kind");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  const size_t n = r->fb_pixels;
  *rgba16f = nullptr;
  *bytes = 0;
  if (n == 0) return 0;
  hipStream_t st = r->stream();
  if (!r->d_rgba16f) HIP_TRY(hipMalloc((void**)&r->d_rgba16f, n * 4 * sizeof(uint16_t)));
  HIP_TRY(launch_rgba16f(r->d_rgba[r->cur], r->d_rgba16f, r->fbsize.current.w, r->fbsize.current.h, flip_vertical ? 1 : 0, st));
  if (mem_kind == OVR_HIP_MEM_HOST) {
    if (!r->h_rgba16f) HIP_TRY(hipHostMalloc((void**)&r->h_rgba16f, n * 4 * sizeof(uint16_t), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(r->h_rgba16f, r->d_rgba16f, n * 4 * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  *rgba16f = mem_kind == OVR_HIP_MEM_HOST ? r->h_rgba16f : r->d_rgba16f;
  *bytes = n * 4 * sizeof(uint16_t);
  return 0;
}

int ovr_hip_get_volume_info(const ovr_hip_renderer* r, ovr_hip_volume_info* out)
{
  if (!r || !out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_volume_info: null argument");
  if (!r->have_volume) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_get_volume_info: no volume was set");
  out->dims[0] = r->vd.nx; out->dims[1] = r->vd.ny; out->dims[2] = r->vd.nz;
  out->value_type = r->value_type;
  out->resident_bytes = (uint64_t)r->volume_bytes;
  out->data_lower = r->data_lower;
  out->data_upper = r->data_upper;
  out->tf_lower = r->P.tf_lower;
  out->tf_upper = r->P.tf_upper;
  return 0;
}

int ovr_hip_swap(ovr_hip_renderer* r)
{
  if (!r) return fail(OVR_HIP_EINVAL, "[hip] null renderer");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  HIP_TRY(hipStreamSynchronize(r->stream())); // device_impl.cpp:105
  r->cur = (r->cur + 1) % 2;                  // safe_swap, optix7_common.h:366-370
  if (r->members.size() > 1) return group_run(r, GW_SWAP);
  return 0;
}

double ovr_hip_render_time_ms(const ovr_hip_renderer* r) { return r ? r->render_time_ms : 0.0; }

int ovr_hip_get_stats(const ovr_hip_renderer* r, ovr_hip_stats* out)
{
  if (!r || !out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_stats: null argument");
  if (r->async_pending) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_get_stats: a frame is still in flight (call ovr_hip_sync)");
  *out = r->stats;
  return 0;
}

int ovr_hip_get_macrocells(ovr_hip_renderer* r, int32_t dims[3], float* minmax_host, float* majorant_host, size_t capacity_cells)
{
  if (!r || !dims) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_macrocells: null argument");
  if (!r->have_volume || !r->have_tfn) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_get_macrocells: needs a volume and a transfer function");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  dims[0] = (r->vd.nx + 15) / 16; dims[1] = (r->vd.ny + 15) / 16; dims[2] = (r->vd.nz + 15) / 16;
  const size_t cells = (size_t)dims[0] * dims[1] * dims[2];
  if (!minmax_host && !majorant_host) return 0;
  if (capacity_cells < cells) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_get_macrocells: output too small");
  hipStream_t st = r->stream();
  if (int e = update_macrocells(r, st)) return e;
  HIP_TRY(hipStreamSynchronize(st));
  if (minmax_host) HIP_TRY(hipMemcpy(minmax_host, r->d_mc_minmax, cells * 2 * sizeof(float), hipMemcpyDeviceToHost));
  if (majorant_host) HIP_TRY(hipMemcpy(majorant_host, r->d_mc_majorant, cells * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ovr_hip_owned_tiles(const ovr_hip_renderer* r, int32_t rank, int32_t* n_tiles)
{
  if (!r || !n_tiles) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_owned_tiles: null argument");
  const ShardP& s = r->shard.current;
  if (rank < 0 || rank >= s.world) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_owned_tiles: rank out of range");
  *n_tiles = count_owned_tiles(r->fbsize.current.w, r->fbsize.current.h, s.tw, s.th, rank, s.world);
  return 0;
}

int ovr_hip_pack_tiles(ovr_hip_renderer* r, float* dst, size_t dst_bytes)
{
  if (!r || !dst) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_pack_tiles: null argument");
  if (r->members.size() > 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_pack_tiles: a device group gathers its tiles itself");
  if (int e = set_device(r)) return e;
  // A frame of the pooled pipeline may still have to be rendered again (request-pool overflow: nothing was written to the
  // framebuffer and only the host can grow the pool).  Until the pool has proven roomy for this configuration (the last frame
  // used at most half of it) the frame is resolved before its tiles are packed; after that the pack is enqueued right behind
  // the frame, and should the improbable happen - an overflow after all - ovr_hip_stats.stale_tiles tells the caller.
  if (r->async_pending && r->P.pool.reqs) {
    if (!r->pool_roomy) { if (int e = finish_frame(r)) return e; }
    else r->packed_early = true; // steady state: the pack runs right behind the frame, nothing waits on the host
  }
  const ShardP& s = r->shard.current;
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  const size_t need = (size_t)count_owned_tiles(W, H, s.tw, s.th, s.rank, s.world) * s.tw * s.th * 4 * sizeof(float);
  if (dst_bytes < need) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_pack_tiles: destination too small");
  HIP_TRY(launch_pack_tiles(r->d_rgba[r->cur], dst, W, H, s.tw, s.th, s.rank, s.world, r->stream()));
  return 0;
}

int ovr_hip_unpack_tiles(ovr_hip_renderer* r, int32_t src_rank, const float* src, size_t src_bytes, float* frame, size_t frame_bytes)
{
  if (!r || !src || !frame) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_tiles: null argument");
  if (int e = set_device(r)) return e;
  const ShardP& s = r->shard.current;
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  if (src_rank < 0 || src_rank >= s.world) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_tiles: rank out of range");
  const size_t need = (size_t)count_owned_tiles(W, H, s.tw, s.th, src_rank, s.world) * s.tw * s.th * 4 * sizeof(float);
  if (src_bytes < need) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_tiles: source too small");
  if (frame_bytes < (size_t)W * H * 4 * sizeof(float)) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_tiles: frame too small");
  HIP_TRY(launch_unpack_tiles(src, frame, W, H, s.tw, s.th, src_rank, s.world, 0, r->stream()));
  return 0;
}

int ovr_hip_unpack_all_tiles(ovr_hip_renderer* r, const float* src, size_t rank_stride_bytes, size_t src_bytes, float* frame, size_t frame_bytes)
{
  if (!r || !src || !frame) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_all_tiles: null argument");
  if (int e = set_device(r)) return e;
  const ShardP& s = r->shard.current;
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  size_t need = 0;
  for (int k = 0; k < s.world; ++k)
    need = std::max(need, (size_t)count_owned_tiles(W, H, s.tw, s.th, k, s.world) * s.tw * s.th * 4 * sizeof(float));
  if (rank_stride_bytes % 16 != 0 || rank_stride_bytes < need) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_all_tiles: rank stride too small or not a multiple of 16");
  if (src_bytes < rank_stride_bytes * (size_t)(s.world - 1) + need) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_all_tiles: source too small");
  if (frame_bytes < (size_t)W * H * 4 * sizeof(float)) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_unpack_all_tiles: frame too small");
  HIP_TRY(launch_unpack_tiles(src, frame, W, H, s.tw, s.th, -1, s.world, rank_stride_bytes / sizeof(float), r->stream()));
  return 0;
}

int ovr_hip_sparse_mask(ovr_hip_renderer* r, int32_t frame_index, int32_t* out_xy, size_t out_bytes, int64_t* n_out)
{
  if (!r || !out_xy || !n_out) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_sparse_mask: null argument");
  if (int e = set_device(r)) return e;
  if (int e = finish_frame(r)) return e;
  if (!r->d_noise) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_sparse_mask: no noise tile was set");
  const int W = r->fbsize.current.w, H = r->fbsize.current.h;
  if (W <= 0 || H <= 0) return fail(OVR_HIP_ESTATE, "[hip] ovr_hip_sparse_mask: framebuffer size not committed");
  if (out_bytes < (size_t)W * H * 2 * sizeof(int32_t)) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_sparse_mask: output too small");
  if (int e = ensure_sparse_buffers(r)) return e;
  hipStream_t st = r->stream();
  HIP_TRY(launch_sparse_mask(make_mask_params(r, frame_index, out_xy), st));
  unsigned long long cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, r->d_sparse_count, sizeof(cnt), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *n_out = (int64_t)cnt;
  return 0;
}

int ovr_hip_pow_floats(ovr_hip_renderer* r, const float* x, const float* y, float* out, int64_t n, int32_t which)
{
  if (!r || !x || !y || !out || n < 0 || which < 0 || which > 1) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_pow_floats: bad arguments");
  if (int e = set_device(r)) return e;
  HIP_TRY(launch_pow(x, y, out, n, which, r->stream()));
  HIP_TRY(hipStreamSynchronize(r->stream()));
  return 0;
}

int ovr_hip_built_for_exact_parity(void) { return built_for_exact_parity(); }

int ovr_hip_tea_floats(ovr_hip_renderer* r, uint32_t* v0v1, float* out, int64_t n)
{
  if (!r || !v0v1 || !out || n < 0) return fail(OVR_HIP_EINVAL, "[hip] ovr_hip_tea_floats: bad arguments");
  if (int e = set_device(r)) return e;
  HIP_TRY(launch_tea(v0v1, out, n, r->stream()));
  HIP_TRY(hipStreamSynchronize(r->stream()));
  return 0;
}

} // extern "C"
