// device_hip.cpp - the reference-side binding of the MI355X backend: `DeviceHIP : ovr::MainRenderer`.
//
// Compiled AGAINST THE REFERENCE'S OWN HEADERS (-I$OVR_ROOT ...; nothing of the reference is copied here) into
// libdevice_hip.so, which exports the one symbol the reference's factory looks up for `--device hip`:
//     extern "C" ovr::MainRenderer* ovr_create_renderer__hip();
// (create_renderer -> LibraryRepository::add("device_hip") -> dlopen("libdevice_hip.so") -> objectFactory<MainRenderer>,
//  reference ovr/renderer.cpp:55-58, ovr/common/dylink/ObjectFactory.h:35-85).  The reference's apps (renderbatch,
// renderapp) run unmodified.  Everything below only forwards to the C ABI of include/ovr_hip.h - it plays the role
// ovr/devices/optix7/device.cpp + device_impl.cpp play for the OptiX device.
#include <ovr/renderer.h>

#include "../include/ovr_hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

void check(int code)
{
  if (code != 0) throw std::runtime_error(ovr_hip_last_error()); // the reference reports device errors as std::runtime_error
}

// The reference links its noise tile into the CUDA device's translation unit (ovr/CMakeLists.txt:67-72, blue_noise.h:74-79),
// out of a plugin's reach.  The same files ship in the reference's data/noise/: the tile is read from $OVR_HIP_NOISE_TILE or,
// failing that, from the names the reference's own (commented-out) file loader used, next to the executable.
// Layout [y][x][t] float32, t = 64 (blue_noise.h:95-99); xy follows from the file size.
bool load_noise_tile(ovr_hip_renderer* h)
{
  std::vector<std::string> candidates;
  if (const char* e = std::getenv("OVR_HIP_NOISE_TILE")) candidates.push_back(e);
  for (const char* name : { "stbn_128x128x64.bin", "blue_64x64x64.bin" }) { // STBN is the reference's default (generate_mask.h:9)
    candidates.push_back(std::string("./") + name);
    candidates.push_back(std::string("./data/noise/") + name);
    if (const char* root = std::getenv("OVR_ROOT")) candidates.push_back(std::string(root) + "/data/noise/" + name);
  }
  for (const std::string& path : candidates) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) continue;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const long xy = bytes > 0 ? std::lround(std::sqrt((double)bytes / (64.0 * sizeof(float)))) : 0;
    std::vector<float> tile;
    bool ok = xy > 0 && (long)(xy * xy * 64 * sizeof(float)) == bytes;
    if (ok) {
      tile.resize((size_t)xy * xy * 64);
      ok = std::fread(tile.data(), 1, (size_t)bytes, f) == (size_t)bytes;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("[hip] " + path + " is not an xy*xy*64 float32 noise tile");
    check(ovr_hip_set_noise_tile(h, tile.data(), (int32_t)xy));
    return true;
  }
  return false;
}

class DeviceHIP : public ovr::MainRenderer {
public:
  DeviceHIP() = default;
  ~DeviceHIP() override { ovr_hip_destroy(h); }
  DeviceHIP(const DeviceHIP&) = delete;
  DeviceHIP& operator=(const DeviceHIP&) = delete;

  // DeviceOptix7::init -> Impl::init -> buildScene (optix7/device.cpp:16-20, device_impl.cpp:283-302)
  void init(int argc, const char** argv) override
  {
    if (h) throw std::runtime_error("[hip] device already initialized!");
    // One GPU (`--hip-device N`, default 0: the reference hard-codes device 0, device_impl.cpp:371-372) or several behind this one
    // MainRenderer: `--hip-devices 0,1,2,...` or OVR_HIP_DEVICES=0,1,2,... (the unmodified apps pass no such flag: the environment
    // variable is their switch).  The group shards the image plane into tiles over the listed devices and gathers them on the first
    // (ovr_hip_create_group); everything below is the same for both.
    std::string list = std::getenv("OVR_HIP_DEVICES") ? std::getenv("OVR_HIP_DEVICES") : "";
    int device_id = 0;
    bool no_skip_flag = false;
    for (int i = 1; i < argc; ++i) {
      if (std::string(argv[i]) == "--hip-no-skip") no_skip_flag = true; // = OVR_HIP_SKIP_EMPTY=0 (a host that forwards its arguments; the reference's
      if (i + 1 >= argc) continue;                                      // own apps reject flags they do not know - args::ParseError - and use the variable)
      if (std::string(argv[i]) == "--hip-device") device_id = std::stoi(argv[i + 1]);
      if (std::string(argv[i]) == "--hip-devices") list = argv[i + 1];
    }
    std::vector<int32_t> devices;
    for (size_t at = 0; at < list.size();) {
      const size_t comma = list.find(',', at);
      const std::string tok = list.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
      if (!tok.empty()) devices.push_back((int32_t)std::stoi(tok));
      if (comma == std::string::npos) break;
      at = comma + 1;
    }
    if (devices.empty()) check(ovr_hip_create(&h, device_id));
    else check(ovr_hip_create_group(&h, devices.data(), (int32_t)devices.size()));
    const auto& v = ovr::parse_single_volume_scene(current_scene, ovr::scene::Volume::STRUCTURED_REGULAR_VOLUME).structured_regular;
    const int32_t dims[3] = { v.data->dims.x, v.data->dims.y, v.data->dims.z };
    const float origin[3] = { v.grid_origin.x, v.grid_origin.y, v.grid_origin.z };
    const float spacing[3] = { v.grid_spacing.x, v.grid_spacing.y, v.grid_spacing.z };
    check(ovr_hip_set_volume(h, v.data->data(), OVR_HIP_MEM_HOST, (int)v.data->type, dims, origin, spacing));
    check(ovr_hip_set_volume_sampling_rate(h, current_scene.volume_sampling_rate)); // device_impl.cpp:298
    check(ovr_hip_set_shading(h, OVR_HIP_SHADE_FULL));                             // what the reference's marcher does
    // The reference builds its macrocell grids for every volume / transfer function (accel/sp_singlemc.cu) but only its path
    // tracer walks them; here the ray marcher skips empty space with them - the frames are bit-identical, so the drop-in
    // device has it on (OVR_HIP_SKIP_EMPTY=0 switches it off, e.g. to count every sample like the reference)
    // (any fps quoted through this device must say which: with skipping the march fetches a fraction of the reference's samples - C3: 889 vs 373 fps)
    const char* skip = std::getenv("OVR_HIP_SKIP_EMPTY");
    const bool skipping = !(no_skip_flag || (skip && skip[0] == '0'));
    check(ovr_hip_set_empty_space_skipping(h, skipping ? 1 : 0));
    if (!(std::getenv("OVR_HIP_QUIET") && std::getenv("OVR_HIP_QUIET")[0] != '0'))
      std::fprintf(stderr, "[hip] empty-space skipping %s\n", skipping ? "ON (bit-identical frames, fewer samples fetched than the reference's marcher; OVR_HIP_SKIP_EMPTY=0 / --hip-no-skip: the reference's exact work)"
                                                                      : "OFF (every sample fetched, like the reference's marcher)");
    // nothing behind this interface reads per-phase device times: no events between the frame's kernels (OVR_HIP_PHASE_TIMING=1 keeps them)
    const char* phases = std::getenv("OVR_HIP_PHASE_TIMING");
    check(ovr_hip_set_phase_timing(h, (phases && phases[0] == '1') ? 1 : 0));
    commit();
  }

  void swap() override { check(ovr_hip_swap(h)); }

  // DeviceOptix7::Impl::commit (device_impl.cpp:113-197): forward every changed TransactionalValue
  void commit() override
  {
    if (params.fbsize.update()) check(ovr_hip_set_fbsize(h, params.fbsize.ref().x, params.fbsize.ref().y));
    if (params.camera.update()) {
      const ovr::scene::Camera& c = params.camera.ref();
      const float from[3] = { c.from.x, c.from.y, c.from.z }, at[3] = { c.at.x, c.at.y, c.at.z }, up[3] = { c.up.x, c.up.y, c.up.z };
      check(ovr_hip_set_camera(h, from, at, up, c.perspective.fovy));
    }
    if (params.tfn.update()) {
      const auto& t = params.tfn.ref();
      check(ovr_hip_set_transfer_function(h, t.tfn_colors.data(), (int32_t)(t.tfn_colors.size() / 3), t.tfn_alphas.data(),
                                          (int32_t)(t.tfn_alphas.size() / 2), t.tfn_value_range.x, t.tfn_value_range.y));
    }
    const bool focus = params.focus_center.update() | params.focus_scale.update() | params.base_noise.update();
    if (focus)
      check(ovr_hip_set_focus(h, params.focus_center.ref().x, params.focus_center.ref().y, params.focus_scale.ref(), params.base_noise.ref()));
    if (params.sample_per_pixel.update()) check(ovr_hip_set_sample_per_pixel(h, params.sample_per_pixel.ref()));
    if (params.path_tracing.update() && params.path_tracing.ref())
      throw std::runtime_error("[hip] path tracing is not part of the ray-marching backend");
    if (params.sparse_sampling.update()) {
      if (params.sparse_sampling.ref() && !have_noise) {
        if (!load_noise_tile(h))
          throw std::runtime_error("[hip] sparse sampling needs the reference's noise tile: set OVR_HIP_NOISE_TILE to data/noise/stbn_128x128x64.bin "
                                   "(or blue_64x64x64.bin), or run next to it");
        have_noise = true;
      }
      check(ovr_hip_set_sparse_sampling(h, params.sparse_sampling.ref()));
    }
    if (params.frame_accumulation.update()) check(ovr_hip_set_frame_accumulation(h, params.frame_accumulation.ref()));
    if (params.volume_sampling_rate.update()) check(ovr_hip_set_volume_sampling_rate(h, params.volume_sampling_rate.get()));
    check(ovr_hip_commit(h));
  }

  // DeviceOptix7::render (optix7/device.cpp:35-43): blocking; elapsed milliseconds are added to render_time
  void render() override
  {
    const auto start = std::chrono::high_resolution_clock::now();
    check(ovr_hip_render(h));
    const auto end = std::chrono::high_resolution_clock::now();
    render_time += std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
    variance = 0.f; // device_impl.cpp:266
  }

  // Impl::mapframe (device_impl.cpp:271-281).  The reference hands out device pointers (DEVICE_CUDA), which only exist
  // in its CUDA build; an app built without OVR_BUILD_CUDA_DEVICES knows DEVICE_CPU only, so the frame is mapped to host
  // memory owned by the backend - what the caller's to_cpu() (cross_device_buffer.h:130-159) would do next anyway.
  // Only the rectangle the volume's box projects into crosses PCIe (ovr_hip_mapframe).  OVR_HIP_MAP_GRAD=0 leaves the gradient layer unset
  // like the reference's OSPRay device does (ospray/device_impl.cpp:814-818) - 43 % of a mapped frame's bytes, read only by renderapp's
  // "gradient" view.
  void mapframe(FrameBufferData* fb) override
  {
    static const bool want_grad = !(std::getenv("OVR_HIP_MAP_GRAD") && std::getenv("OVR_HIP_MAP_GRAD")[0] == '0');
    const float *rgba = nullptr, *grad = nullptr;
    size_t nb_rgba = 0, nb_grad = 0;
    check(ovr_hip_mapframe(h, OVR_HIP_MEM_HOST, &rgba, &nb_rgba, want_grad ? &grad : nullptr, want_grad ? &nb_grad : nullptr));
    fb->rgba->set_data((void*)rgba, nb_rgba, ovr::CrossDeviceBuffer::DEVICE_CPU);
    if (want_grad) fb->grad->set_data((void*)grad, nb_grad, ovr::CrossDeviceBuffer::DEVICE_CPU);
  }

private:
  ovr_hip_renderer* h = nullptr;
  bool have_noise = false;
};

} // namespace

// OVR_REGISTER_OBJECT(MainRenderer, renderer, DeviceHIP, hip) would expand to this (ObjectFactory.h:77-85)
extern "C" ovr::MainRenderer* ovr_create_renderer__hip() { return new DeviceHIP; }
