// plugin_probe.cpp - drives plugin/libdevice_hip.so through the REFERENCE'S OWN MainRenderer interface (compiled against the
// reference's headers and linked with its host code in oracle/_ref, like renderbatch) along a call sequence renderbatch
// does not exercise: sparse sampling with a focus window, several samples per pixel, progressive accumulation and a camera
// change in between.  Test infrastructure only; tests/test_renderbatch_gpu.py compares the dumped frames with what the
// Python host gets from the same C ABI for the same calls.
//   usage: plugin_probe <scene.json> <w> <h> <out.f32>      (writes two w*h*4 float frames)
#include <ovr/renderer.h>
#include <ovr/scene.h>
#include <ovr/serializer/serializer.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace ovr;

static void dump(MainRenderer& ren, std::FILE* f, int w, int h)
{
  MainRenderer::FrameBufferData fb;
  ren.mapframe(&fb);
  const float* frame = (const float*)fb.rgba->to_cpu()->data();
  std::fwrite(frame, sizeof(float), (size_t)w * h * 4, f);
}

// renderapp's render-thread loop (apps/main_app.cpp:244-263: commit, mapframe, swap, render) without its window: what a caller that maps
// every frame gets.  usage: plugin_probe --loop <frames> scene.json w h     prints "loop fps = ..."
#include <chrono>
static int loop_mode(int frames, const char* scene_file, int w, int h, const char* argv0)
{
  Scene scene = scene::create_json_scene(scene_file);
  auto ren = create_renderer("hip");
  ren->set_fbsize(vec2i(w, h));
  ren->set_frame_accumulation(true);
  ren->set_path_tracing(false);
  ren->set_sample_per_pixel(1);
  ren->set_volume_sampling_rate(1.f);   // renderbatch's default (main_batch.cpp:69)
  const char* args[] = { argv0 };
  ren->init(1, args, scene, scene.camera);
  ren->set_camera(scene.camera.from, scene.camera.at, scene.camera.up);
  MainRenderer::FrameBufferData fb;
  double checksum = 0.0;
  auto iterate = [&](int n) {
    for (int i = 0; i < n; ++i) {
      ren->commit();
      ren->mapframe(&fb);
      checksum += ((const float*)fb.rgba->to_cpu()->data())[((size_t)h / 2 * w + w / 2) * 4 + 3]; // the GUI thread reads the mapped frame
      ren->swap();
      ren->render();
    }
  };
  iterate(8);
  const auto t0 = std::chrono::high_resolution_clock::now();
  iterate(frames);
  const double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  std::printf("loop fps = %.2f (%d frames in renderapp's order commit -> mapframe -> swap -> render, %dx%d, %.3f ms per frame, centre alpha sum %.4f)\n",
              frames / s, frames, w, h, s / frames * 1e3, checksum);
  return 0;
}

// Two-thread contract of the interactive app (apps/main_app.cpp:233-278, ovr/common/vidi_async_loop.h:70-87,
// ovr/common/vidi_transactional_value.h:75-103): a GUI thread calls the thread-safe setters at any time while the render thread loops
// commit -> mapframe -> (publish the mapped pointer) -> swap -> render, and the GUI reads the published frame WHILE the next one renders into
// the other buffer set.  usage: plugin_probe --stress <iterations> scene.json w h <out prefix>
//   1. reference pass, one thread: every state (K cameras x M transfer functions; accumulation off, so a frame is a function of its state)
//      rendered once; inputs and frames dumped to <prefix>_states.bin (tests/test_round4_gpu.py renders the same states through the Python host)
//   2. stress: the setter thread picks random states / focus windows at random moments, the render thread hashes every mapped frame - it must
//      be one of the K x M reference frames, never a mixture (a torn rectangle of the cropped mapframe copy, a half-applied commit) - and a
//      reader thread hashes the published buffer during the following render(): it must still be the frame that was mapped
#include <atomic>
#include <mutex>
#include <random>
#include <set>
#include <thread>
static uint64_t frame_hash(const float* p, size_t n_floats)
{
  const uint32_t* u = (const uint32_t*)p;
  uint64_t a = 0, b = 0;
  for (size_t i = 0; i < n_floats; ++i) { // position-weighted sums mod 2^64 (vectorisable in numpy for the Python side)
    a += (uint64_t)u[i] * (2ull * i + 1ull);
    b += ((uint64_t)u[i] ^ 0x9e3779b97f4a7c15ull) * (i * 0x100000001b3ull + 7ull);
  }
  return a ^ (b << 1);
}
static int stress_mode(int iters, const char* scene_file, int w, int h, const char* prefix, const char* argv0)
{
  Scene scene = scene::create_json_scene(scene_file);
  auto ren = create_renderer("hip");
  ren->set_fbsize(vec2i(w, h));
  ren->set_frame_accumulation(false);
  ren->set_path_tracing(false);
  ren->set_sample_per_pixel(1);
  ren->set_volume_sampling_rate(1.f);
  const char* args[] = { argv0 };
  ren->init(1, args, scene, scene.camera);
  const int K = 4, M = 3;
  const auto base = ren->unsafe_get_tfn(); // the scene's transfer function as MainRenderer::set_scene flattened it
  std::vector<vec3f> from(K);
  for (int k = 0; k < K; ++k) from[k] = scene.camera.from * (1.f + 0.08f * (float)k);
  std::vector<std::vector<float>> alphas(M, base.tfn_alphas);
  const float scale[3] = { 1.f, 0.5f, 0.25f };
  for (int m = 0; m < M; ++m)
    for (size_t i = 1; i < alphas[m].size(); i += 2) alphas[m][i] = base.tfn_alphas[i] * scale[m];
  auto set_state = [&](int k, int m) {
    ren->set_camera(from[k], scene.camera.at, scene.camera.up);
    ren->set_transfer_function(base.tfn_colors, alphas[m], base.tfn_value_range);
  };
  const size_t nf = (size_t)w * h * 4;
  MainRenderer::FrameBufferData fb;
  std::set<uint64_t> known;
  {
    std::FILE* f = std::fopen((std::string(prefix) + "_states.bin").c_str(), "wb");
    if (!f) return 3;
    const int32_t hdr[6] = { K, M, (int32_t)(base.tfn_colors.size() / 3), (int32_t)(base.tfn_alphas.size() / 2), w, h };
    std::fwrite(hdr, sizeof(int32_t), 6, f);
    for (int k = 0; k < K; ++k) { std::fwrite(&from[k], sizeof(float), 3, f); std::fwrite(&scene.camera.at, sizeof(float), 3, f); std::fwrite(&scene.camera.up, sizeof(float), 3, f); }
    std::fwrite(base.tfn_colors.data(), sizeof(float), base.tfn_colors.size(), f);
    for (int m = 0; m < M; ++m) std::fwrite(alphas[m].data(), sizeof(float), alphas[m].size(), f);
    std::fwrite(&base.tfn_value_range, sizeof(float), 2, f);
    for (int k = 0; k < K; ++k)
      for (int m = 0; m < M; ++m) {
        set_state(k, m);
        ren->commit();
        ren->render();
        ren->mapframe(&fb);
        const float* p = (const float*)fb.rgba->to_cpu()->data();
        known.insert(frame_hash(p, nf));
        std::fwrite(p, sizeof(float), nf, f);
        ren->swap();
      }
    std::fclose(f);
  }
  std::mutex pub;
  const float* pub_ptr = nullptr;
  uint64_t pub_serial = 0, pub_hash = 0;
  std::atomic<uint64_t> mapping_serial{ 0 };
  std::atomic<bool> stop{ false };
  std::atomic<uint64_t> reads{ 0 }, torn{ 0 }, sets{ 0 };
  std::thread gui([&] {
    std::mt19937 rng(12345);
    while (!stop.load()) {
      const unsigned r = rng();
      const int k = (int)(r % K), m = (int)((r >> 8) % M);
      switch ((r >> 16) % 4) {
      case 0: ren->set_camera(from[k], scene.camera.at, scene.camera.up); break;
      case 1: ren->set_transfer_function(base.tfn_colors, alphas[m], base.tfn_value_range); break;
      case 2: set_state(k, m); break;
      default: ren->set_focus(vec2f(0.25f + 0.5f * (float)((r >> 20) & 255) / 255.f, 0.5f), 0.1f + 0.2f * (float)((r >> 28) & 3), 0.05f); break; // no effect on a dense frame
      }
      sets.fetch_add(1);
      std::this_thread::sleep_for(std::chrono::microseconds(20 + (r >> 24) % 400));
    }
  });
  std::thread reader([&] {
    while (!stop.load()) {
      const float* p; uint64_t serial, hash;
      { std::lock_guard<std::mutex> lk(pub); p = pub_ptr; serial = pub_serial; hash = pub_hash; }
      if (!p) { std::this_thread::yield(); continue; }
      const uint64_t got = frame_hash(p, nf);
      // the buffer of iteration `serial` is written again by the mapframe of iteration serial + 2 (the same set comes round): a read that
      // overlapped it proves nothing - the reference's double buffer has the same window (optix7_common.h:328-414)
      if (mapping_serial.load() >= serial + 2) continue;
      reads.fetch_add(1);
      if (got != hash) torn.fetch_add(1);
    }
  });
  uint64_t checked = 0, unknown = 0;
  std::set<uint64_t> seen;
  for (int i = 1; i <= iters; ++i) {
    ren->commit();
    mapping_serial.store((uint64_t)i);
    ren->mapframe(&fb);
    const float* p = (const float*)fb.rgba->to_cpu()->data();
    const uint64_t hsh = frame_hash(p, nf);
    ++checked;
    if (!known.count(hsh)) ++unknown;
    seen.insert(hsh);
    { std::lock_guard<std::mutex> lk(pub); pub_ptr = p; pub_serial = (uint64_t)i; pub_hash = hsh; }
    ren->swap();
    ren->render();
  }
  stop.store(true);
  gui.join();
  reader.join();
  std::printf("stress: iterations %d frames_checked %llu not_a_reference_frame %llu distinct_frames_seen %zu of %d reader_checks %llu torn %llu setter_calls %llu\n", iters,
              (unsigned long long)checked, (unsigned long long)unknown, seen.size(), K * M, (unsigned long long)reads.load(), (unsigned long long)torn.load(),
              (unsigned long long)sets.load());
  return (unknown == 0 && torn.load() == 0) ? 0 : 4;
}

int main(int ac, char** av)
{
  if (ac >= 7 && std::string(av[1]) == "--stress") {
    try { return stress_mode(std::atoi(av[2]), av[3], std::atoi(av[4]), std::atoi(av[5]), av[6], av[0]); }
    catch (const std::exception& e) { std::fprintf(stderr, "plugin_probe: %s\n", e.what()); return 1; }
  }
  if (ac >= 6 && std::string(av[1]) == "--loop") {
    try { return loop_mode(std::atoi(av[2]), av[3], std::atoi(av[4]), std::atoi(av[5]), av[0]); }
    catch (const std::exception& e) { std::fprintf(stderr, "plugin_probe: %s\n", e.what()); return 1; }
  }
  if (ac < 5) { std::fprintf(stderr, "usage: plugin_probe scene.json w h out.f32 | plugin_probe --loop frames scene.json w h\n"); return 2; }
  const int w = std::atoi(av[2]), h = std::atoi(av[3]);
  try {
    Scene scene = scene::create_json_scene(av[1]);
    auto ren = create_renderer("hip");
    ren->set_fbsize(vec2i(w, h));
    ren->set_frame_accumulation(true);
    ren->set_path_tracing(false);
    ren->set_sample_per_pixel(2);
    ren->set_volume_sampling_rate(scene.volume_sampling_rate);
    const char* argv0[] = { av[0] };
    ren->init(1, argv0, scene, scene.camera);
    ren->commit();
    // sparse sampling around an off-centre focus, three accumulated frames
    ren->set_sparse_sampling(true);
    ren->set_focus(vec2f(0.4f, 0.6f), 0.3f, 0.05f);
    ren->commit();
    for (int i = 0; i < 3; ++i) ren->render();
    std::FILE* f = std::fopen(av[4], "wb");
    if (!f) return 3;
    dump(*ren, f, w, h);
    // back to dense sampling from a moved camera (three-vector form: fovy falls back to 60, renderer.h:149-152), swap in between
    ren->set_sparse_sampling(false);
    ren->set_camera(scene.camera.from * 1.1f, scene.camera.at, scene.camera.up);
    ren->commit();
    ren->render();
    ren->swap();
    ren->render();
    dump(*ren, f, w, h);
    std::fclose(f);
  }
  catch (const std::exception& e) {
    std::fprintf(stderr, "plugin_probe: %s\n", e.what());
    return 1;
  }
  return 0;
}
