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

int main(int ac, char** av)
{
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
