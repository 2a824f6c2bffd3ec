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

int main(int ac, char** av)
{
  if (ac < 5) { std::fprintf(stderr, "usage: plugin_probe scene.json w h out.f32\n"); return 2; }
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
