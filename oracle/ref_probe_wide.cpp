// ref_probe_wide.cpp - the same pins as ref_probe.cpp on WIDE seeded inputs: calls the REAL reference code (compiled in place by
// oracle/build_ref.sh into oracle/_ref/libovr_refhost.so) and writes raw little-endian binary vectors that
// tests/golden/make_ref_probe_wide.py turns into tests/golden/ref_probe_wide.npz.  Contains no reference source.
//   rgba8   image_to_rgba8 (ovr/common/imageio.cpp:146-181) on every float within 6 ulp of k/255 for k = 0..255, special values, 4096 random floats
//   camera  the camera basis (formulas of ovr/devices/optix7/device_impl.cpp:125-144) with the reference's gdt math on 400 random cameras
//   affine  gdt's xfmPoint / xfmVector / xfmNormal / normalize for 200 random instance transforms and points (device_impl.cpp:288-296)
//   exr     ovr::save_image("*.exr") + the reference's load_exr on 16384 floats spread over the half range (imageio.cpp:15-103, tinyexr)
// usage: ref_probe_wide <output directory>
#include <ovr/common/imageio.h>
#include <ovr/common/math_def.h>
#include <ovr/scene.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>
#include <memory>
#include <vector>

using namespace ovr;

std::shared_ptr<uint32_t> image_to_rgba8(const float* input, int width, int height, int ch, int ch_stride, bool flip_vertical);
void load_exr(float** data, int* width, int* height, const char* filename);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static float uni(float a, float b) { return a + (b - a) * (float)(rnd() >> 8) * (1.0f / 16777216.0f); }

template <typename T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v)
{
  FILE* f = fopen((dir + "/" + name).c_str(), "wb");
  if (!f) { perror(name); _exit(3); }
  fwrite(v.data(), sizeof(T), v.size(), f);
  fclose(f);
}

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  { // rgba8
    std::vector<float> in;
    for (int k = 0; k <= 255; ++k) {
      float c = (float)k / 255.f;
      float lo = c, hi = c;
      in.push_back(c);
      for (int u = 0; u < 6; ++u) { lo = std::nextafter(lo, -1.f); hi = std::nextafter(hi, 2.f); in.push_back(lo); in.push_back(hi); }
    }
    for (float v : { 0.f, -0.f, 1.f, 1e-45f, 1e-39f, -1e-39f, 1e30f, -1e30f, INFINITY, -INFINITY, 0.5f, 0.49999997f, 0.50000006f, 254.5f / 255.f, 255.5f / 255.f, 0.9999999f })
      in.push_back(v);
    for (int i = 0; i < 4096; ++i) in.push_back(uni(-0.5f, 1.5f));
    while (in.size() % 4) in.push_back(0.25f);
    const int W = (int)(in.size() / 4), H = 1;
    auto out = image_to_rgba8(in.data(), W, H, 4, 4, false);
    std::vector<uint8_t> o((const uint8_t*)out.get(), (const uint8_t*)out.get() + in.size());
    dump(dir, "rgba8_in.f32", in);
    dump(dir, "rgba8_out.u8", o);
  }
  { // camera
    std::vector<float> in, out;
    for (int i = 0; i < 400; ++i) {
      const float s = (i % 4 == 0) ? 1.f : (i % 4 == 1) ? 1000.f : (i % 4 == 2) ? 0.01f : 50.f;
      const vec3f from(uni(-s, s), uni(-s, s), uni(-s, s)), at(uni(-s, s), uni(-s, s), uni(-s, s));
      vec3f up(uni(-1, 1), uni(-1, 1), uni(-1, 1));
      if (i % 5 == 0) up = vec3f(0.f, 1.f, 0.f);
      const float fovy = uni(5.f, 120.f);
      const int w = 1 + (int)(rnd() % 4000), h = 1 + (int)(rnd() % 2500);
      const float t = 2.f * tan(fovy * 0.5f * (float)M_PI / 180.f);
      const float aspect = w / float(h);
      const vec3f direction = normalize(at - from);
      const vec3f horizontal = t * aspect * normalize(cross(direction, up));
      const vec3f vertical = cross(horizontal, direction) / aspect;
      for (float v : { from.x, from.y, from.z, at.x, at.y, at.z, up.x, up.y, up.z, fovy, (float)w, (float)h }) in.push_back(v);
      for (float v : { from.x, from.y, from.z, direction.x, direction.y, direction.z, horizontal.x, horizontal.y, horizontal.z, vertical.x, vertical.y, vertical.z }) out.push_back(v);
    }
    dump(dir, "camera_in.f32", in);
    dump(dir, "camera_out.f32", out);
  }
  { // affine
    std::vector<float> in, out;
    for (int i = 0; i < 200; ++i) {
      const vec3f origin(uni(-100, 100), uni(-100, 100), uni(-100, 100));
      const vec3f scale(uni(0.05f, 3000.f), uni(0.05f, 3000.f), uni(0.05f, 3000.f));
      const vec3f p(uni(-500, 500), uni(-500, 500), uni(-500, 500));
      const affine3f otw = affine3f::translate(origin) * affine3f::scale(scale);
      const affine3f wto = rcp(otw);
      const vec3f a = xfmPoint(wto, p), b = xfmVector(wto, p), c = xfmNormal(otw, p), d = normalize(p);
      for (float v : { origin.x, origin.y, origin.z, scale.x, scale.y, scale.z, p.x, p.y, p.z }) in.push_back(v);
      for (float v : { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z }) out.push_back(v);
    }
    dump(dir, "affine_in.f32", in);
    dump(dir, "affine_out.f32", out);
  }
  { // exr
    const int W = 1024, H = 4;
    std::vector<float> img(W * H * 4);
    for (size_t i = 0; i < img.size(); ++i) {
      // sign, an exponent from below the smallest subnormal half (2^-25) to above the largest half (2^16), random mantissa -
      // every 4th value with its low 13 mantissa bits set to the tie pattern 0x1000 (exactly between two halves)
      const int e = -27 + (int)(rnd() % 45);
      uint32_t m = rnd() & 0x7fffffu;
      if (i % 4 == 1) m = (m & ~0x1fffu) | 0x1000u;
      if (i % 16 == 3) m = (m & ~0x1fffu) | 0x0fffu;
      if (i % 16 == 7) m = (m & ~0x1fffu) | 0x1001u;
      const uint32_t bits = ((rnd() & 1u) << 31) | ((uint32_t)(e + 127) << 23) | m;
      memcpy(&img[i], &bits, 4);
    }
    const char* path = "/tmp/ovr_ref_probe_wide.exr";
    fflush(stdout);
    const int saved = dup(1);
    (void)freopen("/dev/null", "w", stdout);
    ovr::save_image(std::string(path), (const vec4f*)img.data(), W, H);
    float* back = nullptr;
    int w = 0, h = 0;
    load_exr(&back, &w, &h, path);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    if (w != W || h != H) return 4;
    std::vector<float> unflipped(img.size());
    for (int y = 0; y < H; ++y) memcpy(&unflipped[(size_t)y * W * 4], &back[(size_t)(H - 1 - y) * W * 4], (size_t)W * 4 * sizeof(float));
    dump(dir, "exr_in.f32", img);
    dump(dir, "exr_out.f32", unflipped);
    free(back);
  }
  return 0;
}
