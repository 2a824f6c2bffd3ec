// ref_probe.cpp - calls the REAL reference code (compiled in place by oracle/build_ref.sh) and prints known-answer
// vectors that pin this repo's oracle.  It contains no reference source: it only #includes the reference's headers and
// links its objects (oracle/_ref/libovr_refhost.so).  Output: one JSON document on stdout (tests/golden/ref_probe.json
// is that output, committed as a fixture; regenerate with `make -C oracle golden`).
#include <ovr/common/imageio.h>
#include <ovr/common/math_def.h>
#include <ovr/scene.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <memory>
#include <vector>

using namespace ovr;

// defined at global scope in ovr/common/imageio.cpp:146 (not declared in the header)
std::shared_ptr<uint32_t> image_to_rgba8(const float* input, int width, int height, int ch, int ch_stride, bool flip_vertical);
void load_exr(float** data, int* width, int* height, const char* filename); // ovr/common/imageio.cpp:85-103

static void print_vec(const char* name, const std::vector<double>& v, bool last = false)
{
  printf("  \"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) printf("%s%.9g", i ? ", " : "", v[i]);
  printf("]%s\n", last ? "" : ",");
}

static void print_bits(const char* name, const std::vector<double>& v)
{
  printf("  \"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) printf("%s%.0f", i ? ", " : "", v[i]);
  printf("],\n");
}

int main()
{
  printf("{\n");
  // (1) the reference's only "tonemap": image_to_rgba8 (ovr/common/imageio.cpp:146-181), 4 channels, with and without flip
  {
    const int W = 16, H = 4;
    std::vector<float> img(W * H * 4);
    std::vector<double> in;
    for (int i = 0; i < W * H * 4; ++i) {
      float v;
      switch (i % 8) {
      case 0: v = (float)(i / 8) / 31.f; break;           // ramp
      case 1: v = (float)(i % 256) / 255.f; break;        // exact k/255
      case 2: v = std::nextafter((float)((i * 7) % 256) / 255.f, 0.f); break; // just below k/255
      case 3: v = std::nextafter((float)((i * 3) % 256) / 255.f, 2.f); break; // just above k/255
      case 4: v = -0.25f + 0.01f * (i % 13); break;       // negative
      case 5: v = 1.0f + 0.125f * (i % 5); break;         // >= 1
      case 6: v = 0.5f; break;
      default: v = 0.999999f; break;
      }
      img[i] = v;
      in.push_back(v);
    }
    print_vec("rgba8_input", in);
    for (int flip = 0; flip < 2; ++flip) {
      auto out = image_to_rgba8(img.data(), W, H, 4, 4, flip != 0);
      std::vector<double> o;
      const uint8_t* b = (const uint8_t*)out.get();
      for (int i = 0; i < W * H * 4; ++i) o.push_back(b[i]);
      print_vec(flip ? "rgba8_flipped" : "rgba8_plain", o);
    }
    printf("  \"rgba8_dims\": [%d, %d],\n", W, H);
  }
  // (2) camera basis with the reference's gdt math, formulas of ovr/devices/optix7/device_impl.cpp:125-144
  {
    struct Cam { vec3f from, at, up; float fovy; int w, h; };
    const Cam cams[] = {
      { vec3f(16.f, 16.f, 100.8f), vec3f(16.f, 16.f, 16.f), vec3f(0.f, 1.f, 0.f), 60.f, 64, 48 },
      { vec3f(-34.9f, 41.4f, 40.8f), vec3f(16.f, 16.f, 16.f), vec3f(0.f, 1.f, 0.f), 45.f, 1920, 1080 },
      { vec3f(3.f, -7.f, 2.f), vec3f(0.5f, 0.25f, -1.f), vec3f(0.1f, 0.9f, 0.2f), 33.f, 333, 777 },
    };
    std::vector<double> in, out;
    for (const Cam& c : cams) {
      const float t = 2.f * tan(c.fovy * 0.5f * (float)M_PI / 180.f);
      const float aspect = c.w / float(c.h);
      const vec3f direction = normalize(c.at - c.from);
      const vec3f horizontal = t * aspect * normalize(cross(direction, c.up));
      const vec3f vertical = cross(horizontal, direction) / aspect;
      for (float v : { c.from.x, c.from.y, c.from.z, c.at.x, c.at.y, c.at.z, c.up.x, c.up.y, c.up.z, c.fovy, (float)c.w, (float)c.h }) in.push_back(v);
      for (float v : { c.from.x, c.from.y, c.from.z, direction.x, direction.y, direction.z, horizontal.x, horizontal.y, horizontal.z,
                       vertical.x, vertical.y, vertical.z }) out.push_back(v);
    }
    print_vec("camera_input", in);
    print_vec("camera_basis", out);
  }
  // (3) gdt primitives the ray marcher relies on: normalize, cross, xfmPoint / xfmVector / xfmNormal of the instance
  //     transform translate(origin) * scale(spacing * dims) and its inverse (device_impl.cpp:288-296)
  {
    const vec3f origin(1.5f, -2.f, 0.25f), scale(31.f, 48.5f, 17.f);
    const affine3f otw = affine3f::translate(origin) * affine3f::scale(scale);
    const affine3f wto = rcp(otw);
    const vec3f pts[] = { vec3f(3.f, 4.f, 5.f), vec3f(-10.f, 0.5f, 22.f), vec3f(16.f, 24.f, 8.5f) };
    std::vector<double> o;
    for (const vec3f& p : pts) {
      const vec3f a = xfmPoint(wto, p), b = xfmVector(wto, p), c = xfmNormal(otw, p), d = normalize(p);
      for (float v : { a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z }) o.push_back(v);
    }
    print_vec("xfm_origin_scale", { origin.x, origin.y, origin.z, scale.x, scale.y, scale.z });
    print_vec("xfm_points", { 3, 4, 5, -10, 0.5, 22, 16, 24, 8.5 });
    print_vec("xfm_results", o);
  }
  // (5) EXR output: ovr::save_image("*.exr", vec4f*) (ovr/common/imageio.cpp:264-272: rows flipped, tinyexr asked for HALF pixels)
  //     followed by the reference's own load_exr (imageio.cpp:85-103): what the float -> half step does to each value, as raw
  //     IEEE bit patterns (input float bits -> bits of the float the file holds)
  {
    const int W = 16, H = 5;
    std::vector<float> img(W * H * 4);
    uint32_t state = 12345u;
    for (int i = 0; i < W * H * 4; ++i) {
      state = state * 1664525u + 1013904223u;
      float v;
      switch (i % 16) {
      case 0: v = 1.0f + std::ldexp(1.0f, -11) * (float)(1 + 2 * ((i / 16) % 8)); break;   // exact ties between two halves
      case 1: v = std::ldexp(1.0f, -24) * (float)(i / 16) * 0.5f; break;                   // subnormal halves and ties among them
      case 2: v = 65504.0f + (float)(i / 16); break;                                      // up to and across the overflow tie 65520
      case 3: v = -((float)(i / 16) * 0.37f + 0.001f); break;
      case 4: v = std::ldexp(1.0f, -14) - std::ldexp(1.0f, -26) * (float)(i / 16); break; // around the smallest normal half
      case 5: v = std::ldexp(1.0f, -126) * 0.5f; break;                                   // a float denormal
      case 6: v = 2.0f - std::ldexp(1.0f, -12); break;                                    // mantissa carry into the exponent
      case 7: v = 1.0e9f; break;
      case 8: v = -1.0e9f; break;
      case 9: v = 0.0f; break;
      case 10: v = -0.0f; break;
      default: { uint32_t b = (state >> 9) | 0x3f800000u; float f; memcpy(&f, &b, 4); v = (f - 1.0f) * ((i % 3) ? 1.0f : 300.0f); } break;
      }
      img[i] = v;
    }
    const char* path = "/tmp/ovr_ref_probe.exr";
    fflush(stdout);
    const int saved = dup(1);                     // save_image prints to stdout: keep the JSON clean
    (void)freopen("/dev/null", "w", stdout);
    ovr::save_image(std::string(path), (const vec4f*)img.data(), W, H);
    float* back = nullptr;
    int w = 0, h = 0;
    load_exr(&back, &w, &h, path);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    std::vector<double> in, outbits;
    for (int i = 0; i < W * H * 4; ++i) { uint32_t b; memcpy(&b, &img[i], 4); in.push_back((double)b); }
    // the file holds the FLIPPED image: un-flip so that entry i answers input i
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W * 4; ++x) { uint32_t b; memcpy(&b, &back[(size_t)(H - 1 - y) * W * 4 + x], 4); outbits.push_back((double)b); }
    printf("  \"exr_dims\": [%d, %d, %d, %d],\n", W, H, w, h);
    print_bits("exr_input_bits", in);
    print_bits("exr_roundtrip_bits", outbits);
    free(back);
  }
  // (4) ValueType numbering and sizes (ovr/scene.h:32-73) - part of the C ABI
  {
    std::vector<double> v;
    for (ValueType t : { VALUE_TYPE_UINT8, VALUE_TYPE_INT8, VALUE_TYPE_UINT16, VALUE_TYPE_INT16, VALUE_TYPE_UINT32, VALUE_TYPE_INT32,
                         VALUE_TYPE_FLOAT, VALUE_TYPE_DOUBLE }) {
      v.push_back((double)t);
      v.push_back((double)value_type_size(t));
    }
    print_vec("value_types", v, true);
  }
  printf("}\n");
  return 0;
}
