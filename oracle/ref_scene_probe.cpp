// ref_scene_probe.cpp - runs the REAL reference scene loader (ovr::scene::create_json_scene, reference
// ovr/serializer/serializer_diva.cpp:13-39 -> serializer_vidi3d.cpp:334-408, tfn::loadTransferFunction) on VIDI3D scene files
// and prints what it produces, to pin this repo's own loader (open-volume-renderer_amd/vidi3d.py::read_scene).
// The reference ships scene JSONs but no volume data, so each scene is loaded from a patched copy whose volume is a 2x2x2
// block of zeros (everything except the voxels - transfer function, value range, camera, lights, sampling rate - is
// independent of the volume size).  Contains no reference source; built by oracle/build_ref.sh against the reference's headers.
// usage: ref_scene_probe out.json scene1.json [scene2.json ...]   (the reference's loader chats on stdout, hence a file)
#include <ovr/scene.h>
#include <ovr/serializer/serializer.h>

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

using json = nlohmann::json;

namespace ovr { namespace scene { Scene create_json_scene(std::string filename); } }

static FILE* out = nullptr;
#define printf(...) fprintf(out, __VA_ARGS__)
static void arr(const char* name, const float* v, size_t n, bool last = false)
{
  printf("    \"%s\": [", name);
  for (size_t i = 0; i < n; ++i) printf("%s%.9g", i ? "," : "", v[i]);
  printf("]%s\n", last ? "" : ",");
}

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  out = fopen(argv[1], "w");
  if (!out) return 3;
  const std::string tmp = "/tmp/ovr_ref_scene_probe";
  (void)system(("mkdir -p " + tmp).c_str());
  printf("{\n");
  for (int a = 2; a < argc; ++a) {
    std::ifstream f(argv[a]);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    json root = json::parse(text, nullptr, true, true);
    // patch: tiny zero volume next to the patched scene file
    const size_t elem = 8;
    {
      std::ofstream z(tmp + "/zero.raw", std::ios::binary);
      std::vector<char> zeros(2 * 2 * 2 * elem, 0);
      z.write(zeros.data(), (std::streamsize)zeros.size());
    }
    for (auto& ds : root["dataSource"]) {
      ds["dimensions"] = { { "x", 2 }, { "y", 2 }, { "z", 2 } };
      ds["fileName"] = tmp + "/zero.raw";
      ds["offset"] = 0;
    }
    const std::string patched = tmp + "/scene.json";
    { std::ofstream o(patched); o << root.dump(); }
    ovr::Scene scene = ovr::scene::create_json_scene(patched);

    std::string name = argv[a];
    name = name.substr(name.find_last_of('/') + 1);
    printf("  \"%s\": {\n", name.c_str());
    const auto& vol = ovr::parse_single_volume_scene(scene).structured_regular;
    const auto& tfn = scene.instances[0].models[0].volume_model.transfer_function;
    printf("    \"value_type\": %d,\n", (int)vol.data->type);
    const float sp[3] = { vol.grid_spacing.x, vol.grid_spacing.y, vol.grid_spacing.z }, og[3] = { vol.grid_origin.x, vol.grid_origin.y, vol.grid_origin.z };
    arr("grid_spacing", sp, 3);
    arr("grid_origin", og, 3);
    arr("tfn_color", (const float*)tfn.color->data(), (size_t)tfn.color->dims.v * 4);
    arr("tfn_opacity", (const float*)tfn.opacity->data(), (size_t)tfn.opacity->dims.v);
    const float vr[2] = { tfn.value_range.x, tfn.value_range.y };
    arr("value_range", vr, 2);
    const auto& c = scene.camera;
    const float cam[10] = { c.from.x, c.from.y, c.from.z, c.at.x, c.at.y, c.at.z, c.up.x, c.up.y, c.up.z, c.perspective.fovy };
    arr("camera", cam, 10);
    std::vector<float> lights;
    for (const auto& l : scene.lights) for (float v : { l.directional.direction.x, l.directional.direction.y, l.directional.direction.z, l.color.x, l.color.y, l.color.z }) lights.push_back(v);
    arr("lights", lights.data(), lights.size());
    const float rate[1] = { scene.volume_sampling_rate };
    arr("volume_sampling_rate", rate, 1, true);
    printf("  }%s\n", a + 1 < argc ? "," : "");
  }
  printf("}\n");
  fclose(out);
  return 0;
}
