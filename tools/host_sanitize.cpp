// Host-side hardening run: the OBJ, PNG and scene-file parsers and the scene builder under
// AddressSanitizer and UndefinedBehaviorSanitizer, on the shipped files and on randomly damaged
// copies of them. Needs no GPU. Build and run with tools/run_host_sanitize.sh.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

#include "../include/rt_amd.h"

static std::vector<char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const std::vector<char>& d) {
  std::ofstream f(p, std::ios::binary);
  f.write(d.data(), (std::streamsize)d.size());
}

// flips, cuts, repeats and splices bytes
static std::vector<char> damage(std::vector<char> d, std::mt19937& g) {
  if (d.empty()) return d;
  const int n = 1 + (int)(g() % 8);
  for (int k = 0; k < n && !d.empty(); ++k) {
    const size_t at = g() % d.size();
    switch (g() % 5) {
      case 0: d[at] = (char)g(); break;
      case 1: d.resize(at); break;
      case 2: {
        const size_t a = g() % d.size(), b = a + g() % std::min<size_t>(64, d.size() - a + 1);
        const std::vector<char> piece(d.begin() + a, d.begin() + std::min(b, d.size()));
        d.insert(d.begin() + at, piece.begin(), piece.end());
        break;
      }
      case 3: d.erase(d.begin() + at, d.begin() + std::min(d.size(), at + 1 + g() % 64)); break;
      default: d[at] ^= (char)(1u << (g() % 8)); break;
    }
  }
  return d;
}

int main(int argc, char** argv) {
  const std::string root = argc > 1 ? argv[1] : ".";
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 300;
  const std::string tmp = root + "/gpurun_out/sanitize_tmp";
  std::mt19937 g(12345);
  int ok = 0, refused = 0;

  // 1. the shipped scenes, whole: parse, flatten, build the tree, export
  for (const char* name : {"spot", "lens", "sphere_floor", "ogre"}) {
    rt_objects* objs = nullptr;
    rt_render_params p;
    uint32_t its = 0;
    if (rt_scene_file_load((root + "/scenes/" + name + ".scene").c_str(), &objs, &p, &its) != RT_OK) {
      std::fprintf(stderr, "scene %s: %s\n", name, rt_last_error());
      return 1;
    }
    rt_scene* s = nullptr;
    if (rt_scene_from_objects(objs, &s) != RT_OK) return 1;
    rt_objects_destroy(objs);
    rt_scene_info info;
    rt_scene_get_info(s, &info);
    std::vector<uint32_t> nodes((size_t)info.bvh_nodes * 16);
    std::vector<int32_t> order(info.n_triangles);
    rt_scene_export_bvh(s, nodes.data(), order.data());
    std::vector<double> tv(info.n_triangles * 9);
    rt_scene_export_flat(s, tv.data(), nullptr, nullptr, nullptr, nullptr);
    if (rt_scene_upload(s, 0) == RT_OK) std::fprintf(stderr, "note: a GPU is visible; device calls are not part of this run\n");
    rt_scene_destroy(s);
    std::printf("scene %-12s %8llu triangles %7u nodes\n", name, (unsigned long long)info.n_triangles, info.bvh_nodes);
  }

  // 1b. a scene of many meshes: the flattening, the tree builder and the leaf records all run on
  // several threads here (48 cows, 281 000 triangles)
  {
    rt_objects* objs = nullptr;
    rt_mesh* cow = nullptr;
    if (rt_objects_create(&objs) != RT_OK || rt_mesh_load_obj((root + "/assets/spot_triangulated.obj").c_str(), &cow) != RT_OK) return 1;
    rt_material mat;
    std::memset(&mat, 0, sizeof mat);
    mat.kind = RT_MAT_OPAQUE;
    mat.color_tex = mat.normal_tex = -1;
    mat.diffuse = 1.0;
    mat.ior = 1.0;
    for (int i = 0; i < 48; ++i) {
      rt_mesh* m = nullptr;
      const double t[3] = {(i % 8) * 1.2, 0.0, -2.0 - (i / 8) * 1.2};
      if (rt_mesh_clone(cow, &m) != RT_OK || rt_mesh_translate(m, t) != RT_OK || rt_objects_push_mesh(objs, m, &mat) < 0) {
        std::fprintf(stderr, "48 cows: %s\n", rt_last_error());
        return 1;
      }
      rt_mesh_destroy(m);
    }
    rt_mesh_destroy(cow);
    rt_scene* s = nullptr;
    if (rt_scene_from_objects(objs, &s) != RT_OK) {
      std::fprintf(stderr, "48 cows: %s\n", rt_last_error());
      return 1;
    }
    rt_objects_destroy(objs);
    rt_scene_info info;
    rt_scene_get_info(s, &info);
    std::printf("scene %-12s %8llu triangles %7u nodes\n", "48 cows", (unsigned long long)info.n_triangles, info.bvh_nodes);
    rt_scene_destroy(s);
  }

  // 2. damaged PNG files
  for (const char* rel : {"/assets/spot_texture.png", "/assets/ogre/normalmap.png"}) {
    std::vector<char> good = slurp(root + rel);
    good.resize(std::min<size_t>(good.size(), 200000));  // the head of the file is where the structure is
    for (int r = 0; r < rounds; ++r) {
      spit(tmp + ".png", damage(good, g));
      rt_image* img = nullptr;
      if (rt_image_load_png((tmp + ".png").c_str(), &img) == RT_OK) {
        uint32_t w, h;
        const uint8_t* px;
        rt_image_info(img, &w, &h, &px);
        volatile uint8_t sink = px[(size_t)w * h * 4 - 1];
        (void)sink;
        rt_image_destroy(img);
        ok++;
      } else {
        refused++;
      }
    }
  }
  // small hand-made PNGs of every colour type survive damage too
  // (their bytes come from the writer)
  {
    std::vector<uint8_t> px(16 * 16 * 4);
    for (size_t i = 0; i < px.size(); ++i) px[i] = (uint8_t)(i * 7);
    rt_image_write_png((tmp + "_w.png").c_str(), px.data(), 16, 16);
    std::vector<char> good = slurp(tmp + "_w.png");
    for (int r = 0; r < rounds * 4; ++r) {
      spit(tmp + ".png", damage(good, g));
      rt_image* img = nullptr;
      if (rt_image_load_png((tmp + ".png").c_str(), &img) == RT_OK) {
        rt_image_destroy(img);
        ok++;
      } else {
        refused++;
      }
    }
  }

  // 3. damaged OBJ files
  for (const char* rel : {"/assets/sphere.obj", "/assets/triangle_floor.obj"}) {
    const std::vector<char> good = slurp(root + rel);
    for (int r = 0; r < rounds; ++r) {
      spit(tmp + ".obj", damage(good, g));
      rt_mesh* m = nullptr;
      if (rt_mesh_load_obj((tmp + ".obj").c_str(), &m) == RT_OK) {
        rt_objects* objs = nullptr;
        rt_objects_create(&objs);
        rt_material mat;
        std::memset(&mat, 0, sizeof mat);
        mat.color_tex = mat.normal_tex = -1;
        if (rt_objects_push_mesh(objs, m, &mat) >= 0) {
          rt_scene* s = nullptr;
          if (rt_scene_from_objects(objs, &s) == RT_OK) rt_scene_destroy(s);
        }
        rt_objects_destroy(objs);
        rt_mesh_destroy(m);
        ok++;
      } else {
        refused++;
      }
    }
  }

  // 4. damaged scene files
  {
    const std::vector<char> good = slurp(root + "/scenes/spot.scene");
    for (int r = 0; r < rounds; ++r) {
      spit(root + "/scenes/sanitize_tmp.scene", damage(good, g));
      rt_objects* objs = nullptr;
      if (rt_scene_file_load((root + "/scenes/sanitize_tmp.scene").c_str(), &objs, nullptr, nullptr) == RT_OK) {
        rt_objects_destroy(objs);
        ok++;
      } else {
        refused++;
      }
    }
    std::remove((root + "/scenes/sanitize_tmp.scene").c_str());
  }
  std::printf("damaged inputs: %d accepted, %d refused, no sanitizer report\n", ok, refused);
  return 0;
}
