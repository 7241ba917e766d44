// rt_scenefile.cpp — scene description files.
//
// The reference compiles its scenes in: sphere_scene() and lens_scene() (src/main.rs:1270-1450,
// 1223-1268) are Rust functions that call load_obj, rotate_*/translate, Object::from_mesh and
// Object::sphere, and main() picks one at src/main.rs:1077-1078. This file reads the same calls
// from a text file and replays them through the library's public entry points, so a scene is
// data instead of code. One statement per line, '#' starts a comment, paths are relative to
// the file:
//
//   texture NAME FILE.png
//   mesh FILE.obj [rotate_x DEG | rotate_y DEG | rotate_z DEG | translate X Y Z]... MATERIAL
//   sphere X Y Z RADIUS MATERIAL
//   render [width W] [height H] [spp S] [its N] [seed K] [max_depth D] [fov F] [dof X]
//
//   MATERIAL = opaque [color R G B | color texture NAME] [normal NAME] [metallic M]
//                     [how_metallic H] [diffuse D] [emission R G B]
//            | dielectric IOR
//
// Transforms apply in the order written. Angles are degrees, converted as the reference writes
// them: DEG * PI / 180.0 (src/main.rs:1251, 1340). Objects get their index in file order, which
// is the order Scene::from_objects flattens them in (src/main.rs:752-769).

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rt_types.h"

namespace {

struct MeshDel {
  void operator()(rt_mesh* m) const { rt_mesh_destroy(m); }
};
struct ObjsDel {
  void operator()(rt_objects* o) const { rt_objects_destroy(o); }
};
struct ImgDel {
  void operator()(rt_image* i) const { rt_image_destroy(i); }
};
using MeshPtr = std::unique_ptr<rt_mesh, MeshDel>;

struct Parser {
  std::string file, dir;
  int line_no = 0;
  std::vector<std::string> tok;
  size_t at = 0;
  std::map<std::string, int> textures;
  std::map<std::string, MeshPtr> meshes;  // parsed OBJ files, by resolved path

  int fail(const char* what, const std::string& detail = std::string()) {
    rt_set_error("%s:%d: %s%s%s", file.c_str(), line_no, what, detail.empty() ? "" : " ", detail.c_str());
    return RT_ERR_IO;
  }
  std::string resolve(const std::string& p) const { return (!p.empty() && p[0] == '/') || dir.empty() ? p : dir + "/" + p; }
  bool more() const { return at < tok.size(); }
  bool word(std::string* out) {
    if (!more()) return false;
    *out = tok[at++];
    return true;
  }
  bool number(double* v) {
    if (!more()) return false;
    if (!rt_parse_f64(tok[at], v)) return false;
    ++at;
    return true;
  }
  bool numbers(double* v, int n) {
    for (int i = 0; i < n; ++i)
      if (!number(v + i)) return false;
    return true;
  }

  int material(rt_material* m) {
    *m = rt_material{};
    m->color_tex = m->normal_tex = -1;
    std::string kind;
    if (!word(&kind)) return fail("material expected: opaque ... or dielectric IOR");
    if (kind == "dielectric") {
      m->kind = RT_MAT_DIELECTRIC;
      if (!number(&m->ior)) return fail("dielectric needs an index of refraction");
      if (more()) return fail("unexpected text after the material:", tok[at]);
      return RT_OK;
    }
    if (kind != "opaque") return fail("unknown material", kind);
    m->kind = RT_MAT_OPAQUE;
    m->diffuse = 1.0;
    m->ior = 1.0;  // unused by Opaque
    m->color[0] = m->color[1] = m->color[2] = 1.0;
    std::string key;
    while (word(&key)) {
      if (key == "color") {
        if (more() && tok[at] == "texture") {
          ++at;
          std::string name;
          if (!word(&name) || !textures.count(name)) return fail("color texture: unknown texture", name);
          m->color_tex = textures[name];
          m->color[0] = m->color[1] = m->color[2] = 0.0;  // unused with a texture
        } else if (!numbers(m->color, 3)) {
          return fail("color needs R G B or texture NAME");
        }
      } else if (key == "normal") {
        std::string name;
        if (!word(&name) || !textures.count(name)) return fail("normal: unknown texture", name);
        m->normal_tex = textures[name];
      } else if (key == "metallic") {
        if (!number(&m->metallic)) return fail("metallic needs a number");
      } else if (key == "how_metallic") {
        if (!number(&m->how_metallic)) return fail("how_metallic needs a number");
      } else if (key == "diffuse") {
        if (!number(&m->diffuse)) return fail("diffuse needs a number");
      } else if (key == "emission") {
        if (!numbers(m->emission, 3)) return fail("emission needs R G B");
      } else {
        return fail("unknown material property", key);
      }
    }
    return RT_OK;
  }
};

void split_line(const std::string& s, std::vector<std::string>* out) {
  out->clear();
  size_t i = 0;
  const size_t n = s.size();
  auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; };
  while (i < n) {
    while (i < n && blank(s[i])) ++i;
    if (i < n && s[i] == '#') break;
    const size_t b = i;
    while (i < n && !blank(s[i])) ++i;
    if (i > b) out->emplace_back(s, b, i - b);
  }
}

bool to_u32(double v, uint32_t* out) {
  if (!(v >= 0.0) || v > 4294967295.0 || v != std::floor(v)) return false;
  *out = (uint32_t)v;
  return true;
}

}  // namespace

extern "C" void rt_render_params_default(rt_render_params* p) {
  if (!p) return;
  *p = rt_render_params{};
  p->width = 1600;       // src/main.rs:1051
  p->height = 1200;      // src/main.rs:1052
  p->spp = 10;           // src/main.rs:1044
  p->max_depth = 200;    // src/main.rs:1043
  p->fov_y_deg = 60.0;   // src/main.rs:978
  p->dof_strength = 0.0; // src/main.rs:1048
}

static int rt_scene_file_load_body(const char* path, rt_objects** out, rt_render_params* params, uint32_t* its);
extern "C" int rt_scene_file_load(const char* path, rt_objects** out, rt_render_params* params, uint32_t* its) {
  return rt_guard("rt_scene_file_load", [&] { return rt_scene_file_load_body(path, out, params, its); });
}
static int rt_scene_file_load_body(const char* path, rt_objects** out, rt_render_params* params, uint32_t* its) {
  if (!path || !out) {
    rt_set_error("rt_scene_file_load: null argument");
    return RT_ERR_INVALID;
  }
  FILE* f = std::fopen(path, "rb");
  if (!f) {
    rt_set_error("rt_scene_file_load: cannot open %s", path);
    return RT_ERR_IO;
  }
  std::string content;
  char buf[1 << 14];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) content.append(buf, got);
  std::fclose(f);

  rt_render_params rp;
  rt_render_params_default(&rp);
  uint32_t n_its = 1;

  Parser ps;
  ps.file = path;
  const size_t slash = ps.file.find_last_of('/');
  ps.dir = slash == std::string::npos ? std::string() : ps.file.substr(0, slash);
  rt_objects* raw_objs = nullptr;
  int rc = rt_objects_create(&raw_objs);
  if (rc != RT_OK) return rc;
  std::unique_ptr<rt_objects, ObjsDel> objs(raw_objs);

  size_t pos = 0;
  while (pos <= content.size()) {
    size_t nl = content.find('\n', pos);
    if (nl == std::string::npos) nl = content.size();
    ++ps.line_no;
    split_line(content.substr(pos, nl - pos), &ps.tok);
    pos = nl + 1;
    ps.at = 0;
    std::string head;
    if (!ps.word(&head)) continue;

    if (head == "texture") {
      std::string name, file;
      if (!ps.word(&name) || !ps.word(&file) || ps.more()) return ps.fail("texture needs NAME FILE");
      if (ps.textures.count(name)) return ps.fail("texture defined twice:", name);
      rt_image* raw = nullptr;
      rc = rt_image_load_png(ps.resolve(file).c_str(), &raw);
      if (rc != RT_OK) return rc;
      std::unique_ptr<rt_image, ImgDel> img(raw);
      uint32_t w = 0, h = 0;
      const uint8_t* px = nullptr;
      rt_image_info(raw, &w, &h, &px);
      const int id = rt_objects_add_texture(objs.get(), px, w, h);
      if (id < 0) return id;
      ps.textures[name] = id;
    } else if (head == "mesh") {
      std::string file;
      if (!ps.word(&file)) return ps.fail("mesh needs FILE");
      const std::string full = ps.resolve(file);
      auto it = ps.meshes.find(full);
      if (it == ps.meshes.end()) {
        rt_mesh* raw = nullptr;
        rc = rt_mesh_load_obj(full.c_str(), &raw);
        if (rc != RT_OK) return rc;
        it = ps.meshes.emplace(full, MeshPtr(raw)).first;
      }
      rt_mesh* raw_copy = nullptr;
      rc = rt_mesh_clone(it->second.get(), &raw_copy);
      if (rc != RT_OK) return rc;
      MeshPtr mesh(raw_copy);
      while (ps.more() && ps.tok[ps.at] != "opaque" && ps.tok[ps.at] != "dielectric") {
        std::string op;
        ps.word(&op);
        double v[3];
        if (op == "translate") {
          if (!ps.numbers(v, 3)) return ps.fail("translate needs X Y Z");
          rt_mesh_translate(mesh.get(), v);
        } else if (op == "rotate_x" || op == "rotate_y" || op == "rotate_z") {
          if (!ps.number(v)) return ps.fail("rotation needs an angle in degrees");
          const double angle = v[0] * 3.14159265358979323846264338327950288 / 180.0;
          if (op == "rotate_x") rt_mesh_rotate_x(mesh.get(), angle);
          if (op == "rotate_y") rt_mesh_rotate_y(mesh.get(), angle);
          if (op == "rotate_z") rt_mesh_rotate_z(mesh.get(), angle);
        } else {
          return ps.fail("unknown mesh transform", op);
        }
      }
      rt_material m;
      rc = ps.material(&m);
      if (rc != RT_OK) return rc;
      rc = rt_objects_push_mesh(objs.get(), mesh.get(), &m);
      if (rc < 0) return rc;
    } else if (head == "sphere") {
      double v[4];
      if (!ps.numbers(v, 4)) return ps.fail("sphere needs X Y Z RADIUS");
      rt_material m;
      rc = ps.material(&m);
      if (rc != RT_OK) return rc;
      rc = rt_objects_push_sphere(objs.get(), v, v[3], &m);
      if (rc < 0) return rc;
    } else if (head == "render") {
      std::string key;
      while (ps.word(&key)) {
        double v;
        if (!ps.number(&v)) return ps.fail("render: number expected after", key);
        uint32_t u = 0;
        const bool is_u = to_u32(v, &u);
        if (key == "width" && is_u) rp.width = u;
        else if (key == "height" && is_u) rp.height = u;
        else if (key == "spp" && is_u) rp.spp = u;
        else if (key == "its" && is_u) n_its = u;
        else if (key == "max_depth" && is_u) rp.max_depth = (int32_t)u;
        else if (key == "seed" && v >= 0.0 && v == std::floor(v) && v < 18446744073709551616.0) rp.seed = (uint64_t)v;
        else if (key == "fov") rp.fov_y_deg = v;
        else if (key == "dof") rp.dof_strength = v;
        else return ps.fail("render: unknown setting or bad value for", key);
      }
    } else {
      return ps.fail("unknown statement", head);
    }
  }
  if (params) *params = rp;
  if (its) *its = n_its;
  *out = objs.release();
  return RT_OK;
}
