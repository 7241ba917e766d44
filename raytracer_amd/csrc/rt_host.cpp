// rt_host.cpp — host half of the library: OBJ loading, mesh transforms, object list,
// flattening (Scene::from_objects, src/main.rs:750-774) and the acceleration-structure builder
// that replaces BVH::build of crate bvh 0.6.0 (src/main.rs:771). No GPU is needed for anything
// in this file. Compile with -ffp-contract=off: flattened coordinates must carry the same bits
// the reference's binary64 arithmetic produces.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "rt_types.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void rt_set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }

#ifndef RT_SOURCE_DIGEST
#define RT_SOURCE_DIGEST "unknown"
#endif
extern "C" const char* rt_source_digest(void) { return RT_SOURCE_DIGEST; }

// ------------------------------------------------------------------------------------------
// meshes — PolygonMesh, src/main.rs:566-574
// ------------------------------------------------------------------------------------------
struct D3 {
  double x, y, z;
};
static inline D3 sub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline D3 add(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline D3 crs(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline D3 unit(D3 a) {
  double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return D3{a.x / l, a.y / l, a.z / l};
}

struct rt_mesh {
  std::vector<D3> vertices;
  std::vector<D3> vertex_normals;
  std::vector<double> uvs;          // pairs
  std::vector<uint64_t> tri_vertex; // triples of vertex indices
  std::vector<uint64_t> tri_uv;     // triples of uv indices (the HashMap of src/main.rs:573)
};

// One whitespace-separated token list per line, as str::split_whitespace gives.
static void split_ws(const char* s, size_t n, std::vector<std::string>& out) {
  out.clear();
  size_t i = 0;
  while (i < n) {
    while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\f' || s[i] == '\v')) ++i;
    size_t b = i;
    while (i < n && !(s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\f' || s[i] == '\v')) ++i;
    if (i > b) out.emplace_back(s + b, i - b);
  }
}

static bool parse_f64(const std::string& t, double* v) { return rt_parse_f64(t, v); }

// "12", "12/5", "12/5/7" -> the integers between the slashes.
static bool parse_face_item(const std::string& t, std::vector<uint64_t>& out) {
  out.clear();
  size_t b = 0;
  for (;;) {
    size_t e = t.find('/', b);
    std::string piece = t.substr(b, e == std::string::npos ? std::string::npos : e - b);
    if (piece.empty()) return false;
    uint64_t v = 0;
    if (!rt_parse_u64(piece, &v)) return false;
    out.push_back(v);
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return true;
}

static int rt_mesh_load_obj_body(const char* path, rt_mesh** out);
extern "C" int rt_mesh_load_obj(const char* path, rt_mesh** out) {
  return rt_guard("rt_mesh_load_obj", [&] { return rt_mesh_load_obj_body(path, out); });
}
static int rt_mesh_load_obj_body(const char* path, rt_mesh** out) {
  if (!path || !out) {
    rt_set_error("rt_mesh_load_obj: null argument");
    return RT_ERR_INVALID;
  }
  FILE* f = std::fopen(path, "rb");
  if (!f) {
    rt_set_error("rt_mesh_load_obj: cannot open %s", path);
    return RT_ERR_IO;
  }
  std::string content;
  char buf[1 << 16];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) content.append(buf, got);
  std::fclose(f);

  std::unique_ptr<rt_mesh> m(new (std::nothrow) rt_mesh());
  if (!m) return RT_ERR_OOM;
  std::vector<std::string> parts;
  std::vector<uint64_t> fi[3];
  size_t pos = 0, line_no = 0;
  while (pos < content.size()) {
    size_t nl = content.find('\n', pos);
    size_t len = (nl == std::string::npos ? content.size() : nl) - pos;
    split_ws(content.data() + pos, len, parts);
    pos += len + 1;
    ++line_no;
    if (parts.empty()) continue;
    const std::string& tag = parts[0];
    if (tag == "v") {  // src/main.rs:1146-1149
      D3 v;
      if (parts.size() < 4 || !parse_f64(parts[1], &v.x) || !parse_f64(parts[2], &v.y) || !parse_f64(parts[3], &v.z)) {
        rt_set_error("%s:%zu: bad vertex line", path, line_no);
        return RT_ERR_IO;
      }
      m->vertices.push_back(v);
    } else if (tag == "vt") {  // src/main.rs:1181-1184
      double a, b;
      if (parts.size() < 3 || !parse_f64(parts[1], &a) || !parse_f64(parts[2], &b)) {
        rt_set_error("%s:%zu: bad texture-coordinate line", path, line_no);
        return RT_ERR_IO;
      }
      m->uvs.push_back(a);
      m->uvs.push_back(b);
    } else if (tag == "f") {  // src/main.rs:1150-1180: first three corners only
      if (parts.size() < 4) {
        rt_set_error("%s:%zu: face with fewer than 3 corners", path, line_no);
        return RT_ERR_IO;
      }
      for (int k = 0; k < 3; ++k)
        if (!parse_face_item(parts[1 + k], fi[k]) || fi[k][0] == 0) {
          rt_set_error("%s:%zu: bad face corner '%s'", path, line_no, parts[1 + k].c_str());
          return RT_ERR_IO;
        }
      bool has_uv = fi[0].size() >= 2;  // src/main.rs:1156
      for (int k = 0; k < 3; ++k) {
        m->tri_vertex.push_back(fi[k][0] - 1);
        if (has_uv) {
          if (fi[k].size() < 2 || fi[k][1] == 0) {
            rt_set_error("%s:%zu: face corner without texture index", path, line_no);
            return RT_ERR_IO;
          }
          m->tri_uv.push_back(fi[k][1] - 1);
        } else {
          m->tri_uv.push_back(0);
        }
      }
    }
  }
  // src/main.rs:1189 — the coordinate uv-less faces and spheres resolve to
  m->uvs.push_back(0.0);
  m->uvs.push_back(1.0);

  const size_t nv = m->vertices.size(), nt = m->tri_vertex.size() / 3, nuv = m->uvs.size() / 2;
  for (size_t i = 0; i < nt * 3; ++i) {
    if (m->tri_vertex[i] >= nv || m->tri_uv[i] >= nuv) {
      rt_set_error("%s: face index out of range", path);
      return RT_ERR_IO;
    }
  }
  // src/main.rs:1191-1212: vertex normal = normalised sum of the (normalised) normals of the
  // faces that use the vertex, added in face order.
  std::vector<D3> sum(nv, D3{0.0, 0.0, 0.0});
  for (size_t t = 0; t < nt; ++t) {
    uint64_t i1 = m->tri_vertex[t * 3], i2 = m->tri_vertex[t * 3 + 1], i3 = m->tri_vertex[t * 3 + 2];
    D3 n = unit(crs(sub(m->vertices[i2], m->vertices[i1]), sub(m->vertices[i3], m->vertices[i1])));
    sum[i1] = add(sum[i1], n);
    sum[i2] = add(sum[i2], n);
    sum[i3] = add(sum[i3], n);
  }
  m->vertex_normals.resize(nv);
  for (size_t i = 0; i < nv; ++i) m->vertex_normals[i] = unit(sum[i]);
  *out = m.release();
  return RT_OK;
}

static int rt_mesh_clone_body(const rt_mesh* mesh, rt_mesh** out);
extern "C" int rt_mesh_clone(const rt_mesh* mesh, rt_mesh** out) {
  return rt_guard("rt_mesh_clone", [&] { return rt_mesh_clone_body(mesh, out); });
}
static int rt_mesh_clone_body(const rt_mesh* mesh, rt_mesh** out) {
  if (!mesh || !out) {
    rt_set_error("rt_mesh_clone: null argument");
    return RT_ERR_INVALID;
  }
  rt_mesh* c = new (std::nothrow) rt_mesh(*mesh);
  if (!c) return RT_ERR_OOM;
  *out = c;
  return RT_OK;
}

extern "C" void rt_mesh_destroy(rt_mesh* mesh) { delete mesh; }

extern "C" int rt_mesh_counts(const rt_mesh* mesh, uint64_t* nv, uint64_t* nt, uint64_t* nuv) {
  if (!mesh) {
    rt_set_error("rt_mesh_counts: null mesh");
    return RT_ERR_INVALID;
  }
  if (nv) *nv = mesh->vertices.size();
  if (nt) *nt = mesh->tri_vertex.size() / 3;
  if (nuv) *nuv = mesh->uvs.size() / 2;
  return RT_OK;
}

// Mat3H * Vec3H followed by the homogeneous divide (src/main.rs:156-167, 116-118, 631): every
// row is a four-term sum against (x, y, z, 1), then divided by the w row's value.
static inline D3 apply_rows(const double r[4][4], D3 p) {
  double o[4];
  for (int i = 0; i < 4; ++i) o[i] = r[i][0] * p.x + r[i][1] * p.y + r[i][2] * p.z + r[i][3] * 1.0;
  return D3{o[0] / o[3], o[1] / o[3], o[2] / o[3]};
}
static int rotate(rt_mesh* mesh, const double r[4][4]) {
  if (!mesh) {
    rt_set_error("rt_mesh_rotate: null mesh");
    return RT_ERR_INVALID;
  }
  for (D3& v : mesh->vertices) v = apply_rows(r, v);
  for (D3& n : mesh->vertex_normals) n = apply_rows(r, n);
  return RT_OK;
}
extern "C" int rt_mesh_rotate_x(rt_mesh* mesh, double angle) {  // src/main.rs:144-153, 649-658
  double c = std::cos(angle), s = std::sin(angle);
  const double r[4][4] = {{1.0, 0.0, 0.0, 0.0}, {0.0, c, -s, 0.0}, {0.0, s, c, 0.0}, {0.0, 0.0, 0.0, 1.0}};
  return rotate(mesh, r);
}
extern "C" int rt_mesh_rotate_y(rt_mesh* mesh, double angle) {  // src/main.rs:122-131, 627-636
  double c = std::cos(angle), s = std::sin(angle);
  const double r[4][4] = {{c, 0.0, s, 0.0}, {0.0, 1.0, 0.0, 0.0}, {-s, 0.0, c, 0.0}, {0.0, 0.0, 0.0, 1.0}};
  return rotate(mesh, r);
}
extern "C" int rt_mesh_rotate_z(rt_mesh* mesh, double angle) {  // src/main.rs:133-142, 638-647
  double c = std::cos(angle), s = std::sin(angle);
  const double r[4][4] = {{c, s, 0.0, 0.0}, {-s, c, 0.0, 0.0}, {0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 0.0, 1.0}};
  return rotate(mesh, r);
}
extern "C" int rt_mesh_translate(rt_mesh* mesh, const double t[3]) {  // src/main.rs:660-664
  if (!mesh || !t) {
    rt_set_error("rt_mesh_translate: null argument");
    return RT_ERR_INVALID;
  }
  for (D3& v : mesh->vertices) v = add(v, D3{t[0], t[1], t[2]});
  return RT_OK;
}

// ------------------------------------------------------------------------------------------
// object list — Vec<Object>, src/main.rs:444-469
// ------------------------------------------------------------------------------------------
struct HostObject {
  int32_t kind;
  rt_mesh mesh;
  double centre[3];
  double radius;
  rt_material mat;
};
struct rt_objects {
  std::vector<HostObject> items;
  std::vector<HostTexture> textures;
};

static int rt_objects_create_body(rt_objects** out);
extern "C" int rt_objects_create(rt_objects** out) {
  return rt_guard("rt_objects_create", [&] { return rt_objects_create_body(out); });
}
static int rt_objects_create_body(rt_objects** out) {
  if (!out) {
    rt_set_error("rt_objects_create: null argument");
    return RT_ERR_INVALID;
  }
  *out = new (std::nothrow) rt_objects();
  return *out ? RT_OK : RT_ERR_OOM;
}
extern "C" void rt_objects_destroy(rt_objects* objs) { delete objs; }

static int rt_objects_add_texture_body(rt_objects* objs, const uint8_t* rgba, uint32_t w, uint32_t h);
extern "C" int rt_objects_add_texture(rt_objects* objs, const uint8_t* rgba, uint32_t w, uint32_t h) {
  return rt_guard("rt_objects_add_texture", [&] { return rt_objects_add_texture_body(objs, rgba, w, h); });
}
static int rt_objects_add_texture_body(rt_objects* objs, const uint8_t* rgba, uint32_t w, uint32_t h) {
  if (!objs || !rgba || w == 0 || h == 0) {
    rt_set_error("rt_objects_add_texture: null argument or empty texture");
    return RT_ERR_INVALID;
  }
  HostTexture t;
  t.w = w;
  t.h = h;
  t.rgba.assign(rgba, rgba + (size_t)w * h * 4);
  objs->textures.push_back(std::move(t));
  return (int)objs->textures.size() - 1;
}

static bool material_ok(const rt_material* m, size_t n_textures) {
  if (!m) return false;
  if (m->kind != RT_MAT_OPAQUE && m->kind != RT_MAT_DIELECTRIC) return false;
  if (m->color_tex < -1 || m->normal_tex < -1) return false;
  if (m->color_tex >= (int64_t)n_textures || m->normal_tex >= (int64_t)n_textures) return false;
  return true;
}

static int rt_objects_push_mesh_body(rt_objects* objs, const rt_mesh* mesh, const rt_material* mat);
extern "C" int rt_objects_push_mesh(rt_objects* objs, const rt_mesh* mesh, const rt_material* mat) {
  return rt_guard("rt_objects_push_mesh", [&] { return rt_objects_push_mesh_body(objs, mesh, mat); });
}
static int rt_objects_push_mesh_body(rt_objects* objs, const rt_mesh* mesh, const rt_material* mat) {
  if (!objs || !mesh || !material_ok(mat, objs->textures.size())) {
    rt_set_error("rt_objects_push_mesh: null argument or invalid material");
    return RT_ERR_INVALID;
  }
  HostObject o;
  o.kind = RT_OBJ_MESH;
  o.mesh = *mesh;
  o.centre[0] = o.centre[1] = o.centre[2] = 0.0;
  o.radius = 0.0;
  o.mat = *mat;
  objs->items.push_back(std::move(o));
  return (int)objs->items.size() - 1;
}

static int rt_objects_push_sphere_body(rt_objects* objs, const double centre[3], double radius, const rt_material* mat);
extern "C" int rt_objects_push_sphere(rt_objects* objs, const double centre[3], double radius, const rt_material* mat) {
  return rt_guard("rt_objects_push_sphere", [&] { return rt_objects_push_sphere_body(objs, centre, radius, mat); });
}
static int rt_objects_push_sphere_body(rt_objects* objs, const double centre[3], double radius, const rt_material* mat) {
  if (!objs || !centre || !material_ok(mat, objs->textures.size())) {
    rt_set_error("rt_objects_push_sphere: null argument or invalid material");
    return RT_ERR_INVALID;
  }
  HostObject o;
  o.kind = RT_OBJ_SPHERE;
  o.centre[0] = centre[0];
  o.centre[1] = centre[1];
  o.centre[2] = centre[2];
  o.radius = radius;
  o.mat = *mat;
  objs->items.push_back(std::move(o));
  return (int)objs->items.size() - 1;
}

// Triangle records in leaf order (rt_types.h), from the flat arrays and s->leaf_order.
// Threads the host's scene passes may use: RT_BUILD_THREADS, else min(16, cores).
unsigned rt_build_threads() {
  unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("RT_BUILD_THREADS")) threads = (unsigned)std::max(1, std::atoi(e));
  return threads;
}

// body(piece, begin, end) over [0, n) in at most 64 contiguous pieces, one per thread. The pieces
// write disjoint outputs, so the result does not depend on the number of threads.
template <class F>
static void parallel_ranges(size_t n, size_t min_per_thread, F&& body) {
  const size_t threads =
      std::max<size_t>(1, std::min<size_t>(std::min<size_t>(64, rt_build_threads()), n / std::max<size_t>(1, min_per_thread)));
  if (threads <= 1) {
    body((size_t)0, (size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(threads - 1);
  size_t started = 1;  // pieces [started, threads) found no thread and run here instead
  try {
    for (; started < threads; ++started)
      pool.emplace_back([&, started] { body(started, n * started / threads, n * (started + 1) / threads); });
  } catch (const std::system_error&) {
  }
  body((size_t)0, (size_t)0, n / threads);
  for (size_t k = started; k < threads; ++k) body(k, n * k / threads, n * (k + 1) / threads);
  for (std::thread& t : pool) t.join();
}

// Which spheres get a place in the tree: the first RT_TREE_SPHERES whose centre and radius are
// finite and of ordinary size (the others are tested against every ray, as the reference tests
// all of them) — in scenes of at least RT_SPHERE_TREE_FROM spheres (default 9). With fewer, testing
// every sphere is cheaper than reaching their places: measured on the reference's default scene
// (six spheres) on MI355X, places cost the 3840x2160x256 frame 318 ms against 301 ms — a place's
// visit is a triangle test of its own and lengthens the leaf loops of the waves it occurs in.
// RT_SPHERE_TREE_FROM=n in the environment moves the threshold (1: always, 0 or a huge n: never).
void rt_scene_pick_tree_spheres(rt_scene* s) {
  s->tree_spheres.clear();
  size_t from = 9;
  const char* e = std::getenv("RT_SPHERE_TREE_FROM");
  if (e && *e) {
    const long v = std::atol(e);
    from = v <= 0 ? (size_t)-1 : (size_t)v;
  }
  if (s->spheres.size() < from) return;
  for (size_t i = 0; i < s->spheres.size() && i < (size_t)RT_TREE_SPHERES; ++i) {
    const SphereRec& sp = s->spheres[i];
    bool ok = std::isfinite(sp.r) && std::fabs(sp.r) < 1e150;
    for (int c = 0; c < 3; ++c) ok = ok && std::isfinite(sp.c[c]) && std::fabs(sp.c[c]) < 1e150;
    if (ok) s->tree_spheres.push_back((int32_t)i);
  }
}

// A sphere's place in the tree is found as a triangle's is, from the bounding box of three
// vertices: two opposite corners of the sphere's box and its centre.
void rt_sphere_place_vertices(const SphereRec& sp, double v9[9]) {
  const double r = std::fabs(sp.r);
  for (int c = 0; c < 3; ++c) {
    v9[c] = sp.c[c] - r;
    v9[3 + c] = sp.c[c] + r;
    v9[6 + c] = sp.c[c];
  }
}

void rt_fill_leaf_records(rt_scene* s) {
  const size_t n = s->leaf_order.size();
  const size_t n_tri = s->tri_obj.size();
  s->tri_geom.resize(n);
  s->tri_shade.resize(n);
  parallel_ranges(n, 16384, [s, n_tri](size_t, size_t begin, size_t end) {
    for (size_t slot = begin; slot < end; ++slot) {
      const size_t t = (size_t)s->leaf_order[slot];
      TriGeom& g = s->tri_geom[slot];
      if (t >= n_tri) {  // a sphere's place: a triangle test that cannot succeed, and the sphere's index
        const size_t sphere = t - n_tri;
        std::memset(&g, 0, sizeof g);
        g.flat = ~(int32_t)sphere;
        g.obj = s->spheres[sphere].obj;
        TriShade& sh = s->tri_shade[slot];
        std::memset(&sh, 0, sizeof sh);
        sh.tri = -1;
        sh.obj = s->spheres[sphere].obj;
        continue;
      }
      const double* v = &s->tri_vertices[t * 9];
      for (int c = 0; c < 3; ++c) {
        g.a[c] = v[c];
        g.ab[c] = v[3 + c] - v[c];  // src/main.rs:188
        g.ac[c] = v[6 + c] - v[c];  // src/main.rs:189
      }
      g.flat = (int32_t)t;
      g.obj = s->tri_obj[t];
      TriShade& sh = s->tri_shade[slot];
      for (int k = 0; k < 3; ++k) {
        for (int c = 0; c < 3; ++c) sh.n[k][c] = s->tri_normals[t * 9 + k * 3 + c];
        for (int c = 0; c < 2; ++c) sh.uv[k][c] = s->tri_uvs[t * 6 + k * 2 + c];
      }
      sh.tri = s->tri_idx[t];
      sh.obj = s->tri_obj[t];
    }
  });
}

// Absolute padding of every child box: 2^-20 of the largest finite coordinate in the scene (at
// least 2^-20). It covers the rounding slop of the binary64 triangle test (a hit accepted a hair
// outside the exact triangle) with five orders of magnitude to spare for well-conditioned hits.
double rt_scene_box_pad(const rt_scene* s) {
  const std::vector<double>& v = s->tri_vertices;
  std::vector<double> part(64, 1.0);  // a maximum: the same whatever the pieces
  parallel_ranges(v.size(), 1u << 20, [&](size_t piece, size_t begin, size_t end) {
    double extent = 1.0;
    for (size_t i = begin; i < end; ++i)
      if (std::isfinite(v[i])) extent = std::max(extent, std::fabs(v[i]));
    part[piece] = extent;
  });
  double extent = 1.0;
  for (double x : part) extent = std::max(extent, x);
  for (int32_t i : s->tree_spheres) {
    const SphereRec& sp = s->spheres[(size_t)i];
    for (int c = 0; c < 3; ++c) extent = std::max(extent, std::fabs(sp.c[c]) + std::fabs(sp.r));
  }
  return extent * 9.5367431640625e-07;
}

// The tiled copy of a texture that goes to the device (layout: rt_types.h), on the builder's
// threads; texels of the padding are zero and never addressed.
void rt_tile_texture(const HostTexture& t, std::vector<uint8_t>* out, uint32_t* blocks_x) {
  const uint32_t bx = (t.w + RT_TEX_BLOCK - 1) / RT_TEX_BLOCK, by = (t.h + RT_TEX_BLOCK - 1) / RT_TEX_BLOCK;
  *blocks_x = bx;
  out->assign((size_t)bx * by * RT_TEX_BLOCK * RT_TEX_BLOCK * 4u, 0);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(t.rgba.data());
  uint32_t* dst = reinterpret_cast<uint32_t*>(out->data());
  parallel_ranges(t.h, 64, [&](size_t, size_t begin, size_t end) {
    for (size_t y = begin; y < end; ++y)
      for (uint32_t x = 0; x < t.w; ++x) dst[rt_texel_offset(x, (uint32_t)y, bx)] = src[y * t.w + x];
  });
}

// dst[i] += src[i] on the builder's threads (the host-buffer render entry point adds a frame to the
// caller's canvas with it).
void rt_add_into(double* dst, const double* src, size_t n) {
  parallel_ranges(n, 1u << 18, [=](size_t, size_t begin, size_t end) {
    for (size_t i = begin; i < end; ++i) dst[i] += src[i];
  });
}

// Bounds of the centres of the triangles' boxes (finite ones only), for the Morton grid of the
// device tree builders. Minima and maxima: the same whatever the pieces.
void rt_scene_centre_bounds(const rt_scene* s, double clo[3], double chi[3]) {
  const size_t n = s->tri_obj.size();
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> part(64 * 6);
  for (size_t k = 0; k < 64; ++k)
    for (int c = 0; c < 3; ++c) {
      part[k * 6 + c] = inf;
      part[k * 6 + 3 + c] = -inf;
    }
  parallel_ranges(n, 1u << 18, [&](size_t piece, size_t begin, size_t end) {
    double lo3[3] = {inf, inf, inf}, hi3[3] = {-inf, -inf, -inf};
    for (size_t t = begin; t < end; ++t) {
      const double* v = &s->tri_vertices[t * 9];
      for (int c = 0; c < 3; ++c) {
        const double lo = std::min(v[c], std::min(v[3 + c], v[6 + c])), hi = std::max(v[c], std::max(v[3 + c], v[6 + c]));
        const double m = 0.5 * lo + 0.5 * hi;
        if (std::isfinite(m)) {
          lo3[c] = std::min(lo3[c], m);
          hi3[c] = std::max(hi3[c], m);
        }
      }
    }
    for (int c = 0; c < 3; ++c) {
      part[piece * 6 + c] = lo3[c];
      part[piece * 6 + 3 + c] = hi3[c];
    }
  });
  for (int c = 0; c < 3; ++c) {
    clo[c] = inf;
    chi[c] = -inf;
    for (size_t k = 0; k < 64; ++k) {
      clo[c] = std::min(clo[c], part[k * 6 + c]);
      chi[c] = std::max(chi[c], part[k * 6 + 3 + c]);
    }
    for (int32_t i : s->tree_spheres) {
      clo[c] = std::min(clo[c], s->spheres[(size_t)i].c[c]);
      chi[c] = std::max(chi[c], s->spheres[(size_t)i].c[c]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// acceleration structure
// ------------------------------------------------------------------------------------------
// Binned surface-area-heuristic build over binary64 bounds. The tree differs from the crate's
// (one shape per leaf, 6 buckets): here a leaf holds 1..4 triangles and every interior node
// stores both child boxes, so one 64-byte fetch decides both children. The nearest hit does
// not depend on the shape of the tree as long as no box ever excludes a point its triangles
// can be hit at, so the boxes are rounded outward to f32 and padded (see pad_for()).
namespace {

struct Box {
  double lo[3], hi[3];
};
inline Box box_empty() {
  const double inf = std::numeric_limits<double>::infinity();
  return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}
inline void box_grow(Box& b, const Box& o) {
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = std::min(b.lo[k], o.lo[k]);
    b.hi[k] = std::max(b.hi[k], o.hi[k]);
  }
}
inline double box_half_area(const Box& b) {
  double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  if (!(dx >= 0.0) || !(dy >= 0.0) || !(dz >= 0.0)) return 0.0;
  return dx * dy + dy * dz + dz * dx;
}
inline float f32_down(double x) {
  float f = (float)x;
  if ((double)f > x) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
  return f;
}
inline float f32_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafter(f, std::numeric_limits<float>::infinity());
  return f;
}

const double COST_NODE = 1.0;  // one 64-byte fetch + two f32 slab tests
// A binary64 Moller-Trumbore test costs a wave about twice the instructions of a node visit, but
// measured on MI355X (four scenes, 321 to 1.15 M triangles) trees built with equal costs render
// 2-7 % faster than with 2:1: fuller leaves mean fewer of the divergent traversal steps.
const double COST_TRI = 1.0;
// Bins per axis. Measured on MI355X against 16: the 14x14 grid of Spot meshes (1.15 M triangles, whose
// rows of meshes 16 bins cannot separate) renders 5 % faster, 8 % fewer node visits by its scattered
// rays; the other scenes within 1 %; 64 bins change nothing further.
#ifndef RT_SAH_BINS
#define RT_SAH_BINS 32
#endif
const int N_BINS = RT_SAH_BINS;

// What one subtree build produces besides its root ref: its interior nodes with refs local to
// this vector, and the two tree statistics.
struct Subtree {
  std::vector<BvhNode> nodes;
  uint32_t max_depth = 0;
  uint32_t leaves = 0;
};

// Large subtrees are built on their own threads (the index ranges are disjoint). The tree does
// not depend on how many threads ran: a forked subtree's nodes are spliced in pre-order.
const size_t PARALLEL_MIN_TRIS = 16384;

struct Builder {
  const std::vector<Box>& tb;      // per flat triangle
  const std::vector<double>& cen;  // centroids, 3 per triangle
  std::vector<int32_t>& idx;       // permutation being built = leaf order
  std::vector<int32_t>& scratch;   // same size as idx
  double pad;

  static int levels_needed(size_t n) {  // interior levels a balanced split of n triangles needs
    int l = 0;
    size_t cap = RT_LEAF_MAX;
    while (cap < n) {
      cap *= 2;
      ++l;
    }
    return l;
  }

  void store_child(BvhNode& n, int c, const Box& b, int32_t ref) {
    n.bx[c * 2 + 0] = f32_down(b.lo[0] - pad);
    n.bx[c * 2 + 1] = f32_up(b.hi[0] + pad);
    n.by[c * 2 + 0] = f32_down(b.lo[1] - pad);
    n.by[c * 2 + 1] = f32_up(b.hi[1] + pad);
    n.bz[c * 2 + 0] = f32_down(b.lo[2] - pad);
    n.bz[c * 2 + 1] = f32_up(b.hi[2] + pad);
    n.child[c] = ref;
  }

  // Appends `sub` to `into`, shifting its local node refs; returns the shifted root ref.
  static int32_t splice(Subtree& into, Subtree& sub, int32_t ref) {
    const int32_t off = (int32_t)into.nodes.size();
    for (BvhNode nd : sub.nodes) {
      for (int c = 0; c < 2; ++c)
        if (nd.child[c] >= 0) nd.child[c] += off;
      into.nodes.push_back(nd);
    }
    into.max_depth = std::max(into.max_depth, sub.max_depth);
    into.leaves += sub.leaves;
    return ref >= 0 ? ref + off : ref;
  }

  // Builds [lo,hi) into `out` and returns its ref; depth = interior nodes above this subtree;
  // fork_levels = how many more levels may hand their left child to a new thread.
  int32_t build(size_t lo, size_t hi, uint32_t depth, Box* out_box, Subtree& out, int fork_levels) {
    const size_t n = hi - lo;
    Box bounds = box_empty(), cb = box_empty();
    for (size_t i = lo; i < hi; ++i) {
      const int32_t t = idx[i];
      box_grow(bounds, tb[t]);
      for (int k = 0; k < 3; ++k) {
        cb.lo[k] = std::min(cb.lo[k], cen[(size_t)t * 3 + k]);
        cb.hi[k] = std::max(cb.hi[k], cen[(size_t)t * 3 + k]);
      }
    }
    *out_box = bounds;

    // choose a split
    size_t mid = 0;
    bool split = false;
    const bool must_balance = depth + (uint32_t)levels_needed(n) + 1 >= RT_BVH_MAX_DEPTH;
    if (n > 1 && !must_balance) {
      double best_cost = std::numeric_limits<double>::infinity();
      int best_axis = -1, best_bin = -1;
      // one pass over the triangles fills the bins of all three axes (a triangle's box and centre
      // are read once, not three times)
      Box bb[3][N_BINS];
      size_t cnt[3][N_BINS];
      double c0[3], scale[3];
      bool usable[3];
      for (int axis = 0; axis < 3; ++axis) {
        const double ext = cb.hi[axis] - cb.lo[axis];
        usable[axis] = ext > 0.0 && std::isfinite(ext);
        c0[axis] = cb.lo[axis];
        scale[axis] = usable[axis] ? (double)N_BINS / ext : 0.0;
        for (int b = 0; b < N_BINS; ++b) {
          bb[axis][b] = box_empty();
          cnt[axis][b] = 0;
        }
      }
      for (size_t i = lo; i < hi; ++i) {
        const int32_t t = idx[i];
        const Box& box = tb[t];
        for (int axis = 0; axis < 3; ++axis) {
          if (!usable[axis]) continue;
          int b = (int)((cen[(size_t)t * 3 + axis] - c0[axis]) * scale[axis]);
          b = std::min(std::max(b, 0), N_BINS - 1);
          cnt[axis][b]++;
          box_grow(bb[axis][b], box);
        }
      }
      for (int axis = 0; axis < 3; ++axis) {
        if (!usable[axis]) continue;
        double right_area[N_BINS];
        size_t right_cnt[N_BINS];
        Box acc = box_empty();
        size_t c = 0;
        for (int b = N_BINS - 1; b > 0; --b) {
          box_grow(acc, bb[axis][b]);
          c += cnt[axis][b];
          right_area[b] = box_half_area(acc);
          right_cnt[b] = c;
        }
        acc = box_empty();
        c = 0;
        for (int b = 0; b < N_BINS - 1; ++b) {
          box_grow(acc, bb[axis][b]);
          c += cnt[axis][b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          double cost = box_half_area(acc) * (double)c + right_area[b + 1] * (double)right_cnt[b + 1];
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = axis;
            best_bin = b;
          }
        }
      }
      if (best_axis >= 0) {
        const double area = box_half_area(bounds);
        const double split_cost = COST_NODE + (area > 0.0 ? best_cost / area : (double)n) * COST_TRI;
        if (n > RT_LEAF_MAX || split_cost < (double)n * COST_TRI) {
          const double part_c0 = c0[best_axis], part_scale = scale[best_axis];
          // stable partition through the builder's scratch array (ranges of concurrent subtree
          // builds are disjoint, so they share it without locking)
          size_t nl = lo, nr = lo;
          for (size_t i = lo; i < hi; ++i) {
            const int32_t t = idx[i];
            int b = (int)((cen[(size_t)t * 3 + best_axis] - part_c0) * part_scale);
            b = std::min(std::max(b, 0), N_BINS - 1);
            if (b <= best_bin)
              idx[nl++] = t;
            else
              scratch[nr++] = t;
          }
          std::copy(scratch.begin() + lo, scratch.begin() + nr, idx.begin() + nl);
          mid = nl;
          split = mid > lo && mid < hi;
        }
      }
    }
    if (!split && n > RT_LEAF_MAX) {
      // coincident centroids, non-finite data, or the depth budget: halve by centroid order
      int axis = 0;
      for (int k = 1; k < 3; ++k)
        if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
      mid = lo + n / 2;
      std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t a, int32_t b) {
        double ca = cen[(size_t)a * 3 + axis], cbv = cen[(size_t)b * 3 + axis];
        return ca < cbv || (ca == cbv && a < b);
      });
      split = true;
    }
    if (!split) {
      std::sort(idx.begin() + lo, idx.begin() + hi);  // flat order inside a leaf
      out.leaves++;
      out.max_depth = std::max(out.max_depth, depth);
      return ~(int32_t)(lo * RT_LEAF_MAX + (n - 1));
    }
    const int32_t me = (int32_t)out.nodes.size();
    out.nodes.emplace_back();
    std::memset(&out.nodes[me], 0, sizeof(BvhNode));
    Box lb, rb;
    int32_t l, r;
    if (fork_levels > 0 && n >= PARALLEL_MIN_TRIS) {
      Subtree left, right;
      std::future<int32_t> fl = std::async(std::launch::async, [&] { return build(lo, mid, depth + 1, &lb, left, fork_levels - 1); });
      r = build(mid, hi, depth + 1, &rb, right, fork_levels - 1);
      l = fl.get();
      l = splice(out, left, l);
      r = splice(out, right, r);
    } else {
      l = build(lo, mid, depth + 1, &lb, out, 0);
      r = build(mid, hi, depth + 1, &rb, out, 0);
    }
    store_child(out.nodes[me], 0, lb, l);
    store_child(out.nodes[me], 1, rb, r);
    return me;
  }
};

int build_scene_accel(rt_scene* s) {
  rt_scene_pick_tree_spheres(s);
  const size_t n_tri = s->tri_obj.size();
  const size_t n = n_tri + s->tree_spheres.size();  // what the tree is built over: triangles, then the spheres' places
  s->nodes.clear();
  s->leaf_order.clear();
  s->tri_geom.clear();
  s->tri_shade.clear();
  s->root_ref = RT_REF_SENTINEL;
  s->bvh_depth = 0;
  s->bvh_leaves = 0;
  s->has_tree = true;
  if (n == 0) return RT_OK;

  // RT_BUILD_TIMING=1 prints the elapsed time after each phase of the build to stderr.
  const auto t_start = std::chrono::steady_clock::now();
  const bool timing = std::getenv("RT_BUILD_TIMING") != nullptr;
  auto lap = [&](const char* phase) {
    if (timing)
      std::fprintf(stderr, "[rt build] %-9s %.3f s\n", phase,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  };
  std::vector<Box> tb(n);
  std::vector<double> cen(n * 3);
  for (size_t t = 0; t < n; ++t) {
    Box b = box_empty();
    double place[9];
    if (t >= n_tri) rt_sphere_place_vertices(s->spheres[(size_t)s->tree_spheres[t - n_tri]], place);
    const double* verts = t < n_tri ? &s->tri_vertices[t * 9] : place;
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) {
        double x = verts[k * 3 + c];
        b.lo[c] = std::min(b.lo[c], x);
        b.hi[c] = std::max(b.hi[c], x);
      }
    tb[t] = b;
    for (int c = 0; c < 3; ++c) cen[t * 3 + c] = 0.5 * b.lo[c] + 0.5 * b.hi[c];
  }
  s->leaf_order.resize(n);
  for (size_t i = 0; i < n; ++i) s->leaf_order[i] = (int32_t)i;
  const double pad = rt_scene_box_pad(s);
  std::vector<int32_t> scratch(n);
  Builder b{tb, cen, s->leaf_order, scratch, pad};
  int fork_levels = 0;  // 2^fork_levels subtree builds run at once; RT_BUILD_THREADS caps it
  {
    const unsigned threads = rt_build_threads();
    while ((2u << fork_levels) <= threads && fork_levels < 6) ++fork_levels;
  }
  lap("boxes");
  Subtree whole;
  whole.nodes.reserve(n < PARALLEL_MIN_TRIS ? n : 0);
  Box root_box;
  s->root_ref = b.build(0, n, 0, &root_box, whole, fork_levels);
  lap("tree");
  s->nodes.swap(whole.nodes);
  s->bvh_depth = whole.max_depth;
  s->bvh_leaves = whole.leaves;
  for (int32_t& e : s->leaf_order)  // the k-th sphere of the tree -> n_triangles + its sphere index
    if ((size_t)e >= n_tri) e = (int32_t)(n_tri + (size_t)s->tree_spheres[(size_t)e - n_tri]);
  // Interior nodes stay in the order the builder made them: depth first, a node followed by its
  // left subtree (the same for any thread count, since a forked subtree is spliced in where the
  // sequential build would have put it). Measured against a breadth-first renumbering this is
  // up to 5 % faster to traverse: the near part of a walk down the tree shares cache lines.
  rt_fill_leaf_records(s);
  lap("records");
  return RT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// scenes
// ------------------------------------------------------------------------------------------
extern "C" int rt_scene_create(const rt_scene_desc* d, rt_scene** out) { return rt_scene_create_with(d, RT_TREE_HOST, out); }

// The triangle arrays of a description the library assembled itself (rt_scene_from_objects): the
// scene takes them over instead of copying them.
struct OwnedTriangles {
  std::vector<double> tv, tn, tuv;
  std::vector<int32_t> tobj, tidx;
};
static int scene_create(const rt_scene_desc* d, uint32_t tree, rt_scene** out, OwnedTriangles* owned);

extern "C" int rt_scene_create_with(const rt_scene_desc* d, uint32_t tree, rt_scene** out) {
  return rt_guard("rt_scene_create", [&] { return scene_create(d, tree, out, nullptr); });
}

static int scene_create(const rt_scene_desc* d, uint32_t tree, rt_scene** out, OwnedTriangles* owned) {
  if (tree != RT_TREE_HOST && tree != RT_TREE_NONE) {
    rt_set_error("rt_scene_create: unknown tree option %u", tree);
    return RT_ERR_INVALID;
  }
  if (!d || !out) {
    rt_set_error("rt_scene_create: null argument");
    return RT_ERR_INVALID;
  }
  if (d->n_objects == 0 || !d->object_kind || !d->materials) {
    rt_set_error("rt_scene_create: a scene needs at least one object");
    return RT_ERR_INVALID;
  }
  if (d->n_triangles >= (1ull << (31 - RT_LEAF_BITS))) {
    rt_set_error("rt_scene_create: too many triangles (%llu)", (unsigned long long)d->n_triangles);
    return RT_ERR_INVALID;
  }
  if (d->n_triangles && (!d->tri_vertices || !d->tri_normals || !d->tri_uvs || !d->tri_obj || !d->tri_idx)) {
    rt_set_error("rt_scene_create: null triangle array");
    return RT_ERR_INVALID;
  }
  if (d->n_spheres && (!d->spheres || !d->sphere_obj)) {
    rt_set_error("rt_scene_create: null sphere array");
    return RT_ERR_INVALID;
  }
  if (d->n_textures && !d->textures) {
    rt_set_error("rt_scene_create: null texture array");
    return RT_ERR_INVALID;
  }
  for (uint32_t i = 0; i < d->n_textures; ++i)
    if (!d->textures[i].rgba || d->textures[i].width == 0 || d->textures[i].height == 0) {
      rt_set_error("rt_scene_create: texture %u is empty", i);
      return RT_ERR_INVALID;
    }
  for (uint32_t i = 0; i < d->n_objects; ++i) {
    if (d->object_kind[i] != RT_OBJ_MESH && d->object_kind[i] != RT_OBJ_SPHERE) {
      rt_set_error("rt_scene_create: object %u has unknown kind %d", i, d->object_kind[i]);
      return RT_ERR_INVALID;
    }
    if (!material_ok(&d->materials[i], d->n_textures)) {
      rt_set_error("rt_scene_create: object %u has an invalid material", i);
      return RT_ERR_INVALID;
    }
  }
  for (uint64_t t = 0; t < d->n_triangles; ++t) {
    int32_t o = d->tri_obj[t];
    if (o < 0 || (uint32_t)o >= d->n_objects || d->object_kind[o] != RT_OBJ_MESH || d->tri_idx[t] < 0) {
      rt_set_error("rt_scene_create: triangle %llu names object %d, which is not a mesh", (unsigned long long)t, o);
      return RT_ERR_INVALID;
    }
  }
  int32_t prev = -1;
  uint32_t sphere_objects = 0;
  for (uint32_t i = 0; i < d->n_objects; ++i) sphere_objects += d->object_kind[i] == RT_OBJ_SPHERE;
  if (sphere_objects != d->n_spheres) {
    rt_set_error("rt_scene_create: %u sphere objects but %u sphere records", sphere_objects, d->n_spheres);
    return RT_ERR_INVALID;
  }
  for (uint32_t i = 0; i < d->n_spheres; ++i) {
    int32_t o = d->sphere_obj[i];
    if (o <= prev || (uint32_t)o >= d->n_objects || d->object_kind[o] != RT_OBJ_SPHERE) {
      rt_set_error("rt_scene_create: sphere %u names object %d (must be ascending sphere objects)", i, o);
      return RT_ERR_INVALID;
    }
    prev = o;
  }

  std::unique_ptr<rt_scene> s(new (std::nothrow) rt_scene());
  if (!s) return RT_ERR_OOM;
  try {
    s->object_kind.assign(d->object_kind, d->object_kind + d->n_objects);
    s->materials.assign(d->materials, d->materials + d->n_objects);
    const size_t n = (size_t)d->n_triangles;
    if (n && owned) {
      s->tri_vertices.swap(owned->tv);
      s->tri_normals.swap(owned->tn);
      s->tri_uvs.swap(owned->tuv);
      s->tri_obj.swap(owned->tobj);
      s->tri_idx.swap(owned->tidx);
    } else if (n) {
      s->tri_vertices.assign(d->tri_vertices, d->tri_vertices + n * 9);
      s->tri_normals.assign(d->tri_normals, d->tri_normals + n * 9);
      s->tri_uvs.assign(d->tri_uvs, d->tri_uvs + n * 6);
      s->tri_obj.assign(d->tri_obj, d->tri_obj + n);
      s->tri_idx.assign(d->tri_idx, d->tri_idx + n);
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
      SphereRec r;
      std::memset(&r, 0, sizeof r);
      r.c[0] = d->spheres[i * 4 + 0];
      r.c[1] = d->spheres[i * 4 + 1];
      r.c[2] = d->spheres[i * 4 + 2];
      r.r = d->spheres[i * 4 + 3];
      r.obj = d->sphere_obj[i];
      s->spheres.push_back(r);
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
      HostTexture t;
      t.w = d->textures[i].width;
      t.h = d->textures[i].height;
      t.rgba.assign(d->textures[i].rgba, d->textures[i].rgba + (size_t)t.w * t.h * 4);
      s->textures.push_back(std::move(t));
    }
    if (tree == RT_TREE_HOST || d->n_triangles == 0) {  // a scene of spheres has no tree to wait for
      int rc = build_scene_accel(s.get());
      if (rc != RT_OK) return rc;
    }
  } catch (const std::bad_alloc&) {
    rt_set_error("rt_scene_create: out of host memory");
    return RT_ERR_OOM;
  }
  *out = s.release();
  return RT_OK;
}

// Scene::from_objects, src/main.rs:750-774: meshes are flattened in object order, then
// triangle order; the per-corner normals and texture coordinates that normal_at_hit and
// get_tex_uv_coords would look up (src/main.rs:526-529, 481-483) are gathered here once.
extern "C" int rt_scene_from_objects(const rt_objects* objs, rt_scene** out) {
  return rt_scene_from_objects_with(objs, RT_TREE_HOST, out);
}

static int scene_from_objects(const rt_objects* objs, uint32_t tree, rt_scene** out);
extern "C" int rt_scene_from_objects_with(const rt_objects* objs, uint32_t tree, rt_scene** out) {
  return rt_guard("rt_scene_from_objects", [&] { return scene_from_objects(objs, tree, out); });
}
static int scene_from_objects(const rt_objects* objs, uint32_t tree, rt_scene** out) {
  if (!objs || !out) {
    rt_set_error("rt_scene_from_objects: null argument");
    return RT_ERR_INVALID;
  }
  try {
    std::vector<int32_t> kinds;
    std::vector<rt_material> mats;
    std::vector<double> sph;
    std::vector<int32_t> sobj;
    const size_t n_obj = objs->items.size();
    std::vector<size_t> first(n_obj + 1, 0);  // flat index of every object's first triangle
    for (size_t oi = 0; oi < n_obj; ++oi) {
      const HostObject& o = objs->items[oi];
      kinds.push_back(o.kind);
      mats.push_back(o.mat);
      first[oi + 1] = first[oi];
      if (o.kind == RT_OBJ_SPHERE) {
        sph.insert(sph.end(), {o.centre[0], o.centre[1], o.centre[2], o.radius});
        sobj.push_back((int32_t)oi);
      } else {
        first[oi + 1] += o.mesh.tri_vertex.size() / 3;
      }
    }
    const size_t total = first[n_obj];
    OwnedTriangles own;
    own.tv.resize(total * 9);
    own.tn.resize(total * 9);
    own.tuv.resize(total * 6);
    own.tobj.resize(total);
    own.tidx.resize(total);
    parallel_ranges(n_obj, 16, [&](size_t, size_t begin, size_t end) {
      for (size_t oi = begin; oi < end; ++oi) {
        const HostObject& o = objs->items[oi];
        if (o.kind == RT_OBJ_SPHERE) continue;
        const rt_mesh& m = o.mesh;
        const size_t nt = m.tri_vertex.size() / 3;
        for (size_t t = 0; t < nt; ++t) {
          const size_t f = first[oi] + t;
          for (int k = 0; k < 3; ++k) {
            const D3& v = m.vertices[m.tri_vertex[t * 3 + k]];
            const D3& nn = m.vertex_normals[m.tri_vertex[t * 3 + k]];
            double* pv = &own.tv[f * 9 + k * 3];
            double* pn = &own.tn[f * 9 + k * 3];
            pv[0] = v.x, pv[1] = v.y, pv[2] = v.z;
            pn[0] = nn.x, pn[1] = nn.y, pn[2] = nn.z;
            const uint64_t ui = m.tri_uv[t * 3 + k];
            own.tuv[f * 6 + k * 2] = m.uvs[ui * 2];
            own.tuv[f * 6 + k * 2 + 1] = m.uvs[ui * 2 + 1];
          }
          own.tobj[f] = (int32_t)oi;
          own.tidx[f] = (int32_t)t;
        }
      }
    });
    std::vector<rt_texture> tex;
    for (const HostTexture& t : objs->textures) tex.push_back(rt_texture{t.rgba.data(), t.w, t.h});
    rt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_objects = (uint32_t)kinds.size();
    d.object_kind = kinds.data();
    d.materials = mats.data();
    d.n_triangles = total;
    d.tri_vertices = own.tv.data();
    d.tri_normals = own.tn.data();
    d.tri_uvs = own.tuv.data();
    d.tri_obj = own.tobj.data();
    d.tri_idx = own.tidx.data();
    d.n_spheres = (uint32_t)sobj.size();
    d.spheres = sph.data();
    d.sphere_obj = sobj.data();
    d.n_textures = (uint32_t)tex.size();
    d.textures = tex.data();
    return scene_create(&d, tree, out, &own);
  } catch (const std::bad_alloc&) {
    rt_set_error("rt_scene_from_objects: out of host memory");
    return RT_ERR_OOM;
  }
}

extern "C" void rt_scene_destroy(rt_scene* scene) {
  if (!scene) return;
  if (scene->dev) rt_device_scene_free(scene->dev);
  delete scene;
}

extern "C" int rt_scene_get_info(const rt_scene* s, rt_scene_info* out) {
  if (!s || !out) {
    rt_set_error("rt_scene_get_info: null argument");
    return RT_ERR_INVALID;
  }
  std::memset(out, 0, sizeof *out);
  out->n_triangles = s->tri_obj.size();
  out->n_spheres = (uint32_t)s->spheres.size();
  out->n_objects = (uint32_t)s->object_kind.size();
  out->n_textures = (uint32_t)s->textures.size();
  out->bvh_nodes = (uint32_t)s->nodes.size();
  out->bvh_leaves = s->bvh_leaves;
  out->bvh_depth = s->bvh_depth;
  out->bvh_spheres = s->has_tree ? (uint32_t)s->tree_spheres.size() : 0u;
  if (s->dev) rt_device_scene_info(s->dev, &out->uploaded, &out->device_bytes);
  return RT_OK;
}

extern "C" int rt_scene_export_flat(const rt_scene* s, double* tv, double* tn, double* tuv, int32_t* tobj, int32_t* tidx) {
  if (!s) {
    rt_set_error("rt_scene_export_flat: null scene");
    return RT_ERR_INVALID;
  }
  const size_t n = s->tri_obj.size();
  if (n == 0) return RT_OK;
  if (tv) std::memcpy(tv, s->tri_vertices.data(), n * 9 * sizeof(double));
  if (tn) std::memcpy(tn, s->tri_normals.data(), n * 9 * sizeof(double));
  if (tuv) std::memcpy(tuv, s->tri_uvs.data(), n * 6 * sizeof(double));
  if (tobj) std::memcpy(tobj, s->tri_obj.data(), n * sizeof(int32_t));
  if (tidx) std::memcpy(tidx, s->tri_idx.data(), n * sizeof(int32_t));
  return RT_OK;
}

extern "C" int rt_scene_export_bvh(const rt_scene* s, uint32_t* node_words, int32_t* leaf_order) {
  if (!s) {
    rt_set_error("rt_scene_export_bvh: null scene");
    return RT_ERR_INVALID;
  }
  if (!s->has_tree) {
    rt_set_error("rt_scene_export_bvh: the scene was created without a tree");
    return RT_ERR_INVALID;
  }
  if (node_words && !s->nodes.empty()) std::memcpy(node_words, s->nodes.data(), s->nodes.size() * sizeof(BvhNode));
  if (leaf_order && !s->leaf_order.empty()) std::memcpy(leaf_order, s->leaf_order.data(), s->leaf_order.size() * sizeof(int32_t));
  return RT_OK;
}

// The tree once more on a 16-bit grid over its own bounds (BvhNodeQ): per axis a plane is
// lo + q * step with q in 0..65535. Returns false, and the kernels then walk the binary32 nodes
// everywhere, unless every grid box provably contains the binary32 box it stands for with a
// margin that covers the traversal's own binary32 arithmetic:
//   * a minimum plane lies at least one whole step below, a maximum plane at least one whole step
//     above the binary32 plane (checked plane by plane, nothing is clamped into range);
//   * the kernels form a plane distance as q * (step * inv) + (lo * inv - o * inv), four binary32
//     roundings that each move the plane by at most 2^-24 of the scene's largest coordinate; the
//     grid is only used when that coordinate is at most 2^20 steps (about sixteen times the
//     scene's extent along the axis), which keeps their sum below a quarter of a step. A scene far
//     from the origin relative to its size fails this and keeps its binary32 nodes.
bool rt_quantize_tree(const rt_scene* s, std::vector<BvhNodeQ>* out, float q_lo[3], float q_step[3]) {
  out->clear();
  for (int a = 0; a < 3; ++a) {
    q_lo[a] = 0.f;
    q_step[a] = 1.f;
  }
  const size_t n = s->nodes.size();
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (const BvhNode& nd : s->nodes) {
    const float* planes[3] = {nd.bx, nd.by, nd.bz};
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < 4; ++k) {
        const float v = planes[a][k];
        if (!std::isfinite(v)) return false;  // a box that reaches infinity has no place on a finite grid
        lo[a] = std::min(lo[a], v);
        hi[a] = std::max(hi[a], v);
      }
  }
  out->resize(n);
  for (int a = 0; a < 3; ++a) {
    if (!(lo[a] <= hi[a])) lo[a] = hi[a] = 0.f;  // an empty tree
    const double extent = (double)hi[a] - (double)lo[a];
    // q = 1 .. 65533 span [lo, hi]; 0 is the outward step of the lowest plane, 65534 and 65535 those of the highest
    double step = extent / 65532.0;
    const double bound = std::max(std::fabs((double)lo[a]), std::fabs((double)hi[a]));
    if (!(step > 0.0)) step = std::max(bound * 0x1p-18, 1e-30);  // a flat axis: any step the bound allows
    float stepf = (float)step;
    while ((double)stepf * 65532.0 < extent) stepf = std::nextafterf(stepf, INFINITY);
    if (!std::isfinite(stepf) || !(stepf > 0.f) || (double)stepf < 1e-30) return false;
    if (bound + (double)stepf > (double)stepf * 0x1p20) return false;
    float lof = (float)((double)lo[a] - (double)stepf);
    while ((double)lof + (double)stepf > (double)lo[a]) lof = std::nextafterf(lof, -INFINITY);
    if (!std::isfinite(lof)) return false;
    q_lo[a] = lof;
    q_step[a] = stepf;
    const double l = lof, st = stepf;
    for (size_t i = 0; i < n; ++i) {
      const BvhNode& nd = s->nodes[i];
      const float* planes = a == 0 ? nd.bx : a == 1 ? nd.by : nd.bz;
      for (int k = 0; k < 4; ++k) {
        const bool is_max = (k & 1) != 0;
        const double v = planes[k];
        long q;
        if (is_max) {
          q = (long)std::ceil((v - l) / st) + 1;
          while (l + (double)(q - 1) * st < v) ++q;
        } else {
          q = (long)std::floor((v - l) / st) - 1;
          while (l + (double)(q + 1) * st > v) --q;
        }
        if (q < 0 || q > 65535) return false;
        (*out)[i].q[a * 4 + k] = (uint16_t)q;
      }
    }
  }
  for (size_t i = 0; i < n; ++i) {
    (*out)[i].child[0] = s->nodes[i].child[0];
    (*out)[i].child[1] = s->nodes[i].child[1];
  }
  return true;
}

// Copies the 16-bit grid form of the tree out: nodes as bvh_nodes x 8 32-bit words (BvhNodeQ),
// lo_step = the grid's {lo x, y, z, step x, y, z}; *usable = 0 (and nothing copied) when the scene
// does not qualify for the grid and the kernels walk the binary32 nodes instead.
extern "C" int rt_scene_export_grid_bvh(const rt_scene* s, uint32_t* node_words, float* lo_step, int32_t* usable) {
  return rt_guard("rt_scene_export_grid_bvh", [&] {
    if (!s || !usable) {
      rt_set_error("rt_scene_export_grid_bvh: null argument");
      return RT_ERR_INVALID;
    }
    if (!s->has_tree) {
      rt_set_error("rt_scene_export_grid_bvh: the scene was created without a tree");
      return RT_ERR_INVALID;
    }
    std::vector<BvhNodeQ> qn;
    float lo[3], st[3];
    *usable = rt_quantize_tree(s, &qn, lo, st) ? 1 : 0;
    if (*usable) {
      if (node_words && !qn.empty()) std::memcpy(node_words, qn.data(), qn.size() * sizeof(BvhNodeQ));
      if (lo_step)
        for (int a = 0; a < 3; ++a) {
          lo_step[a] = lo[a];
          lo_step[3 + a] = st[a];
        }
    }
    return RT_OK;
  });
}

extern "C" int rt_scene_export_objects(const rt_scene* s, int32_t* object_kind, rt_material* materials, double* spheres,
                                       int32_t* sphere_obj) {
  if (!s) {
    rt_set_error("rt_scene_export_objects: null scene");
    return RT_ERR_INVALID;
  }
  const size_t n = s->object_kind.size();
  if (object_kind && n) std::memcpy(object_kind, s->object_kind.data(), n * sizeof(int32_t));
  if (materials && n) std::memcpy(materials, s->materials.data(), n * sizeof(rt_material));
  for (size_t i = 0; i < s->spheres.size(); ++i) {
    const SphereRec& r = s->spheres[i];
    if (spheres) {
      spheres[i * 4 + 0] = r.c[0];
      spheres[i * 4 + 1] = r.c[1];
      spheres[i * 4 + 2] = r.c[2];
      spheres[i * 4 + 3] = r.r;
    }
    if (sphere_obj) sphere_obj[i] = r.obj;
  }
  return RT_OK;
}

extern "C" int rt_scene_export_texture(const rt_scene* s, uint32_t id, uint32_t* width, uint32_t* height, uint8_t* rgba) {
  if (!s || id >= s->textures.size()) {
    rt_set_error("rt_scene_export_texture: null scene or texture id out of range");
    return RT_ERR_INVALID;
  }
  const HostTexture& t = s->textures[id];
  if (width) *width = t.w;
  if (height) *height = t.h;
  if (rgba) std::memcpy(rgba, t.rgba.data(), t.rgba.size());
  return RT_OK;
}

// impl Into<Rgba<u8>> for RGBA, src/main.rs:29-37, on canvas / its (src/main.rs:1109).
extern "C" int rt_quantize_rgba8(const double* canvas, uint32_t w, uint32_t h, uint32_t its, uint8_t* out) {
  if (!canvas || !out || its == 0) {
    rt_set_error("rt_quantize_rgba8: null argument or its == 0");
    return RT_ERR_INVALID;
  }
  const double n = (double)its;
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    for (int c = 0; c < 3; ++c) {
      double x = (canvas[i * 3 + c] / n) * 255.0;
      // f64::clamp(0, 255) then `as u8` (truncation; NaN -> 0)
      if (x < 0.0) x = 0.0;
      if (x > 255.0) x = 255.0;
      out[i * 4 + c] = (x == x) ? (uint8_t)x : 0;
    }
    out[i * 4 + 3] = 255;
  }
  return RT_OK;
}
