// rt_oracle.cpp — CPU restatement of the reference's render hot path.  TEST INFRASTRUCTURE ONLY.
//
// Nothing under raytracer_amd/ may include, link, load or call this file; it exists so that
// tests, the smoke check and the bench's cpu_baseline leg have something to compare the HIP
// path against.
//
// PARITY UNPINNED: the reference (skius/raytracer) has no tests, no golden vectors and cannot
// be built in this environment (no Rust toolchain, un-vendored crates, needs a window). This
// file follows src/main.rs function by function, in binary64 with no fused multiply-add
// (compile with -ffp-contract=off), and is pinned only by (1) analytic known-answer vectors
// derived from the reference's formulas and (2) a block-mean comparison with the screenshot
// images/20220206_170941.png the reference's authors committed (mean absolute difference 0.003
// at 16 samples per pixel, no brightness bias; tests/test_oracle_kat.py).
//
// Two deliberate substitutions, both forced:
//   * rand::thread_rng() (src/main.rs:266-267, 859, 871-873, 956, 995-997, 1007-1008) is an
//     OS-seeded, non-reproducible generator. It is replaced by a counter-based generator keyed
//     on (seed, pixel, sample, stage, draw) — see rng_* below. Draw order follows the reference.
//   * crate bvh 0.6.0 (src/main.rs:771, 777-780) is not in the repository. HIT_BRUTE tests every
//     flat triangle (the exact f64 answer, independent of any tree); HIT_REFLIKE models the
//     crate's published behaviour (f32 boxes rounded to nearest, f32-normalised ray, every
//     touched leaf is a candidate, no distance pruning) and is used for the CPU baseline timing.
//
// The unbounded rejection loops of src/main.rs:265-278 and 876-883 are capped at
// REJECT_CAP tries with a defined fallback, identically here and in the HIP kernels.

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt_amd.h"  // POD descriptions only (rt_scene_desc, rt_material, rt_hit, ...)

namespace {

// ------------------------------------------------------------------------------------------
// L0 math — src/main.rs:40-183
// ------------------------------------------------------------------------------------------
struct Vec3 {
  double c[3];
};
struct Vec3H {
  double c[4];
};
struct Mat3H {
  double m[4][4];
};

inline Vec3 V(double x, double y, double z) { return Vec3{{x, y, z}}; }

// src/main.rs:69-71
inline double dot(const Vec3& a, const Vec3& b) { return a.c[0] * b.c[0] + a.c[1] * b.c[1] + a.c[2] * b.c[2]; }
// src/main.rs:73-75
inline double length(const Vec3& a) { return std::sqrt(a.c[0] * a.c[0] + a.c[1] * a.c[1] + a.c[2] * a.c[2]); }
// src/main.rs:77-85
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return V(a.c[1] * b.c[2] - a.c[2] * b.c[1], a.c[2] * b.c[0] - a.c[0] * b.c[2], a.c[0] * b.c[1] - a.c[1] * b.c[0]);
}
// src/main.rs:87-90
inline Vec3 normalize(const Vec3& a) {
  double l = length(a);
  return V(a.c[0] / l, a.c[1] / l, a.c[2] / l);
}
// src/main.rs:93-113
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return V(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return V(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
inline Vec3 operator*(double s, const Vec3& r) { return V(s * r.c[0], s * r.c[1], s * r.c[2]); }

// src/main.rs:65-67, 116-118
inline Vec3H to_homogeneous(const Vec3& a) { return Vec3H{{a.c[0], a.c[1], a.c[2], 1.0}}; }
inline Vec3 homogeneous_divide(const Vec3H& a) { return V(a.c[0] / a.c[3], a.c[1] / a.c[3], a.c[2] / a.c[3]); }

// src/main.rs:122-153
Mat3H rot_y(double theta) {
  double c = std::cos(theta), s = std::sin(theta);
  return Mat3H{{{c, 0.0, s, 0.0}, {0.0, 1.0, 0.0, 0.0}, {-s, 0.0, c, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
}
Mat3H rot_z(double theta) {
  double c = std::cos(theta), s = std::sin(theta);
  return Mat3H{{{c, s, 0.0, 0.0}, {-s, c, 0.0, 0.0}, {0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
}
[[maybe_unused]] Mat3H rot_x(double theta) {
  double c = std::cos(theta), s = std::sin(theta);
  return Mat3H{{{1.0, 0.0, 0.0, 0.0}, {0.0, c, -s, 0.0}, {0.0, s, c, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
}
// src/main.rs:156-167
inline Vec3H operator*(const Mat3H& a, const Vec3H& o) {
  Vec3H r;
  for (int i = 0; i < 4; ++i)
    r.c[i] = a.m[i][0] * o.c[0] + a.m[i][1] * o.c[1] + a.m[i][2] * o.c[2] + a.m[i][3] * o.c[3];
  return r;
}

// ------------------------------------------------------------------------------------------
// L1 primitives — src/main.rs:185-262
// ------------------------------------------------------------------------------------------
const double epsilon = 0.00000001;  // src/main.rs:185

// src/main.rs:187-239. Returns true and (dist,u,v) on a hit.
bool intersects_triangle(const Vec3& origin, const Vec3& dir, const Vec3& a, const Vec3& b, const Vec3& c,
                         double* out_t, double* out_u, double* out_v) {
  Vec3 a_to_b = b - a;
  Vec3 a_to_c = c - a;
  Vec3 u_vec = cross(dir, a_to_c);
  double det = dot(a_to_b, u_vec);
  if (det < epsilon && det > -epsilon) return false;  // :205
  double inv_det = 1.0 / det;
  Vec3 a_to_origin = origin - a;
  double u = dot(a_to_origin, u_vec) * inv_det;
  if (!(u >= 0.0 && u <= 1.0)) return false;  // :218  !(0.0..=1.0).contains(&u)
  Vec3 v_vec = cross(a_to_origin, a_to_b);
  double v = dot(dir, v_vec) * inv_det;
  if (v < 0.0 || u + v > 1.0) return false;  // :228
  double dist = dot(a_to_c, v_vec) * inv_det;
  if (dist > epsilon) {  // :234
    *out_t = dist;
    *out_u = u;
    *out_v = v;
    return true;
  }
  return false;
}

// src/main.rs:241-262
bool intersects_sphere(const Vec3& center, double radius, const Vec3& orig, const Vec3& dir, double* out_t) {
  Vec3 oc = orig - center;
  double a = dot(dir, dir);
  double half_b = dot(oc, dir);
  double c = dot(oc, oc) - radius * radius;
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  double sqrtd = std::sqrt(discriminant);
  double root1 = (-half_b - sqrtd) / a;
  if (root1 < epsilon) {
    double root2 = (-half_b + sqrtd) / a;
    if (root2 < epsilon) return false;
    *out_t = root2;
    return true;
  }
  *out_t = root1;
  return true;
}

// ------------------------------------------------------------------------------------------
// Counter-based generator (replaces rand::thread_rng)
// ------------------------------------------------------------------------------------------
// A path is keyed by (seed, pixel index, sample index); within a path every stage has its own
// draw counter: stage 0 is the camera (src/main.rs:995-1008), stage d+1 is the shading of the
// hit found at recursion depth d (src/main.rs:859-883, 956). A draw is
//   mix64(key + GOLDEN * ((stage << 20) + index + 1))
// with the splitmix64 finaliser as mix64, mapped to [0,1) with 52 bits.
const uint64_t GOLDEN = 0x9e3779b97f4a7c15ull;
const int REJECT_CAP = 64;  // tries per rejection loop before the defined fallback

inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
inline uint64_t rng_path_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
  uint64_t k = mix64(seed + GOLDEN);
  k = mix64(k + GOLDEN * (pixel + 1));
  k = mix64(k + GOLDEN * (sample + 1));
  return k;
}
const uint32_t JITTER_STREAM = 1u << 19;  // first draw index of a shading stage's second stream (a stage has 2^20 indices)
struct Rng {
  uint64_t key;
  uint64_t stage_base;
  uint32_t index;
  void start(uint64_t k, uint32_t stage) {
    key = k;
    stage_base = (uint64_t)stage << 20;
    index = 0;
  }
  double u01() {
    uint64_t z = mix64(key + GOLDEN * (stage_base + (uint64_t)index + 1));
    index += 1;
    // 52 random bits as the mantissa of a double in [1,2), minus 1 (as rand's f64 sampling does)
    uint64_t bits = 0x3ff0000000000000ull | (z >> 12);
    double d;
    std::memcpy(&d, &bits, 8);
    return d - 1.0;
  }
  // rng.gen_range(lo..hi) / gen_range(lo..=hi) — uniform on the interval.
  double range(double lo, double hi) { return lo + (hi - lo) * u01(); }
  // The two gen_range(-1.0..1.0) calls of the Marsaglia loop (src/main.rs:266-267) from ONE
  // draw: its high and low 32 bits, each as the top mantissa bits of a double in [1,2).
  void pair_pm1(double* a, double* b) {
    uint64_t z = mix64(key + GOLDEN * (stage_base + (uint64_t)index + 1));
    index += 1;
    uint64_t ba = 0x3ff0000000000000ull | ((z >> 32) << 20), bb = 0x3ff0000000000000ull | ((z & 0xffffffffull) << 20);
    double da, db;
    std::memcpy(&da, &ba, 8);
    std::memcpy(&db, &bb, 8);
    *a = -1.0 + 2.0 * (da - 1.0);
    *b = -1.0 + 2.0 * (db - 1.0);
  }
};

// src/main.rs:264-279 (Marsaglia), capped.
Vec3 random_unit_sphere_surface(Rng& r) {
  for (int tries = 0; tries < REJECT_CAP; ++tries) {
    double x1, x2;
    r.pair_pm1(&x1, &x2);
    if (x1 * x1 + x2 * x2 >= 1.0) continue;
    double sq = std::sqrt(1.0 - x1 * x1 - x2 * x2);
    double x = 2.0 * x1 * sq;
    double y = 2.0 * x2 * sq;
    double z = 1.0 - 2.0 * (x1 * x1 + x2 * x2);
    return V(x, y, z);
  }
  return V(0.0, 0.0, 1.0);  // fallback after REJECT_CAP rejections (probability (1-pi/4)^64)
}

// ------------------------------------------------------------------------------------------
// L2 sky and optics — src/main.rs:281-333
// ------------------------------------------------------------------------------------------
// src/main.rs:281-314; the first gradient (:282-294) is dead code in the reference.
void interp_sky(const Vec3& dir, double out[3]) {
  Vec3 c1 = V(0.4, 0.6, 0.8);
  Vec3 c2 = V(0.8, 0.8, 0.8);
  Vec3 light_c = V(1.5, 1.5, 1.5);
  Vec3 light_dir = normalize(V(1.0, 2.0, 1.0));
  double t = dot(dir, light_dir);
  t = std::pow((t + 1.0) / 2.0, 6.0);
  Vec3 res = (1.0 - t) * (dir.c[1] * c2 + (1.0 - dir.c[1]) * c1) + t * light_c;
  out[0] = res.c[0];
  out[1] = res.c[1];
  out[2] = res.c[2];
}

// src/main.rs:320-327
Vec3 refract(const Vec3& uv_in, const Vec3& n, double etai_over_etat) {
  Vec3 uv = -1.0 * uv_in;
  double cos_theta = std::fmin(dot(-1.0 * uv, n), 1.0);
  Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
  Vec3 r_out_parallel = -std::sqrt(std::fabs(1.0 - dot(r_out_perp, r_out_perp))) * n;
  return r_out_perp + r_out_parallel;
}

// f64::powi(5) — compiler-rt __powidf2: square-and-multiply from the low bit.
inline double powi5(double a) {
  double r = a;        // bit 0 of 5
  double a2 = a * a;   // bit 1 is 0
  double a4 = a2 * a2; // bit 2 is 1
  return r * a4;
}
// src/main.rs:329-333
double reflectance(double cosine, double ref_idx) {
  double r = (1.0 - ref_idx) / (1.0 + ref_idx);
  double r2 = r * r;
  return r2 + (1.0 - r2) * powi5(1.0 - cosine);
}

// ------------------------------------------------------------------------------------------
// L2b textures / materials — src/main.rs:335-427
// ------------------------------------------------------------------------------------------
struct Image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;
};

// `as u32` on an f64: saturating, NaN -> 0.
inline uint32_t f64_as_u32(double x) {
  if (!(x > 0.0)) return 0u;
  if (x >= 4294967295.0) return 4294967295u;
  return (uint32_t)x;
}
// image::get_pixel panics out of bounds (reachable only on exact u == 1 or v == 0 vertex hits);
// the build clamps to the last texel instead.
inline void get_pixel(const Image& im, uint32_t x, uint32_t y, uint8_t px[4]) {
  if (x >= im.w) x = im.w - 1;
  if (y >= im.h) y = im.h - 1;
  const uint8_t* p = &im.rgba[((size_t)y * im.w + x) * 4];
  px[0] = p[0];
  px[1] = p[1];
  px[2] = p[2];
  px[3] = p[3];
}

struct MaterialO {
  int kind = RT_MAT_OPAQUE;
  int color_tex = -1;   // ColorKind::Texture(id) or SolidColor
  int normal_tex = -1;  // NormalKind::NormalMap(id) or SimpleNormal
  double metallic = 0, diffuse = 0, how_metallic = 0;
  Vec3 emission = V(0, 0, 0);
  Vec3 color = V(0, 0, 0);
  double ior = 1.5;
};

// NormalKind::normal_at — src/main.rs:341-381
Vec3 normal_at(const MaterialO& m, const std::vector<Image>& tex, const Vec3& world_normal, double tex_u, double tex_v) {
  if (m.normal_tex < 0) return world_normal;
  const Image& nm = tex[m.normal_tex];
  uint32_t text_x = f64_as_u32(std::floor(tex_u * (double)nm.w));
  uint32_t text_y = f64_as_u32(std::floor((1.0 - tex_v) * (double)nm.h));
  uint8_t px[4];
  get_pixel(nm, text_x, text_y, px);
  double r = (double)px[0] / 255.0, g = (double)px[1] / 255.0, b = (double)px[2] / 255.0;
  Vec3 ts_normal = normalize(V(r, g, b));
  Vec3 t = cross(world_normal, V(0.0, 1.0, 0.0));
  if (std::fabs(length(t)) < epsilon) t = cross(world_normal, V(0.0, 0.0, 1.0));
  t = normalize(t);
  Vec3 bt = cross(world_normal, t);
  Vec3 map_n = 2.0 * ts_normal - V(1.0, 1.0, 1.0);
  Mat3H tbn = Mat3H{{{t.c[0], bt.c[0], world_normal.c[0], 0.0},
                     {t.c[1], bt.c[1], world_normal.c[1], 0.0},
                     {t.c[2], bt.c[2], world_normal.c[2], 0.0},
                     {0.0, 0.0, 0.0, 1.0}}};
  return normalize(homogeneous_divide(tbn * to_homogeneous(map_n)));
}

// ColorKind::color_at — src/main.rs:392-411
Vec3 color_at(const MaterialO& m, const std::vector<Image>& tex, double u, double v) {
  if (m.color_tex < 0) return m.color;
  const Image& img = tex[m.color_tex];
  uint32_t text_x = f64_as_u32(std::floor(u * (double)img.w));
  uint32_t text_y = f64_as_u32(std::floor((1.0 - v) * (double)img.h));
  uint8_t px[4];
  get_pixel(img, text_x, text_y, px);
  return V((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
}

// ------------------------------------------------------------------------------------------
// L3 scene — src/main.rs:429-815
// ------------------------------------------------------------------------------------------
struct PolygonMesh {  // src/main.rs:566-574
  std::vector<Vec3> vertices;
  std::vector<Vec3> vertice_normals;
  std::vector<std::array<double, 2>> uv_coords;
  std::vector<std::array<size_t, 3>> triangles;
  std::vector<std::array<size_t, 3>> triangle_to_v_to_uv_idx;  // HashMap<usize,[usize;3]> keyed 0..n
};

void mesh_rotate(PolygonMesh& m, const Mat3H& rot) {  // src/main.rs:627-658
  for (auto& v : m.vertices) v = homogeneous_divide(rot * to_homogeneous(v));
  for (auto& n : m.vertice_normals) n = homogeneous_divide(rot * to_homogeneous(n));
}
void mesh_translate(PolygonMesh& m, const Vec3& t) {  // src/main.rs:660-664
  for (auto& v : m.vertices) v = v + t;
}

struct Object {  // src/main.rs:429-448
  int kind = RT_OBJ_MESH;
  PolygonMesh mesh;
  Vec3 center = V(0, 0, 0);
  double radius = 0;
  MaterialO material;
};

struct FlatTriangle {  // src/main.rs:576-582
  Vec3 vertices[3];
  size_t triangle_idx;
  size_t obj_idx;
};

struct Hit {  // src/main.rs:668-681
  bool some = false;
  bool sphere = false;
  double distance = 0, u = 0, v = 0;
  size_t triangle_idx = 0, obj_idx = 0;
};

// src/main.rs:741-745
inline double bary_interp(double a, double b, double c, double u, double v) { return (1.0 - u - v) * a + u * b + v * c; }
inline Vec3 bary_interp(const Vec3& a, const Vec3& b, const Vec3& c, double u, double v) {
  return (1.0 - u - v) * a + u * b + v * c;
}

// --- model of crate bvh 0.6.0 (behavioural; not pinned) ---
struct Aabb32 {
  float lo[3], hi[3];
};
struct RefNode {
  int left = -1, right = -1;  // child node indices; leaf when shape >= 0
  int shape = -1;
  Aabb32 lbox, rbox;
};

enum HitMode { HIT_BRUTE = 0, HIT_REFLIKE = 1 };

struct Scene {  // src/main.rs:747
  std::vector<Object> objects;
  std::vector<FlatTriangle> triangles;
  std::vector<Image> textures;
  std::vector<RefNode> ref_nodes;  // built on first HIT_REFLIKE use
  int ref_root = -1;
};

Aabb32 tri_aabb(const FlatTriangle& t) {  // src/main.rs:594-623: min/max in f64, then `as f32`
  Aabb32 b;
  for (int k = 0; k < 3; ++k) {
    double lo = t.vertices[0].c[k];
    lo = std::fmin(lo, t.vertices[1].c[k]);
    lo = std::fmin(lo, t.vertices[2].c[k]);
    double hi = t.vertices[0].c[k];
    hi = std::fmax(hi, t.vertices[1].c[k]);
    hi = std::fmax(hi, t.vertices[2].c[k]);
    b.lo[k] = (float)lo;
    b.hi[k] = (float)hi;
  }
  return b;
}
inline Aabb32 aabb_empty() {
  Aabb32 b;
  for (int k = 0; k < 3; ++k) {
    b.lo[k] = std::numeric_limits<float>::infinity();
    b.hi[k] = -std::numeric_limits<float>::infinity();
  }
  return b;
}
inline void aabb_grow(Aabb32& a, const Aabb32& o) {
  for (int k = 0; k < 3; ++k) {
    a.lo[k] = std::min(a.lo[k], o.lo[k]);
    a.hi[k] = std::max(a.hi[k], o.hi[k]);
  }
}
inline float aabb_area(const Aabb32& a) {
  float dx = a.hi[0] - a.lo[0], dy = a.hi[1] - a.lo[1], dz = a.hi[2] - a.lo[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}

int ref_build(Scene& s, std::vector<int>& idx, int lo, int hi, const std::vector<Aabb32>& boxes, Aabb32* out_box) {
  int me = (int)s.ref_nodes.size();
  s.ref_nodes.emplace_back();
  if (hi - lo == 1) {
    s.ref_nodes[me].shape = idx[lo];
    *out_box = boxes[idx[lo]];
    return me;
  }
  // centroid bounds, widest axis
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = lo; i < hi; ++i)
    for (int k = 0; k < 3; ++k) {
      float c = 0.5f * (boxes[idx[i]].lo[k] + boxes[idx[i]].hi[k]);
      clo[k] = std::min(clo[k], c);
      chi[k] = std::max(chi[k], c);
    }
  int axis = 0;
  for (int k = 1; k < 3; ++k)
    if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
  float extent = chi[axis] - clo[axis];
  int mid;
  if (extent < 1e-5f) {
    mid = lo + (hi - lo) / 2;
  } else {
    const int NB = 6;
    Aabb32 bb[NB];
    int cnt[NB];
    for (int b = 0; b < NB; ++b) {
      bb[b] = aabb_empty();
      cnt[b] = 0;
    }
    auto bucket_of = [&](int t) {
      float c = 0.5f * (boxes[t].lo[axis] + boxes[t].hi[axis]);
      int b = (int)((c - clo[axis]) / extent * (float)NB);
      return std::min(std::max(b, 0), NB - 1);
    };
    for (int i = lo; i < hi; ++i) {
      int b = bucket_of(idx[i]);
      cnt[b]++;
      aabb_grow(bb[b], boxes[idx[i]]);
    }
    float best = INFINITY;
    int best_split = 0;
    for (int sp = 0; sp < NB - 1; ++sp) {
      Aabb32 l = aabb_empty(), r = aabb_empty();
      int nl = 0, nr = 0;
      for (int b = 0; b <= sp; ++b) {
        aabb_grow(l, bb[b]);
        nl += cnt[b];
      }
      for (int b = sp + 1; b < NB; ++b) {
        aabb_grow(r, bb[b]);
        nr += cnt[b];
      }
      if (nl == 0 || nr == 0) continue;
      float cost = (float)nl * aabb_area(l) + (float)nr * aabb_area(r);
      if (cost < best) {
        best = cost;
        best_split = sp;
      }
    }
    auto first_right = std::stable_partition(idx.begin() + lo, idx.begin() + hi,
                                             [&](int t) { return bucket_of(t) <= best_split; });
    mid = (int)(first_right - idx.begin());
    if (mid == lo || mid == hi) mid = lo + (hi - lo) / 2;
  }
  Aabb32 lb, rb;
  int l = ref_build(s, idx, lo, mid, boxes, &lb);
  int r = ref_build(s, idx, mid, hi, boxes, &rb);
  s.ref_nodes[me].left = l;
  s.ref_nodes[me].right = r;
  s.ref_nodes[me].lbox = lb;
  s.ref_nodes[me].rbox = rb;
  *out_box = lb;
  aabb_grow(*out_box, rb);
  return me;
}

void ref_ensure_built(Scene& s) {
  if (s.ref_root >= 0 || s.triangles.empty()) return;
  std::vector<Aabb32> boxes(s.triangles.size());
  for (size_t i = 0; i < s.triangles.size(); ++i) boxes[i] = tri_aabb(s.triangles[i]);
  std::vector<int> idx(s.triangles.size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
  Aabb32 root_box;
  s.ref_nodes.reserve(2 * idx.size());
  s.ref_root = ref_build(s, idx, 0, (int)idx.size(), boxes, &root_box);
}

struct Ray32 {
  float o[3], inv[3];
  int sign[3];
};
inline bool ray_box(const Ray32& r, const Aabb32& b) {
  float tmin = -INFINITY, tmax = INFINITY;
  for (int k = 0; k < 3; ++k) {
    float near = r.sign[k] ? b.hi[k] : b.lo[k];
    float far = r.sign[k] ? b.lo[k] : b.hi[k];
    float t0 = (near - r.o[k]) * r.inv[k];
    float t1 = (far - r.o[k]) * r.inv[k];
    if (t0 > tmax || t1 < tmin) return false;
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
  }
  return tmax > 0.0f;
}
void ref_traverse(const Scene& s, int node, const Ray32& r, std::vector<int>& out) {
  const RefNode& n = s.ref_nodes[node];
  if (n.shape >= 0) {
    out.push_back(n.shape);
    return;
  }
  if (ray_box(r, n.lbox)) ref_traverse(s, n.left, r, out);
  if (ray_box(r, n.rbox)) ref_traverse(s, n.right, r, out);
}

// Scene::hit — src/main.rs:776-814
Hit scene_hit(const Scene& s, const Vec3& origin, const Vec3& dir, int mode, uint64_t* cand_counter) {
  double least_t = std::numeric_limits<double>::infinity(), least_u = 0.0, least_v = 0.0;
  size_t least_tri = 0, least_obj = 0;
  auto consider = [&](const FlatTriangle& ft) {
    double t, u, v;
    if (intersects_triangle(origin, dir, ft.vertices[0], ft.vertices[1], ft.vertices[2], &t, &u, &v)) {
      if (t < least_t) {  // :791 strict <, first minimal candidate wins
        least_t = t;
        least_u = u;
        least_v = v;
        least_tri = ft.triangle_idx;
        least_obj = ft.obj_idx;
      }
    }
  };
  if (mode == HIT_BRUTE) {
    for (const FlatTriangle& ft : s.triangles) consider(ft);
    if (cand_counter) *cand_counter += s.triangles.size();
  } else if (!s.triangles.empty()) {
    // :777-780 — narrow to f32, normalise the direction in f32, collect every touched leaf.
    Ray32 r;
    float d[3] = {(float)dir.c[0], (float)dir.c[1], (float)dir.c[2]};
    float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) {
      r.o[k] = (float)origin.c[k];
      float dk = d[k] / len;
      r.inv[k] = 1.0f / dk;
      r.sign[k] = dk < 0.0f ? 1 : 0;
    }
    static thread_local std::vector<int> cands;
    cands.clear();
    Aabb32 whole = s.ref_nodes[s.ref_root].shape >= 0 ? tri_aabb(s.triangles[s.ref_nodes[s.ref_root].shape]) : Aabb32();
    if (s.ref_nodes[s.ref_root].shape >= 0) {
      if (ray_box(r, whole)) cands.push_back(s.ref_nodes[s.ref_root].shape);
    } else {
      ref_traverse(s, s.ref_root, r, cands);
    }
    for (int t : cands) consider(s.triangles[t]);
    if (cand_counter) *cand_counter += cands.size();
  }

  Hit least;
  if (least_t != std::numeric_limits<double>::infinity()) {  // :795-799
    least.some = true;
    least.sphere = false;
    least.distance = least_t;
    least.u = least_u;
    least.v = least_v;
    least.triangle_idx = least_tri;
    least.obj_idx = least_obj;
  }
  // :802-810 — all spheres, in object order, strictly nearer replaces
  for (size_t obj_idx = 0; obj_idx < s.objects.size(); ++obj_idx) {
    const Object& obj = s.objects[obj_idx];
    if (obj.kind != RT_OBJ_SPHERE) continue;
    double t;
    if (intersects_sphere(obj.center, obj.radius, origin, dir, &t)) {
      double cur = least.some ? least.distance : std::numeric_limits<double>::infinity();
      if (t < cur) {
        least = Hit();
        least.some = true;
        least.sphere = true;
        least.distance = t;
        least.obj_idx = obj_idx;
      }
    }
  }
  return least;
}

// Object::get_tex_uv_coords — src/main.rs:471-492
void get_tex_uv_coords(const Object& obj, const Hit& hit, double* tu, double* tv) {
  if (!hit.sphere) {
    const PolygonMesh& pm = obj.mesh;
    const auto& m = pm.triangle_to_v_to_uv_idx[hit.triangle_idx];
    auto uv1 = pm.uv_coords[m[0]], uv2 = pm.uv_coords[m[1]], uv3 = pm.uv_coords[m[2]];
    *tu = bary_interp(uv1[0], uv2[0], uv3[0], hit.u, hit.v);
    *tv = bary_interp(uv1[1], uv2[1], uv3[1], hit.u, hit.v);
  } else {
    *tu = 0.0;
    *tv = 1.0;
  }
}

// Object::normal_at_hit — src/main.rs:515-562
Vec3 normal_at_hit(const Scene& s, const Object& obj, const Hit& hit, const Vec3& origin, const Vec3& dir) {
  if (!hit.sphere) {
    const PolygonMesh& pm = obj.mesh;
    const auto& tri = pm.triangles[hit.triangle_idx];
    Vec3 n1 = pm.vertice_normals[tri[0]], n2 = pm.vertice_normals[tri[1]], n3 = pm.vertice_normals[tri[2]];
    Vec3 normal = bary_interp(n1, n2, n3, hit.u, hit.v);  // :531 not normalised
    if (obj.material.kind == RT_MAT_OPAQUE) {
      double tu, tv;
      get_tex_uv_coords(obj, hit, &tu, &tv);  // :538-543 (same arithmetic)
      return normal_at(obj.material, s.textures, normal, tu, tv);
    }
    return normal;
  }
  Vec3 hit_point = origin + hit.distance * dir;
  return normalize(hit_point - obj.center);
}

// ------------------------------------------------------------------------------------------
// L4 integrator — src/main.rs:817-975
// ------------------------------------------------------------------------------------------
const double ANTI_SHADOW_ACNE_LENGTH = epsilon;  // :1041

struct TraceCtx {
  const Scene* scene;
  int mode;
  int max_depth;
  uint64_t key;  // path key
  uint64_t rays = 0;
  uint64_t cands = 0;
};

void raytrace(TraceCtx& cx, const Vec3& origin, const Vec3& dir, int depth, double out[3]) {
  if (depth > cx.max_depth) {  // :818
    out[0] = out[1] = out[2] = 0.0;
    return;
  }
  cx.rays += 1;
  Hit hit = scene_hit(*cx.scene, origin, dir, cx.mode, &cx.cands);
  if (!hit.some) {
    interp_sky(dir, out);  // :826
    return;
  }
  const Object& obj = cx.scene->objects[hit.obj_idx];
  Vec3 hit_pos = origin + hit.distance * dir;               // :831
  Vec3 normal = normal_at_hit(*cx.scene, obj, hit, origin, dir);  // :833
  Vec3 camera_dir = normalize(-1.0 * dir);                  // :835
  const MaterialO& mat = obj.material;
  Rng rng;
  rng.start(cx.key, (uint32_t)depth + 1);

  if (mat.kind == RT_MAT_OPAQUE) {
    double tex_u, tex_v;
    get_tex_uv_coords(obj, hit, &tex_u, &tex_v);                // :847
    Vec3 point = color_at(mat, cx.scene->textures, tex_u, tex_v);  // :848
    Vec3 r;
    if (dot(camera_dir, normal) < 0.0) {  // :851
      r = -1.0 * camera_dir;
    } else {
      r = 2.0 * dot(camera_dir, normal) * normal - camera_dir;
    }
    Vec3 out_orig0 = hit_pos;
    Vec3 reflected_camera = normalize(r);  // :857

    double reflected[3] = {0.0, 0.0, 0.0};
    // NUM_SAMPLES_PER_BOUNCE = 1 (:1045): the loop body runs once
    {
      // The metal jitter and the three unused draws take their uniforms from a stream of their own
      // (indices from JITTER_STREAM on), the hemisphere sample below from index 0 on: where a draw's
      // uniform comes from is this restatement's choice (the reference asks thread_rng), and with
      // two streams the hemisphere sample does not depend on how many pairs the jitter rejected.
      Rng jit = rng;
      jit.index = JITTER_STREAM;
      Vec3 m = mat.metallic * random_unit_sphere_surface(jit);  // :869
      // :871-873 three draws whose values are never used
      (void)jit.range(-mat.diffuse, mat.diffuse);
      (void)jit.range(-mat.diffuse, mat.diffuse);
      (void)jit.range(-mat.diffuse, mat.diffuse);

      Vec3 diffuse_out = V(0.0, 0.0, 0.0);
      bool found = false;
      for (int tries = 0; tries < REJECT_CAP; ++tries) {  // :876-883, capped
        Vec3 random_sphere = random_unit_sphere_surface(rng);
        if (dot(random_sphere, normal) > 0.0) {
          diffuse_out = random_sphere;
          found = true;
          break;
        }
      }
      (void)found;  // fallback: diffuse_out stays (0,0,0), so the sum below is `normal`
      diffuse_out = mat.diffuse * diffuse_out + normal;  // :884
      if (length(diffuse_out) < epsilon) {               // :885
        diffuse_out = normal;
      } else {
        diffuse_out = normalize(diffuse_out);
      }
      Vec3 fuzzy_out = normalize(m + reflected_camera);  // :896
      Vec3 out_dir = mat.how_metallic * fuzzy_out + (1.0 - mat.how_metallic) * diffuse_out;  // :900
      Vec3 out_orig = out_orig0 + ANTI_SHADOW_ACNE_LENGTH * out_dir;                         // :903
      double rr[3];
      raytrace(cx, out_orig, out_dir, depth + 1, rr);  // :905
      reflected[0] += rr[0];
      reflected[1] += rr[1];
      reflected[2] += rr[2];
    }
    reflected[0] /= 1.0;  // :911-913
    reflected[1] /= 1.0;
    reflected[2] /= 1.0;
    double material_loss = 0.0;  // :915
    out[0] = point.c[0] * (reflected[0] * (1.0 - material_loss) + mat.emission.c[0]);  // :925-929
    out[1] = point.c[1] * (reflected[1] * (1.0 - material_loss) + mat.emission.c[1]);
    out[2] = point.c[2] * (reflected[2] * (1.0 - material_loss) + mat.emission.c[2]);
    return;
  }

  // Dielectric — :936-969
  double index_of_refraction = mat.ior;
  bool front_face = dot(normal, camera_dir) > 0.0;
  Vec3 n = front_face ? normal : -1.0 * normal;
  double eta = front_face ? 1.0 / index_of_refraction : index_of_refraction;
  double cos_theta = std::fmin(dot(camera_dir, n), 1.0);
  double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
  bool cannot_refract = eta * sin_theta > 1.0;
  Vec3 ndir;
  // `||` short-circuits in the reference, so the draw only happens when refraction is possible;
  // with a counter-based generator the draw index is the same either way.
  if (cannot_refract || reflectance(cos_theta, eta) > rng.range(0.0, 1.0)) {
    Vec3 r = 2.0 * dot(camera_dir, n) * n - camera_dir;
    ndir = normalize(r);
  } else {
    ndir = normalize(refract(camera_dir, n, eta));
  }
  raytrace(cx, hit_pos + ANTI_SHADOW_ACNE_LENGTH * ndir, ndir, depth + 1, out);  // :966
}

// ------------------------------------------------------------------------------------------
// L5 frame driver — src/main.rs:977-1039
// ------------------------------------------------------------------------------------------
struct Camera {
  double tan_half, aspect;
  uint32_t w, h;
};
Camera make_camera(uint32_t w, uint32_t h, double fov_y_deg) {
  Camera c;
  double fov_y = fov_y_deg * 3.14159265358979323846 / 180.0;  // :978
  c.aspect = (double)w / (double)h;                            // :979
  c.tan_half = std::tan(fov_y / 2.0);
  c.w = w;
  c.h = h;
  return c;
}
// :996-1013 for one sample. `rng` must be started at stage 0.
void camera_ray(const Camera& cam, uint32_t u, uint32_t v, Rng& rng, bool jitter, double dof, Vec3* orig, Vec3* dir) {
  double x_offset = rng.range(-0.3, 0.3);
  double y_offset = rng.range(-0.3, 0.3);
  if (!jitter) {
    x_offset = 0.0;
    y_offset = 0.0;
  }
  double p_x = (2.0 * (((double)u + 0.5 + x_offset) / (double)cam.w) - 1.0) * cam.tan_half * cam.aspect;  // :999
  double p_y = (1.0 - 2.0 * (((double)v + 0.5 + y_offset) / (double)cam.h)) * cam.tan_half;               // :1000
  double orig_x_offset = rng.range(-dof, dof);  // :1007
  double orig_y_offset = rng.range(-dof, dof);  // :1008
  *orig = V(orig_x_offset, orig_y_offset, 0.0);
  Vec3 d = 2.4 * normalize(V(p_x, p_y, -1.0));  // :1011
  *dir = normalize(d - *orig);                  // :1013
}

bool pixel_in_tiles(const rt_render_params& p, uint32_t u, uint32_t v) {
  if (p.tile_size == 0 || p.tile_ranks <= 1) return true;
  uint32_t tiles_x = (p.width + p.tile_size - 1) / p.tile_size;
  uint32_t t = (v / p.tile_size) * tiles_x + (u / p.tile_size);
  return t % p.tile_ranks == p.tile_rank;
}

void render(const Scene& scene, const rt_render_params& p, double* canvas, uint64_t* rays_out, uint64_t* cands_out,
            int threads, int mode) {
  Camera cam = make_camera(p.width, p.height, p.fov_y_deg);
  const uint64_t npix = (uint64_t)p.width * p.height;
  std::atomic<uint64_t> next(0), rays(0), cands(0);
  const uint64_t CHUNK = 256;
  const double divisor = (double)(p.spp_total ? p.spp_total : p.spp);
  auto worker = [&]() {
    uint64_t my_rays = 0, my_cands = 0;
    for (;;) {
      uint64_t begin = next.fetch_add(CHUNK);
      if (begin >= npix) break;
      uint64_t end = std::min(begin + CHUNK, npix);
      for (uint64_t pix = begin; pix < end; ++pix) {
        uint32_t v = (uint32_t)(pix / p.width), u = (uint32_t)(pix % p.width);
        if (!pixel_in_tiles(p, u, v)) continue;
        double total[3] = {0.0, 0.0, 0.0};
        for (uint32_t s = 0; s < p.spp; ++s) {  // :994
          TraceCtx cx;
          cx.scene = &scene;
          cx.mode = mode;
          cx.max_depth = p.max_depth;
          cx.key = rng_path_key(p.seed, pix, (uint64_t)p.first_sample + s);
          Rng rng;
          rng.start(cx.key, 0);
          Vec3 orig, dir;
          camera_ray(cam, u, v, rng, !(p.flags & RT_FLAG_NO_JITTER), p.dof_strength, &orig, &dir);
          double c[3];
          raytrace(cx, orig, dir, 0, c);  // :1013
          // NUM_SAMPLES_FOR_DOF = 1: curr / 1 (:1019), then summed in sample order (:1020-1024)
          total[0] += c[0] / 1.0;
          total[1] += c[1] / 1.0;
          total[2] += c[2] / 1.0;
          my_rays += cx.rays;
          my_cands += cx.cands;
        }
        canvas[pix * 3 + 0] += total[0] / divisor;  // :1029-1037
        canvas[pix * 3 + 1] += total[1] / divisor;
        canvas[pix * 3 + 2] += total[2] / divisor;
      }
    }
    rays += my_rays;
    cands += my_cands;
  };
  if (threads <= 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }
  if (rays_out) *rays_out = rays.load();
  if (cands_out) *cands_out = cands.load();
}

// ------------------------------------------------------------------------------------------
// L7 assets and scenes — src/main.rs:1127-1450
// ------------------------------------------------------------------------------------------
// load_obj — src/main.rs:1130-1221
bool load_obj(const std::string& filename, PolygonMesh* out) {
  std::ifstream f(filename);
  if (!f) return false;
  PolygonMesh m;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::vector<std::string> parts;
    std::string tok;
    while (ss >> tok) parts.push_back(tok);
    if (parts.empty()) continue;  // (the reference would panic on a blank line; its assets have none)
    if (parts[0] == "v") {
      m.vertices.push_back(V(std::strtod(parts[1].c_str(), nullptr), std::strtod(parts[2].c_str(), nullptr),
                             std::strtod(parts[3].c_str(), nullptr)));
    } else if (parts[0] == "f") {
      std::vector<std::vector<size_t>> fi;
      for (size_t i = 1; i < parts.size(); ++i) {
        std::vector<size_t> e;
        std::stringstream ps(parts[i]);
        std::string item;
        while (std::getline(ps, item, '/')) e.push_back((size_t)std::strtoull(item.c_str(), nullptr, 10));
        fi.push_back(e);
      }
      size_t vi1 = fi[0][0] - 1, vi2 = fi[1][0] - 1, vi3 = fi[2][0] - 1;
      size_t uv1 = 0, uv2 = 0, uv3 = 0;
      if (!(fi[0].size() < 2)) {  // :1156
        uv1 = fi[0][1] - 1;
        uv2 = fi[1][1] - 1;
        uv3 = fi[2][1] - 1;
      }
      m.triangles.push_back({vi1, vi2, vi3});
      m.triangle_to_v_to_uv_idx.push_back({uv1, uv2, uv3});
    } else if (parts[0] == "vt") {
      m.uv_coords.push_back({std::strtod(parts[1].c_str(), nullptr), std::strtod(parts[2].c_str(), nullptr)});
    }
  }
  m.uv_coords.push_back({0.0, 1.0});  // :1189

  std::vector<std::vector<Vec3>> verts_n(m.vertices.size());
  for (const auto& t : m.triangles) {  // :1193-1202
    Vec3 v1 = m.vertices[t[0]], v2 = m.vertices[t[1]], v3 = m.vertices[t[2]];
    Vec3 normal = normalize(cross(v2 - v1, v3 - v1));
    verts_n[t[0]].push_back(normal);
    verts_n[t[1]].push_back(normal);
    verts_n[t[2]].push_back(normal);
  }
  for (const auto& lst : verts_n) {  // :1204-1212
    Vec3 normal = V(0.0, 0.0, 0.0);
    for (const Vec3& n : lst) normal = normal + n;
    m.vertice_normals.push_back(normalize(normal));
  }
  *out = std::move(m);
  return true;
}

// Scene::from_objects — src/main.rs:750-774
void scene_flatten(Scene& s) {
  s.triangles.clear();
  for (size_t obj_idx = 0; obj_idx < s.objects.size(); ++obj_idx) {
    const Object& obj = s.objects[obj_idx];
    if (obj.kind != RT_OBJ_MESH) continue;
    for (size_t idx = 0; idx < obj.mesh.triangles.size(); ++idx) {
      const auto& t = obj.mesh.triangles[idx];
      FlatTriangle ft;
      ft.vertices[0] = obj.mesh.vertices[t[0]];
      ft.vertices[1] = obj.mesh.vertices[t[1]];
      ft.vertices[2] = obj.mesh.vertices[t[2]];
      ft.triangle_idx = idx;
      ft.obj_idx = obj_idx;
      s.triangles.push_back(ft);
    }
  }
}

MaterialO opaque(double metallic, double how_metallic, double diffuse, Vec3 color, int color_tex = -1, int normal_tex = -1) {
  MaterialO m;
  m.kind = RT_MAT_OPAQUE;
  m.metallic = metallic;
  m.how_metallic = how_metallic;
  m.diffuse = diffuse;
  m.color = color;
  m.color_tex = color_tex;
  m.normal_tex = normal_tex;
  m.emission = V(0.0, 0.0, 0.0);
  return m;
}
MaterialO dielectric(double ior) {
  MaterialO m;
  m.kind = RT_MAT_DIELECTRIC;
  m.ior = ior;
  return m;
}
Object mesh_object(const PolygonMesh& pm, const MaterialO& mat) {
  Object o;
  o.kind = RT_OBJ_MESH;
  o.mesh = pm;
  o.material = mat;
  return o;
}
Object sphere_object(Vec3 c, double r, const MaterialO& mat) {
  Object o;
  o.kind = RT_OBJ_SPHERE;
  o.center = c;
  o.radius = r;
  o.material = mat;
  return o;
}

const double PI = 3.14159265358979323846;  // std::f64::consts::PI

// The floor object shared by every scene — src/main.rs:1284-1296
Object floor_object(const PolygonMesh& triangle_poly) {
  return mesh_object(triangle_poly, opaque(0.1, 0.0, 1.0, V(0.7, 0.7, 0.7)));
}

// Textures are decoded by the caller (PNG decoding is outside the hot path) and handed in by
// file name.
struct TextureRegistry {
  std::map<std::string, Image> by_name;
};
TextureRegistry g_textures;

int use_texture(Scene& s, const std::string& name) {
  auto it = g_textures.by_name.find(name);
  if (it == g_textures.by_name.end()) return -1000;
  s.textures.push_back(it->second);
  return (int)s.textures.size() - 1;
}

// sphere_scene — src/main.rs:1270-1450 (the default scene: "Spot")
bool build_spot_scene(const std::string& dir, const std::string& spot_texture, Scene* out) {
  Scene s;
  PolygonMesh spot_poly, triangle_poly;
  if (!load_obj(dir + "/spot_triangulated.obj", &spot_poly)) return false;
  if (!load_obj(dir + "/triangle_floor.obj", &triangle_poly)) return false;
  int spot_text = use_texture(s, spot_texture);
  if (spot_text < 0) return false;
  s.objects.push_back(floor_object(triangle_poly));
  PolygonMesh mirror = spot_poly;  // :1339-1351
  mesh_rotate(mirror, rot_y((220.0) * PI / 180.0));
  mesh_translate(mirror, V(1.0, 0.0, -2.5));
  s.objects.push_back(mesh_object(mirror, opaque(0.0, 0.0, 1.0, V(0, 0, 0), spot_text)));
  s.objects.push_back(sphere_object(V(-0.3, -0.2, -2.8), 0.5, opaque(0.0, 1.0, 0.9, V(0.7, 0.7, 0.75))));  // :1366
  s.objects.push_back(sphere_object(V(-1.0, -0.5, -2.8), 0.2, opaque(0.0, 1.0, 0.9, V(1.0, 1.0, 1.0))));   // :1376
  s.objects.push_back(sphere_object(V(-1.5, -0.4, -2.3), 0.3, opaque(0.0, 1.0, 0.9, V(0.0, 0.0, 0.6))));   // :1386
  s.objects.push_back(sphere_object(V(0.3, -0.4, -1.4), 0.3, dielectric(1.5)));                            // :1396
  s.objects.push_back(sphere_object(V(-0.15, -0.6, -2.0), 0.1, opaque(0.0, 0.0, 1.0, V(0.7, 0.0, 0.0))));  // :1409
  s.objects.push_back(sphere_object(V(0.0, 0.3, -1.9), 0.1, dielectric(1.5)));                             // :1419
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// floor + sphere.obj as a mesh — the commented-out object of src/main.rs:1353-1364
bool build_sphere_floor_scene(const std::string& dir, Scene* out) {
  Scene s;
  PolygonMesh sphere_poly, triangle_poly;
  if (!load_obj(dir + "/sphere.obj", &sphere_poly)) return false;
  if (!load_obj(dir + "/triangle_floor.obj", &triangle_poly)) return false;
  s.objects.push_back(floor_object(triangle_poly));
  PolygonMesh sphere = sphere_poly;
  mesh_translate(sphere, V(0.0, 2.0, -7.0));
  s.objects.push_back(mesh_object(sphere, opaque(0.0, 0.0, 0.9, V(1.0, 0.3, 0.0))));
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// floor + ogre — the commented-out object of src/main.rs:1434-1445
bool build_ogre_scene(const std::string& dir, Scene* out) {
  Scene s;
  PolygonMesh ogre_poly, triangle_poly;
  if (!load_obj(dir + "/ogre/bs_rest.obj", &ogre_poly)) return false;
  if (!load_obj(dir + "/triangle_floor.obj", &triangle_poly)) return false;
  int ogre_text = use_texture(s, "ogre/diffuse.png");
  int ogre_normal = use_texture(s, "ogre/normalmap.png");
  if (ogre_text < 0 || ogre_normal < 0) return false;
  s.objects.push_back(floor_object(triangle_poly));
  PolygonMesh ogre = ogre_poly;
  mesh_rotate(ogre, rot_y((20.0) * PI / 180.0));
  mesh_translate(ogre, V(0.0, 0.0, -2.0));
  s.objects.push_back(mesh_object(ogre, opaque(0.0, 0.0, 0.9, V(0, 0, 0), ogre_text, ogre_normal)));
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// lens_scene — src/main.rs:1223-1268
bool build_lens_scene(const std::string& dir, const std::string& spot_texture, Scene* out) {
  Scene s;
  PolygonMesh spot_poly, triangle_poly;
  if (!load_obj(dir + "/spot_triangulated.obj", &spot_poly)) return false;
  if (!load_obj(dir + "/triangle_floor.obj", &triangle_poly)) return false;
  int spot_text = use_texture(s, spot_texture);
  if (spot_text < 0) return false;
  s.objects.push_back(floor_object(triangle_poly));
  PolygonMesh mirror = spot_poly;
  mesh_rotate(mirror, rot_y((220.0) * PI / 180.0));
  mesh_rotate(mirror, rot_z(180.0 * PI / 180.0));
  mesh_translate(mirror, V(0.0, 0.9, -2.2));
  s.objects.push_back(mesh_object(mirror, opaque(0.0, 0.0, 1.0, V(0, 0, 0), spot_text)));
  s.objects.push_back(sphere_object(V(0.0, -0.3, -1.9), 0.3, dielectric(1.5)));
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// floor + nx x nz copies of Spot (the reference has no instancing: copies are duplicated,
// src/main.rs:752-769), spacing 1.2, rows receding in -z.
bool build_spot_grid_scene(const std::string& dir, const std::string& spot_texture, int nx, int nz, Scene* out) {
  Scene s;
  PolygonMesh spot_poly, triangle_poly;
  if (!load_obj(dir + "/spot_triangulated.obj", &spot_poly)) return false;
  if (!load_obj(dir + "/triangle_floor.obj", &triangle_poly)) return false;
  int spot_text = use_texture(s, spot_texture);
  if (spot_text < 0) return false;
  s.objects.push_back(floor_object(triangle_poly));
  PolygonMesh turned = spot_poly;
  mesh_rotate(turned, rot_y((220.0) * PI / 180.0));
  for (int iz = 0; iz < nz; ++iz)
    for (int ix = 0; ix < nx; ++ix) {
      PolygonMesh m = turned;
      double x = ((double)ix - 0.5 * (double)(nx - 1)) * 1.2;
      double z = -2.5 - (double)iz * 1.2;
      mesh_translate(m, V(x, 0.0, z));
      s.objects.push_back(mesh_object(m, opaque(0.0, 0.0, 1.0, V(0, 0, 0), spot_text)));
    }
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// A scene from a flat description: every triangle becomes three private vertices of one mesh
// per object, so that the reference's indexed lookups resolve to the given per-corner data.
bool build_from_desc(const rt_scene_desc& d, Scene* out) {
  Scene s;
  s.objects.resize(d.n_objects);
  for (uint32_t i = 0; i < d.n_textures; ++i) {
    Image im;
    im.w = d.textures[i].width;
    im.h = d.textures[i].height;
    im.rgba.assign(d.textures[i].rgba, d.textures[i].rgba + (size_t)im.w * im.h * 4);
    s.textures.push_back(std::move(im));
  }
  for (uint32_t i = 0; i < d.n_objects; ++i) {
    Object& o = s.objects[i];
    o.kind = d.object_kind[i];
    const rt_material& m = d.materials[i];
    o.material.kind = m.kind;
    o.material.color_tex = m.color_tex;
    o.material.normal_tex = m.normal_tex;
    o.material.metallic = m.metallic;
    o.material.diffuse = m.diffuse;
    o.material.how_metallic = m.how_metallic;
    o.material.emission = V(m.emission[0], m.emission[1], m.emission[2]);
    o.material.color = V(m.color[0], m.color[1], m.color[2]);
    o.material.ior = m.ior;
  }
  for (uint64_t t = 0; t < d.n_triangles; ++t) {
    Object& o = s.objects[d.tri_obj[t]];
    PolygonMesh& pm = o.mesh;
    size_t want = (size_t)d.tri_idx[t];
    if (pm.triangles.size() != want) return false;  // triangle_idx must count up per object
    size_t base = pm.vertices.size();
    for (int k = 0; k < 3; ++k) {
      pm.vertices.push_back(V(d.tri_vertices[t * 9 + k * 3 + 0], d.tri_vertices[t * 9 + k * 3 + 1], d.tri_vertices[t * 9 + k * 3 + 2]));
      pm.vertice_normals.push_back(V(d.tri_normals[t * 9 + k * 3 + 0], d.tri_normals[t * 9 + k * 3 + 1], d.tri_normals[t * 9 + k * 3 + 2]));
      pm.uv_coords.push_back({d.tri_uvs[t * 6 + k * 2 + 0], d.tri_uvs[t * 6 + k * 2 + 1]});
    }
    pm.triangles.push_back({base, base + 1, base + 2});
    pm.triangle_to_v_to_uv_idx.push_back({base, base + 1, base + 2});
  }
  for (uint32_t i = 0; i < d.n_spheres; ++i) {
    Object& o = s.objects[d.sphere_obj[i]];
    o.center = V(d.spheres[i * 4 + 0], d.spheres[i * 4 + 1], d.spheres[i * 4 + 2]);
    o.radius = d.spheres[i * 4 + 3];
  }
  scene_flatten(s);
  *out = std::move(s);
  return true;
}

// ------------------------------------------------------------------------------------------
// Restatement of the HIP kernel's own traversal over the library's exported tree: used to show
// on the CPU that the tree returns the brute-force answer, and to count the nodes and
// triangles a front-to-back, distance-pruned traversal touches per ray (the roofline's
// algorithmic bytes). Node layout: 16 words —
//   [0..3]  child0.min.x child0.max.x child1.min.x child1.max.x   (f32)
//   [4..7]  ... y,  [8..11] ... z
//   [12] child0 ref, [13] child1 ref (i32), [14],[15] unused.
// A ref >= 0 is an interior node index; a ref < 0 encodes a leaf: ~ref = first_slot * 4 +
// (count - 1). Leaf slots index the leaf-order triangle array.
// ------------------------------------------------------------------------------------------
struct BvhCounts {
  uint64_t nodes = 0, tris = 0;
};

inline float f32_up(double x) {  // smallest float >= x
  float f = (float)x;
  if ((double)f < x) f = std::nextafter(f, INFINITY);
  return f;
}

// leaf_order entries from tris.size() on are places of spheres in the tree (entry - tris.size() =
// sphere index): their visit counts as a triangle test (the kernel runs one, on all-zero geometry,
// which fails) and sets the sphere's bit in *sphere_mask.
void bvh_trace(const uint32_t* node_words, uint32_t n_nodes, int32_t root_ref, const int32_t* leaf_order,
               const std::vector<FlatTriangle>& tris, const Vec3& origin, const Vec3& dir, double* best_t,
               double* best_u, double* best_v, int64_t* best_flat, BvhCounts* counts, uint64_t* sphere_mask = nullptr) {
  const float K_FAR = 1.0f + 9.5367431640625e-07f;  // 1 + 2^-20
  float m_lo[3], m_hi[3], inv[3];
  double omax = std::fmax(std::fabs(origin.c[0]), std::fmax(std::fabs(origin.c[1]), std::fabs(origin.c[2])));
  float pad = f32_up(omax * 4.76837158203125e-07);  // 2^-21
  for (int k = 0; k < 3; ++k) {
    float o32 = (float)origin.c[k];
    double d = dir.c[k];
    if (std::fabs(d) < 1e-30) d = std::signbit(d) ? -1e-30 : 1e-30;
    // The kernel takes the hardware's binary32 reciprocal of the rounded component (v_rcp_f32,
    // within one unit in the last place of the exact reciprocal). Restated here as the correctly
    // rounded reciprocal pushed one unit up or down, or left alone, as a hash of the ray decides:
    // the margins of the culling test must hold for every such error.
    inv[k] = 1.0f / (float)d;
    {
      uint64_t bits;
      std::memcpy(&bits, &dir.c[k], 8);
      uint64_t obits;
      std::memcpy(&obits, &origin.c[(k + 1) % 3], 8);
      const uint64_t hsh = mix64(bits ^ (obits * 0x9e3779b97f4a7c15ull) ^ (uint64_t)k);
      if ((hsh & 3u) == 1u) inv[k] = std::nextafter(inv[k], INFINITY);
      if ((hsh & 3u) == 2u) inv[k] = std::nextafter(inv[k], -INFINITY);
    }
    m_lo[k] = -(o32 - pad) * inv[k];
    m_hi[k] = -(o32 + pad) * inv[k];
  }
  *best_t = std::numeric_limits<double>::infinity();
  *best_flat = -1;
  float best32k = INFINITY;
  int32_t stack[128];
  int sp = 0;
  int32_t cur = root_ref;
  const int32_t SENTINEL = INT32_MIN;
  stack[sp++] = SENTINEL;
  auto f = [&](uint32_t w) {
    float x;
    std::memcpy(&x, &w, 4);
    return x;
  };
  for (;;) {
    while (cur >= 0) {
      const uint32_t* n = node_words + (size_t)cur * 16;
      if (counts) counts->nodes++;
      float tn[2], tf[2];
      for (int c = 0; c < 2; ++c) {
        float tnear = 0.0f, tfar = INFINITY;
        for (int k = 0; k < 3; ++k) {
          float bmin = f(n[k * 4 + c * 2 + 0]), bmax = f(n[k * 4 + c * 2 + 1]);
          float ta = std::fmaf(bmin, inv[k], m_hi[k]);
          float tb = std::fmaf(bmax, inv[k], m_lo[k]);
          tnear = std::fmax(tnear, std::fmin(ta, tb));
          tfar = std::fmin(tfar, std::fmax(ta, tb));
        }
        tn[c] = tnear;
        tf[c] = tfar * K_FAR;
      }
      bool h0 = tn[0] <= std::fmin(tf[0], best32k);
      bool h1 = tn[1] <= std::fmin(tf[1], best32k);
      int32_t c0 = (int32_t)n[12], c1 = (int32_t)n[13];
      if (h0 && h1) {
        if (tn[1] < tn[0]) std::swap(c0, c1);
        stack[sp++] = c1;
        cur = c0;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else {
        cur = stack[--sp];
      }
    }
    if (cur == SENTINEL) break;
    uint32_t leaf = (uint32_t)~cur;
    uint32_t first = leaf >> 2, cnt = (leaf & 3u) + 1u;
    for (uint32_t i = 0; i < cnt; ++i) {
      int32_t flat = leaf_order[first + i];
      if (counts) counts->tris++;
      if ((size_t)flat >= tris.size()) {
        if (sphere_mask) *sphere_mask |= 1ull << (((size_t)flat - tris.size()) & 63u);
        continue;
      }
      const FlatTriangle& ft = tris[flat];
      double t, u, v;
      if (intersects_triangle(origin, dir, ft.vertices[0], ft.vertices[1], ft.vertices[2], &t, &u, &v)) {
        if (t < *best_t || (t == *best_t && flat < *best_flat)) {
          *best_t = t;
          *best_u = u;
          *best_v = v;
          *best_flat = flat;
          best32k = f32_up(t) * K_FAR;
        }
      }
    }
    cur = stack[--sp];
  }
  (void)n_nodes;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C entry points for the Python tests (ctypes)
// ------------------------------------------------------------------------------------------
struct orc_scene {
  Scene s;
};

extern "C" {

int orc_register_texture(const char* name, const uint8_t* rgba, uint32_t w, uint32_t h) {
  Image im;
  im.w = w;
  im.h = h;
  im.rgba.assign(rgba, rgba + (size_t)w * h * 4);
  g_textures.by_name[name] = std::move(im);
  return 0;
}

// name: "spot", "sphere_floor", "ogre", "lens", "spot_grid:<nx>x<nz>"
orc_scene* orc_scene_build(const char* name, const char* assets_dir, const char* spot_texture) {
  std::unique_ptr<orc_scene> sc(new orc_scene());
  std::string n(name), dir(assets_dir), st(spot_texture ? spot_texture : "spot_texture.png");
  bool ok = false;
  if (n == "spot") ok = build_spot_scene(dir, st, &sc->s);
  else if (n == "sphere_floor") ok = build_sphere_floor_scene(dir, &sc->s);
  else if (n == "ogre") ok = build_ogre_scene(dir, &sc->s);
  else if (n == "lens") ok = build_lens_scene(dir, st, &sc->s);
  else if (n.rfind("spot_grid:", 0) == 0) {
    int nx = 0, nz = 0;
    if (std::sscanf(n.c_str() + 10, "%dx%d", &nx, &nz) == 2 && nx > 0 && nz > 0) ok = build_spot_grid_scene(dir, st, nx, nz, &sc->s);
  }
  if (!ok) return nullptr;
  return sc.release();
}

orc_scene* orc_scene_from_desc(const rt_scene_desc* d) {
  std::unique_ptr<orc_scene> sc(new orc_scene());
  if (!build_from_desc(*d, &sc->s)) return nullptr;
  return sc.release();
}

void orc_scene_free(orc_scene* sc) { delete sc; }

void orc_scene_counts(const orc_scene* sc, uint64_t* n_tris, uint32_t* n_spheres, uint32_t* n_objects) {
  *n_tris = sc->s.triangles.size();
  uint32_t ns = 0;
  for (const Object& o : sc->s.objects) ns += o.kind == RT_OBJ_SPHERE;
  *n_spheres = ns;
  *n_objects = (uint32_t)sc->s.objects.size();
}

// Flat arrays in the layout of rt_scene_desc (per-corner normals and uvs resolved through the
// mesh's index tables, as normal_at_hit / get_tex_uv_coords would).
void orc_scene_flat(const orc_scene* sc, double* verts, double* normals, double* uvs, int32_t* obj, int32_t* tri) {
  const Scene& s = sc->s;
  for (size_t i = 0; i < s.triangles.size(); ++i) {
    const FlatTriangle& ft = s.triangles[i];
    const PolygonMesh& pm = s.objects[ft.obj_idx].mesh;
    const auto& t = pm.triangles[ft.triangle_idx];
    const auto& m = pm.triangle_to_v_to_uv_idx[ft.triangle_idx];
    for (int k = 0; k < 3; ++k) {
      for (int c = 0; c < 3; ++c) {
        if (verts) verts[i * 9 + k * 3 + c] = ft.vertices[k].c[c];
        if (normals) normals[i * 9 + k * 3 + c] = pm.vertice_normals[t[k]].c[c];
      }
      if (uvs) {
        uvs[i * 6 + k * 2 + 0] = pm.uv_coords[m[k]][0];
        uvs[i * 6 + k * 2 + 1] = pm.uv_coords[m[k]][1];
      }
    }
    if (obj) obj[i] = (int32_t)ft.obj_idx;
    if (tri) tri[i] = (int32_t)ft.triangle_idx;
  }
}

// Spheres (centre, radius, object index) and materials in object order.
void orc_scene_objects(const orc_scene* sc, int32_t* kinds, rt_material* mats, double* spheres4, int32_t* sphere_obj) {
  const Scene& s = sc->s;
  uint32_t ns = 0;
  for (size_t i = 0; i < s.objects.size(); ++i) {
    const Object& o = s.objects[i];
    if (kinds) kinds[i] = o.kind;
    if (mats) {
      rt_material& m = mats[i];
      std::memset(&m, 0, sizeof m);
      m.kind = o.material.kind;
      m.color_tex = o.material.color_tex;
      m.normal_tex = o.material.normal_tex;
      m.metallic = o.material.metallic;
      m.diffuse = o.material.diffuse;
      m.how_metallic = o.material.how_metallic;
      for (int c = 0; c < 3; ++c) {
        m.emission[c] = o.material.emission.c[c];
        m.color[c] = o.material.color.c[c];
      }
      m.ior = o.material.ior;
    }
    if (o.kind == RT_OBJ_SPHERE) {
      if (spheres4) {
        spheres4[ns * 4 + 0] = o.center.c[0];
        spheres4[ns * 4 + 1] = o.center.c[1];
        spheres4[ns * 4 + 2] = o.center.c[2];
        spheres4[ns * 4 + 3] = o.radius;
      }
      if (sphere_obj) sphere_obj[ns] = (int32_t)i;
      ns++;
    }
  }
}

static void fill_hit(const Hit& h, rt_hit* o) {
  if (!h.some) {
    o->t = 0.0;
    o->u = 0.0;
    o->v = 0.0;
    o->obj = -1;
    o->tri = -1;
    return;
  }
  o->t = h.distance;
  o->u = h.sphere ? 0.0 : h.u;
  o->v = h.sphere ? 0.0 : h.v;
  o->obj = (int32_t)h.obj_idx;
  o->tri = h.sphere ? -1 : (int32_t)h.triangle_idx;
}

// Batch Scene::hit. mode: 0 brute force, 1 reference-like candidates.
void orc_trace(orc_scene* sc, int mode, const double* org3, const double* dir3, size_t n, rt_hit* out, int threads) {
  if (mode == HIT_REFLIKE) ref_ensure_built(sc->s);
  std::atomic<size_t> next(0);
  auto worker = [&]() {
    for (;;) {
      size_t b = next.fetch_add(1024);
      if (b >= n) break;
      size_t e = std::min(b + 1024, n);
      for (size_t i = b; i < e; ++i) {
        Hit h = scene_hit(sc->s, V(org3[i * 3], org3[i * 3 + 1], org3[i * 3 + 2]), V(dir3[i * 3], dir3[i * 3 + 1], dir3[i * 3 + 2]), mode, nullptr);
        fill_hit(h, &out[i]);
      }
    }
  };
  if (threads <= 1) worker();
  else {
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }
}

// Pixel-centre camera rays of src/main.rs:999-1013 (x_offset = y_offset = 0, DOF 0).
void orc_camera_rays(uint32_t w, uint32_t h, double fov_y_deg, double* org3, double* dir3) {
  Camera cam = make_camera(w, h, fov_y_deg);
  for (uint32_t v = 0; v < h; ++v)
    for (uint32_t u = 0; u < w; ++u) {
      Rng rng;
      rng.start(0, 0);
      Vec3 o, d;
      camera_ray(cam, u, v, rng, false, 0.0, &o, &d);
      size_t i = (size_t)v * w + u;
      for (int c = 0; c < 3; ++c) {
        org3[i * 3 + c] = o.c[c];
        dir3[i * 3 + c] = d.c[c];
      }
    }
}

// One jittered camera ray exactly as render() would generate it for (pixel, sample).
void orc_camera_ray_sample(const rt_render_params* p, uint32_t u, uint32_t v, uint32_t sample, double org[3], double dir[3]) {
  Camera cam = make_camera(p->width, p->height, p->fov_y_deg);
  Rng rng;
  rng.start(rng_path_key(p->seed, (uint64_t)v * p->width + u, (uint64_t)p->first_sample + sample), 0);
  Vec3 o, d;
  camera_ray(cam, u, v, rng, !(p->flags & RT_FLAG_NO_JITTER), p->dof_strength, &o, &d);
  for (int c = 0; c < 3; ++c) {
    org[c] = o.c[c];
    dir[c] = d.c[c];
  }
}

// render(); canvas is [h][w][3] and is added into. mode as orc_trace.
void orc_render(orc_scene* sc, const rt_render_params* p, double* canvas, uint64_t* rays_out, uint64_t* cands_out, int threads, int mode) {
  if (mode == HIT_REFLIKE) ref_ensure_built(sc->s);
  render(sc->s, *p, canvas, rays_out, cands_out, threads, mode);
}

// Radiance of one path, for per-sample comparisons.
void orc_path_radiance(orc_scene* sc, const rt_render_params* p, uint32_t u, uint32_t v, uint32_t sample, int mode, double rgb[3], uint64_t* rays) {
  if (mode == HIT_REFLIKE) ref_ensure_built(sc->s);
  Camera cam = make_camera(p->width, p->height, p->fov_y_deg);
  TraceCtx cx;
  cx.scene = &sc->s;
  cx.mode = mode;
  cx.max_depth = p->max_depth;
  cx.key = rng_path_key(p->seed, (uint64_t)v * p->width + u, (uint64_t)p->first_sample + sample);
  Rng rng;
  rng.start(cx.key, 0);
  Vec3 o, d;
  camera_ray(cam, u, v, rng, !(p->flags & RT_FLAG_NO_JITTER), p->dof_strength, &o, &d);
  raytrace(cx, o, d, 0, rgb);
  if (rays) *rays = cx.rays;
}

// Known-answer entry points.
int orc_intersects_triangle(const double o[3], const double d[3], const double a[3], const double b[3], const double c[3], double tuv[3]) {
  return intersects_triangle(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(a[0], a[1], a[2]), V(b[0], b[1], b[2]), V(c[0], c[1], c[2]), &tuv[0], &tuv[1], &tuv[2]) ? 1 : 0;
}
int orc_intersects_sphere(const double c[3], double r, const double o[3], const double d[3], double* t) {
  return intersects_sphere(V(c[0], c[1], c[2]), r, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), t) ? 1 : 0;
}
void orc_interp_sky(const double d[3], double rgb[3]) { interp_sky(V(d[0], d[1], d[2]), rgb); }
void orc_refract(const double uv[3], const double n[3], double eta, double out[3]) {
  Vec3 r = refract(V(uv[0], uv[1], uv[2]), V(n[0], n[1], n[2]), eta);
  out[0] = r.c[0];
  out[1] = r.c[1];
  out[2] = r.c[2];
}
double orc_reflectance(double cosine, double ref_idx) { return reflectance(cosine, ref_idx); }
double orc_rng_u01(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t stage, uint32_t index) {
  Rng r;
  r.start(rng_path_key(seed, pixel, sample), stage);
  r.index = index;
  return r.u01();
}
void orc_random_unit_sphere_surface(uint64_t key, uint32_t stage, uint32_t index, double out[3], uint32_t* index_after) {
  Rng r;
  r.start(key, stage);
  r.index = index;
  Vec3 v = random_unit_sphere_surface(r);
  out[0] = v.c[0];
  out[1] = v.c[1];
  out[2] = v.c[2];
  *index_after = r.index;
}
void orc_normal_map(const uint8_t rgb[3], const double world_normal[3], double out[3]) {
  // NormalKind::normal_at with a 1x1 normal map holding `rgb`
  std::vector<Image> tex(1);
  tex[0].w = tex[0].h = 1;
  tex[0].rgba = {rgb[0], rgb[1], rgb[2], 255};
  MaterialO m;
  m.normal_tex = 0;
  Vec3 r = normal_at(m, tex, V(world_normal[0], world_normal[1], world_normal[2]), 0.0, 1.0);
  out[0] = r.c[0];
  out[1] = r.c[1];
  out[2] = r.c[2];
}
int orc_load_obj_counts(const char* path, uint64_t* nv, uint64_t* nt, uint64_t* nuv) {
  PolygonMesh m;
  if (!load_obj(path, &m)) return -1;
  *nv = m.vertices.size();
  *nt = m.triangles.size();
  *nuv = m.uv_coords.size();
  return 0;
}

// Traversal of the library's exported tree (see bvh_trace). out hits use flat->(obj,tri) of the
// oracle scene, then spheres as Scene::hit does. counts2 = {node visits, triangle tests}.
void orc_bvh_trace(orc_scene* sc, const uint32_t* node_words, uint32_t n_nodes, int32_t root_ref, const int32_t* leaf_order,
                   const double* org3, const double* dir3, size_t n, rt_hit* out, uint64_t* counts2, int threads) {
  std::atomic<size_t> next(0);
  std::atomic<uint64_t> cn(0), ct(0);
  const Scene& s = sc->s;
  // the spheres that have a place in this tree (the others are tested against every ray)
  uint64_t in_tree = 0;
  size_t n_spheres = 0;
  for (const Object& ob : s.objects) n_spheres += ob.kind == RT_OBJ_SPHERE ? 1 : 0;
  size_t n_slots = 0;
  {
    // slots = highest slot any leaf names + 1
    auto leaf_end = [&](int32_t ref) {
      if (ref >= 0 || ref == INT32_MIN) return;
      const uint32_t leaf = (uint32_t)~ref;
      n_slots = std::max(n_slots, (size_t)(leaf >> 2) + (leaf & 3u) + 1u);
    };
    leaf_end(root_ref);
    for (uint32_t k = 0; k < n_nodes; ++k) {
      leaf_end((int32_t)node_words[(size_t)k * 16 + 12]);
      leaf_end((int32_t)node_words[(size_t)k * 16 + 13]);
    }
  }
  for (size_t k = 0; k < n_slots; ++k)
    if ((size_t)leaf_order[k] >= s.triangles.size()) in_tree |= 1ull << (((size_t)leaf_order[k] - s.triangles.size()) & 63u);
  auto worker = [&]() {
    BvhCounts bc;
    for (;;) {
      size_t b = next.fetch_add(1024);
      if (b >= n) break;
      size_t e = std::min(b + 1024, n);
      for (size_t i = b; i < e; ++i) {
        Vec3 o = V(org3[i * 3], org3[i * 3 + 1], org3[i * 3 + 2]), d = V(dir3[i * 3], dir3[i * 3 + 1], dir3[i * 3 + 2]);
        Hit h;
        uint64_t reached = 0;
        if (root_ref != INT32_MIN) {
          double t, u = 0, v = 0;
          int64_t flat;
          bvh_trace(node_words, n_nodes, root_ref, leaf_order, s.triangles, o, d, &t, &u, &v, &flat, &bc, &reached);
          if (flat >= 0) {
            h.some = true;
            h.distance = t;
            h.u = u;
            h.v = v;
            h.triangle_idx = s.triangles[flat].triangle_idx;
            h.obj_idx = s.triangles[flat].obj_idx;
          }
        }
        size_t sphere_index = 0;
        for (size_t oi = 0; oi < s.objects.size(); ++oi) {
          const Object& ob = s.objects[oi];
          if (ob.kind != RT_OBJ_SPHERE) continue;
          const size_t si = sphere_index++;
          // a sphere with a place in the tree is tested only if the ray reached that place
          if (si < 64 && ((in_tree >> si) & 1u) && !((reached >> si) & 1u)) continue;
          double t;
          if (intersects_sphere(ob.center, ob.radius, o, d, &t)) {
            double cur = h.some ? h.distance : std::numeric_limits<double>::infinity();
            if (t < cur) {
              h = Hit();
              h.some = true;
              h.sphere = true;
              h.distance = t;
              h.obj_idx = oi;
            }
          }
        }
        fill_hit(h, &out[i]);
      }
    }
    cn += bc.nodes;
    ct += bc.tris;
  };
  if (threads <= 1) worker();
  else {
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }
  if (counts2) {
    counts2[0] = cn.load();
    counts2[1] = ct.load();
  }
}

}  // extern "C"
