// rt_types.h — records shared by the host scene builder and the gfx950 kernels.
//
// Everything the kernels read lives in HBM as arrays of fixed-size, 16-byte-aligned records so
// that one lane fetches one record with a few dwordx4 loads:
//
//   BvhNode   64 B  both child boxes (f32, rounded outward and padded) + both child refs
//   TriGeom   80 B  a, b-a, c-a in binary64 (what src/main.rs:188-189 computes first) + ids
//   TriShade 128 B  per-corner vertex normals and texture coordinates (src/main.rs:526-531,
//                   481-486 with the index lookups already resolved)
//   Sphere    48 B  centre, radius, object index (src/main.rs:437-442)
//   rt_material     one per object (include/rt_amd.h)
//
// TriGeom and TriShade are stored in leaf order: a leaf names a run of 1..4 consecutive slots.
//
// Spheres are not triangles, but the tree culls them like triangles: each of the first
// RT_TREE_SPHERES finite spheres has a slot in some leaf, placed by its bounding box. The slot's
// TriGeom is all zeros (its triangle test can never succeed) with flat = ~sphere index, which tells
// the traversal to note the sphere as a candidate; the sphere tests themselves run after the
// traversal, on the candidates only, in object order (src/main.rs:802-810 tests every sphere).
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#include "../../include/rt_amd.h"

#ifndef RT_LEAF_BITS
#define RT_LEAF_BITS 2                /* a leaf ref keeps (count - 1) in this many low bits    */
#endif
#define RT_LEAF_MAX (1 << RT_LEAF_BITS) /* triangles per leaf                                 */
#define RT_BVH_MAX_DEPTH 62           /* builder guarantee; the traversal stack is sized by it */
#define RT_REF_SENTINEL INT32_MIN     /* bottom of the traversal stack / empty tree          */
#ifndef RT_REJECT_CAP
#define RT_REJECT_CAP 64              /* tries per rejection loop before the fallback        */
#endif
#define RT_TREE_SPHERES 64            /* spheres that can have a place in the tree (one bit each in a ray's candidate mask) */

// A child ref >= 0 is the index of an interior node. A ref < 0 other than RT_REF_SENTINEL is a
// leaf: ~ref = first_slot * RT_LEAF_MAX + (count - 1).
struct alignas(16) BvhNode {
  float bx[4];  // child0.min.x child0.max.x child1.min.x child1.max.x
  float by[4];
  float bz[4];
  int32_t child[2];
  int32_t pad[2];
};

// The same node in half the bytes, for the launches over the ray queue: their rays are incoherent,
// every lane of a wave fetches a different node, and what limits them is the number of load
// instructions the texture addressers process, not arithmetic. The twelve planes are 16-bit
// positions on a uniform grid over the scene's bounds (per axis: lo + q * step), rounded outward by
// at least one grid step beyond the binary32 planes of BvhNode, so the boxes only grow. Only scenes
// for which that can be shown plane by plane get this copy (rt_quantize_tree, rt_host.cpp); the
// others are walked through BvhNode everywhere.
struct alignas(16) BvhNodeQ {
  uint16_t q[12];  // x: child0.min child0.max child1.min child1.max, then y, then z
  int32_t child[2];
};

struct alignas(16) TriGeom {
  double a[3];
  double ab[3];  // b - a
  double ac[3];  // c - a
  int32_t flat;  // index in the reference's flat triangle order (tie-break, src/main.rs:791)
  int32_t obj;   // FlatTriangle.obj_idx
};

struct alignas(16) TriShade {
  double n[3][3];   // vertex normal per corner
  double uv[3][2];  // texture coordinate per corner
  int32_t tri;      // FlatTriangle.triangle_idx
  int32_t obj;
};

struct alignas(16) SphereRec {
  double c[3];
  double r;
  int32_t obj;
  int32_t pad[3];
};

// Textures are stored tiled in HBM: 32x32-texel blocks (4 KB) in row-major block order, each made
// of 4x4 tiles of 8x8 texels (256 B = two cache lines), so that the texels of neighbouring hits
// share lines and pages in both directions (a 20000-texel row is 80 KB: in row-major order the
// texel one row down is 625 lines away). Width and height are padded up to whole blocks.
#define RT_TEX_BLOCK 32u
struct TextureRec {
  const uint8_t* rgba;  // device pointer, RGBA8, tiled as above, row 0 on top
  uint32_t w, h;
  uint32_t blocks_x;    // 32x32 blocks per row of blocks
};
// Offset, in texels, of texel (x, y) in the tiled layout.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline size_t rt_texel_offset(uint32_t x, uint32_t y, uint32_t blocks_x) {
  const size_t block = (size_t)(y >> 5) * blocks_x + (x >> 5);
  const uint32_t tile = ((y >> 3) & 3u) * 4u + ((x >> 3) & 3u);
  return block * 1024u + tile * 64u + (y & 7u) * 8u + (x & 7u);
}

static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
static_assert(sizeof(BvhNodeQ) == 32, "BvhNodeQ must be 32 bytes");
static_assert(sizeof(TriGeom) == 80, "TriGeom must be 80 bytes");
static_assert(sizeof(TriShade) == 128, "TriShade must be 128 bytes");
static_assert(sizeof(SphereRec) == 48, "SphereRec must be 48 bytes");

// Host-side scene: the flat description (owned copies) plus the built tree. The device half
// is created by rt_scene_upload.
#ifdef __cplusplus
#include <string>
#include <vector>

struct HostTexture {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;
};

struct DeviceScene;  // rt_device.hip

struct rt_scene {
  // flat description, reference order
  std::vector<int32_t> object_kind;
  std::vector<rt_material> materials;
  std::vector<double> tri_vertices, tri_normals, tri_uvs;
  std::vector<int32_t> tri_obj, tri_idx;
  std::vector<SphereRec> spheres;
  std::vector<HostTexture> textures;
  // acceleration structure
  std::vector<BvhNode> nodes;
  std::vector<int32_t> leaf_order;  // slot -> flat triangle index, or n_triangles + sphere index for a sphere's place in the tree
  std::vector<int32_t> tree_spheres;  // the spheres that have a place in the tree (ascending sphere indices)
  std::vector<TriGeom> tri_geom;    // leaf order
  std::vector<TriShade> tri_shade;  // leaf order
  int32_t root_ref = RT_REF_SENTINEL;
  uint32_t bvh_depth = 0;
  uint32_t bvh_leaves = 0;
  bool has_tree = false;  // false only between rt_scene_create_with(RT_TREE_NONE) and a device build
  DeviceScene* dev = nullptr;
};

// rt_host.cpp
void rt_set_error(const char* fmt, ...);

// Nothing may unwind across the C boundary: an entry point whose body can throw (container
// growth, mostly) runs it through this.
#include <exception>
#include <new>
template <class F>
static inline int rt_guard(const char* who, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    rt_set_error("%s: out of host memory", who);
    return RT_ERR_OOM;
  } catch (const std::exception& e) {
    rt_set_error("%s: %s", who, e.what());
    return RT_ERR_INVALID;
  } catch (...) {
    rt_set_error("%s: unexpected exception", who);
    return RT_ERR_INVALID;
  }
}
// Number parsing that does not depend on the process locale (strtod does: a host that has called
// setlocale with a comma decimal point could load no asset) and accepts what Rust's str::parse
// accepts (src/main.rs:1147, 1161): decimal notation with an optional sign and exponent, "inf",
// "infinity", "nan"; no hexadecimal floats, no leading or trailing junk.
#include <charconv>
#include <system_error>
static inline bool rt_parse_f64(const std::string& t, double* v) {
  const char* b = t.data();
  const char* e = b + t.size();
  if (b != e && *b == '+') ++b;  // from_chars takes no plus sign; Rust does
  if (b == e || *b == '+') return false;
  const std::from_chars_result r = std::from_chars(b, e, *v, std::chars_format::general);
  return r.ec == std::errc() && r.ptr == e;
}
// An unsigned decimal integer, optionally with a plus sign (Rust's usize parse): no minus, no junk.
static inline bool rt_parse_u64(const std::string& t, uint64_t* v) {
  const char* b = t.data();
  const char* e = b + t.size();
  if (b != e && *b == '+') ++b;
  if (b == e) return false;
  const std::from_chars_result r = std::from_chars(b, e, *v, 10);
  return r.ec == std::errc() && r.ptr == e && *b != '+' && *b != '-';
}
void rt_fill_leaf_records(rt_scene* s);
void rt_scene_pick_tree_spheres(rt_scene* s);
void rt_sphere_place_vertices(const SphereRec& sp, double v9[9]);
double rt_scene_box_pad(const rt_scene* s);
unsigned rt_build_threads();
void rt_add_into(double* dst, const double* src, size_t n);
void rt_tile_texture(const HostTexture& t, std::vector<uint8_t>* out, uint32_t* blocks_x);
void rt_scene_centre_bounds(const rt_scene* s, double clo[3], double chi[3]);
bool rt_quantize_tree(const rt_scene* s, std::vector<BvhNodeQ>* out, float q_lo[3], float q_step[3]);
// rt_device.hip
void rt_device_scene_free(DeviceScene* d);
int rt_device_scene_upload_from(const rt_scene* host, rt_scene* target, int device);
int rt_device_scene_info(const DeviceScene* d, uint32_t* device_plus_one, uint64_t* bytes);
#endif

#endif  // RT_TYPES_H
