// rt_build.hip — the acceleration structure built on the GPU.
//
// The reference builds its tree on the CPU (BVH::build of crate bvh 0.6.0, src/main.rs:771) once
// per scene. rt_scene_rebuild_on_device replaces the host-built tree of a scene by one made on
// the device, for scenes that change between frames: Morton codes of the triangle centroids, one
// radix sort, then either
//   * bottom-up clustering along the sorted order (PLOC, Meister and Bittner 2018; the default), or
//   * the hierarchy of Karras ("Maximizing parallelism in the construction of BVHs, octrees and
//     k-d trees", 2012) with a bottom-up pass for the boxes,
// and a last pass that writes nodes in the library's own format (rt_types.h: two child boxes per
// 64-byte node, leaves of 1..RT_LEAF_MAX triangles). Either takes milliseconds where the host's
// binned-SAH builder takes tenths of a second; the clustered tree traverses 6-17 % slower than
// the host's, the Morton hierarchy 12-27 % slower.
//
// The nearest hit does not depend on the tree (every box contains its triangles: exact binary64
// minima and maxima, rounded outward to binary32 and padded exactly as the host builder does),
// so hit records stay bit-identical to the host-built tree's.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "rt_types.h"

namespace {

#define BUILD_BLOCK 256

#define HIP_TRY_B(expr)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RT_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

struct Box6 {
  double lo[3], hi[3];
};

// 21 bits -> every third bit of a 63-bit word
__device__ __forceinline__ uint64_t spread21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// Per triangle: its binary64 box and the Morton code of the box centre within `clo..chi`.
__global__ void k_boxes_codes(const double* __restrict__ verts, uint32_t n, double clo_x, double clo_y, double clo_z, double sx,
                              double sy, double sz, Box6* __restrict__ boxes, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double* v = verts + (size_t)t * 9;
  Box6 b;
  for (int c = 0; c < 3; ++c) {
    b.lo[c] = fmin(v[c], fmin(v[3 + c], v[6 + c]));
    b.hi[c] = fmax(v[c], fmax(v[3 + c], v[6 + c]));
  }
  boxes[t] = b;
  const double clo[3] = {clo_x, clo_y, clo_z}, sc[3] = {sx, sy, sz};
  uint64_t q[3];
  for (int c = 0; c < 3; ++c) {
    double f = (0.5 * b.lo[c] + 0.5 * b.hi[c] - clo[c]) * sc[c];  // 0 .. 2^21, NaN for non-finite data
    f = fmin(fmax(f, 0.0), 2097151.0);                            // fmax(NaN, 0) = 0
    q[c] = (uint64_t)f;
  }
  keys[t] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  vals[t] = t;
}

// Length of the common prefix of the keys at sorted positions i and j; equal keys are told apart
// by their positions, so that every pair of leaves differs. -1 outside the array.
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

// Child references of the hierarchy: >= 0 internal node, < 0 leaf ~position.
// One thread per internal node (n - 1 of them): its range of sorted positions and its children.
__global__ void k_hierarchy(const uint64_t* __restrict__ keys, int n, int2* __restrict__ child, int2* __restrict__ range,
                            int* __restrict__ parent_node, int* __restrict__ parent_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int first = min(i, j), last = max(i, j);
  const int left = first == gamma ? ~gamma : gamma;
  const int right = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  child[i] = make_int2(left, right);
  range[i] = make_int2(first, last);
  if (left >= 0) parent_node[left] = i;
  else parent_leaf[~left] = i;
  if (right >= 0) parent_node[right] = i;
  else parent_leaf[~right] = i;
  if (i == 0) parent_node[0] = -1;
}

__device__ __forceinline__ Box6 box_union(const Box6& a, const Box6& b) {
  Box6 r;
  for (int c = 0; c < 3; ++c) {
    r.lo[c] = fmin(a.lo[c], b.lo[c]);
    r.hi[c] = fmax(a.hi[c], b.hi[c]);
  }
  return r;
}

// One thread per leaf walks towards the root; the second thread to reach a node computes its box.
__global__ void k_fit(int n, const int2* __restrict__ child, const int* __restrict__ parent_node, const int* __restrict__ parent_leaf,
                      const uint32_t* __restrict__ vals, const Box6* __restrict__ tri_boxes, Box6* node_boxes, int* visits) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int node = parent_leaf[p];
  while (node >= 0) {
    __threadfence();
    if (atomicAdd(&visits[node], 1) == 0) return;  // the sibling subtree is not done yet
    __threadfence();
    const int2 ch = child[node];
    const Box6 a = ch.x >= 0 ? node_boxes[ch.x] : tri_boxes[vals[~ch.x]];
    const Box6 b = ch.y >= 0 ? node_boxes[ch.y] : tri_boxes[vals[~ch.y]];
    node_boxes[node] = box_union(a, b);
    node = parent_node[node];
  }
}

// Nodes that cover more than RT_LEAF_MAX triangles are kept; smaller subtrees become leaves.
__global__ void k_mark(int n, const int2* __restrict__ range, uint32_t* __restrict__ kept) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  kept[i] = (range[i].y - range[i].x + 1) > RT_LEAF_MAX ? 1u : 0u;
}

__global__ void k_emit(int n, const int2* __restrict__ child, const int2* __restrict__ range, const int* __restrict__ parent_node,
                       const uint32_t* __restrict__ kept, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ vals,
                       const Box6* __restrict__ tri_boxes, const Box6* __restrict__ node_boxes, double pad, BvhNode* __restrict__ out,
                       unsigned int* __restrict__ stats /* [0] leaves, [1] deepest leaf */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1 || !kept[i]) return;
  BvhNode nd = {};
  const int2 ch = child[i];
  unsigned int leaves = 0;
  for (int c = 0; c < 2; ++c) {
    const int r = c ? ch.y : ch.x;
    Box6 b;
    int32_t ref;
    if (r < 0) {  // a single triangle
      b = tri_boxes[vals[~r]];
      ref = ~(int32_t)((uint32_t)(~r) * RT_LEAF_MAX + 0u);
      leaves++;
    } else {
      b = node_boxes[r];
      if (kept[r]) {
        ref = (int32_t)slot_of[r];
      } else {
        const int2 rg = range[r];
        ref = ~(int32_t)((uint32_t)rg.x * RT_LEAF_MAX + (uint32_t)(rg.y - rg.x));
        leaves++;
      }
    }
    nd.bx[c * 2 + 0] = __double2float_rd(b.lo[0] - pad);
    nd.bx[c * 2 + 1] = __double2float_ru(b.hi[0] + pad);
    nd.by[c * 2 + 0] = __double2float_rd(b.lo[1] - pad);
    nd.by[c * 2 + 1] = __double2float_ru(b.hi[1] + pad);
    nd.bz[c * 2 + 0] = __double2float_rd(b.lo[2] - pad);
    nd.bz[c * 2 + 1] = __double2float_ru(b.hi[2] + pad);
    nd.child[c] = ref;
  }
  out[slot_of[i]] = nd;
  if (leaves) {
    unsigned int depth = 1;  // interior nodes above a leaf of this node: the node and its ancestors
    for (int a = parent_node[i]; a >= 0; a = parent_node[a]) depth++;
    atomicAdd(&stats[0], leaves);
    atomicMax(&stats[1], depth);
  }
}


// ---- PLOC: parallel locally-ordered clustering (Meister and Bittner 2018) -----------------------
// Bottom-up: the clusters (at first one per triangle, in Morton order) each look RADIUS places
// to either side for the neighbour with which they would share the smallest box; pairs that
// choose each other merge into a node; the array is compacted and the search repeats until one
// cluster is left. Closer to a surface-area-heuristic tree than the Morton hierarchy, at the price
// of some tens of small launches.
//
// Node ids: 0..n-1 are the triangles at their sorted positions, n.. the merged nodes in the order
// of their creation (a parent always has a larger id than its children; the root is the last).
#define PLOC_NONE 0xffffffffu

__device__ __forceinline__ uint32_t ploc_hash(uint32_t a, uint32_t b) {  // of an unordered pair, a < b
  uint32_t h = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u) * 0x85ebca6bu;
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}

__device__ __forceinline__ double box_area(const Box6& b) {  // half the surface; NaN and infinity count as huge
  const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  const double a = dx * dy + dy * dz + dz * dx;
  return a < 1e300 ? a : 1e300;
}

__global__ void k_ploc_init(uint32_t n, const uint32_t* __restrict__ vals, const Box6* __restrict__ tri_boxes, Box6* __restrict__ nbox,
                            uint32_t* __restrict__ ncount, uint32_t* __restrict__ nparent, double* __restrict__ ncost,
                            uint32_t* __restrict__ cluster) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const Box6 b = tri_boxes[vals[p]];
  nbox[p] = b;
  ncost[p] = box_area(b);
  ncount[p] = 1u;
  nparent[p] = PLOC_NONE;
  cluster[p] = p;
}

// The neighbour of cluster i: smallest surface of the joint box; ties (whole regular meshes tie)
// go to the smaller hash of the pair of positions, then to the smaller pair of positions. The
// order is a strict order on unordered pairs, so the best pair overall always chooses each other
// and every round merges something. Boxes with NaN or infinite extent count as infinitely large.
__global__ void k_ploc_nn(uint32_t m, int radius, const uint32_t* __restrict__ cluster, const Box6* __restrict__ nbox,
                          uint32_t* __restrict__ nn) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Box6 bi = nbox[cluster[i]];
  const int lo = max(0, (int)i - radius), hi = min((int)m - 1, (int)i + radius);
  double best_a = 0.0;
  uint32_t best_h = 0;
  int best_j = -1;
  for (int j = lo; j <= hi; ++j) {
    if (j == (int)i) continue;
    const Box6 bj = nbox[cluster[j]];
    const double dx = fmax(bi.hi[0], bj.hi[0]) - fmin(bi.lo[0], bj.lo[0]);
    const double dy = fmax(bi.hi[1], bj.hi[1]) - fmin(bi.lo[1], bj.lo[1]);
    const double dz = fmax(bi.hi[2], bj.hi[2]) - fmin(bi.lo[2], bj.lo[2]);
    double a = dx * dy + dy * dz + dz * dx;
    if (!(a < 1.7e308)) a = 1.7976931348623157e308;  // NaN and infinity
    const uint32_t h = ploc_hash(min(i, (uint32_t)j), max(i, (uint32_t)j));
    // candidates are visited in rising j, so among equal (a, h) the smaller pair of positions is the first seen
    if (best_j < 0 || a < best_a || (a == best_a && h < best_h)) {
      best_a = a;
      best_h = h;
      best_j = j;
    }
  }
  nn[i] = (uint32_t)best_j;
}

// Low word: the cluster stays in the array (alone, or as the merged pair's first member); high
// word: it is the first member of a pair that merges.
__global__ void k_ploc_pair(uint32_t m, const uint32_t* __restrict__ nn, unsigned long long* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t j = nn[i];
  const bool mutual = nn[j] == i;
  const unsigned long long keep = !(mutual && i > j), lead = mutual && i < j;
  flags[i] = keep | (lead << 32);
}

__global__ void k_ploc_apply(uint32_t m, uint32_t base, const uint32_t* __restrict__ nn, const unsigned long long* __restrict__ flags,
                             const unsigned long long* __restrict__ scan, const uint32_t* __restrict__ cluster,
                             uint32_t* __restrict__ cluster_out, Box6* __restrict__ nbox, uint32_t* __restrict__ ncount,
                             uint32_t* __restrict__ nparent, int2* __restrict__ nchild, double* __restrict__ ncost,
                             uint32_t* __restrict__ nleaf, uint32_t* __restrict__ nnodes, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const unsigned long long f = flags[i], sc = scan[i];
  if (!(f & 1ull)) return;
  uint32_t id = cluster[i];
  if (f >> 32) {
    const uint32_t other = cluster[nn[i]];
    const uint32_t node = base + (uint32_t)(sc >> 32);
    const Box6 b = box_union(nbox[id], nbox[other]);
    const uint32_t count = ncount[id] + ncount[other];
    nbox[node] = b;
    ncount[node] = count;
    // Surface-area cost with one unit per box test and per triangle test (the host builder's
    // weights): the subtree becomes one leaf where testing its few triangles outright is no dearer
    // than descending into it.
    const double area = box_area(b);
    const double as_leaf = area * (double)count, as_tree = area + ncost[id] + ncost[other];
    const bool leaf = count <= RT_LEAF_MAX && as_leaf <= as_tree;
    ncost[node] = leaf ? as_leaf : as_tree;
    nleaf[node - n] = leaf ? 1u : 0u;
    // tree nodes in this subtree (none below a leaf), for the depth-first numbering
    nnodes[node - n] = leaf ? 0u : 1u + (id >= n ? nnodes[id - n] : 0u) + (other >= n ? nnodes[other - n] : 0u);
    nparent[node] = PLOC_NONE;
    nchild[node - n] = make_int2((int)id, (int)other);
    nparent[id] = node << 1;
    nparent[other] = (node << 1) | 1u;
    id = node;
  }
  cluster_out[(uint32_t)sc] = id;
}

// Position of a node's first triangle in depth-first order: the triangles of every left sibling
// passed on the way up to the root.
__device__ __forceinline__ uint32_t ploc_first_slot(uint32_t node, uint32_t n, const uint32_t* __restrict__ nparent,
                                                    const int2* __restrict__ nchild, const uint32_t* __restrict__ ncount,
                                                    uint32_t* depth_out) {
  uint32_t off = 0, depth = 0;
  for (uint32_t p = nparent[node]; p != PLOC_NONE; p = nparent[node]) {
    const uint32_t parent = p >> 1;
    if (p & 1u) off += ncount[nchild[parent - n].x];
    node = parent;
    depth++;
  }
  if (depth_out) *depth_out = depth;
  return off;
}

__global__ void k_ploc_order(uint32_t n, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ nparent,
                             const int2* __restrict__ nchild, const uint32_t* __restrict__ ncount, uint32_t* __restrict__ leaf_order) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  leaf_order[ploc_first_slot(p, n, nparent, nchild, ncount, nullptr)] = vals[p];
}

// A merged node stays a node of the tree unless it, or a small ancestor, was priced as a leaf.
__global__ void k_ploc_mark(uint32_t ni, uint32_t n, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nleaf,
                            const uint32_t* __restrict__ nparent, uint32_t* __restrict__ kept) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ni) return;
  uint32_t keep = nleaf[k] ? 0u : 1u;
  for (uint32_t p = nparent[n + k]; keep && p != PLOC_NONE; p = nparent[p >> 1]) {
    const uint32_t a = p >> 1;
    if (ncount[a] > RT_LEAF_MAX) break;
    if (nleaf[a - n]) keep = 0u;
  }
  kept[k] = keep;
}

// Preorder index of a kept node: one for every ancestor passed on the way up, plus the tree nodes
// of every left sibling's subtree. The root is node 0, a node's left subtree follows it directly.
__device__ __forceinline__ uint32_t ploc_node_slot(uint32_t node, uint32_t n, const uint32_t* __restrict__ nparent,
                                                   const int2* __restrict__ nchild, const uint32_t* __restrict__ nnodes, uint32_t* depth_out) {
  uint32_t slot = 0, depth = 0;
  for (uint32_t p = nparent[node]; p != PLOC_NONE; p = nparent[node]) {
    const uint32_t parent = p >> 1;
    slot += 1u;
    if (p & 1u) {
      const uint32_t left = (uint32_t)nchild[parent - n].x;
      if (left >= n) slot += nnodes[left - n];
    }
    node = parent;
    depth++;
  }
  if (depth_out) *depth_out = depth;
  return slot;
}

__global__ void k_ploc_emit(uint32_t ni, uint32_t n, const uint32_t* __restrict__ kept, const int2* __restrict__ nchild,
                            const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nparent, const uint32_t* __restrict__ nnodes,
                            const Box6* __restrict__ nbox, double pad, BvhNode* __restrict__ out, unsigned int* __restrict__ stats) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ni || !kept[k]) return;
  uint32_t depth = 0;
  const uint32_t slot = ploc_node_slot(n + k, n, nparent, nchild, nnodes, &depth);
  BvhNode nd = {};
  const int2 ch = nchild[k];
  unsigned int leaves = 0;
  for (int c = 0; c < 2; ++c) {
    const uint32_t r = (uint32_t)(c ? ch.y : ch.x);
    const Box6 b = nbox[r];
    int32_t ref;
    if (r >= n && kept[r - n]) {
      const uint32_t left = (uint32_t)ch.x;
      ref = (int32_t)(slot + 1u + (c && left >= n ? nnodes[left - n] : 0u));
    } else {
      const uint32_t first = ploc_first_slot(r, n, nparent, nchild, ncount, nullptr);
      ref = ~(int32_t)(first * RT_LEAF_MAX + (ncount[r] - 1u));
      leaves++;
    }
    nd.bx[c * 2 + 0] = __double2float_rd(b.lo[0] - pad);
    nd.bx[c * 2 + 1] = __double2float_ru(b.hi[0] + pad);
    nd.by[c * 2 + 0] = __double2float_rd(b.lo[1] - pad);
    nd.by[c * 2 + 1] = __double2float_ru(b.hi[1] + pad);
    nd.bz[c * 2 + 0] = __double2float_rd(b.lo[2] - pad);
    nd.bz[c * 2 + 1] = __double2float_ru(b.hi[2] + pad);
    nd.child[c] = ref;
  }
  out[slot] = nd;
  if (leaves) {
    atomicAdd(&stats[0], leaves);
    atomicMax(&stats[1], depth + 1u);  // interior nodes above the leaf: this node and its ancestors
  }
}

struct DevMem {  // frees what a failed or finished build allocated
  std::vector<void*> ptrs;
  ~DevMem() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  int get(T** out, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(1, count) * sizeof(T)) != hipSuccess) {
      rt_set_error("rt_scene_rebuild_on_device: out of device memory");
      return RT_ERR_OOM;
    }
    ptrs.push_back(p);
    *out = static_cast<T*>(p);
    return RT_OK;
  }
};

}  // namespace

static int rt_scene_rebuild_on_device_body(rt_scene* s, int device, uint32_t method);
#define BUILD_TOO_DEEP 1 /* internal: the tree would overflow the traversal stack */
// Clustering degenerates on triangle soups whose boxes all overlap (one cluster swallows its
// neighbours one at a time and the tree becomes a chain); the Morton hierarchy cannot be deeper
// than its 63 code bits plus the tie-break, so it takes over when that happens.
static int rebuild_with_fallback(rt_scene* s, int device, uint32_t method) {
  int rc = rt_scene_rebuild_on_device_body(s, device, method);
  if (rc == BUILD_TOO_DEEP && method == RT_BUILD_PLOC) rc = rt_scene_rebuild_on_device_body(s, device, RT_BUILD_LBVH);
  return rc == BUILD_TOO_DEEP ? RT_ERR_INVALID : rc;
}
extern "C" int rt_scene_rebuild_on_device(rt_scene* s, int device) {
  return rt_guard("rt_scene_rebuild_on_device", [&] { return rebuild_with_fallback(s, device, RT_BUILD_DEFAULT); });
}
extern "C" int rt_scene_rebuild_on_device_with(rt_scene* s, int device, uint32_t method) {
  return rt_guard("rt_scene_rebuild_on_device", [&] { return rebuild_with_fallback(s, device, method); });
}

// The clustering rounds of the PLOC build, after the sort. Leaves the node pool (boxes, counts,
// parents, children) filled and returns the number of merged nodes in *n_merged (n - 1).
static int ploc_rounds(DevMem& mem, uint32_t n, const uint32_t* d_vals_sorted, const Box6* d_tri_boxes, Box6* d_nbox, uint32_t* d_ncount,
                       uint32_t* d_nparent, int2* d_nchild, uint32_t* d_nleaf, uint32_t* d_nnodes, uint32_t* rounds_out) {
  int rc;
  uint32_t *d_cluster[2], *d_nn;
  double* d_ncost;
  if ((rc = mem.get(&d_ncost, 2 * (size_t)n - 1))) return rc;
  unsigned long long *d_flags, *d_scan;
  if ((rc = mem.get(&d_cluster[0], n)) || (rc = mem.get(&d_cluster[1], n)) || (rc = mem.get(&d_nn, n)) || (rc = mem.get(&d_flags, n)) ||
      (rc = mem.get(&d_scan, n)))
    return rc;
  size_t scan_bytes = 0;
  HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_scan, (int)n, nullptr));
  char* d_tmp;
  if ((rc = mem.get(&d_tmp, scan_bytes))) return rc;
  const char* re = std::getenv("RT_PLOC_RADIUS");
  const int radius = re && *re ? std::min(64, std::max(1, std::atoi(re))) : 16;
  hipLaunchKernelGGL(k_ploc_init, dim3((n + BUILD_BLOCK - 1) / BUILD_BLOCK), dim3(BUILD_BLOCK), 0, nullptr, n, d_vals_sorted, d_tri_boxes,
                     d_nbox, d_ncount, d_nparent, d_ncost, d_cluster[0]);
  HIP_TRY_B(hipGetLastError());
  uint32_t m = n, base = n, rounds = 0;
  int cur = 0;
  while (m > 1) {
    const dim3 grid((m + BUILD_BLOCK - 1) / BUILD_BLOCK);
    hipLaunchKernelGGL(k_ploc_nn, grid, dim3(BUILD_BLOCK), 0, nullptr, m, radius, d_cluster[cur], d_nbox, d_nn);
    hipLaunchKernelGGL(k_ploc_pair, grid, dim3(BUILD_BLOCK), 0, nullptr, m, d_nn, d_flags);
    HIP_TRY_B(hipGetLastError());
    size_t bytes = scan_bytes;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(d_tmp, bytes, d_flags, d_scan, (int)m, nullptr));
    hipLaunchKernelGGL(k_ploc_apply, grid, dim3(BUILD_BLOCK), 0, nullptr, m, base, d_nn, d_flags, d_scan, d_cluster[cur], d_cluster[cur ^ 1],
                       d_nbox, d_ncount, d_nparent, d_nchild, d_ncost, d_nleaf, d_nnodes, n);
    HIP_TRY_B(hipGetLastError());
    unsigned long long last_flag = 0, last_scan = 0;
    HIP_TRY_B(hipMemcpy(&last_flag, d_flags + (m - 1), sizeof last_flag, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMemcpy(&last_scan, d_scan + (m - 1), sizeof last_scan, hipMemcpyDeviceToHost));
    const unsigned long long total = last_flag + last_scan;  // both words: no carry between them (counts < 2^32)
    const uint32_t kept = (uint32_t)total, merged = (uint32_t)(total >> 32);
    if (merged == 0 || kept + merged != m) {
      rt_set_error("rt_scene_rebuild_on_device: internal error: a clustering round merged %u of %u clusters into %u", merged, m, kept);
      return RT_ERR_HIP;
    }
    base += merged;
    m = kept;
    cur ^= 1;
    rounds++;
  }
  if (base != 2u * n - 1u) {
    rt_set_error("rt_scene_rebuild_on_device: internal error: %u nodes for %u triangles", base, n);
    return RT_ERR_HIP;
  }
  *rounds_out = rounds;
  return RT_OK;
}

static int rt_scene_rebuild_on_device_body(rt_scene* s, int device, uint32_t method) {
  if (method != RT_BUILD_LBVH && method != RT_BUILD_PLOC) {
    rt_set_error("rt_scene_rebuild_on_device: unknown method %u", method);
    return RT_ERR_INVALID;
  }
  if (!s) {
    rt_set_error("rt_scene_rebuild_on_device: null scene");
    return RT_ERR_INVALID;
  }
  const int n_dev = rt_device_count();
  if (n_dev <= 0) {
    rt_set_error("rt_scene_rebuild_on_device: no HIP device is visible; this library has no CPU path");
    return RT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n_dev) {
    rt_set_error("rt_scene_rebuild_on_device: device %d out of range (0..%d)", device, n_dev - 1);
    return RT_ERR_INVALID;
  }
  rt_scene_pick_tree_spheres(s);
  const size_t n_tri = s->tri_obj.size();
  const size_t n = n_tri + s->tree_spheres.size();  // triangles, then the places of the spheres the tree culls
  if (n == 0) return RT_OK;  // nothing to build over
  HIP_TRY_B(hipSetDevice(device));
  const bool timing = std::getenv("RT_BUILD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[rt device build] %-9s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };

  // centre bounds of the triangle boxes, for the quantisation of the Morton codes
  double clo[3], chi[3];
  rt_scene_centre_bounds(s, clo, chi);
  // One scale for the three axes: the cells of the Morton grid are cubes, so neighbours along the
  // curve are neighbours in space whatever the proportions of the scene (a field of meshes 40 times
  // wider than tall, quantised per axis, would be sliced into sheets first).
  double ext_max = 0.0;
  for (int c = 0; c < 3; ++c) {
    if (!(chi[c] >= clo[c])) clo[c] = chi[c] = 0.0;
    const double ext = chi[c] - clo[c];
    if (std::isfinite(ext)) ext_max = std::max(ext_max, ext);
  }
  double scale[3];
  for (int c = 0; c < 3; ++c) scale[c] = ext_max > 0.0 ? 2097152.0 / ext_max : 0.0;
  const double pad = rt_scene_box_pad(s);
  phase("bounds");

  DevMem mem;
  int rc;
  double* d_verts;
  Box6 *d_tri_boxes, *d_node_boxes;
  uint64_t *d_keys, *d_keys_sorted;
  uint32_t *d_vals, *d_vals_sorted, *d_kept, *d_slot;
  int2 *d_child, *d_range;
  int *d_parent_node, *d_parent_leaf, *d_visits;
  unsigned int* d_stats;
  BvhNode* d_nodes;
  const size_t ni = n - 1;  // internal nodes of the hierarchy
  if ((rc = mem.get(&d_verts, n * 9)) || (rc = mem.get(&d_tri_boxes, n)) || (rc = mem.get(&d_node_boxes, ni)) ||
      (rc = mem.get(&d_keys, n)) || (rc = mem.get(&d_keys_sorted, n)) || (rc = mem.get(&d_vals, n)) ||
      (rc = mem.get(&d_vals_sorted, n)) || (rc = mem.get(&d_kept, ni + 1)) || (rc = mem.get(&d_slot, ni + 1)) ||
      (rc = mem.get(&d_child, ni)) || (rc = mem.get(&d_range, ni)) || (rc = mem.get(&d_parent_node, ni + 1)) ||
      (rc = mem.get(&d_parent_leaf, n)) || (rc = mem.get(&d_visits, ni + 1)) || (rc = mem.get(&d_stats, 2)) ||
      (rc = mem.get(&d_nodes, ni)))
    return rc;
  if (n_tri) HIP_TRY_B(hipMemcpy(d_verts, s->tri_vertices.data(), n_tri * 9 * sizeof(double), hipMemcpyHostToDevice));
  if (n > n_tri) {
    std::vector<double> places((n - n_tri) * 9);
    for (size_t k = 0; k < n - n_tri; ++k) rt_sphere_place_vertices(s->spheres[(size_t)s->tree_spheres[k]], &places[k * 9]);
    HIP_TRY_B(hipMemcpy(d_verts + n_tri * 9, places.data(), places.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  phase("copy in");
  const uint32_t grid_n = (uint32_t)((n + BUILD_BLOCK - 1) / BUILD_BLOCK);
  hipLaunchKernelGGL(k_boxes_codes, dim3(grid_n), dim3(BUILD_BLOCK), 0, nullptr, d_verts, (uint32_t)n, clo[0], clo[1], clo[2], scale[0],
                     scale[1], scale[2], d_tri_boxes, d_keys, d_vals);
  HIP_TRY_B(hipGetLastError());

  // sort positions by code (stable, so equal codes stay in flat order)
  size_t tmp_bytes = 0;
  HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (int)n, 0, 63, nullptr));
  char* d_tmp;
  if ((rc = mem.get(&d_tmp, tmp_bytes))) return rc;
  HIP_TRY_B(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (int)n, 0, 63, nullptr));

  phase("sort");
  std::vector<int32_t> leaf_order(n);
  std::vector<BvhNode> nodes;
  int32_t root_ref;
  uint32_t depth = 0, leaves = 1;
  bool order_on_device = false;  // the triangle order differs from the sorted order (PLOC)
  uint32_t* d_leaf_order = nullptr;
  if (n <= (size_t)RT_LEAF_MAX) {
    root_ref = ~(int32_t)(0u * RT_LEAF_MAX + (uint32_t)(n - 1));  // the whole scene is one leaf
  } else if (method == RT_BUILD_PLOC) {
    Box6* d_nbox;
    uint32_t *d_ncount, *d_nparent, *d_nleaf, *d_nnodes;
    if ((rc = mem.get(&d_nbox, 2 * n - 1)) || (rc = mem.get(&d_ncount, 2 * n - 1)) || (rc = mem.get(&d_nparent, 2 * n - 1)) ||
        (rc = mem.get(&d_leaf_order, n)) || (rc = mem.get(&d_nleaf, ni)) || (rc = mem.get(&d_nnodes, ni)))
      return rc;
    uint32_t rounds = 0;
    if ((rc = ploc_rounds(mem, (uint32_t)n, d_vals_sorted, d_tri_boxes, d_nbox, d_ncount, d_nparent, d_child, d_nleaf, d_nnodes, &rounds)) != RT_OK)
      return rc;
    phase("cluster");
    const uint32_t grid_i = (uint32_t)((ni + BUILD_BLOCK - 1) / BUILD_BLOCK);
    hipLaunchKernelGGL(k_ploc_mark, dim3(grid_i), dim3(BUILD_BLOCK), 0, nullptr, (uint32_t)ni, (uint32_t)n, d_ncount, d_nleaf, d_nparent, d_kept);
    HIP_TRY_B(hipGetLastError());
    uint32_t n_kept = 0;  // tree nodes under the root (the last merged node, never a leaf: more than RT_LEAF_MAX triangles)
    HIP_TRY_B(hipMemcpy(&n_kept, d_nnodes + (ni - 1), sizeof n_kept, hipMemcpyDeviceToHost));
    if (n_kept == 0 || n_kept > ni) {
      rt_set_error("rt_scene_rebuild_on_device: internal error: %u tree nodes for %zu triangles", n_kept, n);
      return RT_ERR_HIP;
    }
    HIP_TRY_B(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned int), nullptr));
    hipLaunchKernelGGL(k_ploc_emit, dim3(grid_i), dim3(BUILD_BLOCK), 0, nullptr, (uint32_t)ni, (uint32_t)n, d_kept, d_child, d_ncount, d_nparent,
                       d_nnodes, d_nbox, pad, d_nodes, d_stats);
    hipLaunchKernelGGL(k_ploc_order, dim3(grid_n), dim3(BUILD_BLOCK), 0, nullptr, (uint32_t)n, d_vals_sorted, d_nparent, d_child, d_ncount,
                       d_leaf_order);
    HIP_TRY_B(hipGetLastError());
    unsigned int stats[2] = {0, 0};
    HIP_TRY_B(hipMemcpy(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost));
    if (stats[1] > RT_BVH_MAX_DEPTH) {
      rt_set_error("rt_scene_rebuild_on_device: the clustered tree is %u levels deep (limit %d); the scene keeps its host-built tree",
                   stats[1], RT_BVH_MAX_DEPTH);
      return BUILD_TOO_DEEP;
    }
    phase("emit");
    nodes.resize(n_kept);
    HIP_TRY_B(hipMemcpy(nodes.data(), d_nodes, (size_t)n_kept * sizeof(BvhNode), hipMemcpyDeviceToHost));
    root_ref = 0;
    depth = stats[1];
    leaves = stats[0];
    order_on_device = true;
    if (std::getenv("RT_BUILD_TIMING")) std::fprintf(stderr, "rt_scene_rebuild_on_device: %u clustering rounds\n", rounds);
  } else {
    const uint32_t grid_i = (uint32_t)((ni + BUILD_BLOCK - 1) / BUILD_BLOCK);
    hipLaunchKernelGGL(k_hierarchy, dim3(grid_i), dim3(BUILD_BLOCK), 0, nullptr, d_keys_sorted, (int)n, d_child, d_range, d_parent_node,
                       d_parent_leaf);
    HIP_TRY_B(hipGetLastError());
    HIP_TRY_B(hipMemsetAsync(d_visits, 0, (ni + 1) * sizeof(int), nullptr));
    hipLaunchKernelGGL(k_fit, dim3(grid_n), dim3(BUILD_BLOCK), 0, nullptr, (int)n, d_child, d_parent_node, d_parent_leaf, d_vals_sorted,
                       d_tri_boxes, d_node_boxes, d_visits);
    HIP_TRY_B(hipGetLastError());
    hipLaunchKernelGGL(k_mark, dim3(grid_i), dim3(BUILD_BLOCK), 0, nullptr, (int)n, d_range, d_kept);
    HIP_TRY_B(hipGetLastError());
    size_t scan_bytes = 0;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_kept, d_slot, (int)ni, nullptr));
    char* d_scan;
    if ((rc = mem.get(&d_scan, scan_bytes))) return rc;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_kept, d_slot, (int)ni, nullptr));
    HIP_TRY_B(hipMemsetAsync(d_stats, 0, 2 * sizeof(unsigned int), nullptr));
    hipLaunchKernelGGL(k_emit, dim3(grid_i), dim3(BUILD_BLOCK), 0, nullptr, (int)n, d_child, d_range, d_parent_node, d_kept, d_slot,
                       d_vals_sorted, d_tri_boxes, d_node_boxes, pad, d_nodes, d_stats);
    HIP_TRY_B(hipGetLastError());
    uint32_t last_kept = 0, last_slot = 0;
    unsigned int stats[2] = {0, 0};
    HIP_TRY_B(hipMemcpy(&last_kept, d_kept + (ni - 1), sizeof last_kept, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMemcpy(&last_slot, d_slot + (ni - 1), sizeof last_slot, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMemcpy(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost));
    const size_t n_nodes = (size_t)last_slot + last_kept;
    if (stats[1] > RT_BVH_MAX_DEPTH) {
      rt_set_error("rt_scene_rebuild_on_device: the Morton hierarchy is %u levels deep (limit %d); the scene keeps its host-built tree",
                   stats[1], RT_BVH_MAX_DEPTH);
      return BUILD_TOO_DEEP;
    }
    nodes.resize(n_nodes);
    HIP_TRY_B(hipMemcpy(nodes.data(), d_nodes, n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    root_ref = 0;  // node 0 of the hierarchy is the root, covers everything, and is kept: slot 0
    depth = stats[1];
    leaves = stats[0];
  }
  HIP_TRY_B(hipMemcpy(leaf_order.data(), order_on_device ? d_leaf_order : d_vals_sorted, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int32_t& e : leaf_order)  // the k-th sphere of the tree -> n_triangles + its sphere index
    if ((size_t)e >= n_tri) e = (int32_t)(n_tri + (size_t)s->tree_spheres[(size_t)e - n_tri]);
  phase("copy out");

  // the scene takes the new tree; its device copy, if any, is stale
  if (s->dev) {
    rt_device_scene_free(s->dev);
    s->dev = nullptr;
  }
  s->nodes.swap(nodes);
  s->leaf_order.swap(leaf_order);
  s->root_ref = root_ref;
  s->bvh_depth = depth;
  s->bvh_leaves = leaves;
  s->has_tree = true;
  rt_fill_leaf_records(s);
  phase("records");
  return RT_OK;
}
