// rt_device.hip — device half of the library: scene upload, kernels, and the C entry points
// that launch them. Written for gfx950 only.
//
// Kernels
//   k_trace        batch Scene::hit (src/main.rs:776-814), one ray per lane
//   k_primary      the pixel-centre camera ray of every pixel + Scene::hit (parity buffer)
//   k_paths        render's per-sample body (src/main.rs:995-1019) and raytrace (817-975) as an
//                  iterative loop; lanes that finish a path pull the next (pixel, sample) from a
//                  work counter, so a wave stays full however unequal the path lengths are
//   k_resolve      the per-pixel sums of src/main.rs:1020-1037 in sample order
//
// HBM layout: see rt_types.h. The per-sample radiance buffer is [3][n_paths] binary64, where
// a path index is ((tile * spp + sample) * 64 + pixel-in-tile) over 8x8-pixel tiles, so the 64
// lanes of a wave start on one tile and one sample (coherent primary rays, coalesced stores)
// and k_resolve reads 64 consecutive values per sample.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rt_kernels.hip.h"
#include "rt_types.h"

using namespace rtk;

// ------------------------------------------------------------------------------------------
// device scene
// ------------------------------------------------------------------------------------------
enum { CNT_RAYS = 0, CNT_NODES, CNT_TRIS, CNT_SPHERES, CNT_SHADED, CNT_PATHS, CNT_QUEUE, CNT_N = 8 };

struct DeviceScene {
  int device = -1;
  SceneK k{};
  std::vector<void*> allocs;
  uint64_t bytes = 0;
  uint32_t stack_entries = 2;
  // work buffers, grown on demand
  double* samples = nullptr;
  size_t samples_bytes = 0;
  double* totals = nullptr;
  size_t totals_bytes = 0;
  unsigned long long* counters = nullptr;  // CNT_N words
  // staging for the host-buffer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  rt_timing timing{};
  bool have_timing = false;
  hipStream_t last_stream = nullptr;
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RT_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

void rt_device_scene_free(DeviceScene* d) {
  if (!d) return;
  if (d->device >= 0) (void)hipSetDevice(d->device);
  (void)hipDeviceSynchronize();
  for (void* p : d->allocs) (void)hipFree(p);
  if (d->samples) (void)hipFree(d->samples);
  if (d->totals) (void)hipFree(d->totals);
  if (d->counters) (void)hipFree(d->counters);
  if (d->stage) (void)hipFree(d->stage);
  for (hipEvent_t e : d->ev)
    if (e) (void)hipEventDestroy(e);
  delete d;
}

int rt_device_scene_info(const DeviceScene* d, uint32_t* device_plus_one, uint64_t* bytes) {
  *device_plus_one = (uint32_t)(d->device + 1);
  *bytes = d->bytes;
  return RT_OK;
}

extern "C" int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

template <typename T>
static int upload_array(DeviceScene* d, const T* src, size_t n, const T** out) {
  *out = nullptr;
  if (n == 0) return RT_OK;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, n * sizeof(T)));
  d->allocs.push_back(p);
  d->bytes += n * sizeof(T);
  HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return RT_OK;
}

extern "C" int rt_scene_upload(rt_scene* s, int device) {
  if (!s) {
    rt_set_error("rt_scene_upload: null scene");
    return RT_ERR_INVALID;
  }
  int n = rt_device_count();
  if (n <= 0) {
    rt_set_error("rt_scene_upload: no HIP device is visible; this library has no CPU path");
    return RT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    rt_set_error("rt_scene_upload: device %d out of range (0..%d)", device, n - 1);
    return RT_ERR_INVALID;
  }
  if (s->dev && s->dev->device == device) return RT_OK;
  if (s->dev) {
    rt_device_scene_free(s->dev);
    s->dev = nullptr;
  }
  HIP_TRY(hipSetDevice(device));
  DeviceScene* d = new (std::nothrow) DeviceScene();
  if (!d) return RT_ERR_OOM;
  d->device = device;
  int rc = RT_OK;
  auto fail = [&](int code) {
    rt_device_scene_free(d);
    return code;
  };
  if ((rc = upload_array(d, s->nodes.data(), s->nodes.size(), &d->k.nodes)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->tri_geom.data(), s->tri_geom.size(), &d->k.geom)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->tri_shade.data(), s->tri_shade.size(), &d->k.shade)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->spheres.data(), s->spheres.size(), &d->k.spheres)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->materials.data(), s->materials.size(), &d->k.mats)) != RT_OK) return fail(rc);
  std::vector<TextureRec> trec(s->textures.size());
  for (size_t i = 0; i < s->textures.size(); ++i) {
    const uint8_t* px = nullptr;
    if ((rc = upload_array(d, s->textures[i].rgba.data(), s->textures[i].rgba.size(), &px)) != RT_OK) return fail(rc);
    trec[i] = TextureRec{px, s->textures[i].w, s->textures[i].h};
  }
  if ((rc = upload_array(d, trec.data(), trec.size(), &d->k.tex)) != RT_OK) return fail(rc);
  d->k.root_ref = s->root_ref;
  d->k.n_spheres = (uint32_t)s->spheres.size();
  d->stack_entries = s->bvh_depth + 2;
  if (hipMalloc((void**)&d->counters, CNT_N * sizeof(unsigned long long)) != hipSuccess) {
    rt_set_error("rt_scene_upload: hipMalloc of the counters failed");
    return fail(RT_ERR_HIP);
  }
  (void)hipMemset(d->counters, 0, CNT_N * sizeof(unsigned long long));
  for (int i = 0; i < 4; ++i)
    if (hipEventCreate(&d->ev[i]) != hipSuccess) {
      rt_set_error("rt_scene_upload: hipEventCreate failed");
      return fail(RT_ERR_HIP);
    }
  s->dev = d;
  return RT_OK;
}

static int require_device(rt_scene* s, const char* who) {
  if (!s) {
    rt_set_error("%s: null scene", who);
    return RT_ERR_INVALID;
  }
  if (!s->dev) {
    int rc = rt_scene_upload(s, 0);
    if (rc != RT_OK) return rc;
  }
  if (hipSetDevice(s->dev->device) != hipSuccess) {
    rt_set_error("%s: hipSetDevice(%d) failed", who, s->dev->device);
    return RT_ERR_HIP;
  }
  return RT_OK;
}

static int ensure_bytes(void** p, size_t* have, size_t want) {
  if (*have >= want) return RT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(p, want));
  *have = want;
  return RT_OK;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
extern __shared__ int32_t g_stack[];  // [stack_entries][blockDim.x]

__device__ __forceinline__ void store_hit(const SceneK& sc, const HitK& h, rt_hit* out) {
  rt_hit r;
  if (h.slot >= 0) {
    const TriShade* s = sc.shade + h.slot;
    r.t = h.t;
    r.u = h.u;
    r.v = h.v;
    r.obj = s->obj;
    r.tri = s->tri;
  } else if (h.sphere >= 0) {
    r.t = h.t;
    r.u = 0.0;
    r.v = 0.0;
    r.obj = sc.spheres[h.sphere].obj;
    r.tri = -1;
  } else {
    r.t = 0.0;
    r.u = 0.0;
    r.v = 0.0;
    r.obj = -1;
    r.tri = -1;
  }
  *out = r;
}

__device__ __forceinline__ void flush_counts(const WorkCount& wc, unsigned long long* counters, bool count_all) {
  // one atomic per wave and counter
  unsigned long long v = wc.rays;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counters[CNT_RAYS], v);
  if (count_all) {
    unsigned long long a = wc.nodes, b = wc.tris, c = wc.spheres, e = wc.shaded;
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_down(a, off);
      b += __shfl_down(b, off);
      c += __shfl_down(c, off);
      e += __shfl_down(e, off);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&counters[CNT_NODES], a);
      atomicAdd(&counters[CNT_TRIS], b);
      atomicAdd(&counters[CNT_SPHERES], c);
      atomicAdd(&counters[CNT_SHADED], e);
    }
  }
}

template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_trace(SceneK sc, const double* __restrict__ org, const double* __restrict__ dir,
                                                    size_t n, rt_hit* __restrict__ out, unsigned long long* counters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  WorkCount wc{};
  if (i < n) {
    const d3 o = mk(org[i * 3], org[i * 3 + 1], org[i * 3 + 2]);
    const d3 d = mk(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
    HitK h;
    trace_ray<COUNT>(sc, o, d, g_stack + threadIdx.x, blockDim.x, h, wc);
    wc.rays = 1;
    store_hit(sc, h, out + i);
  }
  flush_counts(wc, counters, COUNT);
}

__global__ __launch_bounds__(RT_BLOCK) void k_primary(SceneK sc, CameraK cam, int32_t* __restrict__ obj, int32_t* __restrict__ tri,
                                                      double* __restrict__ t, double* __restrict__ u, double* __restrict__ v,
                                                      unsigned long long* counters) {
  const size_t npix = (size_t)cam.width * cam.height;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  WorkCount wc{};
  if (i < npix) {
    const uint32_t px = (uint32_t)(i % cam.width), py = (uint32_t)(i / cam.width);
    d3 o, d;
    camera_ray(cam, px, py, 0ull, o, d);
    HitK h;
    trace_ray<false>(sc, o, d, g_stack + threadIdx.x, blockDim.x, h, wc);
    wc.rays = 1;
    rt_hit r;
    store_hit(sc, h, &r);
    obj[i] = r.obj;
    tri[i] = r.tri;
    t[i] = r.t;
    u[i] = r.u;
    v[i] = r.v;
  }
  flush_counts(wc, counters, false);
}

// Which pixels a call renders and how path indices map onto them.
struct TileMap {
  uint32_t width, height;
  uint32_t tiles_x;      // 8x8 tiles per image row (whole-image mode)
  uint32_t big;          // 8x8 tiles per side of a big tile, 0 = whole-image mode
  uint32_t bigs_x;       // big tiles per image row
  uint32_t rank, ranks;  // this call renders big tiles rank, rank + ranks, ...
  uint32_t n_tiles;      // 8x8 tiles this call renders
};

__host__ __device__ inline bool tile_pixel(const TileMap& tm, uint32_t tile, uint32_t lane_pix, uint32_t* x, uint32_t* y) {
  uint32_t tx, ty;
  if (tm.big == 0) {
    tx = tile % tm.tiles_x;
    ty = tile / tm.tiles_x;
  } else {
    const uint32_t per_big = tm.big * tm.big;
    const uint32_t b = tm.rank + (tile / per_big) * tm.ranks;
    const uint32_t j = tile % per_big;
    tx = (b % tm.bigs_x) * tm.big + j % tm.big;
    ty = (b / tm.bigs_x) * tm.big + j / tm.big;
  }
  *x = tx * 8u + (lane_pix & 7u);
  *y = ty * 8u + (lane_pix >> 3);
  return *x < tm.width && *y < tm.height;
}

struct PathParams {
  CameraK cam;
  TileMap tm;
  uint64_t seed;
  uint64_t n_paths;       // n_tiles * spp * 64
  uint32_t spp;           // samples in this launch
  uint32_t first_sample;  // stream index of this launch's sample 0
  int32_t max_depth;
};

template <bool COUNT, bool REGEN>
__global__ __launch_bounds__(RT_BLOCK) void k_paths(SceneK sc, PathParams pp, double* __restrict__ samples,
                                                    unsigned long long* counters) {
  int32_t* stk = g_stack + threadIdx.x;
  const int stride = blockDim.x;
  const int lane = threadIdx.x & 63;
  WorkCount wc{};
  uint32_t n_done = 0;

  bool alive = false;
  bool exhausted = false;  // wave-uniform: the work counter has run out
  uint64_t path = 0, key = 0;
  int depth = 0;
  d3 o = mk(0, 0, 0), dir = mk(0, 0, 0), thr = mk(1, 1, 1), rad = mk(0, 0, 0);

  if (!REGEN) {
    // one path per thread, in path-index order
    path = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    exhausted = true;
    if (path < pp.n_paths) {
      const uint64_t per_tile = (uint64_t)pp.spp * 64u;
      const uint32_t tile = (uint32_t)(path / per_tile);
      const uint32_t rem = (uint32_t)(path % per_tile);
      uint32_t x, y;
      if (tile_pixel(pp.tm, tile, rem & 63u, &x, &y)) {
        key = path_key(pp.seed, (uint64_t)y * pp.tm.width + x, (uint64_t)pp.first_sample + (rem >> 6));
        camera_ray(pp.cam, x, y, key, o, dir);
        alive = true;
      } else {
        samples[path] = 0.0;
        samples[pp.n_paths + path] = 0.0;
        samples[2 * pp.n_paths + path] = 0.0;
      }
    }
  }

  for (;;) {
    if (REGEN && !exhausted) {
      const unsigned long long need = __ballot(!alive);
      if (need) {
        const int n = __popcll(need);
        const int leader = __ffsll((long long)need) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&counters[CNT_QUEUE], (unsigned long long)n);
        base = __shfl(base, leader);
        if (base + (unsigned long long)n >= pp.n_paths) exhausted = true;
        if (!alive) {
          path = base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
          if (path < pp.n_paths) {
            const uint64_t per_tile = (uint64_t)pp.spp * 64u;
            const uint32_t tile = (uint32_t)(path / per_tile);
            const uint32_t rem = (uint32_t)(path % per_tile);
            uint32_t x, y;
            if (tile_pixel(pp.tm, tile, rem & 63u, &x, &y)) {
              key = path_key(pp.seed, (uint64_t)y * pp.tm.width + x, (uint64_t)pp.first_sample + (rem >> 6));
              camera_ray(pp.cam, x, y, key, o, dir);
              thr = mk(1.0, 1.0, 1.0);
              rad = mk(0.0, 0.0, 0.0);
              depth = 0;
              alive = true;
            } else {
              samples[path] = 0.0;
              samples[pp.n_paths + path] = 0.0;
              samples[2 * pp.n_paths + path] = 0.0;
            }
          }
        }
      }
    }
    if (!__any(alive)) {
      if (exhausted) break;
      continue;
    }
    if (alive) {
      HitK h;
      wc.rays++;
      trace_ray<COUNT>(sc, o, dir, stk, stride, h, wc);
      bool finished;
      if (h.slot < 0 && h.sphere < 0) {
        const d3 s = sky(dir);  // src/main.rs:826
        rad = rad + mk(thr.x * s.x, thr.y * s.y, thr.z * s.z);
        finished = true;
      } else {
        if (COUNT) wc.shaded++;
        shade_hit(sc, h, key, depth, o, dir, thr, rad);
        depth += 1;
        finished = depth > pp.max_depth;  // src/main.rs:818: deeper calls return black
      }
      if (finished) {
        samples[path] = rad.x;
        samples[pp.n_paths + path] = rad.y;
        samples[2 * pp.n_paths + path] = rad.z;
        alive = false;
        n_done++;
      }
    }
  }
  flush_counts(wc, counters, COUNT);
  if (COUNT) {
    unsigned long long v = n_done;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) atomicAdd(&counters[CNT_PATHS], v);
  }
}

// Sums a pixel's samples in sample order (src/main.rs:1020-1024). `totals` carries the running
// sum between launches when a call is split into several sample chunks. On the last chunk the
// mean is added to the canvas (src/main.rs:1029-1037), or the plain sum to an f32 buffer.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(TileMap tm, const double* __restrict__ samples, uint64_t n_paths, uint32_t spp,
                                                      double* __restrict__ totals, int first_chunk, int last_chunk, double divisor,
                                                      double* __restrict__ canvas, float* __restrict__ sum_f32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // tile * 64 + pixel-in-tile
  const uint64_t n_pix = (uint64_t)tm.n_tiles * 64u;
  if (i >= n_pix) return;
  const uint32_t tile = (uint32_t)(i >> 6), lp = (uint32_t)(i & 63u);
  uint32_t x, y;
  if (!tile_pixel(tm, tile, lp, &x, &y)) return;
  double tr = 0.0, tg = 0.0, tb = 0.0;
  if (!first_chunk) {
    tr = totals[i];
    tg = totals[n_pix + i];
    tb = totals[2 * n_pix + i];
  }
  const uint64_t base = (uint64_t)tile * spp * 64u + lp;
  for (uint32_t s = 0; s < spp; ++s) {
    const uint64_t p = base + (uint64_t)s * 64u;
    tr += samples[p];
    tg += samples[n_paths + p];
    tb += samples[2 * n_paths + p];
  }
  if (!last_chunk) {
    totals[i] = tr;
    totals[n_pix + i] = tg;
    totals[2 * n_pix + i] = tb;
    return;
  }
  const size_t pix = ((size_t)y * tm.width + x) * 3;
  if (canvas) {
    canvas[pix + 0] += tr / divisor;
    canvas[pix + 1] += tg / divisor;
    canvas[pix + 2] += tb / divisor;
  }
  if (sum_f32) {
    sum_f32[pix + 0] += (float)tr;
    sum_f32[pix + 1] += (float)tg;
    sum_f32[pix + 2] += (float)tb;
  }
}

// ------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------
static size_t stack_lds_bytes(const DeviceScene* d) { return (size_t)d->stack_entries * RT_BLOCK * sizeof(int32_t); }

static int check_params(const rt_render_params* p, const char* who) {
  if (!p) {
    rt_set_error("%s: null parameters", who);
    return RT_ERR_INVALID;
  }
  if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536) {
    rt_set_error("%s: bad image size %ux%u", who, p->width, p->height);
    return RT_ERR_INVALID;
  }
  if (p->max_depth < 0 || p->max_depth > 100000) {
    rt_set_error("%s: bad max_depth %d", who, p->max_depth);
    return RT_ERR_INVALID;
  }
  if (!(p->fov_y_deg > 0.0 && p->fov_y_deg < 180.0)) {
    rt_set_error("%s: bad fov_y_deg %g", who, p->fov_y_deg);
    return RT_ERR_INVALID;
  }
  if (!(p->dof_strength >= 0.0)) {
    rt_set_error("%s: bad dof_strength %g", who, p->dof_strength);
    return RT_ERR_INVALID;
  }
  if (p->tile_size != 0 && p->tile_ranks > 1) {
    if (p->tile_size % 8 != 0 || p->tile_rank >= p->tile_ranks) {
      rt_set_error("%s: tile_size must be a multiple of 8 and tile_rank < tile_ranks", who);
      return RT_ERR_INVALID;
    }
  }
  return RT_OK;
}

static CameraK make_camera(const rt_render_params* p) {
  CameraK c;
  const double fov_y = p->fov_y_deg * 3.14159265358979323846 / 180.0;  // src/main.rs:978
  c.tan_half = std::tan(fov_y / 2.0);
  c.aspect = (double)p->width / (double)p->height;  // src/main.rs:979
  c.dof = p->dof_strength;
  c.width = p->width;
  c.height = p->height;
  c.jitter = (p->flags & RT_FLAG_NO_JITTER) ? 0u : 1u;
  return c;
}

static TileMap make_tilemap(const rt_render_params* p) {
  TileMap tm;
  tm.width = p->width;
  tm.height = p->height;
  tm.tiles_x = (p->width + 7) / 8;
  const uint32_t tiles_y = (p->height + 7) / 8;
  if (p->tile_size == 0 || p->tile_ranks <= 1) {
    tm.big = 0;
    tm.bigs_x = 0;
    tm.rank = 0;
    tm.ranks = 1;
    tm.n_tiles = tm.tiles_x * tiles_y;
  } else {
    tm.big = p->tile_size / 8;
    tm.bigs_x = (p->width + p->tile_size - 1) / p->tile_size;
    const uint32_t bigs_y = (p->height + p->tile_size - 1) / p->tile_size;
    const uint32_t n_big = tm.bigs_x * bigs_y;
    tm.rank = p->tile_rank;
    tm.ranks = p->tile_ranks;
    const uint32_t mine = n_big > tm.rank ? (n_big - tm.rank + tm.ranks - 1) / tm.ranks : 0;
    tm.n_tiles = mine * tm.big * tm.big;
  }
  return tm;
}

// Upper bound of the per-sample buffer; a call with more samples runs in chunks.
static size_t sample_budget_bytes() {
  const char* e = std::getenv("RT_SAMPLE_BUFFER_MIB");
  size_t mib = 8192;
  if (e && *e) {
    long v = std::atol(e);
    if (v > 0) mib = (size_t)v;
  }
  return mib << 20;
}

static int render_impl(rt_scene* s, const rt_render_params* p, double* d_canvas, float* d_sum, hipStream_t stream, const char* who) {
  int rc = require_device(s, who);
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, who)) != RT_OK) return rc;
  if (!d_canvas && !d_sum) {
    rt_set_error("%s: null output buffer", who);
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  d->last_stream = stream;
  d->have_timing = false;
  std::memset(&d->timing, 0, sizeof d->timing);
  HIP_TRY(hipMemsetAsync(d->counters, 0, CNT_N * sizeof(unsigned long long), stream));
  if (p->spp == 0) return RT_OK;

  const TileMap tm = make_tilemap(p);
  if (tm.n_tiles == 0) return RT_OK;
  const uint64_t n_pix = (uint64_t)tm.n_tiles * 64u;
  const size_t budget = sample_budget_bytes();
  uint32_t chunk = (uint32_t)std::min<uint64_t>(p->spp, std::max<uint64_t>(1, budget / (n_pix * 24u)));
  const uint32_t n_chunks = (p->spp + chunk - 1) / chunk;
  if ((rc = ensure_bytes((void**)&d->samples, &d->samples_bytes, (size_t)n_pix * chunk * 24u)) != RT_OK) return rc;
  if (n_chunks > 1 && (rc = ensure_bytes((void**)&d->totals, &d->totals_bytes, (size_t)n_pix * 24u)) != RT_OK) return rc;

  const bool timing = (p->flags & RT_FLAG_TIMING) != 0;
  const bool count = (p->flags & RT_FLAG_COUNT_WORK) != 0;
  const bool regen = (p->flags & RT_FLAG_NO_REGEN) == 0;
  const size_t lds = stack_lds_bytes(d);
  const double divisor = (double)(p->spp_total ? p->spp_total : p->spp);

  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device);
  float ms_paths = 0.f, ms_resolve = 0.f;

  for (uint32_t c = 0; c < n_chunks; ++c) {
    PathParams pp;
    pp.cam = make_camera(p);
    pp.tm = tm;
    pp.seed = p->seed;
    pp.spp = std::min(chunk, p->spp - c * chunk);
    pp.first_sample = p->first_sample + c * chunk;
    pp.n_paths = n_pix * pp.spp;
    pp.max_depth = p->max_depth;
    const uint64_t blocks_all = (pp.n_paths + RT_BLOCK - 1) / RT_BLOCK;
    if (blocks_all > 0x7fffffffull) {
      rt_set_error("%s: too many paths in one launch; lower RT_SAMPLE_BUFFER_MIB", who);
      return RT_ERR_INVALID;
    }
    uint32_t grid;
    if (regen) {
      grid = (uint32_t)std::min<uint64_t>(blocks_all, (uint64_t)cus * 8u);
      HIP_TRY(hipMemsetAsync(d->counters + CNT_QUEUE, 0, sizeof(unsigned long long), stream));
    } else {
      grid = (uint32_t)blocks_all;
    }
    if (timing) HIP_TRY(hipEventRecord(d->ev[0], stream));
    if (regen) {
      if (count) hipLaunchKernelGGL((k_paths<true, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, d->samples, d->counters);
      else hipLaunchKernelGGL((k_paths<false, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, d->samples, d->counters);
    } else {
      if (count) hipLaunchKernelGGL((k_paths<true, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, d->samples, d->counters);
      else hipLaunchKernelGGL((k_paths<false, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, d->samples, d->counters);
    }
    HIP_TRY(hipGetLastError());
    if (timing) HIP_TRY(hipEventRecord(d->ev[1], stream));
    const uint32_t rgrid = (uint32_t)((n_pix + RT_BLOCK - 1) / RT_BLOCK);
    hipLaunchKernelGGL(k_resolve, dim3(rgrid), dim3(RT_BLOCK), 0, stream, tm, d->samples, pp.n_paths, pp.spp, d->totals,
                       (int)(c == 0), (int)(c + 1 == n_chunks), divisor, d_canvas, d_sum);
    HIP_TRY(hipGetLastError());
    if (timing) {
      HIP_TRY(hipEventRecord(d->ev[2], stream));
      HIP_TRY(hipEventSynchronize(d->ev[2]));
      float a = 0.f, b = 0.f;
      HIP_TRY(hipEventElapsedTime(&a, d->ev[0], d->ev[1]));
      HIP_TRY(hipEventElapsedTime(&b, d->ev[1], d->ev[2]));
      ms_paths += a;
      ms_resolve += b;
    }
  }
  if (timing) {
    d->timing.ms_paths = ms_paths;
    d->timing.ms_resolve = ms_resolve;
    d->timing.path_launches = n_chunks;
    d->timing.resolve_launches = n_chunks;
    d->have_timing = true;
  }
  return RT_OK;
}

extern "C" int rt_render_device(rt_scene* s, const rt_render_params* p, double* device_canvas, void* stream) {
  return render_impl(s, p, device_canvas, nullptr, (hipStream_t)stream, "rt_render_device");
}

extern "C" int rt_render_sum_f32_device(rt_scene* s, const rt_render_params* p, float* device_sum, void* stream) {
  return render_impl(s, p, nullptr, device_sum, (hipStream_t)stream, "rt_render_sum_f32_device");
}

extern "C" int rt_render_accumulate(rt_scene* s, const rt_render_params* p, double* canvas, uint64_t* rays_out) {
  int rc = require_device(s, "rt_render_accumulate");
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, "rt_render_accumulate")) != RT_OK) return rc;
  if (!canvas) {
    rt_set_error("rt_render_accumulate: null canvas");
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(double);
  if ((rc = ensure_bytes(&d->stage, &d->stage_bytes, bytes)) != RT_OK) return rc;
  HIP_TRY(hipMemcpy(d->stage, canvas, bytes, hipMemcpyHostToDevice));
  rc = render_impl(s, p, (double*)d->stage, nullptr, nullptr, "rt_render_accumulate");
  if (rc != RT_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(canvas, d->stage, bytes, hipMemcpyDeviceToHost));
  if (rays_out) {
    unsigned long long c[CNT_N];
    HIP_TRY(hipMemcpy(c, d->counters, sizeof c, hipMemcpyDeviceToHost));
    *rays_out = c[CNT_RAYS];
  }
  return RT_OK;
}

extern "C" int rt_trace(rt_scene* s, const double* org3, const double* dir3, size_t n, rt_hit* out) {
  int rc = require_device(s, "rt_trace");
  if (rc != RT_OK) return rc;
  if (n == 0) return RT_OK;
  if (!org3 || !dir3 || !out) {
    rt_set_error("rt_trace: null argument");
    return RT_ERR_INVALID;
  }
  if (n > 0x7fffffffull * RT_BLOCK) {
    rt_set_error("rt_trace: too many rays in one call");
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  const size_t ray_bytes = n * 3 * sizeof(double), hit_bytes = n * sizeof(rt_hit);
  if ((rc = ensure_bytes(&d->stage, &d->stage_bytes, 2 * ray_bytes + hit_bytes)) != RT_OK) return rc;
  double* d_org = (double*)d->stage;
  double* d_dir = d_org + n * 3;
  rt_hit* d_hit = (rt_hit*)(d_dir + n * 3);
  HIP_TRY(hipMemcpy(d_org, org3, ray_bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_dir, dir3, ray_bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d->counters, 0, CNT_N * sizeof(unsigned long long)));
  const uint32_t grid = (uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK);
  hipLaunchKernelGGL((k_trace<true>), dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), nullptr, d->k, d_org, d_dir, n, d_hit, d->counters);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_hit, hit_bytes, hipMemcpyDeviceToHost));
  d->last_stream = nullptr;
  return RT_OK;
}

extern "C" int rt_primary_hits(rt_scene* s, const rt_render_params* p, int32_t* obj, int32_t* tri, double* t, double* u, double* v) {
  int rc = require_device(s, "rt_primary_hits");
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, "rt_primary_hits")) != RT_OK) return rc;
  if (!obj || !tri || !t || !u || !v) {
    rt_set_error("rt_primary_hits: null output");
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  const size_t npix = (size_t)p->width * p->height;
  const size_t bytes = npix * (2 * sizeof(int32_t) + 3 * sizeof(double));
  if ((rc = ensure_bytes(&d->stage, &d->stage_bytes, bytes)) != RT_OK) return rc;
  double* d_t = (double*)d->stage;
  double* d_u = d_t + npix;
  double* d_v = d_u + npix;
  int32_t* d_obj = (int32_t*)(d_v + npix);
  int32_t* d_tri = d_obj + npix;
  rt_render_params q = *p;
  q.flags |= RT_FLAG_NO_JITTER;
  q.dof_strength = 0.0;
  CameraK cam = make_camera(&q);
  HIP_TRY(hipMemset(d->counters, 0, CNT_N * sizeof(unsigned long long)));
  const uint32_t grid = (uint32_t)((npix + RT_BLOCK - 1) / RT_BLOCK);
  hipLaunchKernelGGL(k_primary, dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), nullptr, d->k, cam, d_obj, d_tri, d_t, d_u, d_v, d->counters);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(t, d_t, npix * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(u, d_u, npix * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(v, d_v, npix * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(obj, d_obj, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(tri, d_tri, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
  d->last_stream = nullptr;
  return RT_OK;
}

extern "C" int rt_scene_last_stats(rt_scene* s, rt_stats* out) {
  if (!s || !out) {
    rt_set_error("rt_scene_last_stats: null argument");
    return RT_ERR_INVALID;
  }
  if (!s->dev) {
    rt_set_error("rt_scene_last_stats: the scene has not been uploaded");
    return RT_ERR_NO_DEVICE;
  }
  DeviceScene* d = s->dev;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->last_stream));
  unsigned long long c[CNT_N];
  HIP_TRY(hipMemcpy(c, d->counters, sizeof c, hipMemcpyDeviceToHost));
  out->rays = c[CNT_RAYS];
  out->node_visits = c[CNT_NODES];
  out->tri_tests = c[CNT_TRIS];
  out->sphere_tests = c[CNT_SPHERES];
  out->shaded_hits = c[CNT_SHADED];
  out->paths = c[CNT_PATHS];
  return RT_OK;
}

extern "C" int rt_scene_last_timing(rt_scene* s, rt_timing* out) {
  if (!s || !out) {
    rt_set_error("rt_scene_last_timing: null argument");
    return RT_ERR_INVALID;
  }
  if (!s->dev || !s->dev->have_timing) {
    rt_set_error("rt_scene_last_timing: the last call was not made with RT_FLAG_TIMING");
    return RT_ERR_INVALID;
  }
  *out = s->dev->timing;
  return RT_OK;
}
