// rt_device.hip — device half of the library: scene upload, kernels, and the C entry points
// that launch them. Written for gfx950 only.
//
// Kernels
//   k_trace        batch Scene::hit (src/main.rs:776-814), one ray per lane
//   k_primary      the pixel-centre camera ray of every pixel + Scene::hit (parity buffer)
//   k_wave<STAGE>  render's per-sample body (src/main.rs:995-1019) and raytrace (817-975) in
//                  wavefront form: one launch per bounce over a compacted queue of live paths
//   k_paths        the same loop as ONE kernel (per-lane bounce loop, optionally with lanes that
//                  finish a path pulling the next one from a work counter) — kept for A/B
//   k_plan         one block behind every launch that fills a queue: the extent of the next launch,
//                  written where that launch's kernels read it, and a snapshot of the work counters
//   k_gather       compacts the last queued records so that the tail can run on a side stream
//   k_resolve      the per-pixel sums of src/main.rs:1020-1037 in sample order
//
// HBM layout: see rt_types.h. The per-sample radiance buffer is [n_paths][3] binary64, where
// a path index is ((tile * spp + sample) * 64 + pixel-in-tile) over 8x8-pixel tiles, so the 64
// lanes of a wave start on one tile and one sample (coherent primary rays), a path's radiance
// is one 24-byte store, and k_resolve reads 64 consecutive records per sample.
//
// Atomics: one word takes ~88 atomic adds per microsecond on this chip, and a 1080p x 64 spp
// frame has two million waves, so nothing here lets every wave add to the same word. Work
// counters and queue tails are sharded over RT_SHARDS 128-byte lines; a wave picks its line
// from its block index, the host sums the lines.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include "rt_kernels.hip.h"
#include "rt_types.h"

using namespace rtk;

// ------------------------------------------------------------------------------------------
// device scene
// ------------------------------------------------------------------------------------------
#define RT_SHARDS 1024     /* counter lines; also the most queue segments a launch uses */
#define RT_LINE_WORDS 16   /* 64-bit words per line (128 B)                             */
#define RT_WORK_BATCH 1024 /* path indices a wave takes from the work counter at a time */
#ifndef RT_GRID_NODES
#define RT_GRID_NODES 1
#endif
#define RT_LEAVE_DEFAULT 32      /* of 64: share of a wave's traversing lanes below which node steps pause for the leaves (trav_run) */
#define RT_LONG_STEPS_DEFAULT 8u /* node visits from which a ray counts as long (see k_wave) */
#define RT_GRID_FIRST_MIB 32      /* binary32 node array from which the first stage walks the 16-bit copy too */
#define RT_DEEP_FROM 40          /* stack entries (tree depth + 2) from which the kernels' DEEP form runs (trav_run) */
// words of a line
enum { CNT_RAYS = 0, CNT_NODES, CNT_TRIS, CNT_SPHERES, CNT_SHADED, CNT_PATHS, CNT_Q0, CNT_R0, CNT_Q1, CNT_R1, CNT_WORK, CNT_WI_INNER, CNT_WI_LEAF, CNT_WI_REJECT, CNT_WAVES, CNT_N };
#define RT_COUNTER_BYTES ((size_t)RT_SHARDS * RT_LINE_WORDS * sizeof(unsigned long long))

// What k_plan (below) writes: the extent of the next launch over a queue, and a snapshot of the
// cumulative work counters.
struct alignas(16) PlanRec {
  uint32_t hit_blocks, ray_blocks;  // blocks of RT_BLOCK records per segment: front run, back run
  uint32_t n_hit, n_ray;            // records over all segments
};
struct LogRec {
  unsigned long long w[CNT_N];  // sums of the counter lines when the snapshot was taken
  uint32_t n_hit, n_ray;        // queue records counted by this plan
  uint32_t overflow;            // some segment holds more records than it has room for (never, by construction)
  uint32_t hit_blocks, ray_blocks;  // as in the plan
  uint32_t stamp;                   // written last: the low 32 bits of the snapshot's sequence number, plus one
};
#define RT_PLAN_RECS 4 /* PlanRec per plan slot: [0] the launch, [1] only if it fits `split_at`, [2] only if not */
#define RT_PLAN_SLOTS 4096  /* ring of plans and counter snapshots; harvested long before it wraps */
#define RT_LAUNCH_KEEP 4096 /* launch log entries kept one by one; older ones are folded per kind */
#define RT_EVENT_POOL_FIRST 4096 /* timing events made by a scene's first call with RT_FLAG_TIMING */

struct DeviceScene {
  int device = -1;
  SceneK k{};
  std::vector<void*> allocs;
  uint64_t bytes = 0;
  uint32_t stack_entries = 2;
  bool grid_ok = false;  // the launches over the queue walk qnodes (else nodes, like everything else)
  // The first stage walks qnodes too when the tree is far larger than the caches: its coherent rays are
  // bound by instruction issue, where the twelve conversions a visit cost more than the halved
  // fetches save (ogre 4K first stage 183.4 -> 190.4 ms, Spot 4K 137.7 -> 140.0), until the nodes
  // come from HBM (14x14 grid of Spot meshes, 147 MB of binary32 nodes: 32.4 -> 30.8 ms).
  bool grid_first = false;
  // work buffers, grown on demand
  double* samples = nullptr;
  size_t samples_bytes = 0;
  double* totals = nullptr;
  size_t totals_bytes = 0;
  void* queue[2] = {nullptr, nullptr};  // QueueRec segments, wavefront form
  size_t queue_bytes[2] = {0, 0};
  double* queue_rad[2] = {nullptr, nullptr};  // emitted radiance carried per queue entry
  size_t queue_rad_bytes[2] = {0, 0};
  unsigned long long* counters = nullptr;       // RT_SHARDS lines
  unsigned long long* cur_counters = nullptr;   // the lines the last call counted in (counters, or the static scratch's)
  unsigned long long* host_counters = nullptr;  // pinned mirror
  // Launch extents decided on the device (k_plan) and the snapshots they leave behind.
  PlanRec* plans = nullptr;      // device: RT_PLAN_SLOTS x RT_PLAN_RECS
  LogRec* host_log = nullptr;    // pinned, written by the device: RT_PLAN_SLOTS snapshots
  uint64_t snap_seq = 0;         // snapshots taken so far; slot = seq % RT_PLAN_SLOTS
  uint64_t snap_floor = 0;       // snapshots below this one are no longer referenced
  int64_t last_main_snap = -1;   // latest snapshot of the main / side stream's counters in this run
  int64_t last_tail_snap = -1;
  struct Account {               // a launch whose work is the difference of two snapshots
    size_t log_index;
    int64_t before, after;       // snapshot sequence numbers; before: -1 counters were zero, -2 base_main, -3 base_tail
    int64_t items_from;          // >= 0: items = records counted by that snapshot's plan
  };
  std::vector<Account> accounts;
  unsigned long long base_main[CNT_N] = {};  // the latest snapshots' sums, kept across a harvest (their slots may be reused)
  unsigned long long base_tail[CNT_N] = {};
  // How many bounce launches a frame of a given shape needed before its queue was down to the
  // tail threshold: learnt from the plans the first such frame leaves behind, then reused, so
  // that later frames are queued without the host waiting for anything.
  struct Hist {
    uint64_t key = 0;
    uint32_t rounds = 0;
    std::vector<uint32_t> blocks;  // record blocks per segment of bounce launch 0 .. rounds - 1, then of the tail
    uint64_t used = 0;             // hist_clock when last looked up: the least recently used shape makes room
  };
  std::vector<Hist> hists;         // a few shapes (a call in several sample chunks alone has two)
  uint64_t hist_clock = 0;
  // What a frame queued, kept until its plans have landed in host memory: the history of its shape is
  // made (or refined) from them then — by a later call, or by whoever waits for the scene next —
  // without the frame itself ever waiting.
  struct FrameLearn {
    uint64_t key;
    std::vector<int64_t> round_plans;  // snapshot that sized bounce launch r
    int64_t final_plan;                // snapshot after the last bounce launch: what the tail was handed
    uint32_t seg_blocks, threshold;
    bool handoff;
  };
  std::deque<FrameLearn> learn_q;
  // Overlapped tails: when a call runs in several sample chunks, the thin tail of long paths of
  // one chunk (and that chunk's resolve) runs on a side stream while the next chunk starts.
  hipStream_t side = nullptr;
  hipStream_t own = nullptr;   // stream of the entry points that take host buffers (never the device's null stream:
                               // waiting on that one stalls every other stream of the process)
  hipEvent_t ev_gathered[2] = {nullptr, nullptr};  // main stream: chunk's tail input is complete
  hipEvent_t ev_resolved[2] = {nullptr, nullptr};  // side stream: chunk's samples have been summed
  hipEvent_t ev_join[2] = {nullptr, nullptr};      // rt_scene_stream_wait: caller's stream, side stream; re-recorded per call
  void* tail_in[2] = {nullptr, nullptr};           // compact tail input, one segment
  size_t tail_in_bytes[2] = {0, 0};
  double* tail_rad[2] = {nullptr, nullptr};
  size_t tail_rad_bytes[2] = {0, 0};
  unsigned long long* tail_counters = nullptr;     // counter lines of the side stream's kernels
  bool side_used = false;                          // the last call ran kernels on the side stream
  // A run = one call, or consecutive RT_FLAG_PIPELINE calls, which share their accounting.
  bool run_open = false;                           // the last call was pipelined
  bool resolved_valid[2] = {false, false};         // ev_resolved[q] recorded since the side stream was last idle
  uint32_t chunk_seq = 0;                          // chunks started so far: parity of the double buffers
  struct Pending {                                 // a timed launch whose events have not been read yet
    size_t log_index;
    size_t ev0;
  };
  std::vector<Pending> pending;
  size_t n_events = 0;
  // staging for the host-buffer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  void* host_stage = nullptr;  // pinned
  size_t host_stage_bytes = 0;
  std::vector<hipEvent_t> ev_pool;
  std::vector<rt_launch_info> launches;
  rt_timing timing{};
  bool have_timing = false;
  hipStream_t last_stream = nullptr;
  bool last_stream_valid = false;
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RT_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

// ------------------------------------------------------------------------------------------
// Static work memory for rt_render_device_noalloc: callers that may not allocate (code under
// graph capture, serving loops, harnesses that forbid hipMalloc) get their sample buffer and ray
// queues out of this array instead of hipMalloc'ed buffers. Every GPU the library
// runs on holds its own copy of the array, at its own address, until the process ends; the
// array's name is only a host-side handle, so the address is asked for once per device and kept.
// ------------------------------------------------------------------------------------------
#define RT_SCRATCH_BYTES ((size_t)RT_NOALLOC_SCRATCH_MIB << 20)
#define RT_SCRATCH_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];
static unsigned char* g_scratch_ptr[RT_SCRATCH_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;  // scenes may be uploaded from several host threads

// Address of `device`'s copy of the array (the device is made current). Called when a scene is
// uploaded, so that the render call itself finds the address already there.
static int scratch_address(int device, unsigned char** out) {
  if (device < 0 || device >= RT_SCRATCH_MAX_DEVICES) {
    rt_set_error("device %d is beyond the %d devices the static work memory is tracked for", device, RT_SCRATCH_MAX_DEVICES);
    return RT_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    HIP_TRY(hipSetDevice(device));
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) {
      rt_set_error("hipGetSymbolAddress returned no address for the static work memory on device %d", device);
      return RT_ERR_HIP;
    }
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}

// Bump allocation inside the static array (256-byte granules).
struct Arena {
  unsigned char* base;
  size_t bytes;
  size_t used;
};
static void* arena_take(Arena* a, size_t n) {
  const size_t at = (a->used + 255u) & ~(size_t)255u;
  if (at + n > a->bytes) return nullptr;
  a->used = at + n;
  return a->base + at;
}

void rt_device_scene_free(DeviceScene* d) {
  if (!d) return;
  if (d->device >= 0) (void)hipSetDevice(d->device);
  // wait for the scene's own work only (its streams and the caller's last one), not for the whole device
  if (d->last_stream_valid) (void)hipStreamSynchronize(d->last_stream);
  if (d->side) (void)hipStreamSynchronize(d->side);
  if (d->own) (void)hipStreamSynchronize(d->own);
  for (void* p : d->allocs) (void)hipFree(p);
  if (d->samples) (void)hipFree(d->samples);
  if (d->totals) (void)hipFree(d->totals);
  if (d->counters) (void)hipFree(d->counters);
  if (d->tail_counters) (void)hipFree(d->tail_counters);
  if (d->plans) (void)hipFree(d->plans);
  if (d->host_log) (void)hipHostFree(d->host_log);
  if (d->side) (void)hipStreamDestroy(d->side);
  if (d->own) (void)hipStreamDestroy(d->own);
  for (int q = 0; q < 2; ++q)
    if (d->ev_join[q]) (void)hipEventDestroy(d->ev_join[q]);
  for (int q = 0; q < 2; ++q) {
    if (d->ev_gathered[q]) (void)hipEventDestroy(d->ev_gathered[q]);
    if (d->ev_resolved[q]) (void)hipEventDestroy(d->ev_resolved[q]);
    if (d->tail_in[q]) (void)hipFree(d->tail_in[q]);
    if (d->tail_rad[q]) (void)hipFree(d->tail_rad[q]);
  }
  if (d->stage) (void)hipFree(d->stage);
  if (d->host_stage) (void)hipHostFree(d->host_stage);
  for (int q = 0; q < 2; ++q) {
    if (d->queue[q]) (void)hipFree(d->queue[q]);
    if (d->queue_rad[q]) (void)hipFree(d->queue_rad[q]);
  }
  if (d->host_counters) (void)hipHostFree(d->host_counters);
  for (hipEvent_t e : d->ev_pool) (void)hipEventDestroy(e);
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
  return rt_guard("rt_scene_upload", [&] { return rt_device_scene_upload_from(s, s, device); });
}
// Uploads the arrays of `s` to `device` and hangs the device half on `target` (normally s itself;
// a group hangs one device half per member on otherwise empty scenes).
int rt_device_scene_upload_from(const rt_scene* s, rt_scene* target, int device) {
  if (!s || !target) {
    rt_set_error("rt_scene_upload: null scene");
    return RT_ERR_INVALID;
  }
  int n = rt_device_count();
  if (n <= 0) {
    rt_set_error("rt_scene_upload: no HIP device is visible; this library has no CPU path");
    return RT_ERR_NO_DEVICE;
  }
  if (!s->has_tree) {
    rt_set_error("rt_scene_upload: the scene was created without a tree; build one with rt_scene_rebuild_on_device first");
    return RT_ERR_INVALID;
  }
  if (device < 0 || device >= n) {
    rt_set_error("rt_scene_upload: device %d out of range (0..%d)", device, n - 1);
    return RT_ERR_INVALID;
  }
  if (target->dev && target->dev->device == device) return RT_OK;
  // trav_run addresses nodes and triangles with 32-bit byte offsets from the arrays' bases
  if ((uint64_t)s->tri_geom.size() * sizeof(TriGeom) >= (1ull << 32) || (uint64_t)s->nodes.size() * sizeof(BvhNode) >= (1ull << 32)) {
    rt_set_error("rt_scene_upload: %llu triangles / %llu nodes exceed what the kernels' 32-bit offsets address (%llu triangles, %llu nodes)",
                 (unsigned long long)s->tri_geom.size(), (unsigned long long)s->nodes.size(), (unsigned long long)((1ull << 32) / sizeof(TriGeom)),
                 (unsigned long long)((1ull << 32) / sizeof(BvhNode)));
    return RT_ERR_INVALID;
  }
  if (target->dev) {
    rt_device_scene_free(target->dev);
    target->dev = nullptr;
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
  {
    // The tree once more on a 16-bit grid over its own bounds (BvhNodeQ), if the scene qualifies.
    std::vector<BvhNodeQ> qn;
    d->grid_ok = RT_GRID_NODES && rt_quantize_tree(s, &qn, d->k.q_lo, d->k.q_step);
    const char* no_grid = std::getenv("RT_NO_GRID");
    if (no_grid && *no_grid && *no_grid != '0') d->grid_ok = false;
    if (d->grid_ok && (rc = upload_array(d, qn.data(), qn.size(), &d->k.qnodes)) != RT_OK) return fail(rc);
    d->grid_first = d->grid_ok && s->nodes.size() * sizeof(BvhNode) > ((size_t)RT_GRID_FIRST_MIB << 20);
    const char* qf = std::getenv("RT_QUANT_FIRST");  // 0 / 1 forces the choice (A/B)
    if (qf && *qf) d->grid_first = d->grid_ok && *qf != '0';
  }
  if ((rc = upload_array(d, s->tri_geom.data(), s->tri_geom.size(), &d->k.geom)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->tri_shade.data(), s->tri_shade.size(), &d->k.shade)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->spheres.data(), s->spheres.size(), &d->k.spheres)) != RT_OK) return fail(rc);
  if ((rc = upload_array(d, s->materials.data(), s->materials.size(), &d->k.mats)) != RT_OK) return fail(rc);
  std::vector<TextureRec> trec(s->textures.size());
  for (size_t i = 0; i < s->textures.size(); ++i) {
    const uint8_t* px = nullptr;
    std::vector<uint8_t> tiled;
    uint32_t blocks_x = 0;
    rt_tile_texture(s->textures[i], &tiled, &blocks_x);
    if ((rc = upload_array(d, tiled.data(), tiled.size(), &px)) != RT_OK) return fail(rc);
    trec[i] = TextureRec{px, s->textures[i].w, s->textures[i].h, blocks_x};
  }
  if ((rc = upload_array(d, trec.data(), trec.size(), &d->k.tex)) != RT_OK) return fail(rc);
  {
    // spheres among the first 64 that have no place in the tree: tested against every ray
    unsigned long long in_tree = 0ull;
    for (int32_t i : s->tree_spheres) in_tree |= 1ull << (uint32_t)i;
    const size_t n_low = std::min<size_t>(s->spheres.size(), (size_t)RT_TREE_SPHERES);
    const unsigned long long low = n_low >= 64 ? ~0ull : ((1ull << n_low) - 1ull);
    d->k.sph_always = low & ~in_tree;
    d->k.sph_places = in_tree != 0ull ? 1u : 0u;
  }
  d->k.root_ref = s->root_ref;
  d->k.n_spheres = (uint32_t)s->spheres.size();
  d->k.has_emission = 0;
  {
    // normalize((1, 2, 1)) as src/main.rs:87-90, 307 computes it: every component over the length
    const double l = std::sqrt(1.0 * 1.0 + 2.0 * 2.0 + 1.0 * 1.0);
    d->k.sky_light[0] = 1.0 / l;
    d->k.sky_light[1] = 2.0 / l;
    d->k.sky_light[2] = 1.0 / l;
  }
  {
    const char* e = std::getenv("RT_LEAVE");
    d->k.leave_q = e && *e ? std::max(0, std::min(64, std::atoi(e))) : RT_LEAVE_DEFAULT;
  }
  for (const rt_material& m : s->materials)
    if (m.emission[0] != 0.0 || m.emission[1] != 0.0 || m.emission[2] != 0.0) d->k.has_emission = 1;
  d->stack_entries = s->bvh_depth + 2;
  {
    // RT_DEEP_STACK=0/1 forces the choice (A/B)
    const char* e = std::getenv("RT_DEEP_STACK");
    d->k.deep_stack = e && *e ? (*e != '0' && d->stack_entries > (uint32_t)RT_DEEP_LDS ? 1u : 0u) : (d->stack_entries >= (uint32_t)RT_DEEP_FROM ? 1u : 0u);
    // the same form of the kernels is the one that handles spheres with places in the tree
    if (d->k.sph_places) d->k.deep_stack = 1u;
  }
  if (hipMalloc((void**)&d->counters, RT_COUNTER_BYTES) != hipSuccess) {
    rt_set_error("rt_scene_upload: hipMalloc of the counters failed");
    return fail(RT_ERR_HIP);
  }
  (void)hipMemset(d->counters, 0, RT_COUNTER_BYTES);
  if (hipMalloc((void**)&d->tail_counters, RT_COUNTER_BYTES) != hipSuccess) {
    rt_set_error("rt_scene_upload: hipMalloc of the side-stream counters failed");
    return fail(RT_ERR_HIP);
  }
  (void)hipMemset(d->tail_counters, 0, RT_COUNTER_BYTES);
  {
    unsigned char* sp = nullptr;
    if ((rc = scratch_address(device, &sp)) != RT_OK) return fail(rc);
  }
  // everything a render call needs besides its work buffers exists from here on, so that the call
  // itself creates nothing
  if (hipMalloc((void**)&d->plans, (size_t)RT_PLAN_SLOTS * RT_PLAN_RECS * sizeof(PlanRec)) != hipSuccess ||
      hipHostMalloc((void**)&d->host_log, (size_t)RT_PLAN_SLOTS * sizeof(LogRec), hipHostMallocDefault) != hipSuccess) {
    rt_set_error("rt_scene_upload: allocation of the launch plans failed");
    return fail(RT_ERR_HIP);
  }
  std::memset(d->host_log, 0, (size_t)RT_PLAN_SLOTS * sizeof(LogRec));
  if (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&d->own, hipStreamNonBlocking) != hipSuccess) {
    rt_set_error("rt_scene_upload: hipStreamCreate failed");
    return fail(RT_ERR_HIP);
  }
  for (int q = 0; q < 2; ++q) {
    if (hipEventCreateWithFlags(&d->ev_gathered[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_resolved[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->ev_join[q], hipEventDisableTiming) != hipSuccess) {
      rt_set_error("rt_scene_upload: hipEventCreate failed");
      return fail(RT_ERR_HIP);
    }
  }
  if (hipHostMalloc((void**)&d->host_counters, RT_COUNTER_BYTES, hipHostMallocDefault) != hipSuccess) {
    rt_set_error("rt_scene_upload: hipHostMalloc of the counter mirror failed");
    return fail(RT_ERR_HIP);
  }
  target->dev = d;
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

// Copies the counter lines to the host (waits for `stream`) and sums them per word; the queue
// tails (CNT_Q0/CNT_Q1) are also returned per line.
struct CounterSums {
  unsigned long long w[CNT_N];
};
static void sum_lines(const unsigned long long* lines, CounterSums* sums) {
  for (int s = 0; s < RT_SHARDS; ++s)
    for (int w = 0; w < CNT_N; ++w) sums->w[w] += lines[(size_t)s * RT_LINE_WORDS + w];
}
static int read_counters(DeviceScene* d, hipStream_t stream, CounterSums* sums) {
  HIP_TRY(hipMemcpyAsync(d->host_counters, d->cur_counters ? d->cur_counters : d->counters, RT_COUNTER_BYTES, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  std::memset(sums, 0, sizeof *sums);
  sum_lines(d->host_counters, sums);
  return RT_OK;
}
// Work counters of a whole call: the main stream's lines plus the side stream's. Waits for both.
static int read_call_counters(DeviceScene* d, hipStream_t stream, CounterSums* sums) {
  if (d->side) HIP_TRY(hipStreamSynchronize(d->side));
  int rc = read_counters(d, stream, sums);
  if (rc != RT_OK || !d->side_used) return rc;
  HIP_TRY(hipMemcpyAsync(d->host_counters, d->tail_counters, RT_COUNTER_BYTES, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  CounterSums t;
  std::memset(&t, 0, sizeof t);
  sum_lines(d->host_counters, &t);
  for (int w : {CNT_RAYS, CNT_NODES, CNT_TRIS, CNT_SPHERES, CNT_SHADED, CNT_PATHS, CNT_WI_INNER, CNT_WI_LEAF, CNT_WI_REJECT, CNT_WAVES}) sums->w[w] += t.w[w];
  return RT_OK;
}

// Reads the events of the timed launches recorded so far into the launch log (the caller has
// waited for the streams they were recorded on).
static int harvest_timing(DeviceScene* d) {
  for (const DeviceScene::Pending& pe : d->pending) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev_pool[pe.ev0], d->ev_pool[pe.ev0 + 1]));
    rt_launch_info& li = d->launches[pe.log_index];
    li.ms = ms;
    if (li.kind == 4) {
      d->timing.ms_resolve += ms;
      d->timing.resolve_launches++;
    } else {
      d->timing.ms_paths += ms;
      d->timing.path_launches++;
    }
    d->have_timing = true;
  }
  d->pending.clear();
  return RT_OK;
}

// Bounce launches a frame of n_paths paths is given when nothing is known about its shape: the queue
// after the first stage holds about a quarter of the paths and loses about a quarter per bounce
// (measured on the reference's scenes: 0.5 to 0.8 survive a bounce, the later the more).
static uint32_t guess_rounds(uint64_t n_records, uint32_t threshold) {
  uint32_t r = 0;
  double n = (double)n_records;
  while (n > (double)threshold && r < 48u) {
    n *= 0.75;
    ++r;
  }
  return r;
}

static const LogRec* landed_plan(const DeviceScene* d, int64_t seq) {
  if (seq < 0 || d->snap_seq - (uint64_t)seq > RT_PLAN_SLOTS) return nullptr;  // its slot of the ring has been reused
  const LogRec* lr = d->host_log + ((uint64_t)seq % RT_PLAN_SLOTS);
  if (*reinterpret_cast<const volatile uint32_t*>(&lr->stamp) != (uint32_t)((uint64_t)seq + 1u)) return nullptr;
  return lr;
}

// Makes (or refines) the history of the shapes whose frames' plans have all arrived in host memory,
// oldest frame first; stops at the first frame still in flight. Never waits.
static void learn_from_landed_frames(DeviceScene* d) {
  while (!d->learn_q.empty()) {
    const DeviceScene::FrameLearn& fl = d->learn_q.front();
    const bool lost = fl.final_plan < 0 || d->snap_seq - (uint64_t)fl.final_plan > RT_PLAN_SLOTS;
    const LogRec* last = lost ? nullptr : landed_plan(d, fl.final_plan);
    if (!lost && !last) return;  // still in flight (the plans of a stream land in order)
    if (last) {
      DeviceScene::Hist h;
      h.key = fl.key;
      bool complete = true;
      uint32_t tail_blocks = last->hit_blocks + last->ray_blocks;
      bool below = false;
      for (size_t r = 0; r < fl.round_plans.size() && complete; ++r) {
        const LogRec* lr = landed_plan(d, fl.round_plans[r]);
        if (!lr) {
          complete = false;
          break;
        }
        if ((uint64_t)lr->n_hit + lr->n_ray <= fl.threshold) {  // the tail takes over from here
          tail_blocks = lr->hit_blocks + lr->ray_blocks;
          below = true;
          break;
        }
        h.blocks.push_back(lr->hit_blocks + lr->ray_blocks);
      }
      if (complete) {
        if (!below) {
          // the launches queued did not bring the queue down to the threshold: more of them next time,
          // sized like the last one (queues only shrink)
          const uint32_t more = guess_rounds((uint64_t)last->n_hit + last->n_ray, fl.threshold);
          for (uint32_t e = 0; e < more; ++e) h.blocks.push_back(std::min(fl.seg_blocks, tail_blocks));
        }
        h.rounds = (uint32_t)h.blocks.size();
        h.blocks.push_back(fl.handoff ? 0u : tail_blocks);
        h.used = ++d->hist_clock;
        DeviceScene::Hist* slot = nullptr;
        for (DeviceScene::Hist& x : d->hists)
          if (x.key == h.key) slot = &x;
        if (!slot && d->hists.size() < 8) {
          d->hists.emplace_back();
          slot = &d->hists.back();
        }
        if (!slot) {
          slot = &d->hists[0];
          for (DeviceScene::Hist& x : d->hists)
            if (x.used < slot->used) slot = &x;
        }
        *slot = std::move(h);
      }
    }
    d->learn_q.pop_front();
  }
}

// Fills the launch log entries whose work is the difference of two counter snapshots (the caller
// has waited for the streams the snapshots were taken on), then folds the oldest entries of a long
// run into one entry per kind so that the log stays bounded.
static int harvest_accounts(DeviceScene* d) {
  bool overflow = false;
  for (const DeviceScene::Account& a : d->accounts) {
    const LogRec& after = d->host_log[(uint64_t)a.after % RT_PLAN_SLOTS];
    unsigned long long zero[CNT_N] = {};
    const unsigned long long* before = a.before >= 0 ? d->host_log[(uint64_t)a.before % RT_PLAN_SLOTS].w
                                       : a.before == -2 ? d->base_main : a.before == -3 ? d->base_tail : zero;
    rt_launch_info& li = d->launches[a.log_index];
    li.rays = after.w[CNT_RAYS] - before[CNT_RAYS];
    li.node_visits = after.w[CNT_NODES] - before[CNT_NODES];
    li.tri_tests = after.w[CNT_TRIS] - before[CNT_TRIS];
    li.waves = after.w[CNT_WAVES] - before[CNT_WAVES];
    li.wave_iters_inner = after.w[CNT_WI_INNER] - before[CNT_WI_INNER];
    li.wave_iters_leaf = after.w[CNT_WI_LEAF] - before[CNT_WI_LEAF];
    li.wave_iters_reject = after.w[CNT_WI_REJECT] - before[CNT_WI_REJECT];
    if (a.items_from >= 0) {
      const LogRec& sized = d->host_log[(uint64_t)a.items_from % RT_PLAN_SLOTS];
      li.items = sized.n_hit + sized.n_ray;
    }
    overflow = overflow || after.overflow != 0;
  }
  d->accounts.clear();
  learn_from_landed_frames(d);  // everything queued has finished: every frame's plans are there
  d->learn_q.clear();
  if (d->last_main_snap >= 0) {
    std::memcpy(d->base_main, d->host_log[(uint64_t)d->last_main_snap % RT_PLAN_SLOTS].w, sizeof d->base_main);
    d->last_main_snap = -2;
  }
  if (d->last_tail_snap >= 0) {
    std::memcpy(d->base_tail, d->host_log[(uint64_t)d->last_tail_snap % RT_PLAN_SLOTS].w, sizeof d->base_tail);
    d->last_tail_snap = -3;
  }
  d->snap_floor = d->snap_seq;
  if (d->launches.size() > RT_LAUNCH_KEEP && d->pending.empty()) {
    // entries [0, n_fold) become at most five entries (one per kind) in front of the rest
    const size_t n_fold = d->launches.size() - RT_LAUNCH_KEEP / 2;
    rt_launch_info folded[5];
    std::memset(folded, 0, sizeof folded);
    for (size_t i = 0; i < n_fold; ++i) {
      const rt_launch_info& li = d->launches[i];
      rt_launch_info& f = folded[li.kind < 5 ? li.kind : 4];
      f.kind = li.kind;
      f.ms += li.ms;
      f.folded += li.folded ? li.folded : 1u;
      f.items = (uint32_t)std::min<uint64_t>(0xffffffffull, (uint64_t)f.items + li.items);
      f.rays += li.rays;
      f.node_visits += li.node_visits;
      f.tri_tests += li.tri_tests;
      f.waves += li.waves;
      f.wave_iters_inner += li.wave_iters_inner;
      f.wave_iters_leaf += li.wave_iters_leaf;
      f.wave_iters_reject += li.wave_iters_reject;
    }
    std::vector<rt_launch_info> kept;
    for (const rt_launch_info& f : folded)
      if (f.folded) kept.push_back(f);
    kept.insert(kept.end(), d->launches.begin() + (long)n_fold, d->launches.end());
    d->launches.swap(kept);
  }
  if (overflow) {
    rt_set_error("internal error: a queue segment was handed more records than it has room for");
    return RT_ERR_HIP;
  }
  return RT_OK;
}

// Waits for what the scene has queued on the caller's stream and on the side stream and fills in
// the launch log (durations, work); does not end a run of pipelined calls.
static int settle(DeviceScene* d) {
  HIP_TRY(hipSetDevice(d->device));
  if (d->last_stream_valid) HIP_TRY(hipStreamSynchronize(d->last_stream));
  if (d->side) HIP_TRY(hipStreamSynchronize(d->side));
  d->resolved_valid[0] = d->resolved_valid[1] = false;
  int rc = harvest_timing(d);
  if (rc != RT_OK) return rc;
  d->n_events = 0;  // every event of the pool has been read: reuse them
  return harvest_accounts(d);
}

// settle, and the run (one call, or consecutive RT_FLAG_PIPELINE calls) is over.
static int finish_run(DeviceScene* d) {
  int rc = settle(d);
  d->run_open = false;
  return rc;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
extern __shared__ int32_t g_stack[];  // [stack_entries][blockDim.x]
#ifdef RT_STAMPS
// Diagnostic build only: cycles of wave lifetime by code section and kernel stage, summed over waves.
__device__ unsigned long long g_stamps[3][RT_NSEC];
extern "C" int rt_debug_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 3 * RT_NSEC) != hipSuccess) return RT_ERR_HIP;
  if (reset) {
    unsigned long long zero[3 * RT_NSEC] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero) != hipSuccess) return RT_ERR_HIP;
  }
  return RT_OK;
}
#endif

#ifdef RT_DEBUG_HIST
// Diagnostic build only (-DRT_DEBUG_HIST): how well what is known about a queued hit predicts the cost
// of the ray scattered from it. [object of the hit, clipped to 7][node visits of the ray that found
// it, clipped to 63][node visits of the scattered ray, clipped to 63], over the launches over the queue.
__device__ unsigned long long g_hist[8][64][64];
extern "C" int rt_debug_hist(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hist), sizeof(unsigned long long) * 8 * 64 * 64) != hipSuccess) return RT_ERR_HIP;
  if (reset) {
    static unsigned long long zero[8 * 64 * 64];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_hist), zero, sizeof zero) != hipSuccess) return RT_ERR_HIP;
  }
  return RT_OK;
}
#endif

__device__ __forceinline__ uint32_t linear_block() { return blockIdx.y * gridDim.x + blockIdx.x; }

// The counter line a wave adds to.
__device__ __forceinline__ unsigned long long* my_line(unsigned long long* counters) {
  const uint32_t wave = linear_block() * (RT_BLOCK / 64) + (threadIdx.x >> 6);
  return counters + (size_t)(wave % RT_SHARDS) * RT_LINE_WORDS;
}

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

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// WAVE_TOTALS: wc.rays and n_done are the wave's totals already (the same value in every lane:
// k_wave counts them with ballots, which keeps two registers free), else per-lane counts.
template <bool WAVE_TOTALS = false>
__device__ __forceinline__ void flush_counts(const WorkCount& wc, unsigned long long* counters, bool count_all, uint32_t n_done,
                                             uint32_t n_waves = 1) {
  // one atomic per wave and counter, on the wave's own line
  unsigned long long* line = my_line(counters);
  const unsigned long long r = WAVE_TOTALS ? (unsigned long long)wc.rays : wave_sum(wc.rays);
  if ((threadIdx.x & 63) == 0 && r) atomicAdd(&line[CNT_RAYS], r);
  if (count_all) {
    const unsigned long long a = wave_sum(wc.nodes), b = wave_sum(wc.tris), c = wave_sum(wc.spheres), e = wave_sum(wc.shaded),
                             f = WAVE_TOTALS ? (unsigned long long)n_done : wave_sum(n_done);
    if ((threadIdx.x & 63) == 0) {
      if (a) atomicAdd(&line[CNT_NODES], a);
      if (b) atomicAdd(&line[CNT_TRIS], b);
      if (c) atomicAdd(&line[CNT_SPHERES], c);
      if (e) atomicAdd(&line[CNT_SHADED], e);
      if (f) atomicAdd(&line[CNT_PATHS], f);
    }
    const unsigned long long g = wave_sum(wc.wi_inner), h = wave_sum(wc.wi_leaf), i = wave_sum(wc.wi_reject);
    if ((threadIdx.x & 63) == 0) {
      if (g) atomicAdd(&line[CNT_WI_INNER], g);
      if (h) atomicAdd(&line[CNT_WI_LEAF], h);
      if (i) atomicAdd(&line[CNT_WI_REJECT], i);
      if (n_waves) atomicAdd(&line[CNT_WAVES], (unsigned long long)n_waves);
    }
  }
}

template <bool COUNT, bool DEEP>
__global__ __launch_bounds__(RT_BLOCK) void k_trace(SceneK sc, const double* __restrict__ org, const double* __restrict__ dir,
                                                    size_t n, rt_hit* __restrict__ out, unsigned long long* counters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  WorkCount wc{};
  if (i < n) {
    const d3 o = mk(org[i * 3], org[i * 3 + 1], org[i * 3 + 2]);
    const d3 d = mk(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
    HitK h;
    trace_ray<COUNT, DEEP>(sc, o, d, g_stack + threadIdx.x, blockDim.x, h, wc);
    wc.rays = 1;
    store_hit(sc, h, out + i);
  }
  flush_counts(wc, counters, COUNT, 0);
}

template <bool DEEP>
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
    trace_ray<false, DEEP>(sc, o, d, g_stack + threadIdx.x, blockDim.x, h, wc);
    wc.rays = 1;
    rt_hit r;
    store_hit(sc, h, &r);
    obj[i] = r.obj;
    tri[i] = r.tri;
    t[i] = r.t;
    u[i] = r.u;
    v[i] = r.v;
  }
  flush_counts(wc, counters, false, 0);
}

// Division of any 32-bit unsigned number by a divisor known on the host, as a multiplication
// (Granlund and Montgomery, "Division by invariant integers using multiplication", 1994): on
// wave-uniform operands it stays on the scalar unit, which has no divide.
struct FastDiv {
  uint32_t m, sh1, sh2;
};
static FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{1u, 0u, 0u};
  if (d <= 1u) return f;  // n / 1 = n (a zero divisor never reaches a kernel)
  uint32_t l = 0;
  while (l < 32u && (1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
  f.sh1 = 1u;
  f.sh2 = l - 1u;
  return f;
}
__host__ __device__ inline uint32_t fastdiv(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t q = __umulhi(n, f.m);
#else
  const uint32_t q = (uint32_t)(((unsigned long long)n * f.m) >> 32);
#endif
  return (q + ((n - q) >> f.sh1)) >> f.sh2;
}

// Which pixels a call renders and how path indices map onto them.
struct TileMap {
  uint32_t width, height;
  uint32_t tiles_x;      // 8x8 tiles per image row (whole-image mode)
  uint32_t big;          // 8x8 tiles per side of a big tile, 0 = whole-image mode
  uint32_t bigs_x;       // big tiles per image row
  uint32_t rank, ranks;  // this call renders big tiles rank, rank + ranks, ...
  uint32_t n_tiles;      // 8x8 tiles this call renders
  FastDiv d_tiles_x, d_big, d_per_big, d_bigs_x;
};

__host__ __device__ inline bool tile_pixel(const TileMap& tm, uint32_t tile, uint32_t lane_pix, uint32_t* x, uint32_t* y) {
  uint32_t tx, ty;
  if (tm.big == 0) {
    ty = fastdiv(tile, tm.d_tiles_x);
    tx = tile - ty * tm.tiles_x;
  } else {
    const uint32_t per_big = tm.big * tm.big;
    const uint32_t bi = fastdiv(tile, tm.d_per_big);
    const uint32_t b = tm.rank + bi * tm.ranks;
    const uint32_t j = tile - bi * per_big;
    const uint32_t by = fastdiv(b, tm.d_bigs_x), jy = fastdiv(j, tm.d_big);
    tx = (b - by * tm.bigs_x) * tm.big + (j - jy * tm.big);
    ty = by * tm.big + jy;
  }
  *x = tx * 8u + (lane_pix & 7u);
  *y = ty * 8u + (lane_pix >> 3);
  return *x < tm.width && *y < tm.height;
}

struct PathParams {
  CameraK cam;
  TileMap tm;
  uint64_t seed_key;      // rtk::seed_key(seed)
  uint32_t n_paths;       // n_tiles * spp * 64, < 2^32 per launch
  uint32_t spp;           // samples in this launch
  FastDiv d_spp;
  uint32_t first_sample;  // stream index of this launch's sample 0
  int32_t max_depth;
  uint32_t fuse_bounces;  // bounces the first wavefront stage follows before it queues its hits
  uint32_t long_steps;    // a hit found after at least this many node visits is queued in the back run
};

// Path index -> pixel and sample; starts the path (camera ray, src/main.rs:995-1013).
// Returns false for the padding lanes of tiles that overhang the image.
// `g` = path >> 6 = tile * spp + sample: the same for the 64 paths of a wave, so callers whose
// blocks are one wave pass their block index and the arithmetic stays on the scalar unit.
__device__ __forceinline__ bool start_path(const PathParams& pp, uint32_t g, uint32_t path, uint64_t& key, d3& o, d3& dir) {
  const uint32_t tile = fastdiv(g, pp.d_spp);
  const uint32_t s = g - tile * pp.spp;
  uint32_t x, y;
  if (!tile_pixel(pp.tm, tile, path & 63u, &x, &y)) return false;
  key = path_key(pp.seed_key, (uint64_t)y * pp.tm.width + x, (uint64_t)pp.first_sample + s);
  camera_ray(pp.cam, x, y, key, o, dir);
  return true;
}

__device__ __forceinline__ void store_sample(double* __restrict__ samples, uint32_t n_paths, uint32_t path, const d3 rad) {
  (void)n_paths;
  double* p = samples + (size_t)path * 3;
  p[0] = rad.x;
  p[1] = rad.y;
  p[2] = rad.z;
}

__device__ __forceinline__ ShadeIn shade_input(const HitK& h, const d3 o, const d3 dir) {
  ShadeIn si;
  si.hit_pos = o + h.t * dir;  // src/main.rs:831
  si.u = h.u;
  si.v = h.v;
  si.slot = h.slot;
  si.sphere = h.sphere;
  return si;
}

// ------------------------------------------------------------------------------------------
// Single-kernel form. REGEN: a lane whose path has ended takes the next path index from a work
// counter (in wave-local batches of RT_WORK_BATCH, so the counter sees one atomic per 1024
// paths); otherwise one thread per path.
// ------------------------------------------------------------------------------------------
template <bool COUNT, bool REGEN, bool DEEP>
__global__ __launch_bounds__(RT_BLOCK) void k_paths(SceneK sc, PathParams pp, double* __restrict__ samples,
                                                    unsigned long long* counters) {
  int32_t* stk = g_stack + threadIdx.x;
  const int stride = blockDim.x;
  const int lane = threadIdx.x & 63;
  WorkCount wc{};
  uint32_t n_done = 0;

  bool alive = false;
  bool exhausted = false;              // wave-uniform: the work counter has run out
  uint32_t local_next = 0, local_end = 0;  // wave-uniform: the wave's batch of path indices
  uint32_t path = 0;
  uint64_t key = 0;
  int depth = 0;
  d3 o = mk(0, 0, 0), dir = mk(0, 0, 0), thr = mk(1, 1, 1), rad = mk(0, 0, 0);

  if (!REGEN) {
    const uint64_t p64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    exhausted = true;
    if (p64 < pp.n_paths) {
      path = (uint32_t)p64;
      alive = start_path(pp, path >> 6, path, key, o, dir);
      if (!alive) store_sample(samples, pp.n_paths, path, mk(0.0, 0.0, 0.0));
    }
  }

  for (;;) {
    if (REGEN && !exhausted) {
      unsigned long long need = __ballot(!alive);
      while (need) {
        if (local_next == local_end) {
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&counters[CNT_WORK], (unsigned long long)RT_WORK_BATCH);
          base = __shfl(base, 0);
          if (base >= pp.n_paths) {
            exhausted = true;
            break;
          }
          local_next = (uint32_t)base;
          local_end = (uint32_t)min(base + (unsigned long long)RT_WORK_BATCH, (unsigned long long)pp.n_paths);
        }
        const uint32_t avail = local_end - local_next;
        const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
        const bool mine = !alive && ((need >> lane) & 1ull) && rank < avail;
        if (mine) {
          path = local_next + rank;
          thr = mk(1.0, 1.0, 1.0);
          rad = mk(0.0, 0.0, 0.0);
          depth = 0;
          alive = start_path(pp, path >> 6, path, key, o, dir);
          if (!alive) store_sample(samples, pp.n_paths, path, mk(0.0, 0.0, 0.0));
        }
        local_next += min(avail, (uint32_t)__popcll(need));
        need &= ~__ballot(mine);
        // lanes that drew a padding pixel need another index
        need |= __ballot(mine && !alive);
      }
    }
    if (!__any(alive)) {
      if (exhausted) break;
      continue;
    }
    if (alive) {
      HitK h;
      wc.rays++;
      trace_ray<COUNT, DEEP>(sc, o, dir, stk, stride, h, wc);
      bool finished;
      if (h.slot < 0 && h.sphere < 0) {
        const d3 s = sky(sc, dir);  // src/main.rs:826
        rad = rad + mk(thr.x * s.x, thr.y * s.y, thr.z * s.z);
        finished = true;
      } else {
        if (COUNT) wc.shaded++;
        const ShadeIn si = shade_input(h, o, dir);
        shade_hit<COUNT>(sc, si, key, depth, o, dir, thr, rad, wc);
        depth += 1;
        finished = depth > pp.max_depth;  // src/main.rs:818: deeper calls return black
      }
      if (finished) {
        store_sample(samples, pp.n_paths, path, rad);
        alive = false;
        n_done++;
      }
    }
  }
  flush_counts(wc, counters, COUNT, n_done);
}

// ------------------------------------------------------------------------------------------
// Wavefront form of the same loop: launches over compacted queues of the paths still alive.
//
// Stage 0 starts every path (camera ray) and traces it — primary rays of one 8x8 tile walk the same
// nodes — then shades and traces the first `fuse_bounces` bounces itself: a wave's primary rays hit
// one surface or all miss, so compacting them first would buy nothing and cost a queue record per
// hit. Stage 1 runs over a queue of hit records: shade, trace the scattered ray, append the next
// hit to the other queue. Stage 2 is stage 1 that keeps every path until it ends, for the thin
// tail of long paths. Every path writes its sample exactly once, into its own slot, so results do
// not depend on queue order.
//
// What costs a wave most is one lane walking deep into the tree while the other 63 wait: most
// scattered rays leave the scene within a few node visits, a minority (rays that start inside or
// next to a dense mesh) needs dozens, and a path whose last ray was long will mostly cast a long
// one again. So a hit is filed by the cost of the ray that found it: hits found within
// `long_steps` node visits at the front of a segment, the others at its back, and the next launch
// runs the two runs in separate waves.
//
// A queue is n_seg segments of seg_cap records; segment s keeps its two counts in counter line
// s. Stage 0 block b appends to segment b % n_seg; later stages run on a (n_seg, blocks) grid
// whose blockIdx.x is the segment they read AND append to, so a segment never outgrows its
// first filling (an input record yields at most one output record) and concurrently running
// blocks spread over all counter lines.
// ------------------------------------------------------------------------------------------
struct alignas(16) QueueRec {  // 112 B
  double hit_pos[3];  // the hit point
  double dir[3];      // the direction the path arrived along
  double thr[3];
  double u, v;
  uint64_t key;
  uint32_t path;
  int32_t id;  // >= 0 triangle slot, < 0 ~sphere index
  int32_t depth;
  int32_t pad;
};
static_assert(sizeof(QueueRec) == 112, "QueueRec must be 112 bytes");

// In memory a queue is seven planes of 16-byte pieces (piece p of entry e at planes[p * entries +
// e]): the 64 lanes of a wave, which read or write consecutive entries, then move one contiguous
// kilobyte per instruction instead of touching 56 cache lines with 112-byte strides.
struct QueueView {
  float4* recs;       // 7 planes of `entries` pieces
  uint32_t entries;   // entries per plane (segments * seg_cap)
  double* rad;        // [entries][3], only when the scene has emitters
  uint32_t seg_cap;   // records per segment
  uint32_t hit_word;  // word of a counter line: records at the front of the segment (cheap rays)
  uint32_t ray_word;  // word of a counter line: records at the back of the segment (long rays)
};

union QueueRecPieces {
  QueueRec rec;
  float4 piece[7];
};
__device__ __forceinline__ QueueRec queue_load(const QueueView& q, size_t e) {
  QueueRecPieces u;
#pragma unroll
  for (int p = 0; p < 7; ++p) u.piece[p] = q.recs[(size_t)p * q.entries + e];
  return u.rec;
}
__device__ __forceinline__ void queue_store(const QueueView& q, size_t e, const QueueRec& r) {
  QueueRecPieces u;
  u.rec = r;
#pragma unroll
  for (int p = 0; p < 7; ++p) q.recs[(size_t)p * q.entries + e] = u.piece[p];
}

// One atomic per wave: reserves room for the lanes of `mask` in a segment count; returns the
// calling lane's index within the segment's front (or back) run.
__device__ __forceinline__ uint32_t wave_reserve(unsigned long long* word, unsigned long long mask, int lane) {
  const int leader = __ffsll((long long)mask) - 1;
  unsigned int base = 0;
  if (lane == leader) base = atomicAdd(reinterpret_cast<unsigned int*>(word), (unsigned int)__popcll(mask));
  base = __shfl(base, leader);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// How large the launches over a queue are is decided on the device: after every launch that
// appends to a queue, one block (k_plan) adds up the segment counts and writes the extent of the
// next launch — the blocks of 64 records the fullest segment holds in its front and back run —
// for the kernels of that launch to read. Those kernels run on a fixed grid of (segments, some
// blocks per segment) whose blocks loop over the record blocks of their segment, so the host
// never has to know a count and a whole frame can be queued without waiting for any of it.
// The same block leaves a snapshot of the cumulative work counters in host-visible memory, from
// which the per-launch log is made once the frame is done.

// One block over the RT_SHARDS counter lines. plan[0] = extent of the queue whose
// counts are words in_hit_word / in_ray_word of the first n_seg lines; with split_at > 0 also
// plan[1] = plan[0] if the queue holds at most split_at records, else empty, and plan[2] the other
// way round. zero_word >= 0: words zero_word, zero_word + 1 of every line are set to zero (the
// counts of the queue the next launch appends to). log (may be null) receives the snapshot.
#define RT_PLAN_THREADS 256
__global__ __launch_bounds__(RT_PLAN_THREADS) void k_plan(unsigned long long* counters, uint32_t in_hit_word, uint32_t in_ray_word, int zero_word,
                                                          uint32_t n_seg, uint32_t seg_cap, uint32_t split_at, PlanRec* __restrict__ plan,
                                                          LogRec* __restrict__ log, uint32_t stamp) {
  // Thread t reads words 2p, 2p + 1 (p = t % 8) of lines t / 8, t / 8 + 32, ...: eight threads share
  // a line, every load instruction of a wave covers eight whole lines, the 32 loads of a thread do
  // not depend on each other, and the two counts of a queue (adjacent words, the first one even)
  // meet in one thread.
  static_assert(RT_LINE_WORDS == 16 && RT_SHARDS % (RT_PLAN_THREADS / 8) == 0, "k_plan's mapping of threads to words");
  static_assert(CNT_Q0 % 2 == 0 && CNT_R0 == CNT_Q0 + 1 && CNT_Q1 % 2 == 0 && CNT_R1 == CNT_Q1 + 1, "queue counts are even-aligned pairs");
  constexpr int WAVES = RT_PLAN_THREADS / 64, ROWS = RT_PLAN_THREADS / 8;
  __shared__ unsigned long long s_sum[WAVES][RT_LINE_WORDS];
  __shared__ uint32_t s_max[WAVES][RT_LINE_WORDS];
  __shared__ uint32_t s_over[WAVES];
  const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
  const uint32_t word = 2u * (t & 7u), row0 = t >> 3;
  const bool zero_it = zero_word >= 0 && word == (uint32_t)zero_word;
  const bool counts = word == in_hit_word;  // then word + 1 == in_ray_word
  unsigned long long sum0 = 0ull, sum1 = 0ull;  // of this thread's words: over every line, or over the segments if they are the queue's counts
  uint32_t mx0 = 0u, mx1 = 0u, ov = 0u;
#pragma unroll 8
  for (uint32_t k = 0; k < RT_SHARDS / ROWS; ++k) {
    const uint32_t seg = row0 + k * ROWS;
    ulonglong2* p = reinterpret_cast<ulonglong2*>(counters + (size_t)seg * RT_LINE_WORDS + word);
    ulonglong2 v = *p;
    if (zero_it) *p = make_ulonglong2(0ull, 0ull);  // each word is read by exactly one thread: nobody else still needs it
    if (counts) {
      if (seg >= n_seg) v = make_ulonglong2(0ull, 0ull);
      mx0 = max(mx0, (uint32_t)v.x);
      mx1 = max(mx1, (uint32_t)v.y);
      ov |= (uint64_t)(uint32_t)v.x + (uint32_t)v.y > seg_cap ? 1u : 0u;
    }
    sum0 += v.x;
    sum1 += v.y;
  }
  // lanes l, l + 8, ..., l + 56 hold the same words
  for (int off = 8; off < 64; off <<= 1) {
    sum0 += __shfl_xor(sum0, off);
    sum1 += __shfl_xor(sum1, off);
    mx0 = max(mx0, (uint32_t)__shfl_xor((int)mx0, off));
    mx1 = max(mx1, (uint32_t)__shfl_xor((int)mx1, off));
  }
  for (int off = 32; off > 0; off >>= 1) ov |= (uint32_t)__shfl_xor((int)ov, off);
  if (lane < 8u) {
    s_sum[wave][word] = sum0;
    s_sum[wave][word + 1u] = sum1;
    s_max[wave][word] = mx0;
    s_max[wave][word + 1u] = mx1;
  }
  if (lane == 0) s_over[wave] = ov;
  __syncthreads();
  if (t < RT_LINE_WORDS) {
    unsigned long long x = 0ull;
    uint32_t m = 0u;
    for (int k = 0; k < WAVES; ++k) {
      x += s_sum[k][t];
      m = max(m, s_max[k][t]);
    }
    s_sum[0][t] = x;
    s_max[0][t] = m;
    if (log && t < CNT_N) {
      log->w[t] = x;
      __threadfence_system();  // in host memory before the stamp below says the record is complete
    }
  }
  __syncthreads();
  if (t == 0) {
    uint32_t over = 0;
    for (int k = 0; k < WAVES; ++k) over |= s_over[k];
    PlanRec pr;
    pr.hit_blocks = (s_max[0][in_hit_word] + RT_BLOCK - 1) / RT_BLOCK;
    pr.ray_blocks = (s_max[0][in_ray_word] + RT_BLOCK - 1) / RT_BLOCK;
    pr.n_hit = (uint32_t)s_sum[0][in_hit_word];
    pr.n_ray = (uint32_t)s_sum[0][in_ray_word];
    const PlanRec none{0u, 0u, 0u, 0u};
    plan[0] = pr;
    if (split_at > 0) {
      const bool fits = (uint64_t)pr.n_hit + pr.n_ray <= split_at;
      plan[1] = fits ? pr : none;
      plan[2] = fits ? none : pr;
    }
    if (log) {
      log->n_hit = pr.n_hit;
      log->n_ray = pr.n_ray;
      log->overflow = over;
      log->hit_blocks = pr.hit_blocks;
      log->ray_blocks = pr.ray_blocks;
      __threadfence_system();
      *reinterpret_cast<volatile uint32_t*>(&log->stamp) = stamp;  // the host reads a record only when its stamp matches
    }
  }
}

// STAGE 0: LOOP = false follows exactly one bounce before it queues its hits (pp.fuse_bounces must
// be 1), LOOP = true follows pp.fuse_bounces of them.
// Later stages: LOOP = false: workgroup (s, y) handles record block vb_base + y of segment s, or nothing if the
// plan says the segment has fewer. LOOP = true: it handles blocks vb_base + y, + gridDim.y, ... to
// the end — the form that cannot fall short whatever the counts turn out to be; it holds more
// registers, so the host launches the first form on a grid sized from experience and the second
// behind it for whatever that grid did not cover (normally nothing).
// EMIT = false: for scenes without emitters (sc.has_emission == 0); the radiance a path has
// gathered so far is then zero until its last ray, and is neither carried in registers nor queued.
template <bool COUNT, int STAGE, bool LOOP, bool QUANT, bool EMIT, bool DEEP>
__global__ __launch_bounds__(RT_BLOCK) void k_wave(SceneK sc, PathParams pp, QueueView in, QueueView out, uint32_t n_seg,
                                                   const PlanRec* __restrict__ plan, uint32_t vb_base, double* __restrict__ samples,
                                                   unsigned long long* counters) {
  int32_t* stk = g_stack + threadIdx.x;
  const int stride = blockDim.x;
  const int lane = threadIdx.x & 63;
  WorkCount wc{};
  uint32_t n_done = 0, n_vb = 0;
#ifdef RT_STAMPS
  if (lane == 0) {
    for (int i = 0; i < RT_NSEC; ++i) s_stamp[threadIdx.x >> 6][i] = 0ull;
    s_stamp[threadIdx.x >> 6][RT_NSEC] = clock64();
  }
#endif
  // Stage 0: one block of paths per workgroup. Later stages: record blocks of the workgroup's
  // segment, as many as the plan says there are.
  uint32_t hit_blocks = 0, vb_end = 1;
  if (STAGE != 0) {
    hit_blocks = plan->hit_blocks;
    vb_end = hit_blocks + plan->ray_blocks;
  }
  for (uint32_t vb = (STAGE != 0 ? vb_base : 0u) + blockIdx.y; vb < vb_end; vb += gridDim.y) {
  n_vb++;
  uint32_t seg;
  size_t entry = 0;
  bool alive;
  bool ray_block = false;  // this block reads the back run of its segment (uniform over the block)
  if (STAGE == 0) {
    seg = blockIdx.x % n_seg;
    alive = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x < pp.n_paths;
  } else {
    seg = blockIdx.x;
    const unsigned long long* line = counters + (size_t)seg * RT_LINE_WORDS;
    ray_block = vb >= hit_blocks;
    if (!ray_block) {
      const uint32_t i = vb * RT_BLOCK + threadIdx.x;
      alive = i < (uint32_t)line[in.hit_word];
      entry = (size_t)seg * in.seg_cap + i;
    } else {
      const uint32_t i = (vb - hit_blocks) * RT_BLOCK + threadIdx.x;
      alive = i < (uint32_t)line[in.ray_word];
      entry = (size_t)seg * in.seg_cap + (in.seg_cap - 1u - (alive ? i : 0u));
    }
  }
  uint32_t path = 0;
  uint64_t key = 0;
  int depth = 0;
  d3 o = mk(0, 0, 0), dir = mk(0, 0, 0), thr = mk(1, 1, 1), rad = mk(0, 0, 0);
  uint32_t fuse_left = STAGE == 0 ? pp.fuse_bounces : 0u;
  bool ended = false;  // this lane's path has just ended
#ifdef RT_DEBUG_HIST
  int32_t dbg_prev = 0, dbg_obj = 0;
#endif

  if (alive) {
    if (STAGE == 0) {
      path = blockIdx.x * RT_BLOCK + threadIdx.x;
      alive = start_path(pp, RT_BLOCK == 64 ? blockIdx.x : path >> 6, path, key, o, dir);
      if (!alive) store_sample(samples, pp.n_paths, path, mk(0.0, 0.0, 0.0));
    } else {
      const QueueRec rec_in = queue_load(in, entry);
      const QueueRec* r = &rec_in;
      const d3 p0 = mk(r->hit_pos[0], r->hit_pos[1], r->hit_pos[2]);
      dir = mk(r->dir[0], r->dir[1], r->dir[2]);
      thr = mk(r->thr[0], r->thr[1], r->thr[2]);
      key = r->key;
      path = r->path;
      depth = r->depth;
      if (EMIT) rad = mk(in.rad[entry * 3], in.rad[entry * 3 + 1], in.rad[entry * 3 + 2]);
      ShadeIn si;
      si.hit_pos = p0;
      si.u = r->u;
      si.v = r->v;
      const int32_t id = r->id;
#ifdef RT_DEBUG_HIST
      dbg_prev = r->pad;
      dbg_obj = id >= 0 ? sc.shade[id].obj : sc.spheres[~id].obj;
#endif
      si.slot = id >= 0 ? id : -1;
      si.sphere = id >= 0 ? -1 : ~id;
      if (COUNT) wc.shaded++;
      shade_hit<COUNT, EMIT>(sc, si, key, depth, o, dir, thr, rad, wc);
      depth += 1;
      if (depth > pp.max_depth) {  // src/main.rs:818
        store_sample(samples, pp.n_paths, path, rad);
        alive = false;
        ended = true;
      }
      // What the walk through the tree does not need waits in the record the lane has just read (the
      // throughput, which shading has changed, is written over the old one; key, path and depth are
      // still there) and is read back behind the walk: ten registers fewer where the kernel holds most.
      if (STAGE == 1 && alive) {
        in.recs[(size_t)3 * in.entries + entry] = make_float4(__int_as_float(__double2loint(thr.x)), __int_as_float(__double2hiint(thr.x)),
                                                              __int_as_float(__double2loint(thr.y)), __int_as_float(__double2hiint(thr.y)));
        reinterpret_cast<double*>(in.recs + (size_t)4 * in.entries + entry)[0] = thr.z;
        if (EMIT) {
          in.rad[entry * 3] = rad.x;
          in.rad[entry * 3 + 1] = rad.y;
          in.rad[entry * 3 + 2] = rad.z;
        }
      }
    }
  }

  n_done += (uint32_t)__popcll(__ballot(ended));  // paths ended and rays cast are counted per wave, not per lane: two registers fewer
  ended = false;

  HitK h;
  h.slot = -1;
  h.sphere = -1;
  int32_t steps = 0;  // node visits of the ray that found this lane's hit
  auto trace = [&]() {
    wc.rays += (uint32_t)__popcll(__ballot(alive));
    if (alive) {
      Trav t;
      // The launches over the queue walk the half-size copy of the tree (see BvhNodeQ): they are
      // bound by node-fetch instructions. The tail is a few waves' serial chain, bound by latency:
      // there the binary32 nodes (no conversions between the fetch and the tests) are quicker.
      trav_begin<QUANT>(t, sc, o, dir, stk);
      trav_run<COUNT, QUANT, DEEP>(t, sc, stride, wc);
      steps = t.steps;
      if (STAGE == 1) {
        asm volatile("" ::: "memory");  // the values parked before the walk are read again, not kept
        // and the record's place is worked out again from the lane index (the old value would hold two registers across the walk)
        uint32_t lane_again;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lane_again) : "v"(threadIdx.x));
        const uint32_t i_again = (vb - (ray_block ? hit_blocks : 0u)) * RT_BLOCK + lane_again;
        const size_t entry = (size_t)seg * in.seg_cap + (ray_block ? in.seg_cap - 1u - i_again : i_again);
        const double2 p3 = reinterpret_cast<const double2*>(in.recs)[(size_t)3 * in.entries + entry];
        const double2 p4 = reinterpret_cast<const double2*>(in.recs)[(size_t)4 * in.entries + entry];
        const float4 p5 = in.recs[(size_t)5 * in.entries + entry], p6 = in.recs[(size_t)6 * in.entries + entry];
        thr = mk(p3.x, p3.y, p4.x);
        key = ((uint64_t)(uint32_t)__float_as_int(p5.w) << 32) | (uint32_t)__float_as_int(p5.z);
        path = (uint32_t)__float_as_int(p6.x);
        depth = __float_as_int(p6.z) + 1;
        if (EMIT) rad = mk(in.rad[entry * 3], in.rad[entry * 3 + 1], in.rad[entry * 3 + 2]);
      }
#ifdef RT_DEBUG_HIST
      if (STAGE == 1) atomicAdd(&g_hist[min(max(dbg_obj, 0), 7)][min(max(dbg_prev, 0), 63)][min(steps, 63)], 1ull);
#endif
      finish_hit<COUNT, DEEP>(sc, t, h, wc);
      if (!(h.slot >= 0 || h.sphere >= 0)) {
        const d3 s = sky(sc, dir);  // src/main.rs:826
        rad = rad + mk(thr.x * s.x, thr.y * s.y, thr.z * s.z);
        store_sample(samples, pp.n_paths, path, rad);
        alive = false;
        ended = true;
      }
    }
    n_done += (uint32_t)__popcll(__ballot(ended));
    ended = false;
  };
  // from a trace on `alive` means: has a hit waiting to be shaded
  auto shade = [&]() {
    if (alive) {
      if (COUNT) wc.shaded++;
      const ShadeIn si = shade_input(h, o, dir);
      shade_hit<COUNT, EMIT>(sc, si, key, depth, o, dir, thr, rad, wc);
      depth += 1;
      if (depth > pp.max_depth) {
        store_sample(samples, pp.n_paths, path, rad);
        alive = false;
        ended = true;
      }
    }
    n_done += (uint32_t)__popcll(__ballot(ended));
    ended = false;
  };
  if (STAGE == 0 && !LOOP) {
    // The first stage following exactly one bounce, written out: camera ray, shade, scattered ray.
    // As a loop (below, for any other number of fused bounces) the same code holds 124 registers
    // instead of 92, which costs the kernel a wave per SIMD.
    trace();
    if (__any(alive)) {
      shade();
      trace();
    }
  } else
  for (;;) {
    trace();
    if (STAGE == 2 || (STAGE == 0 && fuse_left > 0)) {
      if (!__any(alive)) break;
      if (STAGE == 0) fuse_left--;
      shade();
      continue;
    }
    break;
  }

  // append: hits found by cheap rays at the front of this block's segment, the others at its back
  if (STAGE != 2) {
    unsigned long long* line = counters + (size_t)seg * RT_LINE_WORDS;
    const bool longer = alive && (uint32_t)steps >= pp.long_steps;
    const unsigned long long hit_mask = __ballot(alive && !longer), ray_mask = __ballot(longer);
    size_t e = 0;
    if (hit_mask) {
      const uint32_t i = wave_reserve(&line[out.hit_word], hit_mask, lane);
      if (alive && !longer) e = (size_t)seg * out.seg_cap + i;
    }
    if (ray_mask) {
      const uint32_t i = wave_reserve(&line[out.ray_word], ray_mask, lane);
      if (longer) e = (size_t)seg * out.seg_cap + (out.seg_cap - 1u - i);
    }
    if (alive) {
      QueueRec rec_out;
      QueueRec* r = &rec_out;
      const d3 p0 = o + h.t * dir;  // src/main.rs:831
      r->hit_pos[0] = p0.x;
      r->hit_pos[1] = p0.y;
      r->hit_pos[2] = p0.z;
      r->dir[0] = dir.x;
      r->dir[1] = dir.y;
      r->dir[2] = dir.z;
      r->thr[0] = thr.x;
      r->thr[1] = thr.y;
      r->thr[2] = thr.z;
      r->u = h.u;
      r->v = h.v;
      r->key = key;
      r->path = path;
      r->id = h.slot >= 0 ? h.slot : ~h.sphere;
      r->depth = depth;
      r->pad = steps;
      queue_store(out, e, rec_out);
      if (EMIT) {
        out.rad[e * 3] = rad.x;
        out.rad[e * 3 + 1] = rad.y;
        out.rad[e * 3 + 2] = rad.z;
      }
    }
  }
  if (!LOOP || STAGE == 0) break;  // the first stage has one block of paths per workgroup
  }  // record blocks of this workgroup
  flush_counts<true>(wc, counters, COUNT, n_done, n_vb);
#ifdef RT_STAMPS
  RT_ACC(wc, 7);
  if (lane == 0)
    for (int i = 0; i < RT_NSEC; ++i) atomicAdd(&g_stamps[STAGE][i], s_stamp[threadIdx.x >> 6][i]);
#endif
}

// Copies the records still queued (both runs of every segment: front and back) into
// one compact segment, whose two counts live in line 0 of `out_counters`. The tail kernel then
// runs from that copy on the side stream while the main stream reuses the queues.
__global__ __launch_bounds__(RT_BLOCK) void k_gather(QueueView in, QueueView out, const PlanRec* __restrict__ plan, int has_rad,
                                                     const unsigned long long* __restrict__ in_counters, unsigned long long* out_counters) {
  // one workgroup per segment; it reserves room for the segment's whole front run, and for its
  // whole back run, with one atomic each (every wave adding to the one tail word would queue up
  // behind the ~88 atomics per microsecond a single word takes)
  if (plan->hit_blocks + plan->ray_blocks == 0u) return;  // nothing to gather, or too much for the tail segment
  const uint32_t seg = blockIdx.x;
  const unsigned long long* line = in_counters + (size_t)seg * RT_LINE_WORDS;
  for (int run = 0; run < 2; ++run) {
    const uint32_t n = (uint32_t)line[run ? in.ray_word : in.hit_word];
    if (n == 0u) continue;
    unsigned int base = 0;
    if (threadIdx.x == 0) base = atomicAdd(reinterpret_cast<unsigned int*>(&out_counters[run ? out.ray_word : out.hit_word]), n);
    base = __shfl(base, 0);
    for (uint32_t i = threadIdx.x; i < n; i += RT_BLOCK) {
      const size_t entry = (size_t)seg * in.seg_cap + (run ? in.seg_cap - 1u - i : i);
      const uint32_t o = base + i;
      if (o >= out.seg_cap) continue;  // cannot happen: the plan is empty unless every record fits
      const size_t e = run ? (size_t)(out.seg_cap - 1u - o) : (size_t)o;
      queue_store(out, e, queue_load(in, entry));
      if (has_rad) {
        out.rad[e * 3] = in.rad[entry * 3];
        out.rad[e * 3 + 1] = in.rad[entry * 3 + 1];
        out.rad[e * 3 + 2] = in.rad[entry * 3 + 2];
      }
    }
  }
}

// Sums a pixel's samples in sample order (src/main.rs:1020-1024). `totals` carries the running
// sum between launches when a call is split into several sample chunks. On the last chunk the
// mean is added to the canvas (src/main.rs:1029-1037), or the plain sum to an f32 buffer.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(TileMap tm, const double* __restrict__ samples, uint32_t n_paths, uint32_t spp,
                                                      double* __restrict__ totals, int first_chunk, int last_chunk, double divisor,
                                                      double* __restrict__ canvas, float* __restrict__ sum_f32, int store) {
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
    const double* p = samples + (base + (uint64_t)s * 64u) * 3u;
    tr += p[0];
    tg += p[1];
    tb += p[2];
  }
  if (!last_chunk) {
    totals[i] = tr;
    totals[n_pix + i] = tg;
    totals[2 * n_pix + i] = tb;
    return;
  }
  const size_t pix = ((size_t)y * tm.width + x) * 3;
  if (canvas) {
    canvas[pix + 0] = (store ? 0.0 : canvas[pix + 0]) + tr / divisor;
    canvas[pix + 1] = (store ? 0.0 : canvas[pix + 1]) + tg / divisor;
    canvas[pix + 2] = (store ? 0.0 : canvas[pix + 2]) + tb / divisor;
  }
  if (sum_f32) {
    sum_f32[pix + 0] = (store ? 0.0f : sum_f32[pix + 0]) + (float)tr;
    sum_f32[pix + 1] = (store ? 0.0f : sum_f32[pix + 1]) + (float)tg;
    sum_f32[pix + 2] = (store ? 0.0f : sum_f32[pix + 2]) + (float)tb;
  }
}

// ------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------
// LDS of a workgroup: the traversal stack, one column per lane. A tree deeper than RT_DEEP_FROM
// entries runs the kernels' DEEP form, which keeps RT_DEEP_LDS entries there.
static size_t stack_lds_bytes(const DeviceScene* d) {
  const uint32_t entries = d->k.deep_stack ? (uint32_t)RT_DEEP_LDS : d->stack_entries;
  return (size_t)entries * RT_BLOCK * sizeof(int32_t);
}

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
  tm.d_tiles_x = make_fastdiv(tm.tiles_x);
  tm.d_big = make_fastdiv(tm.big);
  tm.d_per_big = make_fastdiv(tm.big * tm.big);
  tm.d_bigs_x = make_fastdiv(tm.bigs_x);
  return tm;
}

static size_t env_mib(const char* name, size_t dflt) {
  const char* e = std::getenv(name);
  if (e && *e) {
    long v = std::atol(e);
    if (v > 0) return (size_t)v;
  }
  return dflt;
}

static int event_at(DeviceScene* d, size_t i, hipEvent_t* out) {
  while (d->ev_pool.size() <= i) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    d->ev_pool.push_back(e);
  }
  *out = d->ev_pool[i];
  return RT_OK;
}

// Queue segments of a launch of n_paths paths: enough that concurrently running waves rarely
// share a tail, few enough that a segment still feeds whole blocks.
static uint32_t choose_segments(uint32_t n_paths) {
  const uint32_t blocks = (n_paths + RT_BLOCK - 1) / RT_BLOCK;
  return std::max<uint32_t>(1, std::min<uint32_t>(RT_SHARDS, blocks / 64));
}
static uint64_t queue_entries(uint32_t n_paths) {
  const uint32_t blocks = (n_paths + RT_BLOCK - 1) / RT_BLOCK;
  const uint32_t n_seg = choose_segments(n_paths);
  return (uint64_t)n_seg * ((blocks + n_seg - 1) / n_seg) * RT_BLOCK;
}

// QUANT (the half-size grid nodes) exists for the launches over the queue only (STAGE 1).
template <bool COUNT, int STAGE, bool LOOP, bool QUANT>
static void launch_wave_as(dim3 grid, size_t lds, hipStream_t st, const SceneK& k, const PathParams& pp, const QueueView& in, const QueueView& out,
                           uint32_t n_seg, const PlanRec* plan, uint32_t vb_base, double* samples, unsigned long long* counters) {
  const bool deep = k.deep_stack != 0u;
#define RT_LAUNCH_K(E, D) \
  hipLaunchKernelGGL((k_wave<COUNT, STAGE, LOOP, QUANT, E, D>), grid, dim3(RT_BLOCK), lds, st, k, pp, in, out, n_seg, plan, vb_base, samples, counters)
  if (k.has_emission) {
    if (deep) RT_LAUNCH_K(true, true);
    else RT_LAUNCH_K(true, false);
  } else {
    if (deep) RT_LAUNCH_K(false, true);
    else RT_LAUNCH_K(false, false);
  }
#undef RT_LAUNCH_K
}
template <int STAGE>
static void launch_wave(bool count, bool loop, bool quant, dim3 grid, size_t lds, hipStream_t st, const SceneK& k, const PathParams& pp, const QueueView& in,
                        const QueueView& out, uint32_t n_seg, const PlanRec* plan, uint32_t vb_base, double* samples, unsigned long long* counters) {
#define RT_LAUNCH_AS(C, L, Q) launch_wave_as<C, STAGE, L, Q>(grid, lds, st, k, pp, in, out, n_seg, plan, vb_base, samples, counters)
  if constexpr (STAGE == 1 || STAGE == 0) {
    if (quant) {
      if (!loop) {
        if (count) RT_LAUNCH_AS(true, false, true);
        else RT_LAUNCH_AS(false, false, true);
      } else {
        if (count) RT_LAUNCH_AS(true, true, true);
        else RT_LAUNCH_AS(false, true, true);
      }
      return;
    }
  }
  if (!loop) {
    if (count) RT_LAUNCH_AS(true, false, false);
    else RT_LAUNCH_AS(false, false, false);
  } else {
    if (count) RT_LAUNCH_AS(true, true, false);
    else RT_LAUNCH_AS(false, true, false);
  }
#undef RT_LAUNCH_AS
}

// `arena` != nullptr: the call takes every work buffer out of the arena (the static work memory)
// and makes no allocating call: no second sample buffer, no side stream, no events.
static int render_impl(rt_scene* s, const rt_render_params* p, double* d_canvas, float* d_sum, hipStream_t stream, const char* who,
                       Arena* arena = nullptr) {
  int rc = require_device(s, who);
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, who)) != RT_OK) return rc;
  if (!d_canvas && !d_sum) {
    rt_set_error("%s: null output buffer", who);
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  const bool mega = (p->flags & (RT_FLAG_MEGAKERNEL | RT_FLAG_NO_REGEN)) != 0;
  // RT_FLAG_PIPELINE: the call may return with its tail and resolve still queued on the side
  // stream, so that the next call's first stage overlaps them. Consecutive pipelined calls on one
  // stream form a run whose counters, launch log and timing accumulate.
  const bool pipeline = (p->flags & RT_FLAG_PIPELINE) != 0 && !mega && !arena;
  const bool continuing = pipeline && d->run_open && d->last_stream_valid && d->last_stream == stream;
  if (!continuing) {
    if (d->run_open && (rc = finish_run(d)) != RT_OK) return rc;  // an earlier run still holds the side stream
    // An earlier call may still have kernels queued on another stream; they read the counters and
    // work buffers this call is about to reset.
    if (d->last_stream_valid && d->last_stream != stream) {
      HIP_TRY(hipEventRecord(d->ev_join[0], d->last_stream));
      HIP_TRY(hipStreamWaitEvent(stream, d->ev_join[0], 0));
    }
    d->have_timing = false;
    d->launches.clear();
    d->pending.clear();
    d->accounts.clear();
    d->snap_floor = d->snap_seq;
    d->last_main_snap = -1;
    d->last_tail_snap = -1;
    d->n_events = 0;
    d->side_used = false;
    std::memset(&d->timing, 0, sizeof d->timing);
    // Also for a call that takes its work buffers out of the static array: the counters are the
    // scene's own (allocated at upload), so that the statistics read after the call cannot be
    // overwritten by another scene's call on the same GPU.
    d->cur_counters = d->counters;
    HIP_TRY(hipMemsetAsync(d->cur_counters, 0, RT_COUNTER_BYTES, stream));
    HIP_TRY(hipMemsetAsync(d->tail_counters, 0, RT_COUNTER_BYTES, stream));
  }
  unsigned long long* const counters = d->cur_counters;
  d->last_stream = stream;
  d->last_stream_valid = true;
  d->run_open = pipeline;
  if (p->spp == 0) return RT_OK;

  const TileMap tm = make_tilemap(p);
  if (tm.n_tiles == 0) return RT_OK;
  const bool timing = (p->flags & RT_FLAG_TIMING) != 0 && !arena;  // timing takes events from a growing pool
  if (timing && d->ev_pool.size() < RT_EVENT_POOL_FIRST) {
    // the first timed call of a scene makes the events of a long pipelined run at once (a run of
    // forty 1080p frames brackets about a thousand launches), so that the calls after it create none
    hipEvent_t e;
    if ((rc = event_at(d, RT_EVENT_POOL_FIRST - 1, &e)) != RT_OK) return rc;
  }
  const bool count = (p->flags & RT_FLAG_COUNT_WORK) != 0;
  const bool regen = (p->flags & RT_FLAG_NO_REGEN) == 0;
  const bool emission = d->k.has_emission != 0;

  // Work buffers are bounded (RT_WORK_BUFFER_MIB); a call that needs more runs in sample chunks.
  const uint64_t n_pix = (uint64_t)tm.n_tiles * 64u;
  const size_t per_path = 24u + (mega ? 0u : 2u * (sizeof(QueueRec) + (emission ? 24u : 0u)));
  // Budget of the work buffers: RT_WORK_BUFFER_MIB, else 192 GiB of the MI355X's 288 GB, and in
  // any case no more than four fifths of what is free (counting what the scene already holds).
  size_t budget = env_mib("RT_WORK_BUFFER_MIB", 196608) << 20;
  if (arena) {
    // what is left of the array, less the per-pixel totals of a call in several chunks and the
    // round-up of the queues to whole blocks per segment (at most RT_SHARDS blocks each) and of
    // every piece to 256 bytes
    const size_t left = arena->bytes - arena->used;
    const size_t fixed = (size_t)n_pix * 24u + 2u * (size_t)RT_SHARDS * RT_BLOCK * (sizeof(QueueRec) + 24u) + 8u * 256u;
    if (left < fixed + (size_t)n_pix * per_path) {
      rt_set_error("%s: %llu pixels need %llu bytes of work memory per sample, the static work memory holds %llu (RT_NOALLOC_SCRATCH_MIB = %u); "
                   "render the image in tiles (tile_size, tile_rank, tile_ranks) or use rt_render_device",
                   who, (unsigned long long)n_pix, (unsigned long long)(fixed + n_pix * per_path), (unsigned long long)left, (unsigned)RT_NOALLOC_SCRATCH_MIB);
      return RT_ERR_INVALID;
    }
    budget = left - fixed;
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t held = d->samples_bytes + d->totals_bytes + d->queue_bytes[0] + d->queue_bytes[1] + d->queue_rad_bytes[0] + d->queue_rad_bytes[1];
      budget = std::min<size_t>(budget, (free_b + held) / 5 * 4);
    }
  }
  uint64_t chunk64 = std::max<uint64_t>(1, budget / (n_pix * per_path));
  chunk64 = std::min<uint64_t>(chunk64, 0xfff00000ull / n_pix);
  if (chunk64 == 0) {
    rt_set_error("%s: image too large for one launch", who);
    return RT_ERR_INVALID;
  }
  // A call in several chunks overlaps each chunk's tail with the next chunk, and a pipelined call
  // its last tail with the next call; both take a second sample buffer (24 B per path more).
  const char* no_overlap = std::getenv("RT_NO_OVERLAP");
  bool overlap = !mega && !arena && (pipeline || p->spp > chunk64) && !(no_overlap && *no_overlap);
  if (overlap) chunk64 = std::max<uint64_t>(1, std::min<uint64_t>(budget / (n_pix * (per_path + 24u)), 0xfff00000ull / n_pix));
  const uint32_t chunk = (uint32_t)std::min<uint64_t>(p->spp, chunk64);
  const uint32_t n_chunks = (p->spp + chunk - 1) / chunk;
  overlap = overlap && (n_chunks > 1 || pipeline);
  const uint64_t cap = n_pix * chunk;  // paths per launch
  double* samples_base = nullptr;
  double* totals = nullptr;
  void* queue[2] = {nullptr, nullptr};
  double* queue_rad[2] = {nullptr, nullptr};
  size_t samples_half = 0;  // doubles from one sample buffer to the other (overlap)
  if (arena) {
    bool ok = (samples_base = static_cast<double*>(arena_take(arena, (size_t)cap * 24u))) != nullptr;
    if (n_chunks > 1) ok = ok && (totals = static_cast<double*>(arena_take(arena, (size_t)n_pix * 24u))) != nullptr;
    if (!mega) {
      const uint64_t qe = queue_entries((uint32_t)cap);
      for (int q = 0; q < 2; ++q) {
        ok = ok && (queue[q] = arena_take(arena, (size_t)qe * sizeof(QueueRec))) != nullptr;
        if (emission) ok = ok && (queue_rad[q] = static_cast<double*>(arena_take(arena, (size_t)qe * 24u))) != nullptr;
      }
    }
    if (!ok) {
      rt_set_error("%s: internal error: the static work memory ran out", who);
      return RT_ERR_INVALID;
    }
  } else {
    if ((rc = ensure_bytes((void**)&d->samples, &d->samples_bytes, (size_t)cap * 24u * (overlap ? 2u : 1u))) != RT_OK) return rc;
    if (n_chunks > 1 && (rc = ensure_bytes((void**)&d->totals, &d->totals_bytes, (size_t)n_pix * 24u)) != RT_OK) return rc;
    if (!mega) {
      const uint64_t qe = queue_entries((uint32_t)cap);
      for (int q = 0; q < 2; ++q) {
        if ((rc = ensure_bytes(&d->queue[q], &d->queue_bytes[q], (size_t)qe * sizeof(QueueRec))) != RT_OK) return rc;
        if (emission && (rc = ensure_bytes((void**)&d->queue_rad[q], &d->queue_rad_bytes[q], (size_t)qe * 24u)) != RT_OK) return rc;
      }
    }
    samples_base = d->samples;
    totals = d->totals;
    samples_half = d->samples_bytes / 16u;
    for (int q = 0; q < 2; ++q) {
      queue[q] = d->queue[q];
      queue_rad[q] = d->queue_rad[q];
    }
  }
  const uint32_t tail_threshold = (uint32_t)env_mib("RT_TAIL_THRESHOLD", 262144);
  // One compact segment per parity holds the input of a tail that runs on the side stream: room for
  // twice the threshold (a frame queued from the learnt number of launches may arrive at the tail
  // with somewhat more than the threshold), in whole blocks of records.
  const uint32_t tail_cap = ((2u * tail_threshold + RT_BLOCK - 1) / RT_BLOCK + 1) * RT_BLOCK;
  if (overlap) {
    for (int q = 0; q < 2; ++q) {
      if ((rc = ensure_bytes(&d->tail_in[q], &d->tail_in_bytes[q], (size_t)tail_cap * sizeof(QueueRec))) != RT_OK) return rc;
      if (emission && (rc = ensure_bytes((void**)&d->tail_rad[q], &d->tail_rad_bytes[q], (size_t)tail_cap * 24u)) != RT_OK) return rc;
    }
    d->side_used = true;
  }

  const size_t lds = stack_lds_bytes(d);
  const double divisor = (double)(p->spp_total ? p->spp_total : p->spp);
  const char* fuse_env = std::getenv("RT_FUSE_BOUNCES");
  const uint32_t fuse_bounces = fuse_env && *fuse_env ? (uint32_t)std::max(0, std::atoi(fuse_env)) : 1u;
  // RT_LONG_STEPS=n: hits found after n or more node visits are queued apart from the others (0: all
  // in one run).
  const char* long_env = std::getenv("RT_LONG_STEPS");
  uint32_t long_steps = long_env && *long_env ? (uint32_t)std::max(0, std::atoi(long_env)) : RT_LONG_STEPS_DEFAULT;
  if (long_steps == 0) long_steps = 0x7fffffffu;
  // RT_GRID_BLOCKS: workgroups of the looping form of a launch over a queue
  const uint32_t grid_blocks = (uint32_t)env_mib("RT_GRID_BLOCKS", 4096);
  // RT_SYNC_ROUNDS=1: the host reads every plan of every frame before the launch it sizes (a wait per bounce; A/B)
  const char* sync_env = std::getenv("RT_SYNC_ROUNDS");
  const bool always_watch = sync_env && *sync_env && *sync_env != '0';
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device);

  // Launch log; with timing every launch is bracketed by two events of the pool.
  auto begin_launch = [&](uint32_t kind, uint32_t items, hipStream_t on) -> int {
    rt_launch_info li;
    std::memset(&li, 0, sizeof li);
    li.kind = kind;
    li.items = items;
    d->launches.push_back(li);
    if (timing) {
      hipEvent_t e;
      int r = event_at(d, d->n_events, &e);
      if (r != RT_OK) return r;
      HIP_TRY(hipEventRecord(e, on));
      d->pending.push_back(DeviceScene::Pending{d->launches.size() - 1, d->n_events});
      d->n_events += 2;
    }
    return RT_OK;
  };
  auto end_launch = [&](hipStream_t on) -> int {
    HIP_TRY(hipGetLastError());
    if (timing) {
      hipEvent_t e;
      int r = event_at(d, d->pending.back().ev0 + 1, &e);
      if (r != RT_OK) return r;
      HIP_TRY(hipEventRecord(e, on));
    }
    return RT_OK;
  };
  // Queues a k_plan on `on` over the lines `lines`: the extent of the queue (in_hit_word,
  // in_ray_word) for the next launch and a snapshot of the lines' sums. Returns the snapshot's
  // sequence number in *seq; the plan records are at plan_at(seq).
  auto plan_at = [&](int64_t seq) -> PlanRec* { return d->plans + ((uint64_t)seq % RT_PLAN_SLOTS) * RT_PLAN_RECS; };
  auto take_plan = [&](hipStream_t on, unsigned long long* lines, uint32_t in_hit_word, uint32_t in_ray_word, int zero_word,
                       uint32_t n_seg, uint32_t seg_cap, uint32_t split_at, int64_t* seq) -> int {
    if (d->snap_seq - d->snap_floor >= RT_PLAN_SLOTS - 8) {  // the ring is about to wrap: settle what is in flight
      int r = settle(d);
      if (r != RT_OK) return r;
    }
    *seq = (int64_t)d->snap_seq++;
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(RT_PLAN_THREADS), 0, on, lines, in_hit_word, in_ray_word, zero_word, n_seg, seg_cap, split_at,
                       plan_at(*seq), d->host_log + ((uint64_t)*seq % RT_PLAN_SLOTS), (uint32_t)((uint64_t)*seq + 1u));
    HIP_TRY(hipGetLastError());
    return RT_OK;
  };
  // The work of the launch logged last = snapshot `after` minus the previous snapshot of the same lines.
  auto account_main = [&](int64_t after, int64_t items_from) {
    d->accounts.push_back(DeviceScene::Account{d->launches.size() - 1, d->last_main_snap, after, items_from});
    d->last_main_snap = after;
  };

  for (uint32_t c = 0; c < n_chunks; ++c) {
    PathParams pp;
    pp.cam = make_camera(p);
    pp.tm = tm;
    pp.seed_key = seed_key(p->seed);
    pp.spp = std::min(chunk, p->spp - c * chunk);
    pp.d_spp = make_fastdiv(pp.spp);
    pp.first_sample = p->first_sample + c * chunk;
    pp.n_paths = (uint32_t)(n_pix * pp.spp);
    pp.max_depth = p->max_depth;
    pp.fuse_bounces = fuse_bounces;
    pp.long_steps = long_steps;
    const uint32_t blocks_all = (pp.n_paths + RT_BLOCK - 1) / RT_BLOCK;
    // Every chunk but the last hands its tail and its resolve to the side stream; so does the
    // last chunk of a pipelined call.
    const bool handoff = overlap && (c + 1 < n_chunks || pipeline);
    const int par = overlap ? (int)(d->chunk_seq++ & 1u) : 0;
    double* const samples = samples_base + (overlap ? (size_t)par * samples_half : 0u);
    // this parity's sample buffer and tail input were last read two chunks ago on the side stream
    if (overlap && d->resolved_valid[par]) HIP_TRY(hipStreamWaitEvent(stream, d->ev_resolved[par], 0));
    hipStream_t resolve_on = stream;
    int64_t snap = -1;

    if (mega) {
      uint32_t grid = blocks_all;
      if (regen) {
        grid = std::min<uint32_t>(blocks_all, (uint32_t)cus * 8u * (256u / RT_BLOCK));  // 32 waves per CU
        HIP_TRY(hipMemsetAsync(counters + CNT_WORK, 0, sizeof(unsigned long long), stream));
      }
      if ((rc = begin_launch(3, pp.n_paths, stream)) != RT_OK) return rc;
      if (regen) {
        if (d->k.deep_stack) hipLaunchKernelGGL((k_paths<true, true, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
        else if (count) hipLaunchKernelGGL((k_paths<true, true, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
        else hipLaunchKernelGGL((k_paths<false, true, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
      } else {
        if (d->k.deep_stack) hipLaunchKernelGGL((k_paths<true, false, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
        else if (count) hipLaunchKernelGGL((k_paths<true, false, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
        else hipLaunchKernelGGL((k_paths<false, false, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, d->k, pp, samples, counters);
      }
      if ((rc = end_launch(stream)) != RT_OK) return rc;
      if ((rc = take_plan(stream, counters, CNT_Q0, CNT_R0, -1, 0u, 0u, 0u, &snap)) != RT_OK) return rc;
      account_main(snap, -1);
    } else {
      // wavefront: stage 0 over all paths, then launches over the queued hits
      const uint32_t n_seg = choose_segments(pp.n_paths);
      const uint32_t seg_cap = ((blocks_all + n_seg - 1) / n_seg) * RT_BLOCK;
      QueueView qv[2];
      for (int q = 0; q < 2; ++q)
        qv[q] = QueueView{(float4*)queue[q], n_seg * seg_cap, queue_rad[q], seg_cap, (uint32_t)(q ? CNT_Q1 : CNT_Q0), (uint32_t)(q ? CNT_R1 : CNT_R0)};
      const size_t line_bytes = RT_LINE_WORDS * sizeof(unsigned long long);
      // grid of the looping form of the launches over a queue (it covers whatever the sized grid of
      // the plain form did not, normally nothing): a few workgroups per segment
      const uint32_t seg_blocks = seg_cap / RT_BLOCK;
      const dim3 cgrid(n_seg, std::max<uint32_t>(1u, std::min<uint32_t>(seg_blocks, (grid_blocks + n_seg - 1) / n_seg)));
      // the four counts of every line start at zero (adjacent words)
      HIP_TRY(hipMemset2DAsync(counters + CNT_Q0, line_bytes, 0, 4 * sizeof(unsigned long long), RT_SHARDS, stream));
      if ((rc = begin_launch(0, pp.n_paths, stream)) != RT_OK) return rc;
      // the written-out form for one fused bounce (the default), the looping form for any other number
      launch_wave<0>(count, fuse_bounces != 1u, d->grid_first, dim3(blocks_all), lds, stream, d->k, pp, qv[0], qv[0], n_seg, nullptr, 0u, samples, counters);
      if ((rc = end_launch(stream)) != RT_OK) return rc;
      // the split of the plan (gather into the tail segment, or tail in place) only matters to the
      // launches after the last bounce, and costs nothing on the others
      const uint32_t split_at = handoff ? tail_cap : 0u;
      if ((rc = take_plan(stream, counters, CNT_Q0, CNT_R0, CNT_Q1, n_seg, seg_cap, split_at, &snap)) != RT_OK) return rc;
      account_main(snap, -1);

      // Bounce launches. A frame of a shape seen before is queued blind, with as many launches as
      // that frame needed; a frame of a new shape with a guessed number of launches in the looping form.
      uint64_t key = 0xcbf29ce484222325ull;
      for (uint64_t v : {(uint64_t)pp.n_paths, (uint64_t)pp.spp, (uint64_t)p->width, (uint64_t)p->height, (uint64_t)(uint32_t)p->max_depth,
                         (uint64_t)p->tile_size, (uint64_t)p->tile_rank, (uint64_t)p->tile_ranks, (uint64_t)fuse_bounces, (uint64_t)long_steps,
                         (uint64_t)tail_threshold, (uint64_t)(p->flags & RT_FLAG_NO_JITTER)})
        key = (key ^ v) * 0x100000001b3ull;
      // Three ways to size them. A shape seen before (`known`): as many launches as its last frame
      // needed, grids of what those needed plus an eighth, the looping form behind each for the
      // rest. A new shape: a guessed number of launches, each in the looping form alone, which covers
      // a queue of any length (slower: it holds more registers and fewer workgroups; it is one
      // frame). Either way the host waits for nothing, and the plans the frame leaves behind in host
      // memory become the shape's history as soon as they are all there. RT_SYNC_ROUNDS=1: the
      // host reads every plan before the launch it sizes (exact grids; a wait per bounce).
      learn_from_landed_frames(d);
      const DeviceScene::Hist* known = nullptr;
      if (!always_watch)
        for (DeviceScene::Hist& x : d->hists)
          if (x.key == key) {
            x.used = ++d->hist_clock;
            known = &x;
          }
      const uint32_t blind_rounds = guess_rounds(pp.n_paths / 4u, tail_threshold);
      DeviceScene::FrameLearn fl;
      fl.key = key;
      fl.final_plan = -1;
      fl.seg_blocks = seg_cap / RT_BLOCK;
      fl.threshold = tail_threshold;
      fl.handoff = handoff;
      // Grid of the plain form of launch `r` over a queue, and whether the looping form must follow
      // it to cover what that grid may have missed.
      auto sized_grid = [&](uint32_t r, int64_t plan_seq, uint32_t* gy) -> bool {
        if (always_watch) {  // the host has read the plan: exact, nothing left to cover
          const LogRec& lr = d->host_log[(uint64_t)plan_seq % RT_PLAN_SLOTS];
          *gy = lr.hit_blocks + lr.ray_blocks;
          return false;
        }
        if (!known) {  // the looping form alone
          *gy = 0u;
          return true;
        }
        const uint32_t h = r < known->blocks.size() ? known->blocks[r] : 0u;
        *gy = std::min<uint32_t>(seg_blocks, h + h / 8u + 2u);
        return *gy < seg_blocks;
      };
      uint32_t rounds = 0;
      int cur = 0;
      for (;;) {
        if (always_watch) {
          HIP_TRY(hipStreamSynchronize(stream));
          const LogRec& lr = d->host_log[(uint64_t)snap % RT_PLAN_SLOTS];
          if (lr.overflow || (uint64_t)lr.n_hit + lr.n_ray > pp.n_paths) {
            rt_set_error("%s: internal error: a queue segment holds more than its %u records", who, seg_cap);
            return RT_ERR_HIP;
          }
          if ((uint64_t)lr.n_hit + lr.n_ray <= tail_threshold) break;
        } else if (rounds >= (known ? known->rounds : blind_rounds)) {
          break;
        }
        const int nxt = cur ^ 1;
        uint32_t gy = 0;
        const bool cover = sized_grid(rounds, snap, &gy);
        if ((rc = begin_launch(1, 0u, stream)) != RT_OK) return rc;
        if (gy) launch_wave<1>(count, false, d->grid_ok, dim3(n_seg, gy), lds, stream, d->k, pp, qv[cur], qv[nxt], n_seg, plan_at(snap), 0u, samples, counters);
        if (cover) launch_wave<1>(count, true, d->grid_ok, cgrid, lds, stream, d->k, pp, qv[cur], qv[nxt], n_seg, plan_at(snap), gy, samples, counters);
        if ((rc = end_launch(stream)) != RT_OK) return rc;
        const int64_t sized_by = snap;
        fl.round_plans.push_back(sized_by);
        // counts of the queue just written; the counts of the one just read start the next round at zero
        if ((rc = take_plan(stream, counters, qv[nxt].hit_word, qv[nxt].ray_word, (int)qv[cur].hit_word, n_seg, seg_cap, split_at, &snap)) != RT_OK) return rc;
        account_main(snap, sized_by);
        cur = nxt;
        rounds++;
      }
      fl.final_plan = snap;
      if (!always_watch) {
        if (d->learn_q.size() >= 64) d->learn_q.pop_front();
        d->learn_q.push_back(std::move(fl));
      }
      const int nxt = cur ^ 1;

      // What is left loops to its end in the tail kernel.
      if (handoff) {
        // If it fits the tail segment (plan record 1 is then the extent, record 2 empty) it is
        // compacted into it and the side stream takes it from there, next to the next chunk or
        // frame; otherwise (record 1 empty, record 2 the extent) the tail runs here, in place.
        const QueueView tv{(float4*)d->tail_in[par], tail_cap, d->tail_rad[par], tail_cap, (uint32_t)(par ? CNT_Q1 : CNT_Q0), (uint32_t)(par ? CNT_R1 : CNT_R0)};
        HIP_TRY(hipMemsetAsync(d->tail_counters + tv.hit_word, 0, 2 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_gather, dim3(n_seg), dim3(RT_BLOCK), 0, stream, qv[cur], tv, (const PlanRec*)(plan_at(snap) + 1), (int)emission, counters, d->tail_counters);
        HIP_TRY(hipGetLastError());
        if ((rc = begin_launch(2, 0u, stream)) != RT_OK) return rc;
        launch_wave<2>(count, true, false, cgrid, lds, stream, d->k, pp, qv[cur], qv[nxt], n_seg, plan_at(snap) + 2, 0u, samples, counters);
        if ((rc = end_launch(stream)) != RT_OK) return rc;
        {
          int64_t after;
          if ((rc = take_plan(stream, counters, CNT_Q0, CNT_R0, -1, 0u, 0u, 0u, &after)) != RT_OK) return rc;
          account_main(after, -1);
        }
        HIP_TRY(hipEventRecord(d->ev_gathered[par], stream));
        HIP_TRY(hipStreamWaitEvent(d->side, d->ev_gathered[par], 0));
        // on the side stream: extent of the compact segment (and the side counters before the tail), the tail, the counters after
        int64_t tplan, tafter;
        if ((rc = take_plan(d->side, d->tail_counters, tv.hit_word, tv.ray_word, -1, 1u, tail_cap, 0u, &tplan)) != RT_OK) return rc;
        if ((rc = begin_launch(2, 0u, d->side)) != RT_OK) return rc;
        launch_wave<2>(count, false, false, dim3(1, tail_cap / RT_BLOCK), lds, d->side, d->k, pp, tv, tv, 1u, plan_at(tplan), 0u, samples, d->tail_counters);
        if ((rc = end_launch(d->side)) != RT_OK) return rc;
        if ((rc = take_plan(d->side, d->tail_counters, tv.hit_word, tv.ray_word, -1, 0u, 0u, 0u, &tafter)) != RT_OK) return rc;
        d->accounts.push_back(DeviceScene::Account{d->launches.size() - 1, tplan, tafter, tplan});
        d->last_tail_snap = tafter;
        resolve_on = d->side;
      } else {
        if (always_watch) HIP_TRY(hipStreamSynchronize(stream));  // the last plan may not have been read yet
        uint32_t gy = 0;
        const bool cover = sized_grid(rounds, snap, &gy);
        if ((rc = begin_launch(2, 0u, stream)) != RT_OK) return rc;
        if (gy) launch_wave<2>(count, false, false, dim3(n_seg, gy), lds, stream, d->k, pp, qv[cur], qv[nxt], n_seg, plan_at(snap), 0u, samples, counters);
        if (cover) launch_wave<2>(count, true, false, cgrid, lds, stream, d->k, pp, qv[cur], qv[nxt], n_seg, plan_at(snap), gy, samples, counters);
        if ((rc = end_launch(stream)) != RT_OK) return rc;
        const int64_t sized_by = snap;
        if ((rc = take_plan(stream, counters, CNT_Q0, CNT_R0, -1, 0u, 0u, 0u, &snap)) != RT_OK) return rc;
        account_main(snap, sized_by);
      }
    }
    // The running per-pixel totals pass from chunk to chunk in order: a handed-off chunk resolves
    // on the side stream (after its tail); the last chunk resolves on the caller's stream after
    // everything the side stream still holds.
    if (handoff && resolve_on == stream) {  // single-kernel form: nothing was handed over
      HIP_TRY(hipEventRecord(d->ev_gathered[par], stream));
      HIP_TRY(hipStreamWaitEvent(d->side, d->ev_gathered[par], 0));
      resolve_on = d->side;
    }
    if (overlap && !handoff && d->resolved_valid[par ^ 1]) HIP_TRY(hipStreamWaitEvent(stream, d->ev_resolved[par ^ 1], 0));
    const uint32_t rgrid = (uint32_t)((n_pix + RT_BLOCK - 1) / RT_BLOCK);
    if ((rc = begin_launch(4, (uint32_t)n_pix, resolve_on)) != RT_OK) return rc;
    hipLaunchKernelGGL(k_resolve, dim3(rgrid), dim3(RT_BLOCK), 0, resolve_on, tm, samples, pp.n_paths, pp.spp, totals,
                       (int)(c == 0), (int)(c + 1 == n_chunks), divisor, d_canvas, d_sum, (int)((p->flags & RT_FLAG_STORE) != 0));
    if ((rc = end_launch(resolve_on)) != RT_OK) return rc;
    if (handoff) {
      HIP_TRY(hipEventRecord(d->ev_resolved[par], d->side));
      d->resolved_valid[par] = true;
    }
  }
  return RT_OK;
}

static int rt_render_device_body(rt_scene* s, const rt_render_params* p, double* device_canvas, void* stream);
extern "C" int rt_render_device(rt_scene* s, const rt_render_params* p, double* device_canvas, void* stream) {
  return rt_guard("rt_render_device", [&] { return rt_render_device_body(s, p, device_canvas, stream); });
}
static int rt_render_device_body(rt_scene* s, const rt_render_params* p, double* device_canvas, void* stream) {
  return render_impl(s, p, device_canvas, nullptr, (hipStream_t)stream, "rt_render_device");
}

// The same call as rt_render_device with every work buffer (sample buffer, running totals, both
// ray queues) taken out of the static work memory: no hipMalloc, hipFree, stream or
// event creation. The scene must have been uploaded before (rt_scene_upload allocates).
extern "C" int rt_render_device_noalloc(rt_scene* s, const rt_render_params* p, double* device_canvas, void* stream) {
  return rt_guard("rt_render_device_noalloc", [&] {
    if (!s || !s->dev) {
      rt_set_error("rt_render_device_noalloc: the scene has not been uploaded (rt_scene_upload allocates; this call does not)");
      return s ? RT_ERR_NO_DEVICE : RT_ERR_INVALID;
    }
    Arena arena{nullptr, RT_SCRATCH_BYTES, 0};
    int rc = scratch_address(s->dev->device, &arena.base);
    if (rc != RT_OK) return rc;
    return render_impl(s, p, device_canvas, nullptr, (hipStream_t)stream, "rt_render_device_noalloc", &arena);
  });
}

static int rt_render_sum_f32_device_body(rt_scene* s, const rt_render_params* p, float* device_sum, void* stream);
extern "C" int rt_render_sum_f32_device(rt_scene* s, const rt_render_params* p, float* device_sum, void* stream) {
  return rt_guard("rt_render_sum_f32_device", [&] { return rt_render_sum_f32_device_body(s, p, device_sum, stream); });
}
static int rt_render_sum_f32_device_body(rt_scene* s, const rt_render_params* p, float* device_sum, void* stream) {
  return render_impl(s, p, nullptr, device_sum, (hipStream_t)stream, "rt_render_sum_f32_device");
}

static int rt_render_accumulate_body(rt_scene* s, const rt_render_params* p, double* canvas, uint64_t* rays_out);
extern "C" int rt_render_accumulate(rt_scene* s, const rt_render_params* p, double* canvas, uint64_t* rays_out) {
  return rt_guard("rt_render_accumulate", [&] { return rt_render_accumulate_body(s, p, canvas, rays_out); });
}
static int rt_render_accumulate_body(rt_scene* s, const rt_render_params* p, double* canvas, uint64_t* rays_out) {
  int rc = require_device(s, "rt_render_accumulate");
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, "rt_render_accumulate")) != RT_OK) return rc;
  if (!canvas) {
    rt_set_error("rt_render_accumulate: null canvas");
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(double);
  // The frame is rendered into a zeroed device buffer, copied once into pinned host memory and
  // added to the caller's canvas on the host: 0 + x = x, so canvas + frame rounds exactly as the
  // kernel's own canvas += x would, and the canvas crosses PCIe once, at pinned speed, instead
  // of twice from pageable memory (49.8 MB at 1080p: 5.6 ms -> 2.6 ms).
  if ((rc = ensure_bytes(&d->stage, &d->stage_bytes, bytes)) != RT_OK) return rc;
  if (d->host_stage_bytes < bytes) {
    if (d->host_stage) (void)hipHostFree(d->host_stage);
    d->host_stage = nullptr;
    d->host_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&d->host_stage, bytes, hipHostMallocDefault));
    d->host_stage_bytes = bytes;
  }
  HIP_TRY(hipMemsetAsync(d->stage, 0, bytes, d->own));
  rc = render_impl(s, p, (double*)d->stage, nullptr, d->own, "rt_render_accumulate");
  if (rc != RT_OK) return rc;
  if ((rc = finish_run(d)) != RT_OK) return rc;  // waits for the scene's streams
  HIP_TRY(hipMemcpyAsync(d->host_stage, d->stage, bytes, hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipStreamSynchronize(d->own));
  rt_add_into(canvas, static_cast<const double*>(d->host_stage), bytes / sizeof(double));
  if (rays_out) {
    CounterSums cs;
    if ((rc = read_call_counters(d, d->own, &cs)) != RT_OK) return rc;
    *rays_out = cs.w[CNT_RAYS];
  }
  return RT_OK;
}

static int rt_trace_body(rt_scene* s, const double* org3, const double* dir3, size_t n, rt_hit* out);
extern "C" int rt_trace(rt_scene* s, const double* org3, const double* dir3, size_t n, rt_hit* out) {
  return rt_guard("rt_trace", [&] { return rt_trace_body(s, org3, dir3, n, out); });
}
static int rt_trace_body(rt_scene* s, const double* org3, const double* dir3, size_t n, rt_hit* out) {
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
  if ((rc = finish_run(d)) != RT_OK) return rc;  // waits for the scene's streams
  const size_t ray_bytes = n * 3 * sizeof(double), hit_bytes = n * sizeof(rt_hit);
  if ((rc = ensure_bytes(&d->stage, &d->stage_bytes, 2 * ray_bytes + hit_bytes)) != RT_OK) return rc;
  double* d_org = (double*)d->stage;
  double* d_dir = d_org + n * 3;
  rt_hit* d_hit = (rt_hit*)(d_dir + n * 3);
  HIP_TRY(hipMemcpyAsync(d_org, org3, ray_bytes, hipMemcpyHostToDevice, d->own));
  HIP_TRY(hipMemcpyAsync(d_dir, dir3, ray_bytes, hipMemcpyHostToDevice, d->own));
  d->cur_counters = d->counters;
  HIP_TRY(hipMemsetAsync(d->counters, 0, RT_COUNTER_BYTES, d->own));
  const uint32_t grid = (uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK);
  if (d->k.deep_stack) hipLaunchKernelGGL((k_trace<true, true>), dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), d->own, d->k, d_org, d_dir, n, d_hit, d->counters);
  else hipLaunchKernelGGL((k_trace<true, false>), dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), d->own, d->k, d_org, d_dir, n, d_hit, d->counters);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_hit, hit_bytes, hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipStreamSynchronize(d->own));
  d->last_stream = d->own;
  d->last_stream_valid = true;
  d->side_used = false;
  return RT_OK;
}

static int rt_primary_hits_body(rt_scene* s, const rt_render_params* p, int32_t* obj, int32_t* tri, double* t, double* u, double* v);
extern "C" int rt_primary_hits(rt_scene* s, const rt_render_params* p, int32_t* obj, int32_t* tri, double* t, double* u, double* v) {
  return rt_guard("rt_primary_hits", [&] { return rt_primary_hits_body(s, p, obj, tri, t, u, v); });
}
static int rt_primary_hits_body(rt_scene* s, const rt_render_params* p, int32_t* obj, int32_t* tri, double* t, double* u, double* v) {
  int rc = require_device(s, "rt_primary_hits");
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p, "rt_primary_hits")) != RT_OK) return rc;
  if (!obj || !tri || !t || !u || !v) {
    rt_set_error("rt_primary_hits: null output");
    return RT_ERR_INVALID;
  }
  DeviceScene* d = s->dev;
  if ((rc = finish_run(d)) != RT_OK) return rc;  // waits for the scene's streams
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
  d->cur_counters = d->counters;
  HIP_TRY(hipMemsetAsync(d->counters, 0, RT_COUNTER_BYTES, d->own));
  const uint32_t grid = (uint32_t)((npix + RT_BLOCK - 1) / RT_BLOCK);
  if (d->k.deep_stack) hipLaunchKernelGGL(k_primary<true>, dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), d->own, d->k, cam, d_obj, d_tri, d_t, d_u, d_v, d->counters);
  else hipLaunchKernelGGL(k_primary<false>, dim3(grid), dim3(RT_BLOCK), stack_lds_bytes(d), d->own, d->k, cam, d_obj, d_tri, d_t, d_u, d_v, d->counters);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(t, d_t, npix * sizeof(double), hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipMemcpyAsync(u, d_u, npix * sizeof(double), hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipMemcpyAsync(v, d_v, npix * sizeof(double), hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipMemcpyAsync(obj, d_obj, npix * sizeof(int32_t), hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipMemcpyAsync(tri, d_tri, npix * sizeof(int32_t), hipMemcpyDeviceToHost, d->own));
  HIP_TRY(hipStreamSynchronize(d->own));
  d->last_stream = d->own;
  d->last_stream_valid = true;
  d->side_used = false;
  return RT_OK;
}

// ------------------------------------------------------------------------------------------
// Arithmetic self-test: the kernels divide a vector by its length, and colour bytes by 255, with
// one refined reciprocal per denominator (rtk::div3_by_length, rtk::div3_in_range); this checks on
// the device that the quotients are the bits of the plain IEEE divisions, on pseudo-random and on
// boundary operands.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool same_bits(double a, double b) {
  return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
}
__global__ __launch_bounds__(256) void k_selftest_div(uint64_t n, uint64_t seed, unsigned long long* mismatches) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long bad = 0;
  uint64_t z[3];
  for (int k = 0; k < 3; ++k) z[k] = mix64(seed + RT_GOLDEN * (3 * i + k + 1));
  const uint32_t mode = (uint32_t)(mix64(seed ^ (i * 0x9e3779b97f4a7c15ull + 7)) & 15u);
  d3 a;
  double* comp[3] = {&a.x, &a.y, &a.z};
  for (int k = 0; k < 3; ++k) {
    uint64_t bits = z[k];
    const uint64_t mant = bits & 0xfffffffffffffull, sign = bits & 0x8000000000000000ull;
    uint64_t ex = (bits >> 52) & 0x7ffull;
    if (mode < 8) ex = 1023 - 40 + (ex % 80);                     // moderate magnitudes: the rendering case
    else if (mode == 8) ex = 1023 - 1000 + (ex % 2000);           // nearly the whole range
    else if (mode == 9) ex = (ex % 3 == 0) ? 0 : 1023 - 8 + ex % 16;  // a denormal or zero-exponent component among ordinary ones
    else if (mode == 10) ex = 1023 - 420 + (ex % 40);             // lengths around the lower edge of the fast range
    else if (mode == 11) ex = 1023 + 380 + (ex % 40);             // and around the upper edge
    else if (mode == 12) ex = (k == (int)(bits % 3)) ? 1023 - 950 + (ex % 100) : 1023 + (ex % 4);  // one tiny numerator
    // modes 13..15: raw bits (NaNs, infinities, denormals as they come)
    bits = sign | (ex << 52) | mant;
    if (mode == 13 && (bits & 0xf00) == 0) bits = sign;            // some exact zeros of either sign
    if (mode < 8 && (z[(k + 1) % 3] & 0x3f) == 0) bits = sign;     // axis-aligned vectors
    *comp[k] = __longlong_as_double((long long)bits);
  }
  const double l = length(a);
  const d3 fast = div3_by_length(a, l);
  if (!same_bits(fast.x, a.x / l) || !same_bits(fast.y, a.y / l) || !same_bits(fast.z, a.z / l)) bad++;
  const d3 px = mk((double)(z[0] & 255u), (double)(z[1] & 255u), (double)(z[2] & 255u));
  const d3 c = div3_in_range(px, 255.0);
  if (!same_bits(c.x, px.x / 255.0) || !same_bits(c.y, px.y / 255.0) || !same_bits(c.z, px.z / 255.0)) bad++;
  const unsigned long long total = wave_sum(bad);
  if ((threadIdx.x & 63) == 0 && total) atomicAdd(mismatches, total);
}

extern "C" int rt_selftest_division(uint64_t n, uint64_t seed, uint64_t* mismatches) {
  return rt_guard("rt_selftest_division", [&] {
    if (!mismatches) {
      rt_set_error("rt_selftest_division: null argument");
      return RT_ERR_INVALID;
    }
    if (rt_device_count() <= 0) {
      rt_set_error("rt_selftest_division: no HIP device is visible");
      return RT_ERR_NO_DEVICE;
    }
    unsigned long long* d_bad = nullptr;
    HIP_TRY(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
    int rc = RT_OK;
    auto body = [&]() -> int {
      HIP_TRY(hipMemset(d_bad, 0, sizeof(unsigned long long)));
      const uint64_t per_launch = 1ull << 28;
      for (uint64_t done = 0; done < n; done += per_launch) {
        const uint64_t m = std::min<uint64_t>(per_launch, n - done);
        hipLaunchKernelGGL(k_selftest_div, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, nullptr, m, seed + done, d_bad);
        HIP_TRY(hipGetLastError());
      }
      unsigned long long bad = 0;
      HIP_TRY(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
      *mismatches = bad;
      return RT_OK;
    };
    rc = body();
    (void)hipFree(d_bad);
    return rc;
  });
}

static int rt_scene_wait_body(rt_scene* s);
extern "C" int rt_scene_wait(rt_scene* s) {
  return rt_guard("rt_scene_wait", [&] { return rt_scene_wait_body(s); });
}
static int rt_scene_wait_body(rt_scene* s) {
  if (!s) {
    rt_set_error("rt_scene_wait: null scene");
    return RT_ERR_INVALID;
  }
  if (!s->dev) return RT_OK;
  return finish_run(s->dev);
}

extern "C" int rt_scene_stream_wait(rt_scene* s, void* stream) {
  return rt_guard("rt_scene_stream_wait", [&] {
    if (!s) {
      rt_set_error("rt_scene_stream_wait: null scene");
      return RT_ERR_INVALID;
    }
    DeviceScene* d = s->dev;
    if (!d) return RT_OK;
    HIP_TRY(hipSetDevice(d->device));
    for (int q = 0; q < 2; ++q)
      if (!d->ev_join[q]) HIP_TRY(hipEventCreateWithFlags(&d->ev_join[q], hipEventDisableTiming));
    hipStream_t waiter = (hipStream_t)stream;
    if (d->last_stream_valid && d->last_stream != waiter) {
      HIP_TRY(hipEventRecord(d->ev_join[0], d->last_stream));
      HIP_TRY(hipStreamWaitEvent(waiter, d->ev_join[0], 0));
    }
    if (d->side) {
      HIP_TRY(hipEventRecord(d->ev_join[1], d->side));
      HIP_TRY(hipStreamWaitEvent(waiter, d->ev_join[1], 0));
    }
    return RT_OK;
  });
}

static int rt_scene_last_stats_body(rt_scene* s, rt_stats* out);
extern "C" int rt_scene_last_stats(rt_scene* s, rt_stats* out) {
  return rt_guard("rt_scene_last_stats", [&] { return rt_scene_last_stats_body(s, out); });
}
static int rt_scene_last_stats_body(rt_scene* s, rt_stats* out) {
  if (!s || !out) {
    rt_set_error("rt_scene_last_stats: null argument");
    return RT_ERR_INVALID;
  }
  if (!s->dev) {
    rt_set_error("rt_scene_last_stats: the scene has not been uploaded");
    return RT_ERR_NO_DEVICE;
  }
  DeviceScene* d = s->dev;
  int rc = finish_run(d);
  if (rc != RT_OK) return rc;
  CounterSums cs;
  if ((rc = read_call_counters(d, d->last_stream, &cs)) != RT_OK) return rc;
  out->rays = cs.w[CNT_RAYS];
  out->node_visits = cs.w[CNT_NODES];
  out->tri_tests = cs.w[CNT_TRIS];
  out->sphere_tests = cs.w[CNT_SPHERES];
  out->shaded_hits = cs.w[CNT_SHADED];
  out->paths = cs.w[CNT_PATHS];
  out->waves = cs.w[CNT_WAVES];
  out->wave_iters_inner = cs.w[CNT_WI_INNER];
  out->wave_iters_leaf = cs.w[CNT_WI_LEAF];
  out->wave_iters_reject = cs.w[CNT_WI_REJECT];
  return RT_OK;
}

static int rt_scene_last_timing_body(rt_scene* s, rt_timing* out);
extern "C" int rt_scene_last_timing(rt_scene* s, rt_timing* out) {
  return rt_guard("rt_scene_last_timing", [&] { return rt_scene_last_timing_body(s, out); });
}
static int rt_scene_last_timing_body(rt_scene* s, rt_timing* out) {
  if (!s || !out) {
    rt_set_error("rt_scene_last_timing: null argument");
    return RT_ERR_INVALID;
  }
  if (s->dev && finish_run(s->dev) != RT_OK) return RT_ERR_HIP;
  if (!s->dev || !s->dev->have_timing) {
    rt_set_error("rt_scene_last_timing: the last call was not made with RT_FLAG_TIMING");
    return RT_ERR_INVALID;
  }
  *out = s->dev->timing;
  return RT_OK;
}

static int rt_scene_last_launches_body(rt_scene* s, rt_launch_info* out, uint32_t capacity, uint32_t* n_out);
extern "C" int rt_scene_last_launches(rt_scene* s, rt_launch_info* out, uint32_t capacity, uint32_t* n_out) {
  return rt_guard("rt_scene_last_launches", [&] { return rt_scene_last_launches_body(s, out, capacity, n_out); });
}
static int rt_scene_last_launches_body(rt_scene* s, rt_launch_info* out, uint32_t capacity, uint32_t* n_out) {
  if (!s || !n_out || (capacity && !out)) {
    rt_set_error("rt_scene_last_launches: null argument");
    return RT_ERR_INVALID;
  }
  if (!s->dev) {
    rt_set_error("rt_scene_last_launches: the scene has not been uploaded");
    return RT_ERR_NO_DEVICE;
  }
  if (finish_run(s->dev) != RT_OK) return RT_ERR_HIP;
  const std::vector<rt_launch_info>& l = s->dev->launches;
  *n_out = (uint32_t)l.size();
  for (uint32_t i = 0; i < capacity && i < l.size(); ++i) out[i] = l[i];
  return RT_OK;
}
