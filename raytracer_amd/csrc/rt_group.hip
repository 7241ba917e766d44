// rt_group.hip — one frame over several GPUs driven from ONE process.
//
// The reference is a single process whose render (src/main.rs:977-1039) spreads independent
// pixels and samples over the machine's cores (rayon, src/main.rs:988, 994). A caller shaped
// like that reaches the GPUs of a node through a group: one member per GPU, each with its own
// copy of the read-only scene in its own HBM and its own stream. A render call only queues work
// (its launch extents are decided on the device), so the calling thread queues every member's
// share one after the other and the GPUs run them side by side; events, not host threads, order
// the combination behind the members. Nothing in a frame waits on the host: the combination is
// ordered behind the caller's stream and behind the members by events, every member has two sum
// buffers, so a member renders frame k + 1 while the first GPU still combines frame k, and the
// caller collects a run of frames with rt_group_wait.
//
// Exchange step. Every member leaves a binary32 [H][W][3] sum of its samples in its own HBM.
// The GPUs of a node are fully connected by xGMI links, so instead of a ring the first member's
// GPU simply reads: k_combine runs there, loads each element from every member's buffer through
// peer mappings (all links into that GPU busy at once, no staging copies, nothing written
// remotely), adds them in member order in binary64 and adds the mean to the canvas. With the
// tile split it loads an element only from the member that owns its tile, so one frame's worth
// of data crosses the links in total.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "rt_types.h"

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RT_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

namespace {

struct SumTable {
  const float* sum[RT_GROUP_MAX];
};

// canvas[e] += (sum over members, or the owner's value) / divisor, one binary32 element per lane:
// a wave reads 256 consecutive bytes of each member's buffer.
__global__ __launch_bounds__(256) void k_combine(SumTable t, uint32_t n_members, uint64_t n_elems, uint32_t width,
                                                 uint32_t tile_size, uint32_t tiles_x, double divisor,
                                                 double* __restrict__ canvas) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  double acc = 0.0;
  if (tile_size == 0) {
    for (uint32_t g = 0; g < n_members; ++g) acc += (double)t.sum[g][e];
  } else {
    const uint64_t pixel = e / 3u;
    const uint32_t x = (uint32_t)(pixel % width), y = (uint32_t)(pixel / width);
    const uint32_t owner = ((y / tile_size) * tiles_x + x / tile_size) % n_members;
    acc = (double)t.sum[owner][e];
  }
  canvas[e] += acc / divisor;
}

struct Member {
  rt_scene view;  // no host arrays: only the device half, uploaded from the caller's scene
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // around the member's share of the latest frame, on its stream
  float* sum[2] = {nullptr, nullptr};               // frame k is summed into sum[k & 1]
  size_t sum_bytes[2] = {0, 0};
  bool timed = false;  // ev_begin / ev_end have been recorded since the last rt_group_wait
  // the frames covered by the latest rt_group_wait
  uint64_t rays = 0;
  double ms = 0.0;     // of the latest of them
};

}  // namespace

struct rt_group {
  std::vector<std::unique_ptr<Member>> members;
  // on the first device: the stream the frames are combined on, and per sum-buffer parity the event
  // of the caller's stream a frame starts behind and the event of its combination (which the
  // caller's stream, and the members' next use of that buffer, wait for)
  hipStream_t combine = nullptr;
  hipStream_t own = nullptr;  // the caller's stream of the host-buffer entry point
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_combined[2] = {nullptr, nullptr};
  bool combined_valid[2] = {false, false};
  uint64_t frames = 0;  // queued so far
  std::string failed;   // a frame that failed half queued: reported by the next rt_group_wait
  double* stage = nullptr;  // frame of the host-buffer entry point, on the first device
  size_t stage_bytes = 0;
  void* host_stage = nullptr;  // the same frame in pinned host memory
  size_t host_stage_bytes = 0;
};

static void group_free(rt_group* g) {
  if (!g) return;
  for (auto& m : g->members) {
    if (!m || m->device < 0) continue;
    (void)hipSetDevice(m->device);
    if (m->view.dev) rt_device_scene_free(m->view.dev);
    m->view.dev = nullptr;
    for (int b = 0; b < 2; ++b)
      if (m->sum[b]) (void)hipFree(m->sum[b]);
    if (m->ev_begin) (void)hipEventDestroy(m->ev_begin);
    if (m->ev_end) (void)hipEventDestroy(m->ev_end);
    if (m->stream) (void)hipStreamDestroy(m->stream);
  }
  if (!g->members.empty() && g->members[0] && g->members[0]->device >= 0) {
    (void)hipSetDevice(g->members[0]->device);
    if (g->stage) (void)hipFree(g->stage);
    for (int b = 0; b < 2; ++b) {
      if (g->ev_in[b]) (void)hipEventDestroy(g->ev_in[b]);
      if (g->ev_combined[b]) (void)hipEventDestroy(g->ev_combined[b]);
    }
    if (g->combine) (void)hipStreamDestroy(g->combine);
    if (g->own) (void)hipStreamDestroy(g->own);
  }
  if (g->host_stage) (void)hipHostFree(g->host_stage);
  delete g;
}

static int rt_group_create_body(const rt_scene* scene, const int* devices, uint32_t n, rt_group** out) {
  if (out) *out = nullptr;
  if (!scene || !devices || !out) {
    rt_set_error("rt_group_create: null argument");
    return RT_ERR_INVALID;
  }
  if (n == 0 || n > RT_GROUP_MAX) {
    rt_set_error("rt_group_create: a group has 1..%u members, not %u", RT_GROUP_MAX, n);
    return RT_ERR_INVALID;
  }
  const int n_dev = rt_device_count();
  if (n_dev <= 0) {
    rt_set_error("rt_group_create: no HIP device is visible; this library has no CPU path");
    return RT_ERR_NO_DEVICE;
  }
  for (uint32_t i = 0; i < n; ++i)
    if (devices[i] < 0 || devices[i] >= n_dev) {
      rt_set_error("rt_group_create: device %d out of range (0..%d)", devices[i], n_dev - 1);
      return RT_ERR_INVALID;
    }
  // the first device must be able to read every other member's memory
  const int root = devices[0];
  HIP_TRY(hipSetDevice(root));
  // RT_GROUP_LOG=1 prints which members the first device reaches and how; RT_GROUP_NO_PEER=i makes
  // member i count as unreachable (to exercise the failure path on any machine)
  const char* log_env = std::getenv("RT_GROUP_LOG");
  const bool log = log_env && *log_env && *log_env != '0';
  const char* inject_env = std::getenv("RT_GROUP_NO_PEER");
  const long inject = inject_env && *inject_env ? std::atol(inject_env) : -1;
  for (uint32_t i = 1; i < n; ++i) {
    if ((long)i == inject) {
      rt_set_error("rt_group_create: device %d cannot access the memory of device %d (member %u, RT_GROUP_NO_PEER)", root, devices[i], i);
      return RT_ERR_HIP;
    }
    if (devices[i] == root) {
      if (log) std::fprintf(stderr, "rt_group: member %u on device %d shares the first member's device\n", i, devices[i]);
      continue;
    }
    bool seen = false;
    for (uint32_t j = 1; j < i; ++j) seen = seen || devices[j] == devices[i];
    if (seen) continue;
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, root, devices[i]));
    if (log) std::fprintf(stderr, "rt_group: device %d -> device %d (member %u): peer access %s\n", root, devices[i], i, can ? "yes" : "NO");
    if (!can) {
      rt_set_error("rt_group_create: device %d cannot access the memory of device %d", root, devices[i]);
      return RT_ERR_HIP;
    }
    const hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
      (void)hipGetLastError();
    } else if (e != hipSuccess) {
      rt_set_error("rt_group_create: hipDeviceEnablePeerAccess(%d) from device %d failed: %s", devices[i], root, hipGetErrorString(e));
      return RT_ERR_HIP;
    }
  }
  rt_group* g = new rt_group();
  for (uint32_t i = 0; i < n; ++i) {
    g->members.emplace_back(new Member());
    Member* m = g->members.back().get();
    int rc = rt_device_scene_upload_from(scene, &m->view, devices[i]);  // leaves devices[i] current
    if (rc == RT_OK) {
      m->device = devices[i];
      if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->ev_begin) != hipSuccess ||
          hipEventCreate(&m->ev_end) != hipSuccess) {
        rt_set_error("rt_group_create: hipStreamCreate / hipEventCreate on device %d failed", devices[i]);
        rc = RT_ERR_HIP;
      }
    }
    if (rc != RT_OK) {
      group_free(g);
      return rc;
    }
  }
  HIP_TRY(hipSetDevice(root));
  bool ok = hipStreamCreateWithFlags(&g->combine, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&g->own, hipStreamNonBlocking) == hipSuccess;
  for (int b = 0; b < 2 && ok; ++b)
    ok = hipEventCreateWithFlags(&g->ev_in[b], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&g->ev_combined[b], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    rt_set_error("rt_group_create: hipStreamCreate / hipEventCreate on device %d failed", root);
    group_free(g);
    return RT_ERR_HIP;
  }
  *out = g;
  return RT_OK;
}

extern "C" int rt_group_create(const rt_scene* scene, const int* devices, uint32_t n_devices, rt_group** out) {
  return rt_guard("rt_group_create", [&] { return rt_group_create_body(scene, devices, n_devices, out); });
}

extern "C" void rt_group_destroy(rt_group* group) { group_free(group); }

extern "C" int rt_group_size(const rt_group* group, uint32_t* n_members) {
  if (!group || !n_members) {
    rt_set_error("rt_group_size: null argument");
    return RT_ERR_INVALID;
  }
  *n_members = (uint32_t)group->members.size();
  return RT_OK;
}

extern "C" int rt_group_member_info(const rt_group* group, uint32_t member, int* device, uint64_t* rays, double* ms) {
  if (!group || member >= group->members.size()) {
    rt_set_error("rt_group_member_info: no such member");
    return RT_ERR_INVALID;
  }
  const Member& m = *group->members[member];
  if (device) *device = m.device;
  if (rays) *rays = m.rays;
  if (ms) *ms = m.ms;
  return RT_OK;
}

// Queues one frame: every member's share on that member's stream, the combination on the first
// device behind them and behind what `stream` held when the call was made; `stream` in turn waits
// for the combination. Returns without waiting for any of it.
static int group_queue_frame(rt_group* g, const rt_render_params* p, uint32_t split, double* device_canvas, hipStream_t stream, const char* who) {
  if (!g || !p || !device_canvas) {
    rt_set_error("%s: null argument", who);
    return RT_ERR_INVALID;
  }
  if (split != RT_SPLIT_SAMPLES && split != RT_SPLIT_TILES) {
    rt_set_error("%s: unknown split %u", who, split);
    return RT_ERR_INVALID;
  }
  if (p->tile_rank != 0 || p->tile_ranks > 1) {
    rt_set_error("%s: tile_rank/tile_ranks belong to the group; describe the whole frame", who);
    return RT_ERR_INVALID;
  }
  if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536) {
    rt_set_error("%s: bad image size %ux%u", who, p->width, p->height);
    return RT_ERR_INVALID;
  }
  if (p->max_depth < 0 || p->max_depth > 100000 || !(p->fov_y_deg > 0.0 && p->fov_y_deg < 180.0) || !(p->dof_strength >= 0.0)) {
    rt_set_error("%s: bad max_depth, fov_y_deg or dof_strength", who);  // before anything is queued
    return RT_ERR_INVALID;
  }
  const uint32_t n = (uint32_t)g->members.size();
  uint32_t tile = 0;
  if (split == RT_SPLIT_TILES) {
    tile = p->tile_size ? p->tile_size : 64u;
    if (tile % 8u != 0) {
      rt_set_error("%s: tile_size must be a multiple of 8", who);
      return RT_ERR_INVALID;
    }
  }
  const uint32_t total = p->spp_total ? p->spp_total : p->spp;
  const uint64_t n_elems = (uint64_t)p->width * p->height * 3u;
  const size_t sum_bytes = (size_t)n_elems * sizeof(float);
  const int b = (int)(g->frames & 1u);

  // Whatever the caller queued on its stream before this call (a clear of the canvas, say) comes
  // before the combination; nothing else on the device is waited for.
  Member& root = *g->members[0];
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipEventRecord(g->ev_in[b], stream));

  for (uint32_t i = 0; i < n; ++i) {
    Member& m = *g->members[i];
    rt_render_params mp = *p;
    // the member's frames form a pipelined run: a frame's thin tail overlaps the member's next frame
    mp.flags |= RT_FLAG_PIPELINE | RT_FLAG_STORE;  // STORE: the buffer takes the frame's sums as they are, no clear
    mp.flags &= ~RT_FLAG_TIMING;
    mp.spp_total = total;
    if (split == RT_SPLIT_SAMPLES) {
      const uint32_t lo = (uint32_t)(((uint64_t)p->spp * i) / n), hi = (uint32_t)(((uint64_t)p->spp * (i + 1)) / n);
      mp.first_sample = p->first_sample + lo;
      mp.spp = hi - lo;
      mp.tile_size = 0;
    } else {
      mp.tile_size = tile;
      mp.tile_rank = i;
      mp.tile_ranks = n;
    }
    HIP_TRY(hipSetDevice(m.device));
    if (m.sum_bytes[b] < sum_bytes) {
      // growing a buffer: the combination that last read it must be over (a frame of another size
      // was in flight; rare, and the only wait a frame can meet)
      if (g->combined_valid[b]) HIP_TRY(hipEventSynchronize(g->ev_combined[b]));
      if (m.sum[b]) (void)hipFree(m.sum[b]);
      m.sum[b] = nullptr;
      m.sum_bytes[b] = 0;
      HIP_TRY(hipMalloc((void**)&m.sum[b], sum_bytes));
      m.sum_bytes[b] = sum_bytes;
    }
    // this buffer was last read by the combination of the frame before the previous one
    if (g->combined_valid[b]) HIP_TRY(hipStreamWaitEvent(m.stream, g->ev_combined[b], 0));
    HIP_TRY(hipEventRecord(m.ev_begin, m.stream));
    // a member without samples of its own (more members than samples) contributes zeros
    if (mp.spp == 0 && split == RT_SPLIT_SAMPLES) HIP_TRY(hipMemsetAsync(m.sum[b], 0, sum_bytes, m.stream));
    int rc = rt_render_sum_f32_device(&m.view, &mp, m.sum[b], m.stream);
    if (rc == RT_OK) {
      HIP_TRY(hipEventRecord(m.ev_end, m.stream));
      m.timed = true;
      // the combination starts behind everything this member has queued, its own side stream included
      rc = rt_scene_stream_wait(&m.view, g->combine);
    }
    if (rc != RT_OK) {
      const std::string why = rt_last_error();
      rt_set_error("%s: member %u (device %d): %s", who, i, m.device, why.c_str());
      g->failed = rt_last_error();  // members before this one have their shares queued: the frame is void
      return rc;
    }
  }

  // combine on the first device, behind the caller's stream and every member's share
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipStreamWaitEvent(g->combine, g->ev_in[b], 0));
  SumTable t{};
  for (uint32_t i = 0; i < n; ++i) t.sum[i] = g->members[i]->sum[b];
  // a single member in tile mode owns every tile; n == 1 needs no owner arithmetic either way
  const uint32_t k_tile = (split == RT_SPLIT_TILES && n > 1) ? tile : 0u;
  const uint32_t tiles_x = k_tile ? (p->width + k_tile - 1) / k_tile : 0u;
  const uint64_t blocks = (n_elems + 255u) / 256u;
  hipLaunchKernelGGL(k_combine, dim3((uint32_t)blocks), dim3(256), 0, g->combine, t, n, n_elems, p->width, k_tile, tiles_x,
                     (double)total, device_canvas);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(g->ev_combined[b], g->combine));
  g->combined_valid[b] = true;
  HIP_TRY(hipStreamWaitEvent(stream, g->ev_combined[b], 0));  // what the caller queues next sees the frame in the canvas
  g->frames++;
  return RT_OK;
}

// Waits for every frame queued so far and collects what the members did in them.
static int group_wait(rt_group* g, uint64_t* rays_out) {
  if (!g) {
    rt_set_error("rt_group_wait: null group");
    return RT_ERR_INVALID;
  }
  Member& root = *g->members[0];
  HIP_TRY(hipSetDevice(root.device));
  HIP_TRY(hipStreamSynchronize(g->combine));
  uint64_t rays = 0;
  for (auto& mp : g->members) {
    Member& m = *mp;
    HIP_TRY(hipSetDevice(m.device));
    rt_stats st;
    m.rays = 0;
    m.ms = 0.0;
    if (m.view.dev && m.timed) {  // ends the member's run of pipelined frames: the run's totals
      const int rc = rt_scene_last_stats(&m.view, &st);
      if (rc != RT_OK) return rc;
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, m.ev_begin, m.ev_end));
      m.rays = st.rays;
      m.ms = ms;
      m.timed = false;
    }
    rays += m.rays;
  }
  if (rays_out) *rays_out = rays;
  HIP_TRY(hipSetDevice(root.device));
  if (!g->failed.empty()) {
    rt_set_error("rt_group_wait: an earlier frame was not queued completely: %s", g->failed.c_str());
    g->failed.clear();
    return RT_ERR_INVALID;
  }
  return RT_OK;
}

extern "C" int rt_group_render_device(rt_group* group, const rt_render_params* p, uint32_t split, double* device_canvas, void* stream) {
  return rt_guard("rt_group_render_device",
                  [&] { return group_queue_frame(group, p, split, device_canvas, (hipStream_t)stream, "rt_group_render_device"); });
}

extern "C" int rt_group_wait(rt_group* group, uint64_t* rays_out) {
  return rt_guard("rt_group_wait", [&] { return group_wait(group, rays_out); });
}

static int rt_group_render_accumulate_body(rt_group* g, const rt_render_params* p, uint32_t split, double* canvas, uint64_t* rays_out) {
  if (!g || !p || !canvas) {
    rt_set_error("rt_group_render_accumulate: null argument");
    return RT_ERR_INVALID;
  }
  if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536) {
    rt_set_error("rt_group_render_accumulate: bad image size %ux%u", p->width, p->height);
    return RT_ERR_INVALID;
  }
  const size_t bytes = (size_t)p->width * p->height * 3u * sizeof(double);
  HIP_TRY(hipSetDevice(g->members[0]->device));
  int rc = group_wait(g, nullptr);  // frames queued through rt_group_render_device are not this call's
  if (rc != RT_OK) return rc;
  if (g->stage_bytes < bytes) {
    if (g->stage) (void)hipFree(g->stage);
    g->stage = nullptr;
    g->stage_bytes = 0;
    HIP_TRY(hipMalloc((void**)&g->stage, bytes));
    g->stage_bytes = bytes;
  }
  if (g->host_stage_bytes < bytes) {
    if (g->host_stage) (void)hipHostFree(g->host_stage);
    g->host_stage = nullptr;
    g->host_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&g->host_stage, bytes, hipHostMallocDefault));
    g->host_stage_bytes = bytes;
  }
  // as rt_render_accumulate does: the frame alone crosses PCIe, once, into pinned memory, and is
  // added to the caller's canvas on the host (0 + x = x: the same bits as adding on the device)
  HIP_TRY(hipMemsetAsync(g->stage, 0, bytes, g->own));  // the frame is ordered behind it through the stream
  rc = group_queue_frame(g, p, split, g->stage, g->own, "rt_group_render_accumulate");
  if (rc != RT_OK) {
    (void)group_wait(g, nullptr);  // let what was queued run out; the caller's canvas is untouched
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(g->host_stage, g->stage, bytes, hipMemcpyDeviceToHost, g->own));
  HIP_TRY(hipStreamSynchronize(g->own));
  if ((rc = group_wait(g, rays_out)) != RT_OK) return rc;
  rt_add_into(canvas, static_cast<const double*>(g->host_stage), bytes / sizeof(double));
  return RT_OK;
}

extern "C" int rt_group_render_accumulate(rt_group* group, const rt_render_params* p, uint32_t split, double* canvas,
                                          uint64_t* rays_out) {
  return rt_guard("rt_group_render_accumulate", [&] { return rt_group_render_accumulate_body(group, p, split, canvas, rays_out); });
}
