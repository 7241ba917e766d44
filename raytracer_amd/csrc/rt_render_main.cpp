// rt_render — command-line counterpart of the reference's main() (src/main.rs:1054-1128) without
// the window: build the scene once (src/main.rs:1077), keep one accumulation buffer
// (src/main.rs:1079), call render `its` times, each call adding the mean of `spp` fresh samples
// per pixel (src/main.rs:1099-1100), then show canvas / its quantised to 8 bits
// (src/main.rs:1106-1112) — here written as a PNG file instead of blitted into a window.
//
//   rt_render SCENE_FILE [-o OUT.png] [--width W] [--height H] [--spp S] [--its N] [--seed K]
//             [--max-depth D] [--device I | --devices I,J,... [--split samples|tiles]] [--progress]
//             [--device-tree]
//
// Options override the scene file's `render` statement. --progress rewrites OUT.png after
// every call, as the reference's window refreshes after every render event. --devices renders
// every frame on a group of GPUs (one member per listed device, the first combines the frame).
// --device-tree builds the tree on the GPU instead of the host (large scenes: a tenth of the build
// time, frames somewhat slower).

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace {

int die(const char* what) {
  std::fprintf(stderr, "rt_render: %s: %s\n", what, rt_last_error());
  return 1;
}

int write_frame(const char* path, const double* dev_canvas, std::vector<double>& host, std::vector<uint8_t>& rgba,
                uint32_t w, uint32_t h, uint32_t its, hipStream_t stream) {
  if (hipMemcpyAsync(host.data(), dev_canvas, host.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    std::fprintf(stderr, "rt_render: copying the canvas back failed\n");
    return 1;
  }
  if (rt_quantize_rgba8(host.data(), w, h, its, rgba.data()) != RT_OK) return die("rt_quantize_rgba8");
  if (rt_image_write_png(path, rgba.data(), w, h) != RT_OK) return die("rt_image_write_png");
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const char* scene_path = nullptr;
  std::string out_path = "render.png";
  long long o_w = -1, o_h = -1, o_spp = -1, o_its = -1, o_depth = -1, device = 0;
  unsigned long long o_seed = 0;
  bool has_seed = false, progress = false, device_tree = false;
  std::vector<int> devices;
  uint32_t split = RT_SPLIT_SAMPLES;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&](long long* dst) {
      if (i + 1 >= argc) return false;
      char* end = nullptr;
      *dst = std::strtoll(argv[++i], &end, 10);
      return end && *end == '\0' && *dst >= 0;
    };
    bool ok = true;
    if (a == "--help" || a == "-h") {
      std::printf("usage: rt_render SCENE_FILE [-o OUT.png] [--width W] [--height H] [--spp S] [--its N] [--seed K] [--max-depth D]\n"
                  "                 [--device I | --devices I,J,... [--split samples|tiles]] [--device-tree] [--progress]\n"
                  "Renders a scene description file (grammar: include/rt_amd.h, examples: scenes/) on the GPU: N calls of S\n"
                  "samples per pixel accumulate into one canvas, whose mean is written as a PNG. Options override the scene\n"
                  "file's render statement.\n");
      return 0;
    }
    if (a == "-o" && i + 1 < argc) out_path = argv[++i];
    else if (a == "--width") ok = value(&o_w);
    else if (a == "--height") ok = value(&o_h);
    else if (a == "--spp") ok = value(&o_spp);
    else if (a == "--its") ok = value(&o_its);
    else if (a == "--max-depth") ok = value(&o_depth);
    else if (a == "--device") ok = value(&device);
    else if (a == "--seed" && i + 1 < argc) { o_seed = std::strtoull(argv[++i], nullptr, 10); has_seed = true; }
    else if (a == "--progress") progress = true;
    else if (a == "--device-tree") device_tree = true;
    else if (a == "--devices" && i + 1 < argc) {
      for (const char* c = argv[++i]; ok && *c;) {
        char* end = nullptr;
        const long v = std::strtol(c, &end, 10);
        ok = end != c && v >= 0 && (*end == ',' || *end == '\0');
        devices.push_back((int)v);
        c = *end ? end + 1 : end;
      }
      ok = ok && !devices.empty();
    } else if (a == "--split" && i + 1 < argc) {
      const std::string v = argv[++i];
      ok = v == "samples" || v == "tiles";
      split = v == "tiles" ? RT_SPLIT_TILES : RT_SPLIT_SAMPLES;
    }
    else if (!a.empty() && a[0] != '-' && !scene_path) scene_path = argv[i];
    else ok = false;
    if (!ok) {
      std::fprintf(stderr, "usage: rt_render SCENE_FILE [-o OUT.png] [--width W] [--height H] [--spp S] [--its N] "
                           "[--seed K] [--max-depth D] [--device I | --devices I,J,... [--split samples|tiles]] [--progress] [--device-tree]\n");
      return 2;
    }
  }
  if (!scene_path) {
    std::fprintf(stderr, "rt_render: no scene file given\n");
    return 2;
  }

  rt_objects* objs = nullptr;
  rt_render_params p;
  uint32_t its = 1;
  if (rt_scene_file_load(scene_path, &objs, &p, &its) != RT_OK) return die("rt_scene_file_load");
  if (o_w >= 0) p.width = (uint32_t)o_w;
  if (o_h >= 0) p.height = (uint32_t)o_h;
  if (o_spp >= 0) p.spp = (uint32_t)o_spp;
  if (o_its >= 0) its = (uint32_t)o_its;
  if (o_depth >= 0) p.max_depth = (int32_t)o_depth;
  if (has_seed) p.seed = o_seed;
  if (its == 0 || p.width == 0 || p.height == 0) {
    std::fprintf(stderr, "rt_render: width, height and its must be positive\n");
    return 2;
  }

  const auto t_build = std::chrono::steady_clock::now();
  rt_scene* scene = nullptr;
  if (rt_scene_from_objects_with(objs, device_tree ? RT_TREE_NONE : RT_TREE_HOST, &scene) != RT_OK) return die("rt_scene_from_objects");
  rt_objects_destroy(objs);
  if (device_tree && rt_scene_rebuild_on_device(scene, devices.empty() ? (int)device : devices[0]) != RT_OK)
    return die("rt_scene_rebuild_on_device");
  rt_group* group = nullptr;
  if (!devices.empty()) {
    device = devices[0];
    if (rt_group_create(scene, devices.data(), (uint32_t)devices.size(), &group) != RT_OK) return die("rt_group_create");
  } else if (rt_scene_upload(scene, (int)device) != RT_OK) {
    return die("rt_scene_upload");
  }
  rt_scene_info info;
  rt_scene_get_info(scene, &info);
  const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
  std::fprintf(stderr, "rt_render: %llu triangles, %u spheres, %u tree nodes (built and uploaded in %.3f s)\n",
               (unsigned long long)info.n_triangles, info.n_spheres, info.bvh_nodes, build_s);

  if (hipSetDevice((int)device) != hipSuccess) {
    std::fprintf(stderr, "rt_render: hipSetDevice(%lld) failed\n", device);
    return 1;
  }
  const size_t n = (size_t)p.width * p.height * 3;
  double* canvas = nullptr;
  hipStream_t stream = nullptr;
  if (hipMalloc(&canvas, n * sizeof(double)) != hipSuccess || hipStreamCreate(&stream) != hipSuccess ||
      hipMemsetAsync(canvas, 0, n * sizeof(double), stream) != hipSuccess) {
    std::fprintf(stderr, "rt_render: cannot allocate the %ux%u canvas on device %lld\n", p.width, p.height, device);
    return 1;
  }
  std::vector<double> host(n);
  std::vector<uint8_t> rgba((size_t)p.width * p.height * 4);

  uint64_t rays = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t it = 0; it < its; ++it) {
    rt_render_params q = p;
    q.first_sample = p.first_sample + it * p.spp;  // fresh samples of each pixel's stream per call
    if (group) {  // frame follows frame: the members render the next one while this one is combined
      if (rt_group_render_device(group, &q, split, canvas, stream) != RT_OK) return die("rt_group_render_device");
      if (progress) {  // showing a frame ends the run
        uint64_t r = 0;
        if (rt_group_wait(group, &r) != RT_OK) return die("rt_group_wait");
        rays += r;
        if (write_frame(out_path.c_str(), canvas, host, rgba, p.width, p.height, it + 1, stream)) return 1;
      }
      continue;
    }
    // frame follows frame: let each call's last kernels overlap the next call (a pipelined run)
    q.flags |= RT_FLAG_PIPELINE;
    if (rt_render_device(scene, &q, canvas, stream) != RT_OK) return die("rt_render_device");
    if (progress) {  // showing a frame ends the run: wait, count its rays, write it
      rt_stats st;
      if (rt_scene_last_stats(scene, &st) != RT_OK) return die("rt_scene_last_stats");
      rays += st.rays;
      if (write_frame(out_path.c_str(), canvas, host, rgba, p.width, p.height, it + 1, stream)) return 1;
    }
  }
  if (!progress && group) {
    if (rt_group_wait(group, &rays) != RT_OK) return die("rt_group_wait");
  }
  if (!progress && !group) {  // waits for the run and reports its totals
    rt_stats st;
    if (rt_scene_last_stats(scene, &st) != RT_OK) return die("rt_scene_last_stats");
    rays = st.rays;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) {
    std::fprintf(stderr, "rt_render: the render stream failed\n");
    return 1;
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!progress && write_frame(out_path.c_str(), canvas, host, rgba, p.width, p.height, its, stream)) return 1;
  std::fprintf(stderr, "rt_render: %ux%u, %u x %u spp, %llu rays in %.3f s (%.1f Mrays/s) -> %s\n", p.width, p.height, its, p.spp,
               (unsigned long long)rays, secs, rays / secs / 1e6, out_path.c_str());
  if (group) {
    for (uint32_t i = 0; i < (uint32_t)devices.size(); ++i) {
      int dev = 0;
      uint64_t r = 0;
      double ms = 0.0;
      rt_group_member_info(group, i, &dev, &r, &ms);
      std::fprintf(stderr, "rt_render:   member %u on device %d: %llu rays since the last wait, its share of the last frame took %.2f ms\n", i, dev, (unsigned long long)r, ms);
    }
    rt_group_destroy(group);
  }
  (void)hipFree(canvas);
  (void)hipStreamDestroy(stream);
  rt_scene_destroy(scene);
  return 0;
}
