/* render_loop.c — the reference's main loop (src/main.rs:1077-1112) against the C ABI, in plain
 * C99: build the scene once, keep one accumulation buffer, call render `its` times, show
 * buffer / its quantised to 8 bits (here: write it as a PNG).
 *
 *   gcc -std=c99 -Iinclude examples/render_loop.c -Lraytracer_amd -lrt_amd -Wl,-rpath,$PWD/raytracer_amd -o render_loop
 *   ./render_loop scenes/spot.scene out.png [its]
 *
 * This is the host-buffer form (the canvas crosses PCIe both ways every call); a frame loop that
 * wants the last few per cent keeps the canvas on the device with rt_render_device and
 * RT_FLAG_PIPELINE, as raytracer_amd/csrc/rt_render_main.cpp does. */
#include <stdio.h>
#include <stdlib.h>

#include "rt_amd.h"

static int fail(const char* what) {
  fprintf(stderr, "render_loop: %s: %s\n", what, rt_last_error());
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: render_loop SCENE_FILE OUT.png [its]\n");
    return 2;
  }
  rt_objects* objects = NULL;
  rt_render_params params;
  uint32_t its = 1;
  if (rt_scene_file_load(argv[1], &objects, &params, &its) != RT_OK) return fail("rt_scene_file_load");
  if (argc > 3) its = (uint32_t)strtoul(argv[3], NULL, 10);
  if (its == 0) its = 1;

  rt_scene* scene = NULL; /* let scene = Scene::from_objects(objects), src/main.rs:1077 */
  if (rt_scene_from_objects(objects, &scene) != RT_OK) return fail("rt_scene_from_objects");
  rt_objects_destroy(objects);

  const size_t n = (size_t)params.width * params.height;
  double* buffer = calloc(n * 3, sizeof(double)); /* src/main.rs:1079 */
  uint8_t* rgba = malloc(n * 4);
  if (!buffer || !rgba) return 1;

  uint64_t rays = 0, total = 0;
  for (uint32_t it = 0; it < its; ++it) { /* one render event per iteration, src/main.rs:1099-1100 */
    params.first_sample = it * params.spp;
    if (rt_render_accumulate(scene, &params, buffer, &rays) != RT_OK) return fail("rt_render_accumulate");
    total += rays;
  }
  if (rt_quantize_rgba8(buffer, params.width, params.height, its, rgba) != RT_OK) return fail("rt_quantize_rgba8");
  if (rt_image_write_png(argv[2], rgba, params.width, params.height) != RT_OK) return fail("rt_image_write_png");
  printf("%ux%u, %u x %u spp, %llu rays -> %s\n", params.width, params.height, its, params.spp, (unsigned long long)total, argv[2]);
  free(rgba);
  free(buffer);
  rt_scene_destroy(scene);
  return 0;
}
