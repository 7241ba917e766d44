#!/bin/bash
# Builds the library with AddressSanitizer and UndefinedBehaviorSanitizer on the HOST code only
# (the device code is compiled as usual) and runs tools/host_sanitize.cpp against it. No GPU needed.
#   bash tools/run_host_sanitize.sh [rounds]
#   SAN_KIND=thread bash tools/run_host_sanitize.sh [rounds]    ThreadSanitizer instead (the builder,
#                                                               the flattening and the leaf records run on threads)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/gpurun_out/sanitize_build_${SAN_KIND:-address}
mkdir -p $B
if [ "${SAN_KIND:-address}" = thread ]; then
  SAN="-Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer"
else
  SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
fi
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
cd $R/raytracer_amd/csrc
for f in rt_host.cpp rt_image.cpp rt_scenefile.cpp; do [ $B/$f.o -nt $f ] || /opt/rocm/bin/hipcc $FLAGS $SAN -c -o $B/$f.o $f; done
for f in rt_device.hip rt_build.hip rt_group.hip; do [ $B/$f.o -nt $f ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $SAN -c -o $B/$f.o $f; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -shared -o $B/librt_amd.so $B/*.o -lz
/opt/rocm/bin/hipcc $FLAGS $SAN -o $B/host_sanitize $R/tools/host_sanitize.cpp -L$B -lrt_amd -Wl,-rpath,$B
ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 TSAN_OPTIONS=halt_on_error=1 \
  $B/host_sanitize $R ${1:-300}
