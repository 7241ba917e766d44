#!/bin/bash
# Builds a variant of the library with extra compiler flags for the device file, beside the product:
#   tools/experiments/build_variant.sh <name> [hipcc flags...]  ->  exp_build/librt_<name>.so
# (RT_AMD_LIB=<path> makes the Python mirror load it). Diagnostic switches of the device code:
#   -DRT_STAMPS       cycle stamps per code section (tools/stamps.py)
#   -DRT_DEBUG_HIST   histogram of (hit object, cost of the ray that found the hit, cost of the ray
#                     scattered from it) over the launches over the queue (tools/experiments/hist.py)
#   -DRT_REJECT_CAP=1 rejection loops cut to one try (what their further tries cost; changes the image)
# Needs the product's object files (make -C raytracer_amd/csrc) for everything but the device file.
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
C=raytracer_amd/csrc
mkdir -p exp_build
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -I$C "$@" -Rpass-analysis=kernel-resource-usage -c -o exp_build/rt_device_$NAME.o $C/rt_device.hip 2> exp_build/rt_device_$NAME.resource.txt || { cat exp_build/rt_device_$NAME.resource.txt; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o exp_build/librt_$NAME.so $C/rt_host.o $C/rt_image.o $C/rt_scenefile.o exp_build/rt_device_$NAME.o $C/rt_build.o $C/rt_group.o -lz -lpthread
echo "exp_build/librt_$NAME.so"
