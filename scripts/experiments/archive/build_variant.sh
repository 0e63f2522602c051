#!/bin/bash
# dev: build polmux_amd/lib/libpolmux_hip_<name>.so from a sed-patched copy of plx_ssfm.hip:  build_variant.sh <name> '<sed script>' [extra hipcc flags]
set -e
cd "$(dirname "$0")/../.."
name=$1; script=$2; shift 2
mkdir -p build_abl
root=$(pwd)
sed -e "$script" -e "s|#include \"../../include/polmux_hip.h\"|#include \"$root/include/polmux_hip.h\"|" \
    -e "s|#include \"plx_fft.h\"|#include \"$root/polmux_amd/csrc/plx_fft.h\"|" -e "s|#include \"plx_internal.h\"|#include \"$root/polmux_amd/csrc/plx_internal.h\"|" \
    -e "s|#include \"plx_gateway.h\"|#include \"$root/polmux_amd/csrc/plx_gateway.h\"|" polmux_amd/csrc/plx_ssfm.hip > build_abl/plx_ssfm_$name.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I polmux_amd/csrc "$@" -c build_abl/plx_ssfm_$name.hip -o build_abl/plx_ssfm_$name.o
OBJS=""
for f in polmux_amd/csrc/*.hip; do o=build/$(basename ${f%.hip}).o; [ $(basename $f) = plx_ssfm.hip ] && o=build_abl/plx_ssfm_$name.o; OBJS="$OBJS $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip_$name.so
echo polmux_amd/lib/libpolmux_hip_$name.so
