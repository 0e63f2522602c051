#!/bin/bash
# Build libpolmux_hip.so for gfx950 (hipcc cross-compiles without a GPU).
# usage: scripts/build_lib.sh [--force]   (--force: recompile every source; the default recompiles what is older than its inputs)
set -e
cd "$(dirname "$0")/.."
mkdir -p polmux_amd/lib build
[ "$1" = "--force" ] && rm -f build/*.o
OBJS=""
for f in polmux_amd/csrc/*.hip; do
  o=build/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ polmux_amd/csrc/plx_common.h -nt $o ] || [ polmux_amd/csrc/plx_fft.h -nt $o ] || [ include/polmux_hip.h -nt $o ] || [ polmux_amd/csrc/plx_internal.h -nt $o ] || [ polmux_amd/csrc/plx_gateway.h -nt $o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $PLX_EXTRA_HIPCC_FLAGS -c $f -o $o &
  fi
  OBJS="$OBJS $o"
done
wait
for o in $OBJS; do [ -f $o ] || { echo "build_lib.sh: $o was not built" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip.so
echo polmux_amd/lib/libpolmux_hip.so
