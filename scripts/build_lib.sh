#!/bin/bash
# Build libpolmux_hip.so for gfx950 (hipcc cross-compiles without a GPU): one object per source, compiled in parallel.
# usage: scripts/build_lib.sh [--force]   (--force: recompile every source; the default recompiles what is older than its
#        source or than ANY header of the library)
# A source that fails to compile fails the build: its object is written to <name>.o.tmp and only renamed on success, every
# compiler's exit status is collected, and nothing is linked unless all of them succeeded (a stale object never gets linked).
set -u
cd "$(dirname "$0")/.."
mkdir -p polmux_amd/lib build
[ "${1:-}" = "--force" ] && rm -f build/*.o
rm -f build/*.o.tmp
HDRS="polmux_amd/csrc/*.h include/polmux_hip.h"
# objects whose source no longer exists (a file was renamed or split) must not be linked
for o in build/*.o; do [ -e "$o" ] || continue; [ -f polmux_amd/csrc/$(basename ${o%.o}).hip ] || rm -f $o; done
OBJS=""; PIDS=""; NAMES=""
for f in polmux_amd/csrc/*.hip; do
  o=build/$(basename ${f%.hip}).o
  stale=0
  [ -f $o ] || stale=1
  [ $stale = 0 ] && [ $f -nt $o ] && stale=1
  if [ $stale = 0 ]; then for h in $HDRS; do [ $h -nt $o ] && stale=1; done; fi
  if [ $stale = 1 ]; then
    rm -f $o
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result ${PLX_EXTRA_HIPCC_FLAGS:-} -c $f -o $o.tmp && mv $o.tmp $o ) &
    PIDS="$PIDS $!"; NAMES="$NAMES $f"
  fi
  OBJS="$OBJS $o"
done
fail=0; i=1
for pid in $PIDS; do
  if ! wait $pid; then echo "build_lib.sh: $(echo $NAMES | cut -d' ' -f$i) failed to compile" >&2; fail=1; fi
  i=$((i + 1))
done
[ $fail = 0 ] || { rm -f build/*.o.tmp; exit 1; }
for o in $OBJS; do [ -f $o ] || { echo "build_lib.sh: $o was not built" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o polmux_amd/lib/libpolmux_hip.so || exit 1
echo polmux_amd/lib/libpolmux_hip.so
