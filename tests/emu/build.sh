#!/bin/bash
# Build the emulated (CPU, test-only) copy of the kernels.  SAN=1 adds ASan+UBSan.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/emu/_build
FLAGS="-std=c++20 -O1 -g -DPLX_EMU -Itests/emu -fPIC -shared"
OUT=tests/emu/_build/libpolmux_emu.so
if [ "$SAN" = "1" ]; then FLAGS="$FLAGS -fsanitize=address,undefined -fno-omit-frame-pointer"; OUT=tests/emu/_build/libpolmux_emu_san.so; fi
SRCS=""
for f in polmux_amd/csrc/*.hip; do SRCS="$SRCS -x c++ $f"; done
g++ $FLAGS $SRCS -x c++ tests/emu/hip_emu.cpp -o $OUT -lpthread
echo $OUT
