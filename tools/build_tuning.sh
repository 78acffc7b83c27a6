#!/bin/bash
# Tuning build of the library (ISX_TUNING: honours ISX_DEBUG_MODE / ISX_GRID / ISX_BLOCK) in a scratch copy of the sources, so the
# production objects and libinstrain_amd.so stay untouched:  tools/build_tuning.sh  ->  instrain_amd/libinstrain_amd_tuning$SUFFIX.so
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=${TMPDIR:-/tmp}/isx_tuning_build
rm -rf "$T" && mkdir -p "$T/a/b" && cp -r "$REPO/instrain_amd/csrc" "$T/a/b/csrc" && cp -r "$REPO/include" "$T/a/include"
cd "$T/a/b/csrc" && rm -f *.o
make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I../../include -DISX_TUNING $EXTRA_DEFS" \
     OUT="$REPO/instrain_amd/libinstrain_amd_tuning$SUFFIX.so" "$REPO/instrain_amd/libinstrain_amd_tuning$SUFFIX.so"
