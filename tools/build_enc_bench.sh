#!/bin/sh
# host-only timing harness of the read-level stagers (tools/src/enc_bench.cpp)
set -e
cd "$(dirname "$0")/.."
make -C instrain_amd/csrc seg_encode.o obs_encode.o >/dev/null
mkdir -p tools/bin
g++ -O3 -std=c++17 -I include tools/src/enc_bench.cpp instrain_amd/csrc/seg_encode.o instrain_amd/csrc/obs_encode.o -lpthread -o tools/bin/enc_bench
