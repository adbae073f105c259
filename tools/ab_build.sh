#!/bin/bash
# builds an alternative libcpecan_hip.so with extra -D flags into build_ab/<tag>.so (for A/B timing in ONE gpurun call:
# boxes differ by several % in clock, so variants are only comparable within a call).  usage: tools/ab_build.sh tag -DX=1 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../cpecan_amd/csrc"
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function -Wno-pass-failed -I../../include -I. "$@" -c -o /tmp/ab_$tag.o cpecan_kernels.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_ab/$tag.so cpecan_host.o cpecan_dropin.o cpecan_realign.o /tmp/ab_$tag.o -lm -lgomp -lpthread
echo built build_ab/$tag.so
