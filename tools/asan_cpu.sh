#!/bin/bash
# AddressSanitizer over the HOST code only (GPU ASan is not available on this pool): rebuilds the three C files with
# -fsanitize=address into build_ab/asan/, links them with the regular kernel object, and runs the CPU test-suite on it.
set -e
cd "$(dirname "$0")/.."
make -C cpecan_amd/csrc > /dev/null
mkdir -p build_ab/asan
for f in cpecan_host cpecan_dropin cpecan_realign; do
    gcc -O1 -g -std=c99 -fPIC -w -ffp-contract=off -fopenmp -fsanitize=address -fno-omit-frame-pointer \
        -Iinclude -Icpecan_amd/csrc -c -o build_ab/asan/$f.o cpecan_amd/csrc/$f.c
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/asan/libcpecan_hip.so build_ab/asan/cpecan_host.o \
    build_ab/asan/cpecan_dropin.o build_ab/asan/cpecan_realign.o cpecan_amd/csrc/cpecan_kernels.o -lm -lgomp
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 CPECAN_LIB=$PWD/build_ab/asan/libcpecan_hip.so \
    python -m pytest tests/test_realign_cpu.py tests/test_abi.py -x -q
