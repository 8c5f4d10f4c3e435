#!/bin/bash
# scripts/host_asan.sh -- the HOST side of libvit_hip.so (argument checks, blob parser, plan selection, error paths) under
# AddressSanitizer + UBSan, CPU only: GPU sanitizers are not available on the pool (-fno-gpu-sanitize keeps the device code
# as it is).  Runs tests/test_host.py (no GPU needed) against the instrumented build.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/vit_hip_asan}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p "$OUT"
make -C "$REPO/viterbidecodercpp_amd/csrc" -j8 > "$OUT/make.log" 2>&1          # the PLAN_REG launch objects, uninstrumented
cd "$REPO/viterbidecodercpp_amd/csrc"
$HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -c -o "$OUT/vit_hip_asan.o" vit_hip.hip > "$OUT/compile.log" 2>&1
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o "$OUT/libvit_hip_asan.so" "$OUT/vit_hip_asan.o" build/reg_inst_*.o -ldl > "$OUT/link.log" 2>&1
cd "$REPO"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT VIT_HIP_LIB_PATH=$OUT/libvit_hip_asan.so \
    python -m pytest tests/test_host.py -x -q
