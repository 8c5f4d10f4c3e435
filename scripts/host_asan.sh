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
# the kernel-descriptor reader (csrc/kernel_desc.hpp: ELF / offload-bundle parsing of files on disk) under the same sanitizers, on
# the library itself and on 300 corrupted / truncated copies of it and of a hipcc --genco bundle: no report, exit code 0 or 1 only
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o "$OUT/kd_dump_asan" tests/cpp/kd_dump.cpp -ldl
printf '#include <hip/hip_runtime.h>\nextern "C" __global__ void k(int* p) { p[threadIdx.x] = 1; }\n' > "$OUT/k.hip"
$HIPCC -O3 --offload-arch=gfx950 --genco -o "$OUT/k.hsaco" "$OUT/k.hip" > /dev/null 2>&1
"$OUT/kd_dump_asan" viterbidecodercpp_amd/libvit_hip.so | wc -l
python3 - "$OUT" <<'PY'
import random, subprocess, sys
out = sys.argv[1]
random.seed(1)
srcs = [open(out + "/k.hsaco", "rb").read(), open("viterbidecodercpp_amd/libvit_hip.so", "rb").read()[:400000]]
bad = 0
for trial in range(300):
    b = bytearray(srcs[trial % 2])
    if trial % 3 == 0:
        for _ in range(random.randint(1, 40)):
            b[random.randrange(len(b))] = random.randrange(256)
    elif trial % 3 == 1:
        b = b[:random.randrange(1, len(b))]
    else:
        for _ in range(20):
            b[random.randrange(min(len(b), 4608))] = random.randrange(256)
    open(out + "/fz.bin", "wb").write(b)
    p = subprocess.run([out + "/kd_dump_asan", out + "/fz.bin"], capture_output=True, text=True)
    if "AddressSanitizer" in p.stderr or "runtime error" in p.stderr or p.returncode not in (0, 1):
        bad += 1
        print("trial", trial, p.returncode, p.stderr[:300])
print("kernel_desc fuzz: 300 inputs,", bad, "bad")
sys.exit(1 if bad else 0)
PY
