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
# the library itself and on 450 corrupted / truncated copies of it, of a hipcc --genco bundle and of a whole small host ELF (section-header attacks included): no report, exit code 0 or 1 only
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o "$OUT/kd_dump_asan" tests/cpp/kd_dump.cpp -ldl
printf '#include <hip/hip_runtime.h>\nextern "C" __global__ void k(int* p) { p[threadIdx.x] = 1; }\n' > "$OUT/k.hip"
$HIPCC -O3 --offload-arch=gfx950 --genco -o "$OUT/k.hsaco" "$OUT/k.hip" > /dev/null 2>&1
$HIPCC -O3 --offload-arch=gfx950 -shared -fPIC -o "$OUT/libk.so" "$OUT/k.hip" > /dev/null 2>&1
"$OUT/kd_dump_asan" "$OUT/libk.so" | wc -l
"$OUT/kd_dump_asan" viterbidecodercpp_amd/libvit_hip.so | wc -l
python3 - "$OUT" <<'PY'
import random, subprocess, sys
out = sys.argv[1]
random.seed(1)
# a WHOLE small host ELF with a .hip_fatbin section (section headers included: the first 400000 bytes of libvit_hip.so have none,
# so every trial on them stopped at the first range check) + the bundle + the head of the library
srcs = [open(out + "/k.hsaco", "rb").read(), open(out + "/libk.so", "rb").read(), open("viterbidecodercpp_amd/libvit_hip.so", "rb").read()[:400000]]
import struct
elf = srcs[1]
shoff, shentsize, shnum = struct.unpack_from("<Q", elf, 0x28)[0], struct.unpack_from("<H", elf, 0x3A)[0], struct.unpack_from("<H", elf, 0x3C)[0]
bad = 0
for trial in range(450):
    b = bytearray(srcs[trial % 3])
    if trial % 3 == 1 and trial % 2 == 0:
        # section-header attacks on the host ELF: turn sections into SHT_NOBITS with wild offsets / sizes, break links
        for _ in range(random.randint(1, 6)):
            o = shoff + random.randrange(shnum) * shentsize
            what = random.randrange(4)
            if what == 0: struct.pack_into("<I", b, o + 4, 8)
            elif what == 1: struct.pack_into("<Q", b, o + 24, random.getrandbits(random.choice((20, 40, 63))))
            elif what == 2: struct.pack_into("<Q", b, o + 32, random.getrandbits(random.choice((20, 40, 63))))
            else: struct.pack_into("<I", b, o + 40, random.randrange(2 * shnum))
    elif trial % 3 == 0:
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
print("kernel_desc fuzz: 450 inputs,", bad, "bad")
sys.exit(1 if bad else 0)
PY
