# A/B timing of kernel variants on ONE box: build_ab/*.so (earlier builds) against the tree's library
mkdir -p gpurun_out
: > gpurun_out/r3_ab.log
for rep in 1 2 3; do
for cfg in "2 SOFT16 65536 8192" "2 HARD8 32768 8192" "5 SOFT16 65536 8192" "3 SOFT16 65536 4096"; do
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/$lib python scripts/time_update.py $cfg 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_ab.log
done
done
done
sed 's/.*repo\///' gpurun_out/r3_ab.log | sort
