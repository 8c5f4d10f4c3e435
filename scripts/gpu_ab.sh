# A/B timing of kernel variants on ONE box: build_ab/libvit_hip_base.so (an earlier commit's build) against the tree's library
mkdir -p gpurun_out
: > gpurun_out/r3_ab.log
for rep in 1 2 3; do
for cfg in "2 SOFT16 65536 8192" "2 HARD8 32768 8192" "3 SOFT16 65536 4096" "4 SOFT16 32768 4096"; do
VIT_HIP_LIB_PATH=$PWD/build_ab/libvit_hip_base.so python scripts/time_update.py $cfg 7 >> gpurun_out/r3_ab.log 2>&1
python scripts/time_update.py $cfg 7 >> gpurun_out/r3_ab.log 2>&1
done
done
cat gpurun_out/r3_ab.log
