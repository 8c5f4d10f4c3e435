# timing-only A/B of kernel variants on ONE box: the libraries named in VIT_EXP_LIBS (build_ab/*.so, git-ignored) against
# each other -- update / chainback alone (scripts/time_update.py) and the pipeline line of bench.py
mkdir -p gpurun_out
: > gpurun_out/r3_exp.log
LIBS=${VIT_EXP_LIBS:-viterbidecodercpp_amd/libvit_hip.so}
CFG=${VIT_EXP_CFG:-2 SOFT16 65536 8192}
BENCH=${VIT_EXP_BENCH:-}
for rep in 1 2; do
for lib in $LIBS; do
VIT_HIP_LIB_PATH=$PWD/$lib python scripts/time_update.py $CFG 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_exp.log
if [ -n "$BENCH" ]; then
VIT_HIP_LIB_PATH=$PWD/$lib python bench.py $BENCH --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'bench', round(d['value']), d['unit'], 'ms', round(d['ms_per_step'], 3), 'median', round(d['ms_per_step_median'], 3), 'upd', round(d.get('update_ms', 0), 3), 'cb', round(d.get('chainback_ms', 0), 3), d['config'].get('schedule'))
" >> gpurun_out/r3_exp.log
fi
done
done
sed 's/.*repo\///' gpurun_out/r3_exp.log | sort
