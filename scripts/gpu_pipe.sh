mkdir -p gpurun_out
for rep in 1 2 3 4; do
for prio in 0 1; do
VIT_HIP_PIPELINE_CB_PRIO=$prio python bench.py --config 2 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('config 2 cb_prio=$prio', round(r['value']), r['ms_per_step'], r['ms_per_step_median'], r['update_ms'], r['chainback_ms'])"
done
done
