#!/bin/bash
# round 5, call u: where the one-wavefront PLAN_LDS2 codes (K = 10, 11) lose their time -- PMC passes over one pipeline run each
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for K in 11 12; do
  if [ $K = 11 ]; then G=1845,1995; else G=2787,3645; fi
  for pass in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    set -- $pass; name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $REPO/gpurun_out/pmc_k$K/$name --output-format csv -- python3 $REPO/scripts/time_pipeline.py $K 2 $G SOFT16 16384 2048 3 > $REPO/gpurun_out/pmc_k${K}_$name.log 2>&1
    echo "K$K $name rc=$?"
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for K in (11, 12):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmc_k{K}/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "lds2_update" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); 
    print("K", K, {k: f"{v['v']:.4g}" for k, v in agg.items()})
    a = {k: v["v"] for k, v in agg.items()}
    if a.get("SQ_WAVE_CYCLES"):
        print("  VALU active / wave cycles", a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], "wait_any", a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], "wait_inst_any", a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
              "busy/wavecycles", a["SQ_BUSY_CYCLES"] / a["SQ_WAVE_CYCLES"], "VALU per wave", a["SQ_INSTS_VALU"] / a["SQ_WAVES"], "SALU per wave", a["SQ_INSTS_SALU"] / a["SQ_WAVES"])
PY
