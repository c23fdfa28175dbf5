#!/bin/bash
# PMC passes over the metric kernel (bench.py, default = tiled layout); one counter group per run
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_b_$i
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_b_$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu $BENCH_ARGS > "$GRAFT_REPO_ROOT/gpurun_out/pmc_b_$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/pmc_b_$i.log"; }
  cd "$GRAFT_REPO_ROOT"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_b_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "defcorr_" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in acc.items():
        print(c, "%.5g" % (sum(v) / len(v)), "(%d launches)" % len(v))
PY
