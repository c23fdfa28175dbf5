#!/bin/bash
# PMC passes over the low-memory path (tools/prof_lowmem.py); one counter group per run (no tracing domains besides kernel-trace)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_lm_$i
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_lm_$i" -- python3 "$GRAFT_REPO_ROOT/tools/prof_lowmem.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_lm_$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/pmc_lm_$i.log"; }
  cd "$GRAFT_REPO_ROOT"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_lm_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "lowmem_mfma" not in n: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in acc.items():
        # 5 launches per level, 4 levels, in order
        per = [sum(v[i*5:(i+1)*5]) / 5 for i in range(len(v) // 5)]
        print(c, " ".join("%.4g" % x for x in per))
PY
