#!/bin/bash
# PMC passes over the cooperative low-memory kernel (bench.py --workload lowmem, BASELINE config 4); one counter group per
# run, kernel-trace only beside it.  Averages per launch of the lowmem_coop kernel.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_co_$i
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_co_$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload lowmem --edges 16 --no-cpu --steps 10 --warmup 2 --blocks 1 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_co_$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/pmc_co_$i.log"; }
  cd "$GRAFT_REPO_ROOT"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_co_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "lowmem" not in r["Kernel_Name"]: continue
        acc.setdefault((r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        print("%-60s %-32s launches %3d  mean %.6g" % (k, c, len(v), sum(v) / len(v)))
PY
