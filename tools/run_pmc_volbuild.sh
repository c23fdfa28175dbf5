#!/bin/bash
# PMC passes over the volume build kernels (tools/ab_volbuild.py, 20 edges of 48x64x128): HBM bytes and instruction mix per
# launch of volume_build_kernel<4,false> (fp32 maps) and <4,true> (half maps), and of the library path's post-processing kernel.
# One counter group per run, kernel-trace only beside it.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_vb_$i
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_vb_$i" -- python3 "$GRAFT_REPO_ROOT/tools/ab_volbuild.py" 20 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_vb_$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$GRAFT_REPO_ROOT/gpurun_out/pmc_vb_$i.log"; }
  cd "$GRAFT_REPO_ROOT"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_vb_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "volume_build" not in n and "volume_pyramid" not in n and "volume_pack" not in n: continue
        acc.setdefault((n.split("(")[0].replace("void ", "")[-48:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        print("%-48s %-26s launches %3d  mean %.6g" % (k, c, len(v), sum(v) / len(v)))
PY
