#!/bin/bash
# kernel breakdown of CorrBlock.__init__ at 20 edges: fp32 maps, half maps (library GEMM), half maps (matrix-core build)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in f32 half halfbuild; do
  arg=$mode; export LGU_FUSED_BUILD_HALF=0
  if [ $mode = halfbuild ]; then arg=half; export LGU_FUSED_BUILD_HALF=1; fi
  rm -rf gpurun_out/prof_init_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_init_$mode -- python3 tools/prof_init.py 20 $arg > gpurun_out/prof_init_$mode.log 2>&1 || { tail -5 gpurun_out/prof_init_$mode.log; exit 1; }
  echo "== $mode: $(grep CorrBlock.__init__ gpurun_out/prof_init_$mode.log)"
  f=$(ls gpurun_out/prof_init_$mode/*/*_kernel_stats.csv | tail -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("   %-90s calls %4s  avg %9.1f us  total %8.3f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
