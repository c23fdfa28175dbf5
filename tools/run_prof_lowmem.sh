#!/bin/bash
# rocprofv3 kernel trace of the low-memory path (tools/prof_lowmem.py); summary lands in gpurun_out/prof_lm
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lm
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_lm" -- python3 "$GRAFT_REPO_ROOT/tools/prof_lowmem.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_lm.log" 2>&1 || { tail -5 "$GRAFT_REPO_ROOT/gpurun_out/prof_lm.log"; exit 1; }
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_lm/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
per = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "lowmem" not in n: continue
    per.setdefault(n[:60], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in per.items():
    print(n, len(v), " ".join("%.1f" % x for x in v))
PY
