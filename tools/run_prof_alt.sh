cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_alt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_alt" -- python3 "$GRAFT_REPO_ROOT/tools/prof_alt.py" 2>&1 | grep "AltCorr"
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_alt/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls %4s avg %8.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
