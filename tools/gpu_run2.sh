#!/bin/bash
# run 2: golden vectors from the reference build, full-size comparison vs reference kernels, FETCH_SIZE calibration
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
ls -la oracle/_ref/ || exit 1
timeout -k 10 600 python oracle/gen_golden.py gpurun_out/golden > gpurun_out/gen_golden.log 2>&1 || { tail -20 gpurun_out/gen_golden.log; exit 1; }
tail -3 gpurun_out/gen_golden.log
timeout -k 10 600 python tools/compare_ref.py > gpurun_out/compare_ref.jsonl 2> gpurun_out/compare_ref.err || { tail -20 gpurun_out/compare_ref.err; exit 1; }
cat gpurun_out/compare_ref.jsonl
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/calib_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/calib_fetch.py" > "$GRAFT_REPO_ROOT/gpurun_out/calib_fetch.log" 2>&1 || { tail -5 "$GRAFT_REPO_ROOT/gpurun_out/calib_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/calib_write" -- python3 "$GRAFT_REPO_ROOT/tools/calib_fetch.py" > "$GRAFT_REPO_ROOT/gpurun_out/calib_write.log" 2>&1 || exit 1
tail -3 "$GRAFT_REPO_ROOT/gpurun_out/calib_fetch.log"
echo ALL_DONE
