#!/bin/bash
# run 5: tile-staged lowMem kernel parity + timing vs reference
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python tools/compare_ref.py > gpurun_out/compare_ref.jsonl 2> gpurun_out/compare_ref.err || { tail -20 gpurun_out/compare_ref.err; exit 1; }
cat gpurun_out/compare_ref.jsonl
echo ALL_DONE
