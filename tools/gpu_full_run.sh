#!/bin/bash
# run 7: full GPU suite, final-form bench (default + probe), rocprof kernel trace + PMC passes of the same command
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json | cut -c1-1500
timeout -k 10 300 python bench.py --no-cpu --probe > gpurun_out/bench_probe.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/bench_probe.json'));print('probe value',round(d['value'],1),'ms',round(d['ms_per_step'],4))"
rm -rf gpurun_out/prof_*
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log" 2>&1 || { echo rocprof trace failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1 || { echo rocprof fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1 || { echo rocprof write failed; exit 1; }
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python tools/compare_ref.py > gpurun_out/compare_ref.jsonl 2> gpurun_out/compare_ref.err || { tail -20 gpurun_out/compare_ref.err; exit 1; }
cat gpurun_out/compare_ref.jsonl | cut -c1-400
echo ALL_DONE
