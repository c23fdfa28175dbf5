#!/bin/bash
# first GPU validation run: parity tests, smoke, bench (variants), rocprof kernel trace + PMC
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -30 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -3 || exit 1
for v in 0 1 3 2; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu --variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err || { echo "bench variant $v failed"; tail -5 gpurun_out/bench_v$v.err; exit 1; }
  cat gpurun_out/bench_v$v.json
done
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log" 2>&1 || { echo rocprof trace failed; tail -5 "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1 || { echo rocprof fetch failed; tail -5 "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1 || { echo rocprof write failed; exit 1; }
find "$GRAFT_REPO_ROOT/gpurun_out" -name "*.csv" | head -30
echo ALL_DONE
