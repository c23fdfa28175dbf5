#!/bin/bash
# run 3: parity tests of the lattice/probe kernel, goldens (small), bench (+probe), compare vs reference, profile
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
rm -rf gpurun_out/golden
timeout -k 10 600 python oracle/gen_golden.py gpurun_out/golden > gpurun_out/gen_golden.log 2>&1 || { tail -20 gpurun_out/gen_golden.log; exit 1; }
tail -1 gpurun_out/gen_golden.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu > gpurun_out/bench_noprobe.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench_noprobe.json | cut -c1-400
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu --probe > gpurun_out/bench_probe.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench_probe.json | cut -c1-400
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu --variant 2 > gpurun_out/bench_v2.json 2> gpurun_out/bench.err || exit 1
timeout -k 10 600 python tools/compare_ref.py > gpurun_out/compare_ref.jsonl 2> gpurun_out/compare_ref.err || { tail -20 gpurun_out/compare_ref.err; exit 1; }
cat gpurun_out/compare_ref.jsonl
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_trace.log" 2>&1 || { echo rocprof trace failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1 || { echo rocprof fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1 || { echo rocprof write failed; exit 1; }
echo ALL_DONE
