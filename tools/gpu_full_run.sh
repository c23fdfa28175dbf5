#!/bin/bash
# Full GPU evidence run: GPU test suite, smoke, the default bench line (cold-cache headline + extra), the lowmem and
# backend workloads, interleaved A/B of the metric-kernel variants, rocprofv3 kernel-trace + PMC passes of the bench
# command in cold mode for both layouts, SQ/TCP counters, low-memory path kernel trace, comparison with the reference's
# own kernels.  Raw output under gpurun_out/; tools/collect_profiles.py turns it into the tracked summaries under profiles/.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
timeout -k 10 500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cut -c1-1500 gpurun_out/bench.json
timeout -k 10 300 python bench.py --workload lowmem --edges 16 > gpurun_out/bench_lowmem.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --workload backend --steps 16 --warmup 8 > gpurun_out/bench_backend.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --workload backend --steps 16 --warmup 8 --chunk-loop > gpurun_out/bench_backend_loop.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python bench.py --workload backend --steps 16 --warmup 8 --ba-split > gpurun_out/bench_backend_split.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python tools/ab_cold.py 0,7,6 6 tiled,rowmajor 0,1 > gpurun_out/ab_final.jsonl 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/ab_final.jsonl
timeout -k 10 300 python tools/ab_lowmem.py 1,1:c,2:c > gpurun_out/ab_lowmem.jsonl 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 300 python tools/e2e_calls.py > gpurun_out/e2e_calls.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
{ timeout -k 10 200 python tools/prof_init.py 20 f32 && timeout -k 10 200 python tools/prof_init.py 20 half && LGU_FUSED_BUILD_HALF=1 timeout -k 10 200 python tools/prof_init.py 20 half | sed 's/half=True/half=True (LGU_FUSED_BUILD_HALF=1)/'; } 2> gpurun_out/bench.err | grep "CorrBlock.__init__" > gpurun_out/prof_init.txt || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/prof_init.txt
timeout -k 10 200 python tools/ab_volbuild.py 20 > gpurun_out/ab_volbuild.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
timeout -k 10 200 python tools/prof_altcall.py 2> gpurun_out/bench.err | grep "per call" > gpurun_out/prof_altcall.txt || { tail -5 gpurun_out/bench.err; exit 1; }
cat gpurun_out/prof_altcall.txt
python -c "import json
for n in ('bench_lowmem','bench_backend','bench_backend_loop','bench_backend_split'):
    d=json.load(open('gpurun_out/%s.json'%n)); print(n,'value',round(d['value'],1),'ms',round(d['ms_per_step'],4), d.get('phases_ms_max_over_ranks'))"
rm -rf gpurun_out/prof_trace* gpurun_out/prof_fetch* gpurun_out/prof_write* gpurun_out/prof_lm
cd /tmp
for lay in tiled rowmajor; do
  sfx=""; [ $lay = rowmajor ] && sfx="_rm"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace$sfx" -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu --no-extra --no-backend --cache cold --layout $lay > "$R/gpurun_out/prof_trace$sfx.log" 2>&1 || { echo rocprof trace $lay failed; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_fetch$sfx" -- python3 "$R/bench.py" --steps 20 --warmup 2 --no-cpu --no-extra --no-backend --cache cold --layout $lay > "$R/gpurun_out/prof_fetch$sfx.log" 2>&1 || { echo rocprof fetch $lay failed; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/prof_write$sfx" -- python3 "$R/bench.py" --steps 20 --warmup 2 --no-cpu --no-extra --no-backend --cache cold --layout $lay > "$R/gpurun_out/prof_write$sfx.log" 2>&1 || { echo rocprof write $lay failed; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_lm" -- python3 "$R/tools/prof_lowmem.py" > "$R/gpurun_out/prof_lm.log" 2>&1 || { echo rocprof lowmem failed; exit 1; }
cd "$R"
BENCH_ARGS="--no-extra --no-backend --cache cold" bash tools/run_pmc_bench.sh > gpurun_out/pmc_cold.txt 2>&1
cat gpurun_out/pmc_cold.txt
timeout -k 10 600 python tools/compare_ref.py > gpurun_out/compare_ref.jsonl 2> gpurun_out/compare_ref.err || { tail -20 gpurun_out/compare_ref.err; exit 1; }
cut -c1-400 gpurun_out/compare_ref.jsonl
echo ALL_DONE
