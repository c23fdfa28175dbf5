#!/bin/bash
# run 4: A/B of the register-gather kernel (variant 0) vs the LDS-DMA staged kernel (variant 1) vs generic (2)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_gpu.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
for v in 0 1 2; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu --variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/bench_v$v.json'));print('variant',$v,'value',round(d['value'],1),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],4))"
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu --variant $v --probe > gpurun_out/bench_v${v}_probe.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/bench_v${v}_probe.json'));print('variant',$v,'probe value',round(d['value'],1),'ms',round(d['ms_per_step'],4))"
  [ $v -eq 1 ] && break
done
cd /tmp
for v in 0; do
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch_v$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu --variant $v > "$GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log" 2>&1 || { echo rocprof fetch failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_write_v$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu --variant $v > "$GRAFT_REPO_ROOT/gpurun_out/prof_write.log" 2>&1 || { echo rocprof write failed; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sq_v$v" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 2 --no-cpu --variant $v > "$GRAFT_REPO_ROOT/gpurun_out/prof_sq.log" 2>&1 || { echo rocprof sq failed; tail -5 "$GRAFT_REPO_ROOT/gpurun_out/prof_sq.log"; }
done
echo ALL_DONE
