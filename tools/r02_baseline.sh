#!/bin/bash
# Round-2 baseline: cold-cache bench line with extras, then SQ/TCP/TCC counter passes in cold mode.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python bench.py --no-cpu > gpurun_out/r02_bench0.json 2> gpurun_out/r02_bench0.err || { tail -20 gpurun_out/r02_bench0.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench0.json'))
print('headline', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), d['roofline']['device_ms_per_step_blocks'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(k, round(v['device_ms_per_step'],4), round(v['Mpix_edges_per_s'],1), v.get('roofline_frac'))
    else: print(k, v)
PY
BENCH_ARGS="--no-extra" bash tools/run_pmc_bench.sh > gpurun_out/r02_pmc_cold.txt 2>&1
cat gpurun_out/r02_pmc_cold.txt
