#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/bench_torchrun_w1.json 2> gpurun_out/bench_torchrun_w1.err
echo rc=$?
tail -3 gpurun_out/bench_torchrun_w1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_torchrun_w1.json").read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], [k for k in d.keys()])
for k in ("exchange", "config5_backend_error"):
    if k in d: print(k, d[k])
print(d.get("extra", {}).get("config5_backend_n1", {}).get("phases_ms_max_over_ranks"))
PY
