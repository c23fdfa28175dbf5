#!/bin/bash
# The driver's launch form at world size 1 with LGU_REHEARSE_COLLECTIVES=1: every collective of the sharded step (padded
# all-gathers of target / weight / damping, the all-reduces of sharded_ba_split, the replica check) is issued over RCCL on
# the one GPU of this pool.  1: the default line; 2: the backend workload with the BA split by edge owner.
cd "$GRAFT_REPO_ROOT"
export LGU_REHEARSE_COLLECTIVES=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra > gpurun_out/bench_torchrun_w1.json 2> gpurun_out/bench_torchrun_w1.err
echo rc=$?
tail -2 gpurun_out/bench_torchrun_w1.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload backend --steps 8 --warmup 4 --ba-split > gpurun_out/bench_torchrun_w1_split.json 2> gpurun_out/bench_torchrun_w1_split.err
echo rc=$?
tail -2 gpurun_out/bench_torchrun_w1_split.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_torchrun_w1.json").read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], d.get("exchange"))
c5 = d.get("extra", {}).get("config5_backend_n1") or d.get("strong_scaling_config5")
print(c5 and (c5.get("phases_ms_max_over_ranks"), c5.get("replicas_agree")))
s = json.loads(open("gpurun_out/bench_torchrun_w1_split.json").read().strip().splitlines()[-1])
print(s["ms_per_step"], s["phases_ms_max_over_ranks"], s["replicas_agree"])
PY
