#!/bin/bash
# GPU evidence run of the low-memory path (second call beside tools/gpu_full_run.sh, which fills the 20-minute limit of a
# gpurun call on its own): interleaved A/B of the cooperative kernel's scheduling modes against the one-wave-per-block
# kernels, per-phase stamps (diagnostic build), PMC passes of both, kernel trace of the dense BA at backend size.
# Raw output under gpurun_out/; tools/collect_profiles.py copies the summaries into profiles/.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/ab_lowmem_coop.py ";LGU_LOWMEM_COOP_SPLIT=0;LGU_LOWMEM_COOP_SPLIT=1;LGU_LOWMEM_COOP=0" > gpurun_out/ab_lowmem_coop.jsonl 2> gpurun_out/lm.err || { tail -5 gpurun_out/lm.err; exit 1; }
cat gpurun_out/ab_lowmem_coop.jsonl
timeout -k 10 200 python tools/diag/run_co_stamps.py > gpurun_out/co_stamps.txt 2>&1 || { tail -5 gpurun_out/co_stamps.txt; exit 1; }
tail -5 gpurun_out/co_stamps.txt
{ echo "== cooperative kernel (csrc/lowmem_coop.hip), bench.py --workload lowmem =="; bash tools/run_pmc_coop.sh; echo "== one-wave-per-block kernels (csrc/lowmem_mfma.hip, LGU_LOWMEM_COOP=0) =="; LGU_DEBUG_KNOBS=1 LGU_LOWMEM_COOP=0 bash tools/run_pmc_coop.sh; } > gpurun_out/pmc_lowmem.txt 2>&1
cat gpurun_out/pmc_lowmem.txt
rm -rf gpurun_out/prof_ba
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_ba" -- python3 "$R/tools/prof_ba.py" > "$R/gpurun_out/prof_ba.log" 2>&1 || { echo rocprof ba failed; exit 1; }
cd "$R"
grep "ba call" gpurun_out/prof_ba.log | tail -1
f=$(find gpurun_out/prof_ba -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-200 > gpurun_out/ba_kernel_stats.csv
cat gpurun_out/ba_kernel_stats.csv
echo LOWMEM_DONE
