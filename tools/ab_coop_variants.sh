#!/bin/bash
# A/B of build-time variants of the cooperative low-memory kernel (tools/build_variant.py -> build/ab/liblgu_<name>.so,
# loaded through LGU_LIB_PATH; "default" = the in-tree library): parity tests of the kernel once per variant, BASELINE
# config 4 timed at 16 and 64 edges in two passes, and one PMC pass of the instruction counters per variant.
#   bash tools/ab_coop_variants.sh default base ...      (PMC=0 skips the counter pass, TESTS=0 the parity tests)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export LGU_DEBUG_KNOBS=1 TMPDIR=/tmp
sel() { if [ "$1" = default ]; then unset LGU_LIB_PATH; else export LGU_LIB_PATH="$GRAFT_REPO_ROOT/build/ab/liblgu_$1.so"; [ -f "$LGU_LIB_PATH" ] || { echo "no library for $1"; exit 1; }; fi; }
for n in "$@"; do
  sel $n
  if [ "${TESTS:-1}" = 1 ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_glue_reference.py -m gpu -q -p no:cacheprovider -k 'coop or lowmem or call_many or config5 or offset_rows or altcorr' > gpurun_out/ab_tests_$n.log 2>&1
    echo "== $n tests: $(tail -1 gpurun_out/ab_tests_$n.log)"
    grep "^FAILED" gpurun_out/ab_tests_$n.log | cut -c1-200    # every failing case by name: a variant that fails parity is not a candidate
  fi
done
for pass in 1 2; do
  for n in "$@"; do
    sel $n
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 16 2>/dev/null | cut -c1-120) $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 64 2>/dev/null | cut -c1-120)"
  done
done
if [ "${PMC:-1}" = 1 ]; then
  for n in "$@"; do
    sel $n
    rm -rf gpurun_out/pmc_ab_$n
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$n" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload lowmem --edges 16 --no-cpu --steps 10 --warmup 2 --blocks 1 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$n.log" 2>&1 ) || { echo "pmc pass of $n failed"; tail -3 gpurun_out/pmc_ab_$n.log; }
    python3 - "$n" <<'PY'
import csv, glob, collections, sys
n = sys.argv[1]
fs = glob.glob("gpurun_out/pmc_ab_%s/**/*counter_collection.csv" % n, recursive=True)
acc = collections.OrderedDict()
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if "lowmem_coop" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {c: sum(v) / len(v) for c, v in acc.items()}
if m:
    w = m.get("SQ_WAVES", 1)
    print("== %s counters per launch: %s | per wave: VALU %.0f SALU %.0f LDS %.0f" % (n, {c: round(v) for c, v in m.items()}, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w))
PY
  done
fi
