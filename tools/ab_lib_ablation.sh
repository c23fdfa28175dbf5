#!/bin/bash
# Timing-only ablations of the cooperative low-memory kernel (build/ab/liblgu_<name>.so made from a copy of round 2's source with
# its -DCO_ABL_* switches — the switches were removed from the source in round 3, the record is profiles/r02_lowmem_coop_ablation.txt; results are
# WRONG by construction, no tests are run): which part of a wave life the time of BASELINE config 4 follows.  The
# variants are loaded through LGU_LIB_PATH; the in-tree library is never touched.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export LGU_DEBUG_KNOBS=1
for pass in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset LGU_LIB_PATH; else export LGU_LIB_PATH="$GRAFT_REPO_ROOT/build/ab/liblgu_$n.so"; [ -f "$LGU_LIB_PATH" ] || exit 1; fi
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 2>&1 | tail -1 | cut -c1-160)"
  done
done
