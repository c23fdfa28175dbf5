#!/bin/bash
# A/B of BUILD-TIME variants of the library.  Variants are built OUT of the package tree
#     python lgu-slam_amd/_build.py build/ab/liblgu_<name>.so -DCO_PF=3 ...
# (their flags end up in lgu_version()) and loaded from there through LGU_LIB_PATH: the in-tree lgu-slam_amd/liblgu_corr.so
# is never overwritten, so nothing that runs afterwards can pick up an experiment by accident.  For every variant the
# cooperative kernel's parity tests run once and tools/ab_lowmem_coop.py times BASELINE config 4; two passes so that drift
# of the box shows.  "default" names the in-tree library.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export LGU_DEBUG_KNOBS=1
for pass in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset LGU_LIB_PATH; else export LGU_LIB_PATH="$GRAFT_REPO_ROOT/build/ab/liblgu_$n.so"; [ -f "$LGU_LIB_PATH" ] || exit 1; fi
    if [ $pass = 1 ]; then
      timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "coop or lowmem or call_many or config5 or offset_rows" 2>&1 | tail -1 || exit 1
    fi
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 2>/dev/null | cut -c1-160)" || exit 1
  done
done
