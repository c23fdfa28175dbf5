#!/bin/bash
# A/B of BUILD-TIME variants of the library (build/ab/liblgu_<name>.so, made with LGU_EXTRA_HIPCC_FLAGS): for every variant the
# library is copied into place, the cooperative kernel's parity tests run once and tools/ab_lowmem_coop.py times BASELINE
# config 4; two passes so that drift of the box shows.  The default build is restored at the end.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
names="$@"
for pass in 1 2; do
  for n in $names; do
    cp build/ab/liblgu_$n.so lgu-slam_amd/liblgu_corr.so || exit 1
    if [ $pass = 1 ]; then
      timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "coop or lowmem or call_many or config5 or offset_rows" 2>&1 | tail -1 || exit 1
    fi
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 2>/dev/null | cut -c1-120)" || exit 1
  done
done
cp build/ab/liblgu_default.so lgu-slam_amd/liblgu_corr.so
