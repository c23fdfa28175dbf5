#!/bin/bash
# Timing-only ablations of the cooperative low-memory kernel (build/ab/liblgu_<name>.so made with -DCO_ABL_*: results are
# WRONG by construction, no tests are run): which part of a wave life the time of BASELINE config 4 follows.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for pass in 1 2; do
  for n in "$@"; do
    cp build/ab/liblgu_$n.so lgu-slam_amd/liblgu_corr.so || exit 1
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_lowmem_coop.py '' 2>&1 | tail -1 | cut -c1-120)"
  done
done
cp build/ab/liblgu_default.so lgu-slam_amd/liblgu_corr.so
