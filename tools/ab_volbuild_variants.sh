#!/bin/bash
# tools/ab_volbuild.py under each experiment library build/ab/liblgu_<name>.so ("default" = the in-tree library), two passes.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for pass in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset LGU_LIB_PATH; else export LGU_LIB_PATH="$GRAFT_REPO_ROOT/build/ab/liblgu_$n.so"; [ -f "$LGU_LIB_PATH" ] || exit 1; fi
    echo "== $n pass $pass: $(timeout -k 10 200 python tools/ab_volbuild.py 20 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print("fp32 build %.3f ms  half build %.3f ms  (library paths %.3f / %.3f)" % (d["matrix_core_build_ms"], d["half_matrix_core_build_ms"], d["matmul_plus_fused_postprocessing_ms"], d["half_matmul_plus_fused_postprocessing_ms"]))')" || exit 1
  done
done
