#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_nms_roialign.py -x -q -m gpu -k "row_streaming" 2>&1 | tail -15
for f in ${FORMS:-0 8}; do
  echo "== form $f"
  timeout 300 python3 tools/roi_fwd_bench.py --tune 7=$f 2>&1 | tail -2
done
