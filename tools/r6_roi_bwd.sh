#!/bin/bash
# RoIAlign3D backward: parity tests, then the launch times of the mask / bbox extractors per MR3D_TUNE_ROIALIGN_FORM value
REPO=${GRAFT_REPO_ROOT:-$PWD}
O=$REPO/gpurun_out/r6_roi_bwd
mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests/test_gpu_nms_roialign.py -x -q -m gpu -k "roi_align or roialign" 2>&1 | tail -5 | tee $O/pytest.txt
for f in ${FORMS:-0 1 4 5 6}; do
  echo "== form $f" | tee -a $O/bwd.txt
  timeout 300 python3 tools/roi_fwd_bench.py --backward --tune 7=$f 2>&1 | grep -E "backward" | tee -a $O/bwd.txt
done
