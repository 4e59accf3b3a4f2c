#!/bin/bash
# round 6: parity + A/B of the RoIAlign3D forward instances (MR3D_TUNE_ROIALIGN_FORM = key 7) on one box
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r6_roi_ab
mkdir -p $O
python -m pytest tests/test_gpu_nms_roialign.py tests/test_gpu_roialign_gradcheck.py "tests/test_gpu_fullsize_parity.py::test_full_size_roialign_forward_backward_vs_oracle" -x -q -m gpu > $O/pytest.log 2>&1
tail -5 $O/pytest.log
rm -f $O/ab.log
for f in ${FORMS:-1 0 3 4 5}; do
  echo "== form $f" >> $O/ab.log
  python tools/roi_fwd_bench.py --tune 7=$f 2>&1 | grep -v amdgpu.ids >> $O/ab.log
done
cat $O/ab.log
if [ -f tools/ab/rows_stamps.so ]; then
  for f in ${STAMP_FORMS:-0}; do MR3D_LIB=$PWD/tools/ab/rows_stamps.so python tools/diag/roi_rows_stamps.py $f 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.log; done
fi
