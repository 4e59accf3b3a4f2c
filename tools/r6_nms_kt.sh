#!/bin/bash
# NMS3D: parity tests, bench and per-kernel durations (rocprofv3 kernel trace of tools/nms_bench.py)
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
python -m pytest tests/test_gpu_nms_roialign.py tests/test_gpu_proposal_path.py -x -q -m gpu -k "nms or proposal" 2>&1 | tail -2
python tools/nms_bench.py 2>&1 | grep -v amdgpu
O=$REPO/gpurun_out/r6_nms_kt
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/kt
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 $REPO/tools/nms_bench.py 20 > /dev/null 2> $O/kt.err
KS=$(find $O/kt -name '*kernel_stats.csv' | head -1)
python3 - "$KS" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nms3d' in r['Name']:
        print(f"{r['Name'][28:80]:52s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
rm -rf $O/kt
