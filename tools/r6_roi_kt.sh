#!/bin/bash
# kernel-trace durations of the RoIAlign3D forward for a list of MR3D_TUNE_ROIALIGN_FORM values (python-free device times)
REPO=${GRAFT_REPO_ROOT:-$PWD}
O=$REPO/gpurun_out/r6_roi_kt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in ${FORMS:-0 1 4}; do
  rm -rf $O/kt$f
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt$f -o kt -- python3 $REPO/tools/roi_fwd_bench.py --tune 7=$f $ARGS > /dev/null 2> $O/kt$f.err
  KS=$(find $O/kt$f -name '*kernel_stats.csv' | head -1)
  echo "== form $f"; python3 - "$KS" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'roi_align' in r['Name']:
        print(f"{r['Name'][28:92]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
  rm -rf $O/kt$f
done
