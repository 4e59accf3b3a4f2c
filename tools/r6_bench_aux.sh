#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r6_bench
mkdir -p $O
python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-cpu-baseline} 2>$O/bench_aux.err | tail -1 > $O/bench_aux.json
python - <<'PY'
import json
j = json.load(open('gpurun_out/r6_bench/bench_aux.json'))
print('value', round(j['value'], 2), 'ms', round(j['ms_per_step'], 3), 'issued', round(j['roofline']['whole_step_issued_frac'], 3))
a = j.get('aux', {})
print({k: a[k] for k in a if 'roialign' in k and ('_us' in k or 'frac' in k) or 'nms3d' in k and 'us' in k or 'step_ms' in k})
PY
