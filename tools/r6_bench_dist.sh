#!/bin/bash
# round 6: the bench line a multi-GPU run prints, exercised on ONE GPU (MR3D_BENCH_FORCE_DIST=1: RCCL communicator of one rank)
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r6_bench
mkdir -p $O
MR3D_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err
tail -c 3000 $O/bench_force_dist.json
python bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
python - <<'PY'
import json
for f in ('bench_force_dist', 'bench_plain'):
    try:
        j = json.loads([l for l in open(f'gpurun_out/r6_bench/{f}.json') if l.startswith('{')][-1])
        r = j['roofline']
        print(f, 'value', round(j['value'], 2), 'ms', round(j['ms_per_step'], 3), 'dom', r['kernel'], 'frac', round(r['frac'], 3), 'issued_frac', round(r['whole_step_issued_frac'], 3),
              'alg_frac', round(r['whole_step_frac'], 3), 'next', [(n['kernel'], round(n['frac'], 3)) for n in r['next_kernels']])
        if 'allreduce' in j:
            print('  allreduce', {k: v for k, v in j['allreduce'].items() if k != 'note'})
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 $O/bench_plain.err
