#!/usr/bin/env python3
"""Which kernels of a traced step are torch's own (copies, fills, adds, cats, ...)?  Aggregates a rocprofv3 kernel trace by
(kernel name, grid size) for everything that is not one of the library's kernels.  usage: glue_kernels.py <kernel_trace.csv> <steps>"""
import csv
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    agg = defaultdict(lambda: [0, 0.0])
    ours = 0.0
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if 'at::native' in name or 'rocclr' in name or 'Cijk' in name or 'hipblas' in name.lower():
            short = name.split('<')[0].replace('void ', '').replace('at::native::', '')
            if 'elementwise' in name or 'vectorized' in name:
                for key in ('FillFunctor', 'CUDAFunctor_add', 'direct_copy', 'MulFunctor', 'DivFunctor', 'CompareFunctor', 'sigmoid', 'where', 'clamp', 'BUnaryFunctor', 'AUnaryFunctor', 'BinaryFunctor'):
                    if key in name:
                        short += ':' + key
                        break
            k = (short[:70], r.get('Grid_Size', r.get('Grid_Size_X', '?')))
            agg[k][0] += 1
            agg[k][1] += dur
        else:
            ours += dur
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print(f'torch / runtime kernels: {tot / steps:.1f} us per step in {sum(v[0] for v in agg.values()) / steps:.0f} launches; library kernels: {ours / steps:.1f} us per step')
    for (name, grid), (n, us) in rows[:40]:
        print(f'{us / steps:9.1f} us/step  {n / steps:6.1f} x  grid {grid:>10s}  {name}')


if __name__ == '__main__':
    main()
