#!/usr/bin/env python3
"""Timeline of ONE step out of a rocprofv3 --kernel-trace CSV: per hardware queue the busy time, the idle gaps between
kernels and the kernels that bound the step (nothing else running).  usage: python tools/timeline.py <kernel_trace.csv> [step]
The step boundaries are the launches of prep_weights_kernel for the 1-layer (stem) descriptor -- the first kernel of a forward."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows))
    starts = [k for k, e in enumerate(ev) if e[2].startswith('stem_fwd')]
    if len(starts) < 3:
        print('need >= 3 steps in the trace'); return
    a, b = starts[which], starts[which + 1]
    # walk back to the first kernel of that forward (the weight preparation) and forward likewise
    while a > 0 and ev[a - 1][2].startswith(('prep_weights', 'stem_pack')):
        a -= 1
    while b > 0 and ev[b - 1][2].startswith(('prep_weights', 'stem_pack')):
        b -= 1
    step = ev[a:b]
    t0, t1 = step[0][0], max(e[1] for e in step)
    print(f'step wall {1e-3 * (t1 - t0):.1f} us, {len(step)} kernels')
    per_q = defaultdict(list)
    for s, e, n, q in step:
        per_q[q].append((s, e, n))
    for q, lst in per_q.items():
        busy = sum(e - s for s, e, _ in lst)
        print(f'queue {q}: {len(lst)} kernels, busy {1e-3 * busy:.1f} us, span {1e-3 * (lst[0][0] - t0):.1f} .. {1e-3 * (lst[-1][1] - t0):.1f} us')
    # device-idle time and exclusive time per kernel name (time during which it is the ONLY kernel running)
    pts = sorted([(s, 1, n) for s, e, n, _ in step] + [(e, -1, n) for s, e, n, _ in step])
    active, last, idle, excl, shared = {}, t0, 0, defaultdict(int), defaultdict(int)
    for t, d, n in pts:
        dt = t - last
        if dt > 0:
            if not active:
                idle += dt
            elif len(active) == 1:
                excl[next(iter(active))] += dt
            else:
                for k in active:
                    shared[k] += dt
        last = t
        if d > 0:
            active[n] = active.get(n, 0) + 1
        else:
            active[n] -= 1
            if not active[n]:
                del active[n]
    print(f'device idle (no kernel running) {1e-3 * idle:.1f} us')
    print('exclusive time (only this kernel name running) / overlapped time, top 25:')
    for n, v in sorted(excl.items(), key=lambda kv: -kv[1])[:25]:
        print(f'  {1e-3 * v:9.1f} us excl  {1e-3 * shared.get(n, 0):9.1f} us overlapped  {n}')
    if len(sys.argv) > 3 and sys.argv[3] == 'dump':
        # every kernel of the step in start order with the device-idle gap in front of it (for the launch-bound head phases)
        end = t0
        for s_, e_, n, q in step:
            gap = s_ - end
            print(f'  {1e-3 * (s_ - t0):9.1f} +{1e-3 * (e_ - s_):7.1f} us  gap {1e-3 * max(gap, 0):6.1f}  q{q}  {n}')
            end = max(end, e_)
        return
    # phases: where the main queue sits at a few milestones
    marks = ('stem_fwd', 'maxpool_fwd', 'subsample2', 'maxpool_bwd', 'stem_wgrad', 'finish_affine')
    for s, e, n, q in step:
        if n.startswith(marks):
            print(f'  t={1e-3 * (s - t0):9.1f} .. {1e-3 * (e - t0):9.1f} us  q{q}  {n}')


if __name__ == '__main__':
    main()
