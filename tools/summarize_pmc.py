#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv.
usage: python tools/summarize_pmc.py <counter_collection.csv> [min_calls]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:90]


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(sys.argv[1])):
        a = acc[r['Kernel_Name']][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
    names = sorted({c for k in acc.values() for c in k})
    print('| kernel | launches | ' + ' | '.join(names) + ' |')
    print('|---|---|' + '---|' * len(names))
    rows = []
    for k, cs in acc.items():
        n = max(v[1] for v in cs.values())
        rows.append((-cs.get('SQ_WAVE_CYCLES', cs[names[0]])[0], k, n, cs))
    for _, k, n, cs in sorted(rows)[:40]:
        print(f'| {short(k)} | {n} | ' + ' | '.join(f'{cs[c][0] / max(cs[c][1], 1):.4g}' if c in cs else '-' for c in names) + ' |')


if __name__ == '__main__':
    main()
