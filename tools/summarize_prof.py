#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into the small
text summaries committed under profiles/.

usage: python tools/summarize_prof.py <kernel_stats.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>] > out.md

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so the read side is DOUBLED
(an upper bound for kernels whose reads are narrower than 16 B/lane); WRITE_SIZE is exact for 16-B stores
and fp32 atomics."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:110]


def main():
    stats = list(csv.DictReader(open(sys.argv[1])))
    pmc = {}
    for path, key in zip(sys.argv[2:4], ('FETCH_SIZE', 'WRITE_SIZE')):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == key:
                a = acc[r['Kernel_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
        pmc[key] = {k: v[0] / v[1] for k, v in acc.items()}
    total = sum(float(r['TotalDurationNs']) for r in stats)
    if not pmc:                                             # timing only: no counter passes were given
        print('| kernel | calls | total ms | avg us | % |')
        print('|---|---|---|---|---|')
        for r in sorted(stats, key=lambda r: -float(r['TotalDurationNs']))[:40]:
            print(f'| {short(r["Name"])} | {r["Calls"]} | {float(r["TotalDurationNs"]) / 1e6:.3f} | '
                  f'{float(r["AverageNs"]) / 1e3:.1f} | {100 * float(r["TotalDurationNs"]) / total:.1f} |')
        return
    print('| kernel | calls | total ms | avg us | % | HBM read MiB/launch (2x FETCH_SIZE) | HBM write MiB/launch | GB/s (r+w) |')
    print('|---|---|---|---|---|---|---|---|')
    for r in sorted(stats, key=lambda r: -float(r['TotalDurationNs']))[:40]:
        n = r['Name']
        avg_us = float(r['AverageNs']) / 1e3
        rd = 2 * pmc.get('FETCH_SIZE', {}).get(n, float('nan')) / 1024
        wr = pmc.get('WRITE_SIZE', {}).get(n, float('nan')) / 1024
        bw = (rd + wr) * 1048576 / (avg_us * 1e-6) / 1e9 if rd == rd and wr == wr else float('nan')
        print(f'| {short(n)} | {r["Calls"]} | {float(r["TotalDurationNs"]) / 1e6:.3f} | {avg_us:.1f} | '
              f'{100 * float(r["TotalDurationNs"]) / total:.1f} | {rd:.1f} | {wr:.1f} | {bw:.0f} |')


if __name__ == '__main__':
    main()
