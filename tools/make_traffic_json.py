#!/usr/bin/env python3
"""Build profiles/<round>/traffic.json (HBM bytes per launch per bench.py kernel key) from the two rocprofv3 PMC
passes of `bench.py` (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, collected in separate runs as
MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B
request of a wide read, so the read side is doubled (see tools/summarize_prof.py).

usage: python tools/make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> "<source note>" > traffic.json"""
import csv
import json
import re
import sys
from collections import defaultdict


def key_of(name):
    """rocprof kernel name (template instance) -> the kernel key bench.py / _hip.op_kernel_key uses"""
    n = name.replace('(anonymous namespace)::', '')
    m = re.search(r'conv3d_fwd_kernel<FwdCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (true|false))?', n)
    if m:
        k, sd, sh, sw, cc, nb, par = m.group(1), m.group(4), m.group(5), m.group(6), m.group(10), m.group(11), m.group(12)
        if par == 'true':
            return f'conv3d_dgrad_s2k3<cc{cc},nb{nb}>'
        return f'conv3d_fwd<k{k},s{sd}{sh}{sw},cc{cc},nb{nb}>'
    m = re.search(r'conv3d_wgrad_kernel<WgCfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)', n)
    if m:
        return f'conv3d_wgrad<k{m.group(1)},s{m.group(4)}{m.group(5)}{m.group(6)},cc{m.group(7)},nb{m.group(8)}>'
    m = re.search(r'pointwise_(fwd|wgrad)_kernel<Pw(?:Wg)?Cfg<(\d+), (\d+)', n)
    if m:
        ci, co = int(m.group(2)), int(m.group(3))
        cc = 64 if ci % 64 == 0 else (32 if ci % 32 == 0 else 16)
        nb = 64 if co % 64 == 0 else (32 if co % 32 == 0 else 16)
        return f'conv3d_{m.group(1)}<k1,s111,cc{cc},nb{nb}>'
    # the two-dimensional Winograd forms of the 64 -> 64 3x3x3 convs (conv3d_wino2d.hip): one key per kernel (round 6)
    if 'conv3d_wino2d_wide_kernel' in n:
        return 'conv3d_fwd<k3,s111,cc16,nb64,wino2d-wide>'
    if 'conv3d_wino2d_kernel' in n:
        return 'conv3d_fwd<k3,s111,cc16,nb64,wino2d>'
    m = re.search(r'conv3d_wino2d_rows_kernel<(\d+)', n)
    if m:
        return None      # (16-wide F(4,3)-along-W rows kernels: several layer shapes share one instance; no per-key traffic)
    for pat, key in (('stem_fwd', 'stem_fwd'), ('stem_wgrad_gather_kernel', 'stem_wgrad_gather'), ('stem_wgrad_kernel', 'stem_wgrad'), ('maxpool_fwd_kernel', 'maxpool_fwd'),
                     ('maxpool_bwd_kernel', 'maxpool_bwd'), ('upsample_adjoint_add_kernel', 'upsample_adjoint_add')):
        if pat in n:
            return key
    return None


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            k = key_of(r['Kernel_Name'])
            if k:
                acc[k][0] += float(r['Counter_Value']) * 1024.0
                acc[k][1] += 1
    return acc


def main():
    rd, wr = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'source': sys.argv[3] if len(sys.argv) > 3 else '', 'kernels': {}}
    for k in sorted(set(rd) & set(wr)):
        out['kernels'][k] = {'hbm_read_bytes_per_launch': 2.0 * rd[k][0] / rd[k][1],
                             'hbm_write_bytes_per_launch': wr[k][0] / wr[k][1], 'launches_sampled': rd[k][1]}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
