import csv, sys
from collections import OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if pat in r['Kernel_Name']]
# consecutive runs of the same kernel name = one timed() block
blocks = []
for name, us in seq:
    import re
    m = re.search(r'(roi_align3d_\w+(<[^(]*>)?)', name)
    short = (m.group(1) if m else name)[:60]
    if blocks and blocks[-1][0] == short:
        blocks[-1][1].append(us)
    else:
        blocks.append((short, [us]))
for short, v in blocks:
    v = v[1:] if len(v) > 3 else v
    print(f'{short:62s} n={len(v):3d}  median {sorted(v)[len(v)//2]:7.1f} us  min {min(v):7.1f}  max {max(v):7.1f}')
