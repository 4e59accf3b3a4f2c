#!/usr/bin/env python3
"""median / min duration per grid size of the kernels whose name contains a substring, from a rocprofv3 --kernel-trace csv.
usage: python tools/kernel_trace_stats.py <kernel_trace.csv> <name substring> [label]"""
import csv,sys,statistics as st,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
g=collections.defaultdict(list)
for r in rows: g[(r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in g.items(): print(sys.argv[3] if len(sys.argv)>3 else '', k, len(v), "median us", st.median(v)/1e3, "min", min(v)/1e3)
