import csv,sys,statistics
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
# last forward run: take the last 70 kernels of the queue with most kernels
import collections
byq=collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
q=max(byq,key=lambda k:len(byq[k]))
ks=byq[q][-64:]
prev=None; gaps=[]
for r in ks:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if prev is not None: gaps.append(((s-prev)/1000, r["Kernel_Name"][:40]))
    prev=e
print('queue',q,'kernels',len(ks),'gap sum %.1f us, median %.1f, zero-gaps %d'%(sum(g[0] for g in gaps), statistics.median(g[0] for g in gaps), sum(1 for g in gaps if g[0]<1.5)))
print(' '.join('%.0f'%g[0] for g in gaps))
