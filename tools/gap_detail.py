import csv, sys
from collections import defaultdict
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n))
rows.sort()
sel=[i for i,r in enumerate(rows) if "pca_select" in r[2]]
# last PCA run
runs=[[sel[0]]]
for a,b in zip(sel,sel[1:]):
    if rows[b][0]-rows[a][0]>5_000_000: runs.append([])
    runs[-1].append(b)
run=runs[-1]
gap=defaultdict(list)
for a,b in zip(run[12:],run[13:]):
    for i in range(a,b):
        g=(rows[i+1][0]-rows[i][1])/1e3
        gap[(rows[i][2],rows[i+1][2])].append(g)
tot=0
for k,v in sorted(gap.items(), key=lambda kv:-sum(kv[1])):
    print(f"{k[0]:28s} -> {k[1]:28s} n={len(v):3d} mean gap {sum(v)/len(v):6.1f} us  total {sum(v)/1e3:.2f} ms")
    tot+=sum(v)
print("total gaps in tail", tot/1e3, "ms")
