import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
scan=[r for r in rows if "pq_scan_v3" in r["Kernel_Name"] and int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1)==32768]
t0=int(scan[-20]["Start_Timestamp"]); t1=int(scan[-1]["End_Timestamp"])
g=collections.defaultdict(list)
for r in rows:
    s=int(r["Start_Timestamp"])
    if s<t0-2_000_000 or s>t1: continue
    n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("asl::","")[:44]
    g[n].append((int(r["End_Timestamp"])-s)/1e6)
print("window ms per step", (t1-t0)/1e6/19)
for n,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print(f"{n:44s} n {len(v):4d} avg {sum(v)/len(v):7.4f}")
