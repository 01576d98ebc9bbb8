import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('asl::', '')
    wg = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    g[(n[:48], wg)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
for (n, wg), v in rows[:45]:
    print(f'{n:48s} wgs {wg:8d} launches {len(v):4d} avg {sum(v)/len(v):8.4f} ms total {sum(v):9.2f}')
