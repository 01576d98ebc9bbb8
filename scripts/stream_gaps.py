"""Measurement helper: idle gaps on the stream that carries scan + rescoring, from a rocprofv3
kernel trace of the pipelined bench.  python scripts/stream_gaps.py <x_kernel_trace.csv>"""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
scan = [r for r in rows if 'pq_scan_v3_kernel' in r['Kernel_Name'] or 'flat_inv_scan_kernel' in r['Kernel_Name']]
full = [r for r in scan if int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) == 16384]
qid = collections.Counter(r['Queue_Id'] for r in full).most_common(1)[0][0]
onq = sorted((r for r in rows if r['Queue_Id'] == qid), key=lambda r: r['s'])
# the last 6 full-size scans delimit 5 steady-state steps
marks = [r for r in onq if r in full][-6:]
t0, t1 = marks[0]['s'], marks[-1]['s']
seg = [r for r in onq if t0 <= r['s'] < t1]
busy = sum(r['e'] - r['s'] for r in seg)
print(f'queue {qid}: {len(seg)} kernels in {(t1 - t0) / 1e6:.3f} ms (5 steps: {(t1 - t0) / 5e6:.3f} ms per step), busy {busy / 5e6:.3f} ms per step, '
      f'idle {(t1 - t0 - busy) / 5e6:.3f} ms per step')
per = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev = None
for r in seg:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('asl::', '')[:48]
    per[name][0] += 1
    per[name][1] += (r['e'] - r['s']) / 1e3
    if prev is not None:
        per[name][2] += max(0, r['s'] - prev['e']) / 1e3      # idle time in FRONT of this kernel
    prev = r
print(f'{"kernel":50s} {"launches/step":>13s} {"us/step":>10s} {"idle in front, us/step":>24s}')
for name, (n, us, gap) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f'{name:50s} {n / 5:13.1f} {us / 5:10.1f} {gap / 5:24.1f}')
other = [r for r in rows if r['Queue_Id'] != qid and t0 <= r['s'] < t1]
oq = collections.defaultdict(float)
for r in other:
    oq[r['Kernel_Name'].split('(')[0].replace('void ', '').replace('asl::', '')[:48]] += (r['e'] - r['s']) / 5e3
print('other queues in the same window (us per step):', {k: round(v, 1) for k, v in sorted(oq.items(), key=lambda kv: -kv[1])[:8]})
