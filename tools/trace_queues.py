"""Print one steady-state step of a rocprofv3 kernel trace with one column per HSA queue (stream)."""
import csv, glob, os, sys
f = sys.argv[1] if len(sys.argv) > 1 else max(glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('scal::', '') for r in rows]
idx = [i for i, n in enumerate(names) if n == 'k_pre']
a, b = idx[-4], idx[-3]
t0 = int(rows[a]['Start_Timestamp'])
qs = sorted(set(r['Queue_Id'] for r in rows[a:b]))
last = {}
for i in range(a, b):
    r = rows[i]; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp']); q = qs.index(r['Queue_Id'])
    gap = (s - last[q]) / 1e3 if q in last else 0
    last[q] = e
    print(f"{(s-t0)/1e3:8.1f} {'   '*q}Q{q} {names[i][:24]:24s} dur {(e-s)/1e3:7.2f} gap {gap:6.2f}")
print(f, 'span', (int(rows[b]['Start_Timestamp']) - t0) / 1e3)
