"""Print the ordered kernel trace of one steady-state step from a rocprofv3 kernel_trace.csv (newest under gpurun_out/prof by default)."""
import csv, glob, os, sys
f = sys.argv[1] if len(sys.argv) > 1 else max(glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('scal::', '') for r in rows]
idx = [i for i, n in enumerate(names) if n == 'k_pre']
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
tot = 0
for i in range(a, b):
    s = int(rows[i]['Start_Timestamp']); e = int(rows[i]['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s-t0)/1e3:8.1f} {names[i][:28]:28s} dur {(e-s)/1e3:7.2f} gap {gap:6.2f} grid {rows[i]['Grid_Size_X']:>8s} wg {rows[i]['Workgroup_Size_X']}")
    prev_end = e; tot += (e - s)
print(f, 'step span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'busy', tot / 1e3, 'launches', b - a)
