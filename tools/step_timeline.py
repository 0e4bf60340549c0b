"""Prints the kernel timeline of the median-length train step found in a rocprofv3 kernel trace (csv)."""
import csv, glob, sys, os
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)   # newest run
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if i > 0 and 'k_adam' in rows[i - 1]['Kernel_Name']]     # a step ends with the Adam update
# the step of median length (the last one of a run ends at a host synchronisation: its successor starts late)
spans = sorted((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']), a, b) for a, b in zip(marks[:-1], marks[1:]))
_, a, b = spans[len(spans) // 2]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {r['Kernel_Name'][:90]}")
print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} launches")
