"""Prints the kernel timeline of the last full train step found in a rocprofv3 kernel trace (csv)."""
import csv, glob, sys, os
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)   # newest run
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_sequence_pack' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {r['Kernel_Name'][:90]}")
print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} launches")
