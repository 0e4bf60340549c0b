"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text table."""
import csv, glob, sys, os
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = max(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)   # newest run
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# source: {f}\n# total kernel time {tot/1e6:.3f} ms over the run; per-step figures divide by {steps} steps')
print(f'{"kernel":72s} {"calls":>6s} {"avg_us":>9s} {"us/step":>9s} {"pct":>6s}')
for r in rows[:40]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e3/steps:9.1f} {float(r['Percentage']):6.2f}")
