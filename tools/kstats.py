"""Top kernels of a rocprofv3 kernel_stats.csv (names contain commas: csv module).  usage: python tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:n]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {100 * float(r['TotalDurationNs']) / tot:5.1f}% {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {name[:110]}")
