#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv as microseconds per bench step:  kstats.py file.csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / steps / 1e3:.1f} us/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.2f}%")
