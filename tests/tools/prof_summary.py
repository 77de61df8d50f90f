"""Per-step kernel time breakdown from a rocprofv3 kernel_stats CSV: python tests/tools/prof_summary.py stats.csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = 0.0
for r in rows:
    us = float(r['TotalDurationNs']) / 1e3 / steps
    tot += us
    if us >= 1.0:
        print('%8.1f us/step  %6.0f calls  avg %8.2f us  %s' % (us, float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, r['Name'][:70]))
print('%8.1f us/step total GPU kernel time' % tot)
