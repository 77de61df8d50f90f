"""Summarise a rocprofv3 --kernel-trace csv per (kernel name, grid size): calls, mean / min duration."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][-48:]
        grid = r.get('Grid_Size') or r.get('Grid_Size_X') or '?'
        acc[(name, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in acc.values())
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
with open(os.path.join(d, 'per_grid.txt'), 'w') as out:
    for (name, grid), v in rows[:40]:
        line = '%-50s grid %-9s calls %5d  mean %8.1f us  min %8.1f us  %5.1f %%' % (name, grid, len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot)
        print(line)
        out.write(line + '\n')
