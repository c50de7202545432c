#!/usr/bin/env python
"""Kernel statistics over the TIMED launches only, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --warmup W --no-extra`: bench.py launches every timed kernel S (settle, default 50) + W (warm-up) +
K (timed) times in that order, so the last K dispatches of a kernel are the ones between the barriers.

    python tools/timed_launch_stats.py <..._kernel_trace.csv> K > profiles/rNN_bench_timed_launches_stats.csv"""
import collections
import csv
import statistics
import sys

trace, steps = sys.argv[1], int(sys.argv[2])
runs = collections.defaultdict(list)
for row in csv.DictReader(open(trace)):
    runs[row['Kernel_Name']].append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
out = csv.writer(sys.stdout)
out.writerow(['Name', 'Calls', 'TimedCalls', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev', 'AllCallsAverageNs'])
for name, spans in sorted(runs.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    spans.sort()
    durations = [e - s for s, e in spans]
    timed = durations[-steps:] if len(durations) >= steps else durations
    out.writerow([name, len(durations), len(timed), '%.1f' % statistics.mean(timed), min(timed), max(timed),
                  '%.1f' % (statistics.pstdev(timed) if len(timed) > 1 else 0.0),
                  '%.1f' % statistics.mean(durations)])
