#!/usr/bin/env python3
"""Idle time BETWEEN the kernels of a step, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace -d DIR --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-b1 --no-cpu-baseline
    python3 tools/graph_gaps.py DIR

Sorts all dispatches by start time, finds the steady-state steps (the last occurrences of the step's first kernel name),
and prints for one step: kernels, busy time (sum of durations), span (first start to last end), the sum of the gaps
end(i) -> start(i+1), and the gap histogram.  A hipGraph replay still pays the command processor's dependent-dispatch
latency per node; this is the number that says how much of a step that is."""
import csv
import glob
import sys
import collections

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# the step's first kernel: the patchify launch of the network graph
first = [i for i, n in enumerate(names) if "k_patchify" in n]
if len(first) < 4:
    sys.exit("no steady-state steps found")
# two k_patchify launches per step (one per view batch) or one: take the distance between alternate occurrences
per_step = 2 if (first[-1] - first[-2]) < (first[-2] - first[-3]) * 0.5 or (first[-2] - first[-3]) < (first[-1] - first[-2]) * 0.5 else 1
starts = first[::per_step]
a, b = starts[-2], starts[-1]
step = rows[a:b]
busy = sum(e - s for s, e, _ in step)
span = step[-1][1] - step[0][0]
gaps = [max(0, step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)]
hist = collections.Counter(min(g // 500, 20) for g in gaps)
print(f"{f}\nkernels in the step: {len(step)}   busy {busy / 1e6:.3f} ms   span {span / 1e6:.3f} ms   sum of gaps {sum(gaps) / 1e6:.3f} ms "
      f"(mean {sum(gaps) / len(gaps) / 1e3:.2f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)")
print("gap histogram (0.5 us bins, last = >= 10 us):", " ".join(f"{k * 0.5:.1f}:{hist[k]}" for k in sorted(hist)))
big = sorted(((g, step[i][2][:60], step[i + 1][2][:60]) for i, g in enumerate(gaps)), reverse=True)[:12]
for g, n0, n1 in big:
    print(f"  {g / 1e3:7.1f} us after {n0}  before {n1}")
