"""Inter-kernel gaps of a loop from a rocprofv3 kernel trace: for each consecutive pair of dispatches (by start time) the idle time
between the end of one and the start of the next, averaged per (previous kernel -> next kernel) pair, plus the busy / idle
totals per iteration.   usage: python tools/gap_analysis.py <kernel_trace.csv> <iterations>     Measurement helper."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
N = int(sys.argv[2])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
# steady state: the last N iterations' worth of dispatches — find the per-iteration count from k_composite_bwd launches
bwd = [i for i, r in enumerate(rows) if "k_composite_bwd" in r[2]]
first = bwd[-N - 1] + 1 if len(bwd) > N else 0
last = bwd[-1] + 1
seg = rows[first:last]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(seg, seg[1:]):
    gaps[(short(a[2]), short(b[2]))].append(b[0] - a[1])
    busy += a[1] - a[0]
busy += seg[-1][1] - seg[-1][0]
span = seg[-1][1] - seg[0][0]
print("iterations %d  span %.1f us/iter  busy %.1f us/iter  idle %.1f us/iter  dispatches/iter %.2f" % (N, span / N / 1e3, busy / N / 1e3, (span - busy) / N / 1e3, len(seg) / N))
for (a, b), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(g) >= N // 2:
        print("  %6.2f us x %5.2f/iter = %6.2f us/iter   %s -> %s" % (sum(g) / len(g) / 1e3, len(g) / N, sum(g) / N / 1e3, a, b))
