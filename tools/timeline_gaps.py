"""Idle time between dependent kernels of a score-network forward, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace -d DIR -o fwd --output-format csv -- python3 tools/forward_only.py 9
    python3 tools/timeline_gaps.py DIR            # analyses the LAST forward of the run

Prints: wall time of the forward (first kernel start .. last kernel end), the union of kernel intervals (GPU busy), the idle
remainder, and per kernel name: launches, summed duration, and the summed idle gap in FRONT of its launches (start minus the
latest end of anything before it, when positive)."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# forwards are separated by the input conv of the network: conv with Ci = 32 is the first conv launch; simpler: split the
# trace at the pack_nchw_to_nhwc kernel (first kernel of forward_rows)
starts = [i for i, r in enumerate(rows) if "pack_nchw_to_nhwc" in r[2]]
assert len(starts) >= 2, "need at least two forwards in the trace"
a = starts[-1]
fw = rows[a:]
t0, t1 = fw[0][0], max(r[1] for r in fw)
busy, cur_s, cur_e = 0, fw[0][0], fw[0][1]
per = defaultdict(lambda: [0, 0, 0])
latest_end = fw[0][0]
for s, e, n in fw:
    m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
    name = (m.group(1) + (m.group(2) or "")) if m else n[:60]
    p = per[name]
    p[0] += 1
    p[1] += e - s
    if s > latest_end:
        p[2] += s - latest_end
    latest_end = max(latest_end, e)
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"last forward: {len(fw)} kernels, wall {(t1 - t0) / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
print(f"{'kernel':58s} {'n':>4s} {'sum us':>9s} {'gap-before us':>13s} {'avg gap':>8s}")
for name, (n, dur, gap) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{name[:58]:58s} {n:4d} {dur / 1e3:9.1f} {gap / 1e3:13.1f} {gap / 1e3 / n:8.2f}")
