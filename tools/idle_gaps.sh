#!/bin/bash
# Kernel trace of a few B=9 forwards: how much of a forward's wall time has NO kernel running (launch gaps, drains)?
OUT=${1:-gpurun_out/gaps}
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/run.log" 2>&1 || echo "trace failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# last forward = last third of the launches after the weight-packing phase: take the final 1/3 of the kernels of the 3 forwards
fw = [r for r in rows if "pack_weights" not in r[2] and "absmax" not in r[2]]
n = len(fw) // 3
last = fw[-n:]
t0, t1 = last[0][0], max(e for _, e, _ in last)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in last:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = sorted(((last[i + 1][0] - max(e for _, e, _ in last[:i + 1][-4:])) for i in range(len(last) - 1)), reverse=True)
print(f"{n} kernels in the last forward, span {(t1 - t0) / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms ({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %), sum of kernel durations {sum(e - s for s, e, _ in last) / 1e6:.3f} ms")
PY
find "$OUT/trace" -name '*kernel_trace.csv' -delete
