#!/bin/bash
# Memory-side PMC passes over the standalone conv harness.  Usage: tools/pmc_conv_mem.sh <shape index> <outdir>
set -e
SHAPE=${1:-1}
OUT=${2:-gpurun_out/pmc_conv_mem}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
# ONE counter per pass (derived TCC/TCP sums exceed the per-pass hardware budget when combined: rocprofv3 then aborts
# and hangs in finalisation), each pass under its own timeout
for set in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o pass --output-format csv -- ./tools/conv_bench_split 0 3 "$SHAPE" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed: $(grep -m1 -i 'error code' $OUT/p$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "split_rr" if "conv_split_rr_kernel" in k else "split" if "conv_split_kernel" in k else "f32" if "conv_igemm_kernel" in k else None
        if not name: continue
        a = acc[(name, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        d = dur[name]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
for name, (t, n) in sorted(dur.items()): print(f"{name} avg kernel time {t / n / 1e3:.1f} us")
for (name, c), (v, n) in sorted(acc.items()):
    print(f"{name:6s} {c:36s} avg {v / n:16.1f}  (n={n})")
PY
