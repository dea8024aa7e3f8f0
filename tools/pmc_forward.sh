#!/bin/bash
# SQ counter passes over B=9 score-network forwards (tools/forward_only.py 9, side stream off), aggregated for the kernels whose
# name contains <substring> (per grid size).  Counters in separate passes, --kernel-trace only.
# Usage: tools/pmc_forward.sh <kernel-name substring> <outdir>
SUB=${1:-attention_f16_kernel}
OUT=${2:-gpurun_out/pmc_fwd}
mkdir -p "$OUT"
export TMPDIR=/tmp
export EVC_OVERLAP_SKIP=0
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o pass --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, sys, collections
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]: continue
        name = "grid %s x %s x %s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = acc[(name, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        d = dur[name]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
for name, (t, n) in sorted(dur.items()): print(f"{name} avg kernel time {t / n / 1e3:.1f} us")
for (name, c), (v, n) in sorted(acc.items()):
    print(f"{name:28s} {c:28s} avg {v / n:16.1f}  (n={n})")
PY
