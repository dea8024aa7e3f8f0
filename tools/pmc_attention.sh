#!/bin/bash
# PMC passes (separate passes, --kernel-trace only) over the attention kernels at the score network's shapes (tools/attn_ab.py B):
# per kernel template instance the mean of every counter.  Usage: tools/pmc_attention.sh [B] > profiles/rNN_attention_pmc_sq.log
B=${1:-9}
OUT=gpurun_out/attn_pmc
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pass --output-format csv -- python3 tools/attn_ab.py $B > $OUT.p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: [0.0, 0]); dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attention\w*(<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        name = m.group(1) + f" grid={r.get('Grid_Size', '?')}"
        a = acc[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        d = dur[name]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
for name, (t, n) in sorted(dur.items()): print(f"{name}: avg kernel time {t / n / 1e3:.1f} us over {n} launches")
for (name, c), (v, n) in sorted(acc.items()): print(f"{name:60s} {c:28s} avg {v / n:16.1f}  (n={n})")
PY
rm -rf $OUT $OUT.p*.log
