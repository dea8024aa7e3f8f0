#!/bin/bash
# PMC passes over the standalone conv harness (one layer shape, both arithmetics).  Counters in separate passes,
# --kernel-trace only (no other trace domains).  Usage: tools/pmc_conv.sh <shape index> <outdir>
set -e
SHAPE=${1:-1}
OUT=${2:-gpurun_out/pmc_conv}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o pass --output-format csv -- ${BENCH:-./tools/conv_bench_r03} ${LAYOUT:-6} 3 "$SHAPE" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = ("2d_bf16x6" if "conv_split_2d_kernel<3" in k else "2d_f16x3" if "conv_split_2d_kernel<2" in k else
                "wide_f16x3" if "conv_wide_kernel" in k else
                "rr_bf16x6" if "conv_split_rr_kernel<3" in k else "rr_f16x3" if "conv_split_rr_kernel<2" in k else
                "sn_f16x3" if "conv_splitn_kernel" in k else "split_bf16x6" if "conv_split_kernel" in k else
                "f32" if "conv_igemm_kernel" in k else None)
        if not name: continue
        a = acc[(name, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        d = dur[name]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
for name, (t, n) in sorted(dur.items()): print(f"{name} avg kernel time {t / n / 1e3:.1f} us")
for (name, c), (v, n) in sorted(acc.items()):
    print(f"{name:12s} {c:28s} avg {v / n:16.1f}  (n={n})")
PY
