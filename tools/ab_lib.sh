#!/bin/bash
# Same-box A/B of two builds of libevc_hip.so on the bench workload (alternating, 2 timed steps each).
# Usage: tools/ab_lib.sh <outdir> <libA.so> <libB.so> [rounds]
OUT=${1:-gpurun_out/ab_lib}; A=$2; B=$3; R=${4:-2}
mkdir -p "$OUT"
for r in $(seq 1 $R); do
  for v in A B; do
    so=$A; [ $v = B ] && so=$B
    EVC_HIP_SO="$so" python bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/${v}_r$r.json" 2> "$OUT/${v}_r$r.err" || echo "run failed"
    python - "$OUT/${v}_r$r.json" "$so" "$r" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} round {sys.argv[3]}: {j['value']:.3f} frames/s, conv {j['roofline']['achieved']:.1f} TFLOP/s, conv ms/forward {j['roofline'].get('conv_ms_per_forward')}")
PY
  done
done
