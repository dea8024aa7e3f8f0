#!/bin/bash
# Same-box A/B of the convolution's K-split tail on the bench workload (2 timed steps each, alternating).
OUT=${1:-gpurun_out/ab_tail}
mkdir -p "$OUT"
for r in 1 2; do
  for v in 0 1; do
    EVC_CONV_OPTIONS="tail_split=$v" python bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/tail${v}_r$r.json" 2> "$OUT/tail${v}_r$r.err" || echo "run failed"
    python - "$OUT/tail${v}_r$r.json" "$v" "$r" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"tail_split={sys.argv[2]} round {sys.argv[3]}: {j['value']:.3f} frames/s, conv {j['roofline']['achieved']:.1f} TFLOP/s, {j['roofline'].get('conv_ms_per_forward')}")
PY
  done
done
