#!/bin/bash
# A/B of the side-stream skip path (EVC_OVERLAP_SKIP), one short step each, same box, two rounds.
for r in 1 2; do for o in 0 1; do
  EVC_OVERLAP_SKIP=$o timeout -k 10 250 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_overlap_$o.json
  python - $o <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/ab_overlap_{o}.json") if l.startswith("{")][-1])
print("overlap_skip", o, d["value"], "frames/s", d["ms_per_step"], "ms/step", flush=True)
PY
done; done
