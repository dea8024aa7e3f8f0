#!/bin/bash
# A/B of concurrent clip groups (bench.py --groups): one short step each, same box.
for g in ${GROUPS_LIST:-1 2 3}; do
  timeout -k 10 250 python bench.py --groups $g --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_groups_$g.json
  python - $g <<'PY'
import json, sys
g = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/ab_groups_{g}.json") if l.startswith("{")][-1])
print("groups", g, d["value"], "frames/s", d["ms_per_step"], "ms/step", flush=True)
PY
done
