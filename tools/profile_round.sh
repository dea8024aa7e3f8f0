#!/bin/bash
# Round-end evidence: kernel-trace stats of the default bench command + HBM traffic PMC passes (one counter per pass).
OUT=${1:-gpurun_out/prof_final}
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${ONLY_PMC:-0}" != "1" ]; then
echo "kernel trace of: python3 bench.py"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > "$OUT/bench_profiled.log" 2>&1 || echo "trace run failed"
grep -m1 '^{"metric"' "$OUT/bench_profiled.log" > "$OUT/bench_profiled.json" || true
cut -c1-160 "$OUT/bench_profiled.json"
find "$OUT/trace" -name '*kernel_trace.csv' -delete      # hundreds of thousands of rows: keep the stats summary only
fi
if [ "${SKIP_PMC:-0}" = "1" ]; then exit 0; fi
if [ "${ONLY_PMC:-0}" = "1" ]; then :; fi
for c in FETCH_SIZE WRITE_SIZE; do
  echo "pmc pass: $c"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/pmc_$c.log" 2>&1 || echo "pmc pass $c failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]
        if "conv_split_rr_kernel<3" in k or "conv_split_kernel<2, 3" in k or "conv_split_kernel<1, 3" in k:
            v = per.setdefault("conv_split_rr_kernel|conv_split_kernel<TN=3>", [0.0, 0]); v[0] += float(r["Counter_Value"]); v[1] += 1
    for k, (v, n) in per.items(): res.setdefault(k, {})[c] = (v / n, n)
print(json.dumps(res))
json.dump(res, open(out + "/pmc_summary.json", "w"))
PY
