#!/bin/bash
# Round evidence: (1) rocprofv3 kernel-trace stats of the default bench command, (2) HBM-traffic PMC passes (one counter
# per pass, --kernel-trace only) over a few B=9 forwards, summarised into profiles/r02_conv_f16x3_pmc.json with the
# sha of the kernel source they were taken on (bench.py reports `roofline.traffic` only when that sha matches).
OUT=${1:-gpurun_out/prof_r02}
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${ONLY_PMC:-0}" != "1" ]; then
echo "kernel trace of: python3 bench.py --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err" || echo "trace run failed"
cut -c1-200 "$OUT/bench_profiled.json"
find "$OUT/trace" -name '*kernel_trace.csv' -delete      # hundreds of thousands of rows: keep the stats summary only
fi
if [ "${SKIP_PMC:-0}" = "1" ]; then exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  echo "pmc pass: $c"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/pmc_$c.log" 2>&1 || echo "pmc pass $c failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, hashlib, json, re, sys
out = sys.argv[1]
src = "extreme-video-compression-with-prediction-using-pre-trainded-diffusion-models-_amd/csrc/conv_igemm.hip"
sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
fam = re.compile(r"conv_split_rr_kernel<2, \d, 3,|conv_splitn_kernel<2, \d, 3,|conv_split_2d_kernel<2, 3,")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and fam.search(r["Kernel_Name"]):
            tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot / max(n, 1), n)
if len(res) == 2:
    fetch_kib, n = res["FETCH_SIZE"]; write_kib, _ = res["WRITE_SIZE"]
    js = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/profile_round.sh) -- python3 tools/forward_only.py 9",
          "kernel": "conv_split_rr_kernel<2>|conv_splitn_kernel<2><TN=3>", "batch": 9, "launches": n, "source_sha": sha,
          "FETCH_SIZE_avg_KiB": round(fetch_kib, 2), "WRITE_SIZE_avg_KiB": round(write_kib, 2),
          "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
          "hbm_bytes_per_launch": int(round((2 * fetch_kib + write_kib) * 1024))}
    json.dump(js, open(out + "/r02_conv_f16x3_pmc.json", "w"), indent=1)
    print(json.dumps(js))
PY
