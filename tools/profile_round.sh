#!/bin/bash
# Round evidence, all on B=9 score-network forwards (tools/forward_only.py 9, res-block side stream OFF so kernels do not
# overlap and per-kernel durations are their own):
#   (1) rocprofv3 --kernel-trace --stats          -> profiles/${TAG}_forward_b9_kernel_stats.csv (+ .json: source sha, batch)
#   (2) --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass (gfx950 correction: FETCH x2) for the dominant kernel instance
#                                                 -> profiles/${TAG}_conv_f16x3_pmc.json  (bench.py: roofline.traffic)
#   (3) --pmc SQ_INSTS_VALU_MFMA_MOPS? no: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  -> MFMA-busy share and the
#       clock the chip holds under this kernel (GRBM_GUI_ACTIVE / 8 / duration), written into the .json of (1)
# bench.py quotes (1) and (2) only while the sha of csrc/conv_igemm.hip equals the one recorded here.
# (4) optionally (WITH_BENCH=1) the kernel-trace stats of the whole default bench command (side stream on).
TAG=${TAG:-r04}
OUT=${1:-gpurun_out/prof_$TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
export EVC_OVERLAP_SKIP=0
echo "(1) kernel trace of: python3 tools/forward_only.py 9 (EVC_OVERLAP_SKIP=0)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/trace.log" 2>&1 || echo "trace run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "(2) pmc pass: $c"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/pmc_$c.log" 2>&1 || echo "pmc pass $c failed"
done
echo "(3) pmc pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o fwd --output-format csv -- python3 tools/forward_only.py 9 > "$OUT/pmc_sq.log" 2>&1 || echo "pmc pass SQ failed"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, hashlib, json, re, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
src = "extreme-video-compression-with-prediction-using-pre-trainded-diffusion-models-_amd/csrc/conv_igemm.hip"
sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
st = glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)
# THE dominant kernel: the convolution kernel instance with the largest share of the GPU time of the profiled forwards
DOM, best = None, -1.0
for r in (csv.DictReader(open(st[0])) if st else []):
    n = r["Name"]
    if "conv_" in n and "Kernel" not in n and not any(k in n for k in ("absmax", "pack", "reduce")) and float(r["TotalDurationNs"]) > best:
        best = float(r["TotalDurationNs"])
        DOM = re.search(r"(conv_\w+<[^>]*>)", n).group(1)
meta = {"command": "EVC_OVERLAP_SKIP=0 rocprofv3 --kernel-trace --stats -- python3 tools/forward_only.py 9 (4 forwards: 1 in ScoreNet "
                   "warm-up of the label table + 3; side stream off)", "batch": 9, "source_sha": sha, "dominant_kernel": DOM}
if st:
    shutil.copy(st[0], f"{out}/{tag}_forward_b9_kernel_stats.csv")
    for r in csv.DictReader(open(st[0])):
        if DOM in r["Name"]:
            meta["dominant_avg_us"] = round(float(r["AverageNs"]) / 1e3, 2); meta["dominant_calls"] = int(r["Calls"])
            meta["dominant_share_of_gpu_time_pct"] = float(r["Percentage"])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and DOM in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot / max(n, 1), n)
if len(res) == 2:
    fetch_kib, n = res["FETCH_SIZE"]; write_kib, _ = res["WRITE_SIZE"]
    js = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, tools/profile_round.sh) -- python3 tools/forward_only.py 9",
          "kernel": DOM, "batch": 9, "launches": n, "source_sha": sha,
          "FETCH_SIZE_avg_KiB": round(fetch_kib, 2), "WRITE_SIZE_avg_KiB": round(write_kib, 2),
          "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
          "hbm_bytes_per_launch": int(round((2 * fetch_kib + write_kib) * 1024))}
    json.dump(js, open(out + f"/{tag}_conv_f16x3_pmc.json", "w"), indent=1)
    print(json.dumps(js))
f = glob.glob(f"{out}/pmc_sq/**/*counter_collection.csv", recursive=True)
if f:
    acc, dur, n = {}, 0.0, 0
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if DOM not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or (r["Start_Timestamp"], r["End_Timestamp"])
        if key not in seen:
            seen.add(key); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); n += 1
    if n and "GRBM_GUI_ACTIVE" in acc:
        meta["held_clock_ghz"] = round(acc["GRBM_GUI_ACTIVE"] / 8.0 / dur, 3)       # summed over the 8 XCDs; ns -> GHz
        meta["held_clock_note"] = "GRBM_GUI_ACTIVE / 8 / kernel duration over the dominant kernel's launches under the profiler"
    if n and "SQ_VALU_MFMA_BUSY_CYCLES" in acc and "GRBM_GUI_ACTIVE" in acc:
        # MFMA pipe busy cycles summed over SIMDs / (1024 SIMDs x elapsed shader cycles)
        meta["mfma_busy_frac"] = round(acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * acc["GRBM_GUI_ACTIVE"] / 8.0), 3)
    meta["sq_counters_sum"] = {k: v for k, v in acc.items()}
json.dump(meta, open(out + f"/{tag}_forward_b9_kernel_stats.json", "w"), indent=1)
print(json.dumps(meta))
PY
if [ "${WITH_BENCH:-0}" = "1" ]; then
  unset EVC_OVERLAP_SKIP
  echo "(4) kernel trace of: python3 bench.py --no-cpu-baseline"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/bench_trace" -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err" || echo "bench trace run failed"
  find "$OUT/bench_trace" -name '*kernel_trace.csv' -delete
  cp "$(find "$OUT/bench_trace" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv" 2>/dev/null
fi
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
