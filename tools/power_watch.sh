#!/bin/bash
# Sample every GPU's average power and shader clock from sysfs (read-only; ordinary user) every ~100 ms while "$@" runs; the log's last
# line names the card whose power moved most (the one the command ran on) with its idle / peak / final power and clock range.
# Usage: tools/power_watch.sh out.log cmd args...
OUT=$1; shift
"$@" &
PID=$!
HS=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
echo "# columns: t_ms then per hwmon: power_W:sclk_MHz   hwmons: $(echo $HS | tr '\n' ' ')" > "$OUT"
T0=$(date +%s%N)
while kill -0 $PID 2>/dev/null; do
  L="$(( ($(date +%s%N) - T0) / 1000000 ))"
  for H in $HS; do
    P=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null || echo 0)
    F=$(cat $H/freq1_input 2>/dev/null || echo 0)
    L="$L $((P / 1000000)):$((F / 1000000))"
  done
  echo "$L" >> "$OUT"
  sleep 0.1
done
wait $PID
python3 - "$OUT" <<'PY'
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
if rows:
    n = len(rows[0]) - 1
    best = max(range(n), key=lambda j: max(int(r[1 + j].split(":")[0]) for r in rows) - min(int(r[1 + j].split(":")[0]) for r in rows))
    pw = [int(r[1 + best].split(":")[0]) for r in rows]; ck = [int(r[1 + best].split(":")[1]) for r in rows]
    open(sys.argv[1], "a").write(f"# busiest hwmon index {best}: power first {pw[0]} W, max {max(pw)} W, last {pw[-1]} W; sclk min {min(ck)} max {max(ck)} last {ck[-1]} MHz; {len(rows)} samples over {rows[-1][0]} ms\n")
PY
