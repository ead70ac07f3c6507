#!/bin/bash
# Samples power and clocks (rocm-smi) while a command runs: is a kernel running against the power limit?
# usage (GPU box): tools/power_probe.sh OUTFILE -- python3 tools/dev_crt_probe.py headline
OUT=$1; shift; [ "$1" = "--" ] && shift
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.05; done ) > $OUT.samples 2>&1 &
SPID=$!
"$@" > $OUT.cmd 2>&1
kill $SPID 2>/dev/null; wait $SPID 2>/dev/null
python3 - $OUT <<'PY'
import sys, json, re
rows = []
for line in open(sys.argv[1] + ".samples"):
    try: d = json.loads(line)
    except Exception: continue
    c = d.get("card0", {})
    pw = next((float(v) for k, v in c.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
    sclk = next((v for k, v in c.items() if "sclk" in k.lower()), None)
    rows.append((pw, sclk))
pws = [p for p, _ in rows if p is not None]
print(f"{len(rows)} samples; power W: min {min(pws):.0f} median {sorted(pws)[len(pws)//2]:.0f} max {max(pws):.0f}" if pws else f"{len(rows)} samples, no power field: {rows[:2]}")
print("sclk samples:", sorted(set(str(s) for _, s in rows))[:12])
PY
tail -2 $OUT.cmd
