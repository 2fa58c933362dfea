#!/bin/bash
# power, clocks and temperature of the GPU while bench.py runs (rocm-smi sampled every 0.2 s; reading needs no privileges)
# usage (GPU box, repo root): bash tools/power_trace.sh <outdir>
OUT=$1; mkdir -p $OUT
rocm-smi --showpower --showclocks --showtemp --showmaxpower > $OUT/smi_before.txt 2>&1
( for i in $(seq 1 120); do echo "== $(date +%s.%N)"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (edge|junction|memory|HBM)" ; sleep 0.2; done ) > $OUT/smi_trace.txt 2>&1 &
SMI=$!
sleep 2
python3 bench.py --no-cpu --steps 4 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
sleep 1
kill $SMI 2>/dev/null
python3 - <<PY
import re
t = re.split(r"^== (?=\\d)", open("$OUT/smi_trace.txt").read(), flags=re.M)[1:]
t0 = None
for blk in t[::2]:
    ts = float(blk.split()[0]); t0 = t0 or ts
    g = lambda pat: (re.search(pat, blk) or [None, None])[1]
    print("t=%5.1f s  power %s W  sclk %s  mclk %s  fclk %s MHz  junction %s C  memory %s C" % (ts - t0, g(r"Package Power \\(W\\):\\s*([0-9.]+)"),
          g(r"sclk[^(]*\\(([0-9]+)Mhz\\)"), g(r"mclk[^(]*\\(([0-9]+)Mhz\\)"), g(r"fclk[^(]*\\(([0-9]+)Mhz\\)"), g(r"junction\\)[^:]*:\\s*([0-9.]+)"), g(r"memory\\)[^:]*:\\s*([0-9.]+)")))
PY
head -30 $OUT/smi_before.txt
