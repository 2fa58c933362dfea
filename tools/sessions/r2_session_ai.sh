#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ai
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o small -- python3 $R/tools/small_sizes.py 24 > $OUT/small_24.txt 2>&1
cd $R
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, re, collections
rows = list(csv.DictReader(open("$F")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 300 kernels of the CG (steady state)
names = []
for r in rows:
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); names.append(m.group(1) if m else r["Kernel_Name"][:20])
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
sel = [i for i, n in enumerate(names) if n in ("k_spmv", "k_step", "k_update", "k_spmv2")]
sel = sel[len(sel)//2:len(sel)//2 + 600]
for a, b in zip(sel[:-1], sel[1:]):
    if b != a + 1: continue
    dur[names[a]].append((int(rows[a]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    gap[names[a] + "->" + names[b]].append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
import statistics as st
for k, v in dur.items(): print("kernel %-10s median %.2f us  (n=%d)" % (k, st.median(v), len(v)))
for k, v in gap.items(): print("gap    %-20s median %.2f us  (n=%d)" % (k, st.median(v), len(v)))
PY
cat $OUT/small_24.txt | tail -2
rm -rf $OUT/trace
