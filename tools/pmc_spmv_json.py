#!/usr/bin/env python3
"""profiles/rNN/pmc_spmv.json (what bench.py quotes as roofline.traffic) from a tools/pmc_run.sh summary.
usage: pmc_spmv_json.py <pmc_summary.txt> <out.json> <n> <spmv_bytes_of_the_streamed_format> "<note>" """
import json
import re
import sys

summary, out, n, alg, note = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
for ln in open(summary):
    m = re.match(r"(k_spmv<double, 1, 9>)\s+launches\s+(\d+).*FETCH_SIZE/launch\s+([\d.]+) KiB\s+WRITE_SIZE/launch\s+([\d.]+) KiB", ln)
    if m:
        fetch, write = float(m.group(3)), float(m.group(4))
        traffic = int((2.0 * fetch + write) * 1024)
        json.dump({"workload_n": n, "value_stream": 0, "kernel": m.group(1), "launches": int(m.group(2)),
                   "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write, "fetch_correction": 2.0,
                   "traffic_bytes_per_launch": traffic, "algorithmic_bytes_streamed_format": alg,
                   "traffic_over_algorithmic": traffic / alg,
                   "note": note + "  gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md): doubled."},
                  open(out, "w"), indent=1)
        print(open(out).read())
        break
else:
    sys.exit("no k_spmv<double, 1, 9> line in " + summary)
