#!/bin/bash
# Read-request size breakdown at the L2 -> fabric boundary (exact bytes, no FETCH_SIZE heuristics)
# and L2 hit/miss, per SpMV variant and value stream.  usage: bash tools/pmc_reqsize.sh <outdir>
OUT=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $R/$OUT/REQ -o pmc -- python3 $R/tools/fx48_variants.py 148 > $R/$OUT/req.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --kernel-trace --output-format csv -d $R/$OUT/HIT -o pmc -- python3 $R/tools/fx48_variants.py 148 > $R/$OUT/hit.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, re
for sub in ("REQ", "HIT"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_spmv<[^>]*>)", row["Kernel_Name"])
            if m: acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        d = {c: sum(v) / len(v) for c, v in acc[k].items()}
        if sub == "REQ":
            n32, n64, n128, n = d.get("TCC_EA0_RDREQ_32B_sum", 0), d.get("TCC_EA0_RDREQ_64B_sum", 0), d.get("TCC_EA0_RDREQ_128B_sum", 0), d.get("TCC_EA0_RDREQ_sum", 0)
            print("%-34s RDREQ %.4e  32B %.4e  64B %.4e  128B %.4e  -> %.3f GB (other-size requests: %.3e)" %
                  (k, n, n32, n64, n128, (32 * n32 + 64 * n64 + 128 * n128) / 1e9, n - n32 - n64 - n128))
        else:
            print("%-34s %s" % (k, "  ".join("%s %.4e" % (c, v) for c, v in sorted(d.items()))))
PY
