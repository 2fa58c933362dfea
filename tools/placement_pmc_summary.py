"""Table for the placement question from the passes of tools/placement_pmc.sh: per counter, the SpMV launches of the
three phases (cross-paired, self-paired, cross-paired; 2 warm-up launches of each phase dropped), summed and per
L2 channel (DIMENSION_INSTANCE 0..15 x DIMENSION_XCC 0..7).  usage: python tools/placement_pmc_summary.py <dir>"""
import glob, json, os, sys
import numpy as np
root = sys.argv[1]
print("%-34s %-6s %9s %14s %9s %9s %9s" % ("counter", "phase", "ms/launch", "sum/launch", "max/mean", "min/mean", "cv"))
for cdir in sorted(glob.glob(os.path.join(root, "TCC_*"))):
    f = os.path.join(cdir, "pmc_results.json")
    if not os.path.exists(f):
        continue
    d = json.load(open(f))["rocprofiler-sdk-tool"][0]
    names = {k["kernel_id"]: k["truncated_kernel_name"] or k["kernel_name"] for k in d["kernel_symbols"]}
    recs = [r for r in d["callback_records"]["counter_collection"]
            if "k_spmv" in names.get(r["dispatch_data"]["dispatch_info"]["kernel_id"], "")]
    recs.sort(key=lambda r: r["dispatch_data"]["start_timestamp"])
    n = len(recs) // 3
    for p, label in enumerate(("cross", "self", "cross")):
        ph = recs[p * n + 2:(p + 1) * n]
        if not ph:
            continue
        v = np.array([[x["value"] for x in r["records"]] for r in ph])          # [launch, 128]
        ms = np.mean([(r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / 1e6 for r in ph])
        per = v.mean(axis=0)
        print("%-34s %-6s %9.4f %14.4e %9.3f %9.3f %9.4f" % (os.path.basename(cdir), label, ms, v.sum(axis=1).mean(),
              per.max() / per.mean(), per.min() / per.mean(), per.std() / per.mean()))
