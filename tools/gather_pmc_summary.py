"""Per-launch averages of every counter tools/gather_pmc.sh collected for k_spmv (launches that did work only)."""
import csv, glob, os, re, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "g*_k*"))):
    k = d.rsplit("_k", 1)[1]
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_spmv" not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in acc.items():
        s = sorted(v); med = s[len(s) // 2]
        work = [x for x in v if x >= 0.05 * med] if med > 0 else v
        res[c][k] = (sum(work) / max(len(work), 1), len(work))
    dur = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_spmv" in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    if dur:
        s = sorted(dur); med = s[len(s) // 2]
        work = [x for x in dur if x >= 0.3 * med]
        res["(duration us, this pass) " + os.path.basename(d).split("_")[0]][k] = (sum(work) / len(work), len(work))
print("%-52s %18s %18s %8s" % ("counter (per k_spmv launch)", "whole box", "40 % knocked out", "ratio"))
for c in sorted(res):
    a = res[c].get("0", (0, 0))[0]; b = res[c].get("0.4", (0, 0))[0]
    print("%-52s %18.1f %18.1f %8.3f" % (c, a, b, b / a if a else float("nan")))
