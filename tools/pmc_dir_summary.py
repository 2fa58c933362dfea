"""Per-launch averages of the counters (and the duration) of the SpMV kernels in one rocprofv3 --pmc output directory
(launches that did work only: early-exit launches behind a converged solve are dropped)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_spmv" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_spmv" in row["Kernel_Name"]:
            dur[row["Kernel_Name"].split("(")[0][-60:]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
for (k, c), v in sorted(acc.items()):
    s = sorted(v); med = s[len(s) // 2]
    work = [x for x in v if x >= 0.05 * med] if med > 0 else v
    print("%-28s %-62s %-28s %16.1f  (%d launches)" % (os.path.basename(d), k, c, sum(work) / max(len(work), 1), len(work)))
for k, v in sorted(dur.items()):
    s = sorted(v); med = s[len(s) // 2]
    work = [x for x in v if x >= 0.3 * med]
    print("%-28s %-62s %-28s %16.1f  (%d launches)" % (os.path.basename(d), k, "duration us", sum(work) / max(len(work), 1), len(work)))
