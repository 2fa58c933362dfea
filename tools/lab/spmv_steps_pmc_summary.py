"""Table of the counter passes of tools/lab/spmv_steps_pmc.sh: one row per counter, one column per step kernel (per-launch
averages; with --kernel-trace the same files give the launch durations)."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
def step(name):
    m = re.search(r"k_steps<(\d+)>|k_stepsILi(\d+)E", name)
    return int(m.group(1) or m.group(2)) if m else None
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = step(row["Kernel_Name"])
        if s is not None:
            acc[(row["Counter_Name"], s)].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = step(row["Kernel_Name"])
        if s is not None:
            dur[s].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
steps = sorted({s for _, s in acc})
print("%-40s" % "counter (per launch)" + "".join("%16s" % ("S%d" % s) for s in steps))
for c in sorted({c for c, _ in acc}):
    print("%-40s" % c + "".join("%16.4g" % (sum(acc[(c, s)]) / max(len(acc[(c, s)]), 1)) for s in steps))
print("%-40s" % "duration under the profiler, us" + "".join("%16.1f" % (sum(dur[s]) / max(len(dur[s]), 1)) for s in steps))
