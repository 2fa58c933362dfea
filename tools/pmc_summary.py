"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel: average per launch over the
launches that did work (dispatches whose counter is below 5 % of the kernel's median are the
early-exit launches queued behind a converged CG solve; they are counted separately)."""
import csv, glob, os, re, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, C, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != C:
                continue
            kn = row["Kernel_Name"]
            m = re.search(r"(k_\w+(<[^>]*>)?)", kn)
            name = m.group(1) if m else kn[:44]
            acc[name].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)
        med = v[len(v) // 2]
        work = [x for x in v if x >= 0.05 * med] if med > 0 else v
        res[k][C] = (sum(work) / max(len(work), 1), len(work), len(v))
with open(os.path.join(out, "pmc_summary.txt"), "w") as fo:
    for k in sorted(res, key=lambda k: -res[k].get("FETCH_SIZE", (0, 0))[0]):
        f = res[k].get("FETCH_SIZE", (0, 0, 0)); w = res[k].get("WRITE_SIZE", (0, 0, 0))
        line = "%-44s launches %6d (of %6d)  FETCH_SIZE/launch %14.1f KiB  WRITE_SIZE/launch %14.1f KiB" % (k, f[1], f[2], f[0], w[0])
        print(line); fo.write(line + "\n")
