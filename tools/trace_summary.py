"""Per-kernel summary of a rocprofv3 kernel trace that separates the launches that did work from
the early-exit ones (a converged CG is followed by up to two chunks of launches that return at
once; `--stats` averages them in).  usage: python tools/trace_summary.py <*_kernel_trace.csv>"""
import csv, re, sys, collections
import numpy as np
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    acc[m.group(1) if m else n[:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("%-34s %6s %9s | %6s %10s %10s" % ("kernel", "calls", "avg ms", "work", "avg ms", "median ms"))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    med = np.median(v)
    work = v[v > max(0.02, 0.1 * med)] if med > 0.05 else v     # early exits take 3-5 us
    print("%-34s %6d %9.4f | %6d %10.4f %10.4f" % (k, len(v), v.mean(), len(work), work.mean(), np.median(work)))
