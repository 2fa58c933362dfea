"""A/B the SpMV variants in ONE process (interleaved rounds, median + min): rule 24."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5, 8]
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
info = K.info()
nb = info["n_blocks"]; nloc = info["n_block_rows"]
bytes_alg = nb * 76 + nloc * 52
x = np.random.default_rng(0).standard_normal(job.n_dof)
ref = None
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        ctx.set_option(hip.OPT_SPMV_VARIANT, v)
        res[v].append(K.spmv_bench(20))
for v in variants:
    ctx.set_option(hip.OPT_SPMV_VARIANT, v)
    y = K.spmv_local(x) if n <= 100 else None
    if v == 0: ref = y
    ok = "" if y is None or v == 8 else (" maxrel %.1e" % (np.abs(y - ref).max() / np.abs(ref).max()))
    t = np.array(res[v])
    print("variant %d: median %.4f ms min %.4f ms -> %.0f GB/s (%.1f%% of 8 TB/s)%s" %
          (v, np.median(t), t.min(), bytes_alg / np.median(t) / 1e6, bytes_alg / np.median(t) / 1e6 / 80, ok))

# the same operator in scalar CSR (fp64 + int32 columns), CSR-vector kernel
if n <= 160:
    ms, nbytes, diff = K.csr_spmv_bench(20)
    print("scalar CSR: %.4f ms for %.3f GB -> %.0f GB/s (%.1f%% of 8 TB/s); product differs from BSELL-64 by %.1e" %
          (ms, nbytes / 1e9, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, diff))

# context: what a plain device copy reaches on THIS device (read + write bytes / time)
a = torch.empty(1 << 29, dtype=torch.float64, device="cuda")   # 4 GiB
b = torch.empty_like(a)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch copy 4 GiB -> 4 GiB: %.3f ms = %.0f GB/s (read+write)" % (ms, 2 * a.numel() * 8 / ms / 1e6))
s_ = a.sum(); torch.cuda.synchronize()
e0.record()
for _ in range(10):
    s_ = a.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch sum of 4 GiB (read only): %.3f ms = %.0f GB/s" % (ms, a.numel() * 8 / ms / 1e6))
