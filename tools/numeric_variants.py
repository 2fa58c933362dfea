"""k_numeric A/B (round 4): the LDS-accumulator kernel against the register form, one PROCESS per variant (the variant is
read once per process from STAN_NUMERIC_VARIANT: 0 = k_numeric, 1 / 2 / 3 = k_numeric_reg with the register budget of
4 / 3 / 2 waves per SIMD).  Prints numeric / symbolic phase times (HIP events) and a bit-level fingerprint of K
(the product K x for a seeded x through the library's plain SpMV): every variant must print the same fingerprint.
usage: numeric_variants.py [n=148] [variants=0,1,2,3] ; child mode: numeric_variants.py --child n"""
import hashlib
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import torch  # noqa
    from stan_amd import hip, problem
    n = int(sys.argv[2])
    job = problem.cube_job(n) if n > 0 else None
    ctx = hip.Context(0)
    ctx.set_option(hip.OPT_PLACEMENT_TRIES, 1)
    ctx.set_profiling(True)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in
         dict(xyz=job.xyz, dof=job.node_dof, conn=job.conn, mat=job.elem_mat, typ=job.elem_type, red=job.red).items()}
    times = []
    for rep in range(5):
        K = ctx.assemble_hex8_dev(job.xyz.shape[0], d["xyz"].data_ptr(), d["dof"].data_ptr(), job.conn.shape[0], d["conn"].data_ptr(),
                                  d["mat"].data_ptr(), d["typ"].data_ptr(), job.mat_E_nu, job.n_dof, d["red"].data_ptr())
        p = ctx.profile()
        times.append((p["numeric_ms"], p["symbolic_ms"]))
        if rep < 4:
            K.free()
    x = np.random.default_rng(1).standard_normal(job.n_red)
    y = K.spmv(x)
    print("variant %s n %d: numeric ms %s | symbolic ms %s | K fingerprint %s" %
          (os.environ.get("STAN_NUMERIC_VARIANT", "0"), n, " ".join("%.2f" % t[0] for t in times),
           " ".join("%.2f" % t[1] for t in times), hashlib.sha1(y.tobytes()).hexdigest()[:16]), flush=True)
    K.free(); ctx.close()
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "148"
    for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(","):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=dict(os.environ, STAN_NUMERIC_VARIANT=v))
