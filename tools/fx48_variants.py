"""spmv_bench of the FIXED-48 and fp64 streams under several kernel variants, for a rocprofv3
--pmc FETCH_SIZE pass (kernel names carry the variant).  usage: python3 tools/fx48_variants.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
for prec, name in ((hip.PREC_FIXED48, "fixed48"), (hip.PREC_FP64, "fp64"), (hip.PREC_MIXED, "fp32")):
    for v in (0, 1, 9, 12):
        ctx.set_option(hip.OPT_SPMV_VARIANT, v)
        print(name, "variant", v, "%.4f ms" % min(K.spmv_bench(20, prec) for _ in range(3)))
