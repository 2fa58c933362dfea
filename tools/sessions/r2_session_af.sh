#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_af
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
for tries in 4 8; do
STAN_LAB_PLACEMENT_FORCE_MISSES=99 python3 - > $OUT/move_$tries.txt 2>&1 <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, time
from stan_amd import hip, problem
job = problem.cube_job(148)
free0 = torch.cuda.mem_get_info()[0]
ctx = hip.Context(0); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0); ctx.set_option(hip.OPT_PLACEMENT_TRIES, $tries); ctx.set_profiling(True)
res = []
for step in range(3):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    p = ctx.profile()
    U, rep = K.cg_solve(job.F, 1e-8)
    p2 = ctx.profile()
    res.append(U)
    print("step %d (forced misses, tries $tries): candidates %d, vectors moved %d, probe kept %.4f slowest %.4f, its %d, in-CG SpMV %.4f ms, in use %.2f GB" %
          (step, p["placement_candidates"], p["placement_moved_vectors"], p["placement_ms_best"], p["placement_ms_worst"], rep["iterations"],
           p2["spmv_ms_total"] / p2["spmv_launches"], (free0 - torch.cuda.mem_get_info()[0]) / 1e9))
    K.free()
assert all(np.array_equal(res[0], r) for r in res[1:])
ctx.close()
print("after close: %.2f GB in use" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e9))
PY
cat $OUT/move_$tries.txt | grep -E "step|after|Error|error" 
done
unset STAN_HIP_LIB
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -3
python3 bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['placement_search'])"
