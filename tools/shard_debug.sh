#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/sd
for W in 1 2; do
STAN_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2951$W tests/sharded_worker.py 12 /tmp/sd 1 2>&1 | grep -vE "amdgpu.ids|^$|Gloo" | tail -5
python - <<PY
import numpy as np, sys
sys.path.insert(0,'.')
from oracle import pyoracle as O
from stan_amd import problem
job=problem.cube_job(12,jitter=0.05)
rc,A=O.assemble(job.xyz,job.node_dof,job.conn,job.elem_mat,job.elem_type,job.mat_E_nu,job.red)
Uo,rep=O.cg(A,job.F,1e-7); Ux,_=O.cg(A,job.F,1e-12)
d=np.load('/tmp/sd/rank0.npz')
print("world $W", 'its',d['its'],'term',d['term'],'oracle',rep, 'err vs 1e-7 oracle', abs(d['U']-Uo).max()/abs(Uo).max(), 'vs exact', abs(d['U']-Ux).max()/abs(Ux).max(), 'rows', d['rows'])
PY
done
