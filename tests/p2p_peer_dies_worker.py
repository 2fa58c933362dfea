"""One rank of a two-process peer-to-peer solve whose PEER DIES between publishing its vectors and its first
reduction (tests/fake_rccl's FAKE_RCCL_EXIT_AFTER_BCAST_GROUPS hook ends rank 1 there).  The survivor's stream sits
in a wait for an arrival that never comes; the library must notice (STAN_P2P_STALL_S), release this rank's own
counters, drain the queue and return STAN_E_COMM -- not block in hipStreamSynchronize with a spinning wavefront
(ADVICE r03, p2p.hip:70).  Launched by tests/test_gpu_sharded.py under torch.distributed.run (gloo, both on GPU 0)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402
from stan_amd import hip, problem  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
job = problem.cube_job(10, jitter=0.05)
ctx = hip.Context(0)
box = [ctx.unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, 0)
ctx.comm_init(rank, world, box[0])
ctx.set_option(hip.OPT_COMM_P2P, 1)        # broadcast group 1: the ranks' mailbox handles
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
U, rep = K.cg_solve(job.F, 1e-6)           # group 2: the vectors of solve 1; group 3: its result rows; both ranks finish it
assert rep["terminationtype"] in (1, 7), rep
print("rank %d: first solve done (%d its)" % (rank, rep["iterations"]), flush=True)
t0 = time.time()
try:
    K.cg_solve(job.F, 1e-6)                # group 4: rank 1 publishes its vectors and is gone
    print("rank %d: SECOND SOLVE RETURNED NORMALLY" % rank, flush=True)
    os._exit(7)
except hip.StanHipError as e:
    took = time.time() - t0
    print("rank %d: SURVIVOR code %d after %.1f s: %s" % (rank, e.code, took, e), flush=True)
    ok = e.code == hip.E_COMM and "released" in str(e)
    # a later solve on the broken exchange is refused at once, not stalled again
    t1 = time.time()
    try:
        K.cg_solve(job.F, 1e-6)
        ok = False
    except hip.StanHipError as e2:
        ok = ok and e2.code == hip.E_COMM and time.time() - t1 < 5.0
    os._exit(0 if ok else 8)   # (no teardown: the communicator's destroy barrier would wait for the dead peer)
