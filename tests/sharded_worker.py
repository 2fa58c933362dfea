"""One rank of the sharded solve, for tests/test_gpu_sharded.py (launched by torch.distributed.run
with the gloo control plane; every rank drives GPU 0 through tests/fake_rccl)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from stan_amd import hip, problem  # noqa: E402

spec, out_dir, overlap = sys.argv[1], sys.argv[2], int(sys.argv[3])
p2p = len(sys.argv) > 4 and sys.argv[4] == "p2p"   # exchanges peer to peer between the PROCESSES (HIP IPC)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
if spec.startswith("fuzz:"):   # tests/fuzz.py job: shuffled wire order, several neighbours, empty ranks
    from tests import fuzz
    job = fuzz.random_job(int(spec[5:]))
elif spec.startswith("rev:"):  # a solid of revolution: collapsed hexes and a high-valence axis (the assembly's slow paths on a shard)
    from tests import fuzz
    job = fuzz.random_revolved_job(int(spec[4:]))
elif spec.startswith("bench:"):   # the bench-mode job of a golden fixture (tests/golden/bench_mode_<n>.npz): no jitter
    job = problem.cube_job(int(spec[6:]))
else:
    job = problem.cube_job(int(spec), jitter=0.05)
ctx = hip.Context(0)
box = [ctx.unique_id() if rank == 0 else None]
dist.broadcast_object_list(box, 0)
ctx.comm_init(rank, world, box[0])
ctx.set_option(hip.OPT_OVERLAP_HALO, overlap)
if p2p:
    ctx.set_option(hip.OPT_COMM_P2P, 1)    # a collective: the IPC handles travel over the communicator
    assert ctx.comm_info()["p2p"]
    ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
info = K.info()
if spec.startswith("bench:"):     # bench mode: merit stop off, 1e-8, fp64 only; the whole U comes back on every rank
    ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    # SHARDED_WORKER_MAXITS: a bound for the run of a deliberately broken library (whose loop may never converge)
    U, rep = K.cg_solve(job.F, 1e-8, int(os.environ.get("SHARDED_WORKER_MAXITS", "0")))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), U=U, its=rep["iterations"], term=rep["terminationtype"],
             rows=np.array([info["row_begin"], info["row_end"], info["n_halo"]]))
    K.free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
res = {}
for tag, eps, prec in (("fp64", 1e-6, hip.PREC_FP64), ("mixed", 1e-5, hip.PREC_MIXED),
                       ("fixed48", 1e-6, hip.PREC_FIXED48)):
    U, rep = K.cg_solve(job.F, eps, precision_mode=prec)
    res[tag] = (U, rep)
    if p2p:
        pr = ctx.profile()
        print("P2P rank %d %s: %s waits %d" % (rank, tag, rep, pr["loop_stream_waits"]), flush=True)
        assert pr["loop_collectives"] == 0 and pr["loop_stream_waits"] > 0, (pr["loop_collectives"], pr["loop_stream_waits"])
# round 5: a reduced-precision solve below what the fp32 entries carry -- the fp64 check (a sharded product with its halo
# exchange and an all-reduce / mailbox sum of its own) asks for refinement passes on both transports; every rank takes the
# same decisions (the iteration counts agree or the exchanges would not)
if spec.isdigit():    # (the jittered cube: well conditioned; the fuzz meshes may not refine to 1e-10 at all)
    ctx.set_profiling(True)
    ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)     # (alglib's merit rule would end the loop near 1e-6, type 7)
    Ut, rept = K.cg_solve(job.F, 1e-10, precision_mode=hip.PREC_MIXED)
    prt = ctx.profile()
    assert rept["terminationtype"] == 1 and rept["rel_residual"] <= 1e-10, rept
    assert prt["refine_passes"] >= 2 and prt["fp64_products"] >= prt["refine_passes"], (prt["refine_passes"], prt["fp64_products"])
    U64t, rep64t = K.cg_solve(job.F, 1e-10)
    ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert np.abs(Ut - U64t).max() <= 1e-6 * np.abs(U64t).max()
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), U=res["fp64"][0], Um=res["mixed"][0],
         Ux=res["fixed48"][0], its_x=res["fixed48"][1]["iterations"],
         its=res["fp64"][1]["iterations"], term=res["fp64"][1]["terminationtype"],
         rows=np.array([info["row_begin"], info["row_end"], info["n_halo"]]))
K.free()
ctx.close()
dist.barrier()
dist.destroy_process_group()
