#!/usr/bin/env python3
"""bench.py -- the hot path (HEX8 assembly + Jacobi-scaled CG to 1e-8) on MI355X.

  python bench.py --gpus N --steps K --warmup W           (any N: with N > 1 and no launcher around it, this
                                                          process starts the N rank processes itself as fresh
                                                          children -- it never touches the GPU -- relays rank 0's
                                                          one JSON line and their exit code)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the same ranks)
  python bench.py --gpus N --one-process                  (ONE process driving N devices through
                                                          stan_hip_init_multi: what Solver.cs:18-69's single
                                                          process would use; host-pointer entries)

A "step" is one pass of the hot path over one synthetic structured HEX8 cube:
  stan_hip_assemble_hex8_dev (symbolic + numeric assembly of K)  +
  stan_hip_cg_solve_dev      (diagonal scaling + CG until ||r|| <= 1e-8 ||b||)
with the mesh arrays, DOF table and load vector already resident in HBM.  Workload:
BASELINE.json's headline, the ~10 M-DOF cube (n = 148 -> 9 923 847 DOF).  N > 1 shards
the block rows of the SAME cube over the ranks (strong scaling; halo exchange + all-reduce
over RCCL inside the CG).

Prints ONE JSON line (rank 0): metric DOF/s = nDOF * K / t, plus
  roofline     achieved HBM GB/s of the dominant kernel (the BSELL-64 SpMV), from HIP events
               around every SpMV launch of the timed solves, against 8 TB/s -- and against what a read-only
               sweep of K's resident values reaches on THIS box (stream_GBs, frac_of_stream);
  cpu_baseline the CPU oracle (a port of the reference algorithm) on a bounded sample;
  secondary    (N = 1, default workload) BASELINE.json's other single-GPU configs, 100^3 ... 400^3, as child processes
               BEHIND the measured line: a signal or a stall there prints the line as measured (bench_legs.py).
The merit-function stop of ALGLIB's lincg (termination type 7) is switched OFF here and in
the CPU baseline: with it the reference algorithm gives up near 1e-7 on cubes of this size
and would never reach the 1e-8 the metric names.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench_launch as BL  # noqa: E402  (watchdog, self-launcher, one-process form, transport probe)
import bench_legs as LEGS  # noqa: E402  (CPU baseline, secondary single-GPU legs)
from bench_launch import HBM_PEAK_GBS, METRIC  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser(description="Options beyond --gpus / --steps / --warmup select other workloads and library "
                                             "switches (include/stan_hip.h STAN_OPT_*); the defaults are the headline.")
    A = ap.add_argument
    A("--gpus", type=int, default=1)
    A("--steps", type=int, default=2)
    A("--warmup", type=int, default=1)
    A("--size", dest="n", type=int, default=148, help="cube edge in elements (148 -> ~10 M DOF)")
    A("--eps", type=float, default=1e-8)
    A("--etype", type=int, default=2, help="2 = HEX8_G2, 1 = HEX8_G1")
    A("--knockout", type=float, default=0.0, help="fraction of the cube's elements knocked out at random (irregular mesh; not the headline)")
    A("--max-its", type=int, default=0, help="LinSolverIterMax; a capped run reports per-iteration timings only (value null)")
    A("--mixed", action="store_true", help="fp32 matrix copy / fp64 vectors, refined to eps in fp64 terms (STAN_OPT_CG_REFINE)")
    A("--fixed48", action="store_true", help="fp64 arithmetic on the 48-bit fixed-point stream of the scaled matrix")
    A("--then-fixed48", action="store_true",
      help="N = 1: after the fp64 steps, the same steps with the FIXED-48 stream on the same resident model (`then_fixed48`)")
    A("--refine", type=int, default=-1, help="STAN_OPT_CG_REFINE: 0 fp64 check only, 1 refinement passes (default), 2 + fp64 refresh products")
    A("--single-reduce", action="store_true", help="STAN_OPT_CG_SINGLE_REDUCE: Chronopoulos-Gear loop (not the oracle's recurrences)")
    A("--sell-sigma", type=int, default=0, help="STAN_OPT_SELL_SIGMA: sorting window in slices (0 = library default; profiles/r03/SELL_C_SIGMA.md)")
    A("--spmv-variant", type=int, default=-1, help="STAN_OPT_SPMV_VARIANT: -1 = the library's choice; 0 / 9 / 12")
    A("--spmv-small-rows", type=int, default=-1, help="STAN_OPT_SPMV_SMALL: block rows up to which a slice belongs to a workgroup (-1 = library's 150 000)")
    A("--fold", type=int, default=-1, help="STAN_OPT_ROW_FOLDING: -1 auto, 0 never, 1 always (fold.hip)")
    A("--placement-tries", type=int, default=32, help="STAN_OPT_PLACEMENT_TRIES: candidates the placement search may time in the first warm-up assembly")
    A("--placement-fraction", type=float, default=0.75, help="share of free device memory the placement search may hold (0 = library default, a quarter)")
    A("--pool-fraction", type=float, default=0.9,
      help="share of free device memory the block pool may keep parked between steps (this process owns its GPU; 0 = library default, half)")
    A("--cpu-n", type=int, default=56, help="cube edge of the CPU-baseline sample (56: ~20 s; 100: ~2 min; 148: ~8 min, ~60 GB)")
    A("--no-cpu", action="store_true")
    A("--p2p", action="store_true", help="STAN_OPT_COMM_P2P (N > 1): reductions and halo exchanges peer to peer (HIP IPC), no RCCL launch in the loop")
    A("--one-process", action="store_true", help="ONE process drives the N devices through stan_hip_init_multi (host-pointer entries: copies inside the step)")
    A("--watchdog", type=float, default=900.0, help="seconds without progress after which a JSON error line is printed, exit code 3 (0 = off)")
    A("--no-p2p-probe", action="store_true", help="N > 1: skip the capped {classic, single-reduction} x {RCCL, peer to peer} comparison (config.p2p_probe)")
    A("--probe-its", type=int, default=200, help="iterations of each capped probe solve")
    A("--probe-watchdog", type=float, default=90.0, help="seconds without progress after which the probe is given up (the line stands, code 0)")
    A("--probe-child", action="store_true", help=argparse.SUPPRESS)   # internal: a rank of the probe's own process group
    A("--no-secondary", action="store_true",
      help="N = 1, default workload: skip the `secondary` legs (100^3, 200^3, FIXED-48, console driver, PMC traffic, k_recover, 400^3)")
    A("--secondary-budget", type=float, default=600.0, help="seconds all secondary legs may take together")
    return ap


def dtype_text(args):
    return ("f32 matrix / f64 vectors (refined until the FP64 residual meets eps)" if args.mixed else
            "f64 (matrix streamed as 48-bit fixed point)" if args.fixed48 else "f64")


class RankRun:
    """What one rank of a measured run holds: device, control plane, synthetic job, library context (with its
    communicator when sharded), the inputs resident in HBM, and `step()` = one pass of the hot path."""

    def __init__(self, args, dog):
        import numpy as np
        import torch
        import torch.distributed as dist
        from stan_amd import hip, problem
        self.args, self.dog, self.torch, self.dist, self.hip, self.np = args, dog, torch, dist, hip, np
        self.rank = rank = int(os.environ.get("RANK", "0"))
        self.world = world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        # STAN_BENCH_BACKEND / STAN_BENCH_DEVICE: test hooks (tests/test_gpu_sharded.py runs this
        # file with several ranks on ONE GPU over gloo + tests/fake_rccl); the driver uses neither
        backend = os.environ.get("STAN_BENCH_BACKEND", "nccl")
        self.dev_index = dev_index = int(os.environ.get("STAN_BENCH_DEVICE", local_rank))
        torch.cuda.set_device(dev_index)
        self.dev = dev = torch.device("cuda", dev_index)
        self.ctl = ctl = dev if backend == "nccl" else torch.device("cpu")   # where control-plane tensors live
        dog.touch("process group")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
        # ---- synthetic job (host: mesh, Database.AssignDOF, BC tables; outside the timed region)
        dog.touch("host set-up (mesh, AssignDOF, BC tables)")
        self.job = job = (problem.perforated_job(args.n, args.knockout, etype=args.etype) if args.knockout > 0
                          else problem.cube_job(args.n, etype=args.etype))
        dog.touch("context + communicator")
        self.ctx = ctx = hip.Context(dev_index)
        if world > 1:
            uid = torch.zeros(128, dtype=torch.uint8, device=ctl)
            if rank == 0:
                uid = torch.tensor(list(ctx.unique_id()), dtype=torch.uint8, device=ctl)
            dist.broadcast(uid, 0)
            ctx.comm_init(rank, world, bytes(uid.cpu().tolist()))
        ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
        ctx.set_option(hip.OPT_PLACEMENT_TRIES, max(1, min(64, args.placement_tries)))
        # The library, living inside a foreign host process, lets its placement search hold a quarter of the free device
        # memory at most (11 candidates at 148^3); this process owns its GPU: three quarters (the runs of one memory group
        # can be 150 GB long, placement.hip).  The search runs in the first warm-up step, never in the timed region.
        self.placement_budget = 0
        shared_gpu = bool(os.environ.get("STAN_BENCH_DEVICE"))   # (ranks sharing one GPU in the tests: library defaults)
        if args.placement_fraction > 0 and not shared_gpu:
            self.placement_budget = int(args.placement_fraction * torch.cuda.mem_get_info(dev)[0])
            ctx.set_option(hip.OPT_PLACEMENT_MAX_BYTES, self.placement_budget)
        if args.pool_fraction > 0 and not shared_gpu:
            ctx.set_option(hip.OPT_POOL_MAX_BYTES, int(args.pool_fraction * torch.cuda.mem_get_info(dev)[0]))
        if args.single_reduce:
            ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 1)
        if args.sell_sigma > 0:
            ctx.set_option(hip.OPT_SELL_SIGMA, args.sell_sigma)
        if args.fold != -1:
            ctx.set_option(hip.OPT_ROW_FOLDING, args.fold)
        if args.spmv_variant >= 0:
            ctx.set_option(hip.OPT_SPMV_VARIANT, args.spmv_variant)
        if args.refine >= 0:
            ctx.set_option(hip.OPT_CG_REFINE, args.refine)
        if args.spmv_small_rows >= 0:
            ctx.set_option(hip.OPT_SPMV_SMALL, args.spmv_small_rows)
        if args.p2p and world > 1:
            ctx.set_option(hip.OPT_COMM_P2P, 1)
        ctx.set_profiling(True)
        self.comm = ctx.comm_info()   # which transport the sharded loop runs over (a SCALE line should say)
        # inputs resident in HBM before the timed region; a rank of a sharded run holds only the elements
        # that touch its rows (the node arrays stay whole: any node may be a halo column)
        conn, emat, etyp = job.conn, job.elem_mat, job.elem_type
        if world > 1:
            from stan_amd import host
            mine = host.partition_elements(job.node_index, job.conn, world, rank)
            conn, emat, etyp = job.conn[mine], job.elem_mat[mine], job.elem_type[mine]
        self.conn = conn
        self.d_xyz = torch.from_numpy(job.xyz).to(dev)
        self.d_dof = torch.from_numpy(job.node_dof).to(dev)
        self.d_conn = torch.from_numpy(np.ascontiguousarray(conn)).to(dev)
        self.d_mat = torch.from_numpy(np.ascontiguousarray(emat)).to(dev)
        self.d_typ = torch.from_numpy(np.ascontiguousarray(etyp)).to(dev)
        self.d_red = torch.from_numpy(job.red).to(dev)
        self.d_F = torch.from_numpy(job.F).to(dev)
        self.d_U = torch.zeros(job.n_red, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        self.prec = hip.PREC_MIXED if args.mixed else hip.PREC_FIXED48 if args.fixed48 else hip.PREC_FP64

    def step(self, max_its=None):
        job, args = self.job, self.args
        K = self.ctx.assemble_hex8_dev(job.xyz.shape[0], self.d_xyz.data_ptr(), self.d_dof.data_ptr(),
                                       self.conn.shape[0], self.d_conn.data_ptr(), self.d_mat.data_ptr(),
                                       self.d_typ.data_ptr(), job.mat_E_nu, job.n_dof, self.d_red.data_ptr())
        rep = K.cg_solve_dev(self.d_F.data_ptr(), self.d_U.data_ptr(), args.eps,
                             args.max_its if max_its is None else max_its, self.prec)
        prof = self.ctx.profile()
        info = K.info()
        K.free()
        return rep, prof, info

    def stream_probe(self, reps=10):
        """The yardstick of the roofline: the library's own read-only sweep over K's RESIDENT fp64
        values in the product's access pattern (stan_hip_stream_bench) on a freshly assembled K of this workload, outside
        the timed region.  GB/s, or None (sharded runs, reduced-precision streams: the fp64 values are not what they stream)."""
        if self.world > 1 or self.prec != self.hip.PREC_FP64:
            return None
        job = self.job
        K = self.ctx.assemble_hex8_dev(job.xyz.shape[0], self.d_xyz.data_ptr(), self.d_dof.data_ptr(),
                                       self.conn.shape[0], self.d_conn.data_ptr(), self.d_mat.data_ptr(),
                                       self.d_typ.data_ptr(), job.mat_E_nu, job.n_dof, self.d_red.data_ptr())
        try:
            ms, nbytes = K.stream_bench(reps)
        finally:
            K.free()
        return {"GBs": nbytes / (ms * 1e-3) / 1e9, "ms_per_sweep": ms, "bytes_per_sweep": nbytes} if ms > 0 else None

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def gather(self, values):
        """one row of floats per rank -> list of rows (every rank gets all)"""
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.ctl)
        allr = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(allr, t)
        return [[float(v) for v in a.cpu().tolist()] for a in allr]

    def close(self):
        self.ctx.close()
        if self.world > 1:
            self.dist.destroy_process_group()


def measure(R, dog):
    """Warm-up, the timed region (EXACTLY args.steps steps between two barrier + device-sync points, max over ranks) and the
    report: returns (line dict on rank 0 / None elsewhere, converged)."""
    args, torch, dist = R.args, R.torch, R.dist
    rank, world, job, ctx, ctl, comm, conn = R.rank, R.world, R.job, R.ctx, R.ctl, R.comm, R.conn
    for i in range(args.warmup):
        dog.touch("warm-up step %d" % (i + 1))
        R.step()
    dog.touch("barrier before the timed steps")
    R.sync()
    t0 = time.perf_counter()
    spmv_ms = spmv_n = spmv2_ms = spmv2_n = asm_ms = cg_ms = red_ms = red_n = halo_ms = halo_n = 0.0
    for i in range(args.steps):
        dog.touch("timed step %d" % (i + 1))
        rep, prof, info = R.step()
        dog.touch("timed step %d done" % (i + 1), step_done=True)
        red_ms += prof["comm_reduce_ms_total"]; red_n += prof["comm_reduce_calls"]
        halo_ms += prof["comm_halo_ms_total"]; halo_n += prof["comm_halo_calls"]
        spmv_ms += prof["spmv_ms_total"]; spmv_n += prof["spmv_launches"]
        spmv2_ms += prof["spmv2_ms_total"]; spmv2_n += prof["spmv2_launches"]
        asm_ms += prof["assemble_ms"]; cg_ms += prof["cg_ms"]
    dog.touch("barrier after the timed steps")
    R.sync()
    dt = time.perf_counter() - t0
    per_rank = rank_devices = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what every rank measured on its own shard: SpMV rate against ITS bytes, exchange times per call
        per_rank = R.gather([prof["spmv_bytes"] / (spmv_ms / max(spmv_n, 1) * 1e-3) / 1e9 if spmv_ms > 0 else 0.0,
                             spmv_ms / max(spmv_n, 1), red_ms / max(red_n, 1) * 1e3, halo_ms / max(halo_n, 1) * 1e3,
                             float(info["n_halo"]), float(info["row_end"] - info["row_begin"])])
        # which physical device every rank drove and what its communicator says (a SCALE line must show that the
        # N ranks sat on N different GPUs of one communicator): HIP ordinal, PCI bus id, ncclCommCount / UserRank
        ordinal, bus = ctx.device_info()
        mine_txt = json.dumps({"rank": rank, "hip_device": ordinal, "pci_bus_id": bus, "comm_ranks": comm["comm_ranks"],
                               "comm_rank": comm["comm_rank"], "pid": os.getpid()}).encode()
        buf = torch.zeros(256, dtype=torch.uint8, device=ctl)
        buf[:len(mine_txt)] = torch.tensor(list(mine_txt), dtype=torch.uint8)
        allb = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(allb, buf)
        rank_devices = [json.loads(bytes(b.cpu().tolist()).rstrip(b"\0").decode()) for b in allb]
    dog.touch("report")
    ok = rep["terminationtype"] == 1 and rep["rel_residual"] <= args.eps
    if rank != 0:
        return None, ok
    stream = R.stream_probe()     # outside the timed region: what a read-only sweep of K's values reaches on THIS box
    avg_ms = spmv_ms / max(spmv_n, 1)
    # HBM traffic per launch: the committed PMC passes of this workload (tools/pmc_run.sh) unless this run's own
    # secondary leg measures it (main) -- counters cannot be read from inside the measured process
    traffic = traffic_source = None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_spmv*.json"))):
        d = json.load(open(f))
        if d.get("workload_n") == args.n and world == 1 and d.get("value_stream", 0) == prof["value_stream"]:
            traffic, traffic_source = d["traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
    achieved = prof["spmv_bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # SURVEY section 8d prices the REDUCED system (fixed DOFs squeezed out); this library keeps them as identity rows
    # (DESIGN.md section 2), so a launch moves slightly more.  Both fractions are reported; the x-clamped cube's reduced
    # block count is (3n-2)(3n+1)^2.
    frac_reduced = csr_equiv = None
    if args.etype == 2 and world == 1 and avg_ms > 0 and args.knockout == 0:
        blocks_red = (3 * args.n - 2) * (3 * args.n + 1) ** 2
        blk_bytes = 40 if args.mixed else 60 if args.fixed48 else 76
        bytes_red = blocks_red * blk_bytes + job.n_red * 16 + (job.n_red // 3) * 4
        frac_reduced = bytes_red / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        csr_equiv = (12 * 9 * blocks_red + 20 * job.n_red) / (avg_ms * 1e-3) / 1e9   # scalar CSR: 12 nnz + 20 N
    if world == 1:
        transport = "one rank"
    else:   # which RCCL: version, the FILE the entry points came from, and whether the host (torch) had mapped it already
        transport = "RCCL %s (ncclGetVersion %d; %s%s), communicator of %d ranks, this = rank %d%s" % (
            "" if comm["rccl_version"] else "stand-in", comm["rccl_version"], comm["library"] or "no library",
            ", shared with the host process" if comm["library_reused"] else "", comm["comm_ranks"], comm["comm_rank"],
            ", peer to peer" if comm["p2p"] else "")
    out = {
        "metric": METRIC, "value": job.n_dof * args.steps / dt, "unit": "DOF/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        # non-default library options this number depends on (DESIGN.md sections 3, 4)
        "merit_stop": False, "placement_tries": args.placement_tries, "placement_max_bytes": R.placement_budget,
        "dtype": dtype_text(args), "data": "synthetic",
        "config": {"workload": "%d^3 HEX8_G%d cube%s, %d DOF, clamp %s, PointLoad (0,0,50) on x=n; fp64 Jacobi-scaled CG to %.0e" %
                               (args.n, args.etype,
                                " with %.0f %% of its elements knocked out (irregular mesh)" % (100 * args.knockout)
                                if args.knockout > 0 else "", job.n_dof, "x=0" if args.etype == 2 else "x=0, y=0, z=0", args.eps),
                   "cg_loop": "single-reduction (Chronopoulos-Gear)" if args.single_reduce else "classic (alglib lincg recurrences)",
                   "loop_kernel_launches_per_iteration": prof["loop_kernel_launches"] / max(prof["loop_iterations_enqueued"], 1),
                   "n_dof": job.n_dof, "n_reduced": job.n_red,
                   "blocks_3x3_rank0": info["n_blocks"], "cg_iterations": rep["iterations"],
                   "termination_type": rep["terminationtype"], "rel_residual": rep["rel_residual"], "converged": bool(ok),
                   # reduced-precision streams: rel_residual is the FP64 residual of the returned point (one product
                   # on the fp64 values); the loop's own recurrence, the passes and the fp64 products it took
                   "rel_residual_recurrence": prof["rel_residual_recurrence"], "refine_passes": prof["refine_passes"],
                   "fp64_products_per_step": prof["fp64_products"],
                   "assemble_ms": asm_ms / args.steps, "cg_ms": cg_ms / args.steps,
                   "matrix_format": "BSELL-64 3x3 blocks (%s values + block cols as 16-bit offsets from per-slot bases -- one, or "
                                    "two where a slice mixes row lengths -- in %.1f %% of the slots, int32 in the rest)" %
                                    ("fp32" if args.mixed else "48-bit fixed-point" if args.fixed48 else "fp64",
                                     100.0 * prof["col_slots_packed"] / max(info["n_slots"], 1)),
                   # SELL-C-sigma: slots streamed per structural block - 1 (padded slots are streamed like real ones)
                   "ell_padding": info["n_slots"] * 64.0 / max(info["n_blocks"], 1) - 1.0,
                   "sell_sigma": info["sell_sigma"], "repacked_streams": "folded rows" if prof["repacked_streams"] else "none",
                   "folded_slots_over_padded_slots": info["folded_slots_permille"] / 1000.0 if info["folded_slots_permille"] else None,
                   "parallelism": "rows sharded x%d" % world, "transport": transport,
                   "elements_on_rank0": int(conn.shape[0]),
                   # the block pool keeps K's arrays between steps; with tries > 1 the first assembly picks the
                   # fastest-streaming of several hipMalloc blocks
                   "placement_tries": args.placement_tries,
                   "placement_search": {"candidates_timed": prof["placement_candidates"],
                                        "probe_ms_kept": prof["placement_ms_best"], "probe_ms_slowest": prof["placement_ms_worst"],
                                        "vectors_moved_instead": prof["placement_moved_vectors"] == 1,
                                        "product_vectors_moved": prof["placement_moved_vectors"] == 2},
                   # SURVEY section 8d assembly bytes: coords + connectivity read, K written once
                   "assembly_GBs": (conn.shape[0] * (192 + 32) + info["n_slots"] * 64 * 72)
                                   / (asm_ms / args.steps * 1e-3) / 1e9 if asm_ms > 0 else None,
                   # the same section's flop figure: the dense B'DB costs ~75 kflop per G2 element (an eighth per G1 element);
                   # the kernels exploit B's sparsity and recompute an element once per incident row, so this is the rate
                   # of the WORK DEFINED, not of the instructions issued
                   "assembly_GFLOPs_dense_equivalent": conn.shape[0] * (75e3 if args.etype == 2 else 75e3 / 8)
                                                       / (asm_ms / args.steps * 1e-3) / 1e9 if asm_ms > 0 else None},
        "roofline": {"bound": "hbm", "kernel": "k_spmv (BSELL-64 SpMV + fused p.Ap)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     # the counter traffic moved in this run's launch time (what the memory system really did; `frac`
                     # prices algorithmic bytes only)
                     "traffic_rate_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic and avg_ms > 0 else None,
                     "frac_reduced_system_bytes": frac_reduced, "csr_equivalent_GBs": csr_equiv,
                     # the yardstick: a read-only sweep of K's resident values in the product's access pattern on THIS box
                     # (stan_hip_stream_bench); the product moves columns, a gather and a store on top of it
                     "stream_GBs": stream["GBs"] if stream else None,
                     "stream_ms_per_sweep": stream["ms_per_sweep"] if stream else None,
                     "frac_of_stream": achieved / stream["GBs"] if stream else None,
                     # bytes of the format actually streamed (packed columns: 74 B per block, not 76);
                     # frac_reduced_system_bytes prices SURVEY section 8d's 76-B formula instead
                     "bytes_per_launch": prof["spmv_bytes"], "avg_launch_ms": avg_ms, "launches": int(spmv_n),
                     # refresh iterations: A p and A x from one matrix pass (not in the average)
                     "two_product_launches": int(spmv2_n), "two_product_avg_ms": (spmv2_ms / spmv2_n if spmv2_n else None)},
    }
    if per_rank is not None:
        fr = [p[0] / HBM_PEAK_GBS for p in per_rank]
        out["roofline"]["per_rank"] = {"frac_min": min(fr), "frac_max": max(fr), "frac": fr, "avg_launch_ms": [p[1] for p in per_rank]}
        # stream time per exchange (events around each, profiling only): RCCL launch -> sums available
        out["config"]["exchange"] = {"allreduce_us_per_call": [p[2] for p in per_rank], "halo_us_per_call": [p[3] for p in per_rank],
                                     "allreduces_per_step": red_n / args.steps, "halo_exchanges_per_step": halo_n / args.steps,
                                     "halo_block_rows": [int(p[4]) for p in per_rank], "owned_block_rows": [int(p[5]) for p in per_rank]}
        out["config"]["ranks"] = rank_devices
        out["config"]["distinct_devices"] = len(set(d["pci_bus_id"] for d in rank_devices))
    if not ok:   # a step that did not reach eps is not a step of this metric
        out["value"] = None
        out["error"] = "CG ended with type %d at %.3e (> eps %.0e): no DOF/s reported" % (
            rep["terminationtype"], rep["rel_residual"], args.eps)
    return out, ok


def main():
    BL.ensure_built()
    args = build_parser().parse_args()
    if args.one_process:
        return BL.run_one_process(args)
    if args.probe_child:
        return BL.probe_child_main(args, RankRun)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(BL.launch_ranks(args, sys.argv[1:]))

    import torch  # noqa: F401
    # armed after the import: the first `import torch` on a fresh box pages the image in for a minute or two
    dog = BL.Watchdog(args.watchdog, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), args)
    R = RankRun(args, dog)
    rank, world = R.rank, R.world
    # STAN_BENCH_TEST_HANG_RANK: test hook (tests/test_gpu_sharded.py): that rank never starts its steps
    if os.environ.get("STAN_BENCH_TEST_HANG_RANK", "") == str(rank):
        dog.stop()
        time.sleep(3600)
    out, ok = measure(R, dog)
    if rank == 0:
        if not args.no_cpu and world == 1:
            dog.stop()   # the CPU sample is bounded by its size, not by the watchdog
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = LEGS.cpu_baseline(args.cpu_n, args.eps)
            if args.watchdog > 0:
                dog.arm()
        else:
            out["cpu_baseline"] = None
        # the CPU port on the WORKLOAD itself (not the in-run sample): a committed run of tests/golden/
        # make_bench_mode_golden.py on a GPU box's host (8 min at 148^3: --cpu-n 148 repeats it here).  When ONE
        # speed-up is quoted, quote value / cpu_baseline.value: that one is of this run.
        at_wl = LEGS.cpu_at_workload(args.n) if args.knockout == 0 and args.etype == 2 else None
        if at_wl and ok:
            out["cpu_baseline_at_workload"] = at_wl
            out["speedup_vs_cpu_at_workload"] = out["value"] / at_wl["value"]
        dog.touch("cpu baseline done")
    # --then-fixed48: the same steps once more with the FIXED-48 value stream on the SAME resident model (one host set-up
    # for both: the 400^3 secondary leg)
    if args.then_fixed48 and world == 1 and ok and R.prec == R.hip.PREC_FP64:
        args.fixed48, R.prec = True, R.hip.PREC_FIXED48
        out2, ok2 = measure(R, dog)
        out["then_fixed48"] = {k: out2[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline") if k in out2}
        if not ok2:
            out["then_fixed48"]["error"] = out2.get("error")
        args.fixed48, R.prec = False, R.hip.PREC_FP64
    line = json.dumps(out) if rank == 0 else None
    # From here on everything is OPTIONAL work behind a finished measurement, in child processes: whatever happens in there
    # (a stall, a GPU fault, an abort inside RCCL or an IPC mapping, an outer timeout's SIGTERM), the line measured above is
    # printed and the exit code is 0 (bench_launch.hold_line).
    if world > 1 and not args.no_p2p_probe and not args.p2p and ok:
        BL.hold_line(dog, line)
        dog.optional = True
        dog.bound = float(args.probe_watchdog) * (len(BL.PROBE_LEGS) + 1) + 300.0   # (the children carry the real bound)
        dog.arm()                # the probe is bounded even when the run was not
        probe = None
        try:
            R.ctx.set_option(R.hip.OPT_POOL_MAX_BYTES, 0)   # every rank's parked blocks go back: the children share these GPUs
            probe = BL.run_probe_children(args, R, dog)
        except Exception as e:   # noqa: BLE001  (an optional extra must not cost the line)
            sys.stderr.write("bench.py: rank %d: transport probe given up: %s\n" % (rank, e))
        if rank == 0 and probe is not None:
            out["config"]["p2p_probe"] = probe
            out["config"]["recommended_transport"] = probe["recommended"]
            line = json.dumps(out)
        BL.hold_line(dog, line)
    # N = 1, the default workload: BASELINE.json's other single-GPU configurations behind the headline (child processes)
    if (world == 1 and rank == 0 and ok and not args.no_secondary and args.n == 148 and args.etype == 2 and args.knockout == 0
            and not args.mixed and not args.fixed48 and not args.then_fixed48 and args.max_its == 0):
        BL.hold_line(dog, line)
        dog.optional = True
        dog.bound = args.secondary_budget + 120.0
        dog.arm()
        try:
            # the context's pool gives its parked blocks back first: the legs are processes of their own on this GPU
            R.ctx.set_option(R.hip.OPT_POOL_MAX_BYTES, 0)
            out["secondary"] = []

            def attach(leg):   # the held line grows leg by leg: a signal or a stall keeps what is done
                out["secondary"].append(leg)
                if leg.get("traffic_bytes_per_launch") and "error" not in leg:   # counter traffic measured by this run
                    t_, ms_ = leg["traffic_bytes_per_launch"], out["roofline"]["avg_launch_ms"]
                    out["roofline"].update(traffic=t_, traffic_source="PMC passes of THIS run (secondary leg: rocprofv3 --pmc FETCH_SIZE / "
                                           "--pmc WRITE_SIZE over one step of the same workload in a child process; FETCH doubled)",
                                           traffic_rate_GBs=t_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None,
                                           traffic_over_algorithmic=t_ / out["roofline"]["bytes_per_launch"])
                BL.hold_line(dog, json.dumps(out))
            LEGS.secondary_legs(args, dog, attach)
            line = json.dumps(out)
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench.py: secondary legs given up: %s\n" % e)
            line = dog.held_line or line
        BL.hold_line(dog, line)
    if rank == 0:
        print(line, flush=True)
        BL.release_line(dog)
    # the measurement is out: a teardown that stalls (a peer that is gone) ends quietly with code 0, never a second line
    dog.held_line, dog.optional = None, True
    dog.touch("teardown")
    R.close()
    dog.stop()
    if rank == 0 and not ok:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
