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
               around every SpMV launch of the timed solves, against 8 TB/s;
  cpu_baseline the CPU oracle (a port of the reference algorithm) on a bounded sample.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def effective_cores():
    """CPUs this process may really use: affinity mask and cgroup CPU quota, not os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(n, eps, return_u=False):
    """Oracle = port of the reference algorithm (parallel K_e, serial locked scatter into a
    hash table, serial symmetric-upper CG), timed on this host's cores.  return_u: also the
    oracle's displacements and report (tools/cpu_sizes.py compares the GPU's with them)."""
    from oracle import pyoracle as O
    from stan_amd import problem
    job = problem.cube_job(n)
    threads = min(8, effective_cores())
    t0 = time.perf_counter()
    rc, A = O.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                       job.mat_E_nu, job.red, n_threads=threads)
    t1 = time.perf_counter()
    U, rep = O.cg(A, job.F, eps, merit_stop=False)
    t2 = time.perf_counter()
    base = {"value": job.n_dof / (t2 - t0), "unit": "DOF/s", "cores": threads, "kind": "port",
            "sample": "%d^3 HEX8_G2 cube, %d DOF: assembly %.2f s (K_e on %d threads, serial "
                      "scatter) + CG to %.0e %.2f s (%d its, serial: what the reference does)" %
                      (n, job.n_dof, t1 - t0, threads, eps, t2 - t1, rep["iterations"])}
    # second, labelled number: same arithmetic, the CG's matrix-vector
    # product on all cores (NOT what alglib does)
    allc = effective_cores()
    O.set_mv_threads(allc)
    t3 = time.perf_counter()
    U2, rep2 = O.cg(A, job.F, eps, merit_stop=False)
    t4 = time.perf_counter()
    O.set_mv_threads(1)
    base_all = {"value": job.n_dof / ((t1 - t0) + (t4 - t3)), "unit": "DOF/s", "cores": allc,
                "kind": "port", "sample": "same sample, CG matrix-vector product on %d OpenMP "
                "threads: CG %.2f s (%d its)" % (allc, t4 - t3, rep2["iterations"])}
    if return_u:
        return base, base_all, U, rep
    return base, base_all


def cpu_at_workload(n):
    """The committed run of the CPU port on the n^3 workload itself (profiles/r*/cpu_at_workload.json, written from
    tests/golden/make_bench_mode_golden.py's log on a GPU box's host cores): value, cores, seconds, source."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_at_workload.json")), reverse=True):
        try:
            for e in json.load(open(f)):
                if e.get("n") == n:
                    return dict(e["cpu_port"], source=os.path.relpath(f, ROOT), n_dof=e["n_dof"],
                                iterations=e["oracle_iterations"])
        except (OSError, ValueError, KeyError):
            continue
    return None


METRIC = "DOF/s (assembly+CG to 1e-8) on 10M-DOF HEX8 cube; SpMV GB/s vs HBM peak"


class Watchdog:
    """A multi-GPU run that stops making progress (a rank that never joins, a collective that never
    returns) must end by itself with a line that says where: a daemon thread checks the time since the
    last `touch`; past the bound it prints ONE JSON error line and leaves with os._exit(3) -- the process is
    never re-executed, a GPU process must not be.  Rank r waits 3 r seconds longer, so that rank 0 (whose
    exit makes the launcher end the others) reports first when every rank is stuck."""

    def __init__(self, bound_s, rank, world, args):
        import threading
        self.bound, self.rank, self.world, self.args = float(bound_s), rank, world, args
        self.phase, self.t_last, self.steps_done = "start", time.time(), 0
        # set while an OPTIONAL extra (the peer-to-peer probe) runs behind a finished measurement: a stall there
        # must not cost the line -- rank 0 prints it unchanged and every rank leaves with code 0
        self.held_line = None
        self.optional = False
        self.enabled = bound_s > 0
        if self.enabled:
            threading.Thread(target=self._run, daemon=True).start()

    def touch(self, phase, step_done=False):
        self.phase, self.t_last = phase, time.time()
        if step_done:
            self.steps_done += 1

    def stop(self):
        self.enabled = False

    def _run(self):
        while self.enabled:
            time.sleep(0.5)
            idle = time.time() - self.t_last
            if self.enabled and self.optional and idle > self.bound + 3.0 * self.rank:
                try:
                    if self.held_line is not None:
                        sys.stdout.write(self.held_line + "\n")
                        sys.stdout.flush()
                    sys.stderr.write("bench.py: rank %d: the optional phase '%s' made no progress for %.0f s; "
                                     "the measured line stands\n" % (self.rank, self.phase, idle))
                finally:
                    os._exit(0)
            if self.enabled and idle > self.bound + 3.0 * self.rank:
                line = {"metric": METRIC, "value": None, "unit": "DOF/s", "n_gpus": self.world,
                        "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None,
                        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                        "error": "watchdog: rank %d made no progress for %.0f s in phase '%s' after %d completed "
                                 "step(s); exiting (code 3)" % (self.rank, idle, self.phase, self.steps_done),
                        "watchdog": {"rank": self.rank, "phase": self.phase, "idle_s": idle,
                                     "bound_s": self.bound, "steps_done": self.steps_done}}
                try:
                    sys.stdout.write(json.dumps(line) + "\n")
                    sys.stdout.flush()
                finally:
                    os._exit(3)


def error_line(args, world, msg, **extra):
    line = {"metric": METRIC, "value": None, "unit": "DOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "error": msg}
    line.update(extra)
    return json.dumps(line)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (how the driver types it for N = 1).
    This process becomes the launcher: it has not touched the GPU and never does (no torch import, no HIP
    call); it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a
    FRESH child in a process group of its own, relays the one JSON line (rank 0's measurement, or a rank's
    watchdog line) and leaves with the child's exit code.  The ranks carry their own progress watchdog; the
    bound here is only the backstop for a launcher that never returns: the group that was started -- exactly
    that one, by its id -- is killed and an error line printed.  Nothing is ever re-executed."""
    import signal
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL and the peer-to-peer mappings need it
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    lines = []

    def relay():
        for ln in proc.stdout:
            if ln.startswith("{") and '"metric"' in ln:
                lines.append(ln.strip())
            else:
                sys.stderr.write(ln)

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    bound = None if args.watchdog <= 0 else (args.watchdog + 30.0) * (args.warmup + args.steps + 6) + 600.0
    try:
        rc = proc.wait(timeout=bound)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)     # the group started above, nothing else
        proc.wait()
        print(lines[0] if lines else error_line(args, args.gpus, "launcher: the %d rank processes did not end within "
                                                "%.0f s and were killed" % (args.gpus, bound)), flush=True)
        return 3
    t.join(10.0)
    if lines:
        print(lines[0], flush=True)
        return rc
    print(error_line(args, args.gpus, "launcher: the rank processes ended with code %d and printed no line "
                     "(their stderr is above)" % rc), flush=True)
    return rc if rc != 0 else 5


def ensure_built():
    """The native libraries normally arrive prebuilt in the tree; from a bare checkout local rank 0
    compiles them (hipcc, ~2 min) while the other ranks wait for the files.  No fallback: without
    them nothing runs."""
    need = [os.path.join(ROOT, "stan_amd", "lib", "libstan_hip.so"),
            os.path.join(ROOT, "stan_amd", "lib", "libstan_host.so"),
            os.path.join(ROOT, "oracle", "libstan_oracle.so")]
    if all(os.path.exists(f) for f in need):
        return
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__ as g
        g.build()
    else:
        t0 = time.time()
        while not all(os.path.exists(f) for f in need):
            if time.time() - t0 > 900:
                raise SystemExit("bench.py: native libraries were not built within 15 min")
            time.sleep(2.0)
        # (the Makefiles link to a temporary name and rename: a file that exists is complete)


def run_one_process(args):
    """--one-process: the form the reference's single process (Solver.cs:18-69) would use on a multi-GPU node --
    stan_hip_init_multi returns ONE handle that drives N devices (one worker thread and one communicator rank per
    device inside the library, multi.hip); the calls are the single-GPU calls with HOST pointers (device pointers
    belong to one device), so a step here includes the upload of the mesh shards and of F and the download of U:
    the PCIe-inclusive rate, reported as such, never bench.py's headline (which keeps its inputs resident)."""
    import numpy as np
    import torch  # noqa: F401  first: one shared HIP runtime
    from stan_amd import hip, problem
    n = args.gpus
    dog = Watchdog(args.watchdog, 0, n, args)
    dog.touch("host set-up (mesh, AssignDOF, BC tables)")
    job = (problem.perforated_job(args.n, args.knockout, etype=args.etype) if args.knockout > 0
           else problem.cube_job(args.n, etype=args.etype))
    # STAN_BENCH_DEVICE: test hook -- every rank of the handle on that one GPU (over tests/fake_rccl)
    hook = os.environ.get("STAN_BENCH_DEVICE")
    devices = [int(hook)] * n if hook is not None else list(range(n))
    dog.touch("stan_hip_init_multi")
    ctx = hip.Context(devices=devices)
    ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    if args.single_reduce:
        ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 1)
    if args.p2p and n > 1:
        ctx.set_option(hip.OPT_COMM_P2P, 1)
    ctx.set_profiling(True)
    prec = hip.PREC_MIXED if args.mixed else hip.PREC_FIXED48 if args.fixed48 else hip.PREC_FP64

    def step():
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        U, rep = K.cg_solve(job.F, args.eps, args.max_its, prec)
        prof, info = ctx.profile(), K.info()
        K.free()
        return U, rep, prof, info

    for i in range(args.warmup):
        dog.touch("warm-up step %d" % (i + 1))
        step()
    t0 = time.perf_counter()
    asm_ms = cg_ms = spmv_ms = spmv_n = 0.0
    for i in range(args.steps):
        dog.touch("timed step %d" % (i + 1))
        U, rep, prof, info = step()      # the calls return when the devices are done
        dog.touch("timed step %d done" % (i + 1), step_done=True)
        asm_ms += prof["assemble_ms"]; cg_ms += prof["cg_ms"]
        spmv_ms += prof["spmv_ms_total"]; spmv_n += prof["spmv_launches"]
    dt = time.perf_counter() - t0
    ok = rep["terminationtype"] == 1 and rep["rel_residual"] <= args.eps
    avg_ms = spmv_ms / max(spmv_n, 1)
    achieved = prof["spmv_bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {"metric": METRIC, "value": job.n_dof * args.steps / dt if ok else None, "unit": "DOF/s", "n_gpus": n,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "merit_stop": False,
           "dtype": ("f32 matrix / f64 vectors" if args.mixed else
                     "f64 (matrix streamed as 48-bit fixed point)" if args.fixed48 else "f64"),
           "data": "synthetic",
           "config": {"workload": "%d^3 HEX8_G%d cube, %d DOF; fp64 Jacobi-scaled CG to %.0e" %
                                  (args.n, args.etype, job.n_dof, args.eps),
                      "process_model": "ONE process, %d device(s) through stan_hip_init_multi (worker thread + "
                                       "communicator rank per device); host-pointer entries: every step uploads "
                                       "mesh, F and downloads U (PCIe-inclusive)" % n,
                      "transport": "peer to peer (mailboxes + arrival counters)" if args.p2p and n > 1 else
                                   ("RCCL" if n > 1 else "one rank"),
                      "devices": devices, "n_dof": job.n_dof, "cg_iterations": rep["iterations"],
                      "termination_type": rep["terminationtype"], "rel_residual": rep["rel_residual"],
                      "converged": bool(ok),
                      # device-side phases of rank 0 (events on its stream); the rest of ms_per_step is host + PCIe
                      "assemble_ms_rank0": asm_ms / args.steps, "cg_ms_rank0": cg_ms / args.steps,
                      "u_max": float(np.abs(U).max()), "parallelism": "rows sharded x%d" % n},
           "roofline": {"bound": "hbm", "kernel": "k_spmv (BSELL-64 SpMV + fused p.Ap), rank 0's shard",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "bytes_per_launch": prof["spmv_bytes"], "avg_launch_ms": avg_ms,
                        "launches": int(spmv_n)},
           "cpu_baseline": None}
    if not ok:
        out["error"] = "CG ended with type %d at %.3e (> eps %.0e): no DOF/s reported" % (
            rep["terminationtype"], rep["rel_residual"], args.eps)
    print(json.dumps(out), flush=True)
    dog.stop()
    ctx.close()
    if not ok:
        raise SystemExit(4)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=148, help="cube edge in elements (148 -> ~10 M DOF)")
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--mixed", action="store_true", help="fp32 matrix / fp64 vectors (refined to eps in fp64 terms: STAN_OPT_CG_REFINE)")
    ap.add_argument("--fixed48", action="store_true",
                    help="fp64 arithmetic on the 48-bit fixed-point stream of the scaled matrix")
    ap.add_argument("--refine", type=int, default=-1,
                    help="STAN_OPT_CG_REFINE for --mixed / --fixed48: 0 fp64 check only, 1 refinement passes (library default), "
                         "2 + fp64 refresh products")
    ap.add_argument("--etype", type=int, default=2, help="2 = HEX8_G2, 1 = HEX8_G1")
    ap.add_argument("--max-its", type=int, default=0,
                    help="LinSolverIterMax; a capped run reports per-iteration timings only (value null)")
    ap.add_argument("--single-reduce", action="store_true",
                    help="STAN_OPT_CG_SINGLE_REDUCE: Chronopoulos-Gear loop (not the oracle's recurrences)")
    ap.add_argument("--cpu-n", type=int, default=56,
                    help="cube edge of the CPU-baseline sample (56: ~20 s of CPU work; 100 = BASELINE config 2's size, "
                         "~2 min; 148 = the bench workload itself, ~8 min and ~60 GB: profiles/r03/cpu_sizes_n148_*.jsonl)")
    ap.add_argument("--sell-sigma", type=int, default=0,
                    help="STAN_OPT_SELL_SIGMA: sorting window of the matrix layout in slices (0 = library default 1: rows sorted "
                         "inside each slice only; up to 32: less padding, less gather locality -- profiles/r03/SELL_C_SIGMA.md)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--placement-tries", type=int, default=32,
                    help="STAN_OPT_PLACEMENT_TRIES: candidates the allocation-by-search of K's value array may time in "
                         "the first (warm-up) assembly (library default 16; 1 = plain allocation)")
    ap.add_argument("--placement-fraction", type=float, default=0.75,
                    help="fraction of the free device memory the placement search may hold while it runs "
                         "(STAN_OPT_PLACEMENT_MAX_BYTES; 0 = the library's default, a quarter)")
    ap.add_argument("--pool-fraction", type=float, default=0.9,
                    help="fraction of the free device memory the context's block pool may keep parked between steps "
                         "(STAN_OPT_POOL_MAX_BYTES; the library's own default -- half the device -- is for a library "
                         "inside a foreign host; this process owns its GPU.  At 400^3 the fp64 values and their fp32 "
                         "copy are 126 + 63 GB: with half the device one of them went back to the driver every step)")
    ap.add_argument("--p2p", action="store_true",
                    help="STAN_OPT_COMM_P2P (N > 1): the CG's reductions and halo exchanges go peer to peer between the "
                         "rank processes (HIP IPC mappings; no RCCL launch in the loop) instead of over RCCL")
    ap.add_argument("--spmv-variant", type=int, default=-1,
                    help="STAN_OPT_SPMV_VARIANT: -1 = the library's choice; 0 / 9 / 12")
    ap.add_argument("--spmv-small-rows", type=int, default=-1,
                    help="STAN_OPT_SPMV_SMALL: block-row limit below which a slice belongs to a workgroup (k_spmv_small) instead of a "
                         "wavefront (-1 = the library's 150 000; 0 = never)")
    ap.add_argument("--fold", type=int, default=-1,
                    help="STAN_OPT_ROW_FOLDING: -1 = auto (library default), 0 = never, 1 = long rows always lend their tails to the "
                         "idle slots of their slice (fold.hip)")
    ap.add_argument("--knockout", type=float, default=0.0,
                    help="not the headline workload: the cube with this fraction of its elements knocked out at random "
                         "(an irregular mesh: row lengths vary; SELL-C-sigma evidence, profiles/r03)")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="seconds without progress (set-up, a warm-up step, a timed step) after which the run prints a "
                         "JSON error line and exits with code 3 (0 = off)")
    ap.add_argument("--one-process", action="store_true",
                    help="ONE process drives the N devices through stan_hip_init_multi (one worker thread and one "
                         "communicator rank per device inside the library) -- the form a single-process host like the "
                         "reference's Solver.Main would use; host-pointer entries, so the copies are inside the step")
    ap.add_argument("--no-p2p-probe", action="store_true",
                    help="N > 1: skip the capped comparison of the loop forms and transports ({classic, single-reduction} x "
                         "{RCCL, peer to peer}) that is attached to the line as config.p2p_probe after the timed steps")
    ap.add_argument("--probe-its", type=int, default=200, help="iterations of each capped probe solve")
    ap.add_argument("--probe-watchdog", type=float, default=90.0,
                    help="seconds without progress after which the probe is given up (the measured line is printed "
                         "unchanged, exit code 0)")
    ap.add_argument("--probe-child", action="store_true", help=argparse.SUPPRESS)   # internal: a rank of the probe's own process group
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1, default workload: skip the secondary legs (100^3, 200^3, FIXED-48, the console driver, "
                         "k_recover) that are attached to the line as `secondary` after the timed steps")
    ap.add_argument("--secondary-budget", type=float, default=240.0, help="seconds all secondary legs may take together")
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


# ---- N > 1: the loop forms and transports side by side, in a process group of their own -------------------------------------
PROBE_LEGS = (("classic_rccl", 0, 0), ("classic_p2p", 0, 1), ("single_reduce_rccl", 1, 0), ("single_reduce_p2p", 1, 1))


def probe_report(legs, capped_at):
    """What the four legs say, and what a host should select on THIS node: a form that is measurably (3 %) faster than the
    library defaults and agrees with them.  Agreement on a capped solve = the same iteration count and a residual within a
    tolerance (1e-9 for the classic loop over another transport: only the order of a handful of partial sums differs;
    1e-3 for the single-reduction form, whose recurrences round differently) -- NOT bit equality: RCCL's ring / tree order
    is not rank order for N >= 3, so `same_residual_bits_classic` is information, never a gate."""
    base = legs["classic_rccl"]

    def agrees(leg, tol):
        return (leg["iterations"] == base["iterations"] and
                abs(leg["rel_residual"] - base["rel_residual"]) <= tol * abs(base["rel_residual"]))
    best = min(legs, key=lambda k: legs[k]["ms_per_iteration"])
    ok = {k: agrees(v, 1e-9 if k.startswith("classic") else 1e-3) for k, v in legs.items()}
    opts = {"classic_rccl": "library defaults", "classic_p2p": "STAN_OPT_COMM_P2P=1",
            "single_reduce_rccl": "STAN_OPT_CG_SINGLE_REDUCE=1", "single_reduce_p2p": "STAN_OPT_CG_SINGLE_REDUCE=1 + STAN_OPT_COMM_P2P=1"}
    best_ok = min((k for k in legs if ok[k]), key=lambda k: legs[k]["ms_per_iteration"])   # (the defaults agree with themselves)
    rec = best_ok if legs[best_ok]["ms_per_iteration"] < 0.97 * base["ms_per_iteration"] else "classic_rccl"
    return {"capped_at_iterations": capped_at, "legs": legs,
            "same_residual_bits_classic": legs["classic_rccl"]["rel_residual"] == legs["classic_p2p"]["rel_residual"],
            "agrees_with_classic_rccl": ok,
            "time_over_classic_rccl": {k: v["ms_per_iteration"] / base["ms_per_iteration"] for k, v in legs.items()},
            "fastest": best, "recommended": "%s (%s)" % (rec, opts[rec])}


def probe_child_main(args):
    """One rank of the probe's own process group (started by run_probe_children as a fresh child of a measured rank):
    the sharded loop as {classic, single-reduction} x {RCCL: 2 / 1 all-reduce launches + 1 grouped send/recv per iteration,
    peer to peer: STAN_OPT_COMM_P2P, mailboxes + arrival counters through HIP IPC, no collective launch in the loop}
    on the same capped solve (--probe-its iterations, one warm-up solve each): ms per iteration (slowest rank), stream
    time per reduction point and halo exchange, launches / collectives / stream waits per iteration.  Rank 0 prints one
    line {"probe_result": ...}.  A crash or a stall here costs the parent nothing but the probe."""
    dog = Watchdog(args.probe_watchdog, int(os.environ.get("RANK", "0")), args.gpus, args)
    dog.optional = True          # a stall ends this child quietly (code 0, no line): the parent sees no result
    args.p2p = False
    R = RankRun(args, dog)
    hip = R.hip
    legs = {}
    for name, sr, p2p in PROBE_LEGS:
        if os.environ.get("STAN_BENCH_TEST_HANG_PROBE", "") == str(R.rank) and p2p:   # test hooks
            time.sleep(3600)
        if os.environ.get("STAN_BENCH_TEST_CRASH_PROBE", "") == str(R.rank) and p2p:
            os.abort()
        dog.touch("transport probe: %s set-up" % name)
        R.ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, sr)
        R.ctx.set_option(hip.OPT_COMM_P2P, p2p)      # (a collective call: every rank makes it)
        for i in range(2):
            dog.touch("transport probe: %s capped solve %d" % (name, i + 1))
            rep_, prof_, _ = R.step(max_its=args.probe_its)
        R.sync()
        its = max(1, rep_["iterations"])
        enq = max(prof_["loop_iterations_enqueued"], 1)
        rows = R.gather([prof_["cg_ms"] / its,
                         prof_["comm_reduce_ms_total"] / max(prof_["comm_reduce_calls"], 1) * 1e3,
                         prof_["comm_halo_ms_total"] / max(prof_["comm_halo_calls"], 1) * 1e3,
                         prof_["loop_kernel_launches"] / enq, prof_["loop_collectives"] / enq, prof_["loop_stream_waits"] / enq,
                         float(rep_["iterations"]), rep_["rel_residual"]])
        legs[name] = {"ms_per_iteration": max(r_[0] for r_ in rows),
                      "ms_per_iteration_per_rank": [r_[0] for r_ in rows],
                      "reduction_us_per_call": [r_[1] for r_ in rows],
                      "halo_us_per_call": [r_[2] for r_ in rows],
                      "kernel_launches_per_iteration": rows[0][3],
                      "collectives_per_iteration": rows[0][4],
                      "stream_waits_per_iteration": rows[0][5],
                      "iterations": int(rows[0][6]), "rel_residual": rows[0][7],
                      "every_rank_same_residual_bits": len(set(r_[7] for r_ in rows)) == 1}
    dog.touch("transport probe: report")
    if R.rank == 0:
        print(json.dumps({"probe_result": probe_report(legs, args.probe_its)}), flush=True)
    dog.stop()
    R.ctx.set_option(hip.OPT_COMM_P2P, 0)
    R.close()


def run_probe_children(args, R, dog):
    """The measured ranks start the probe as FRESH child processes (one per rank, a process group of their own on a port
    of its own) and wait for them: whatever happens in there -- a stall, a GPU fault, a segfault in an IPC mapping,
    an RCCL abort -- happens to the children (in-process, a hard failure took the measured line with it).
    Returns rank 0's {"probe_result": ...} dict or None."""
    import signal
    import socket
    import subprocess
    torch, dist = R.torch, R.dist
    port = torch.zeros(1, dtype=torch.int64, device=R.ctl)
    if R.rank == 0:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port[0] = sk.getsockname()[1]
        sk.close()
    dist.broadcast(port, 0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}   # (the child ranks host their own store)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port.item())))
    keep = ["--gpus", str(args.gpus), "--size", str(args.n), "--eps", str(args.eps), "--etype", str(args.etype),
            "--knockout", str(args.knockout), "--probe-its", str(args.probe_its), "--probe-watchdog", str(args.probe_watchdog),
            "--placement-tries", "1", "--steps", "1", "--warmup", "0", "--no-cpu"]
    if args.mixed:
        keep.append("--mixed")
    if args.fixed48:
        keep.append("--fixed48")
    cmd = [sys.executable, os.path.abspath(__file__), "--probe-child"] + keep
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE if R.rank == 0 else subprocess.DEVNULL, text=True, env=env,
                            cwd=ROOT, start_new_session=True)
    bound = args.probe_watchdog * (len(PROBE_LEGS) + 1) + 240.0     # backstop; the children carry their own watchdog
    t0 = time.time()
    out = ""
    try:
        while True:
            try:
                out, _ = proc.communicate(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                dog.touch("transport probe (child processes)")
                if time.time() - t0 > bound:
                    raise
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)     # the child started above, nothing else
        except OSError:
            pass
        proc.wait()
        sys.stderr.write("bench.py: rank %d: probe child did not end within %.0f s and was killed\n" % (R.rank, bound))
        return None
    if proc.returncode != 0:
        sys.stderr.write("bench.py: rank %d: probe child ended with code %d; the measured line stands\n" % (R.rank, proc.returncode))
    if R.rank != 0:
        return None
    for ln in (out or "").splitlines():
        if ln.startswith("{") and '"probe_result"' in ln:
            try:
                return json.loads(ln)["probe_result"]
            except ValueError:
                return None
    return None


# ---- N = 1: the other single-GPU configurations behind the headline, each in a process of its own ----------------------------
def _child_json(cmd, timeout, env=None, marker='"metric"', cwd=None):
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd or ROOT, env=env,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return None, "timed out after %.0f s" % timeout
    for ln in out.splitlines():
        if ln.startswith("{") and marker in ln:
            try:
                return json.loads(ln), None
            except ValueError:
                pass
    return None, "rc %d, no line; stderr tail: %s" % (proc.returncode, (err or "")[-300:].replace("\n", " | "))


def _bench_leg(extra, timeout):
    d, why = _child_json([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu",
                          "--no-secondary"] + extra, timeout)
    if d is None:
        return {"error": why}
    c, r = d["config"], d["roofline"]
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "workload": c["workload"],
            "cg_iterations": c["cg_iterations"], "termination_type": c["termination_type"], "rel_residual": c["rel_residual"],
            "assemble_ms": c["assemble_ms"], "cg_ms": c["cg_ms"],
            "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_reduced_system_bytes",
                                           "bytes_per_launch", "avg_launch_ms", "launches")},
            "speedup_vs_cpu_at_workload": d.get("speedup_vs_cpu_at_workload")}


def _console_leg(n, timeout):
    """stan_solver --json on a generated n^3 STdb (GUI defaults: CG, tol 1e-6, alglib's merit stop on): the reference's
    console entry point end to end -- read, AssignDOF, BC tables, assembly + CG on the GPU, stress recovery, export of
    the results into the file (Solver.cs:18-217, 454-462) -- with its phase times."""
    import tempfile
    import numpy as np
    from stan_amd import host
    from stan_amd.cube import cube_bcs, cube_mesh
    xyz, conn = cube_mesh(n)
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, "HEX8_G2")
    spc, ld, f = cube_bcs(n)
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(tol=1e-6)
    tmp = tempfile.mkdtemp(prefix="stan_bench_")
    path = os.path.join(tmp, "cube%d.STdb" % n)
    try:
        d.write_stdb(path)
        size_in = os.path.getsize(path)
        del d
        exe = os.path.join(ROOT, "stan_amd", "bin", "stan_solver")
        t0 = time.perf_counter()
        line, why = _child_json([exe, "--json", path], timeout, marker='"t_wall_s"')
        wall = time.perf_counter() - t0
        if line is None:
            return {"error": why}
        line.update(workload="stan_solver --json on a generated %d^3 HEX8_G2 STdb (CG, tol 1e-6, merit stop on)" % n,
                    process_wall_s=wall, input_MB=size_in / 1e6, output_MB=os.path.getsize(path) / 1e6)
        return line
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_per_launch(csv_files, counter, kernel_regex=r"k_spmv<double, 1, 9>"):
    """(average counter value per working launch, working launches, dispatches) of one kernel from rocprofv3's
    *counter_collection.csv files.  A launch queued behind a converged solve returns at once: dispatches whose counter is
    below 5 % of the kernel's median are not launches that did work."""
    import csv
    import re
    vals = []
    for f in csv_files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter and re.search(kernel_regex, row.get("Kernel_Name", "")):
                vals.append(float(row["Counter_Value"]))
    if not vals:
        return None
    vals.sort()
    med = vals[len(vals) // 2]
    work = [v for v in vals if v >= 0.05 * med] if med > 0 else vals
    return sum(work) / len(work), len(work), len(vals)


def _pmc_leg(timeout_each):
    """HBM-side traffic of the dominant kernel MEASURED BY THIS RUN: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc
    WRITE_SIZE: separate passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over one step of the headline
    workload in a child process; per launch over the launches that did work (a launch queued behind a converged solve
    returns at once: counters below 5 % of the kernel's median), FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)."""
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="stan_pmc_", dir="/tmp")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-secondary"]
            line, why = _child_json(cmd, timeout_each, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            if line is None:
                return {"error": "%s pass: %s" % (counter, why)}
            res[counter] = pmc_per_launch(glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True), counter)
            if res[counter] is None:
                return {"error": "%s pass: no k_spmv<double, 1, 9> dispatch in the counter file" % counter}
        fetch, write = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
        return {"kernel": "k_spmv<double, 1, 9>", "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                "fetch_correction": 2.0, "launches": res["FETCH_SIZE"][1], "dispatches": res["FETCH_SIZE"][2],
                "traffic_bytes_per_launch": int((2.0 * fetch + write) * 1024)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def secondary_legs(args, dog):
    """BASELINE.json's other single-GPU configurations, measured by the driver's own run: each leg is a
    fresh child process with a bound of its own; a leg that fails leaves {"error": ...} and the headline untouched."""
    t_end = time.time() + args.secondary_budget
    legs = []

    def left():
        return max(5.0, t_end - time.time())
    plan = [("config 2: 100^3 fp64", lambda: _bench_leg(["--size", "100"], min(90.0, left()))),
            ("config 3: 200^3 fp64 (HBM-roofline SpMV run)", lambda: _bench_leg(["--size", "200"], min(150.0, left()))),
            ("148^3, FIXED-48 value stream", lambda: _bench_leg(["--size", "148", "--fixed48"], min(90.0, left()))),
            ("console driver end to end, 148^3", lambda: _console_leg(148, min(120.0, left()))),
            ("HBM traffic of k_spmv from PMC counters, 148^3 (two rocprofv3 passes)", lambda: _pmc_leg(min(90.0, left()))),
            ("k_recover (stress recovery) at 148^3",
             lambda: (lambda d, why: d if d is not None else {"error": why})(
                 *_child_json([sys.executable, os.path.join(ROOT, "tools", "recover_time.py"), "148", "10"], min(60.0, left()),
                              marker='"kernel"')))]
    for name, fn in plan:
        dog.touch("secondary leg: " + name)
        if time.time() > t_end:
            legs.append({"leg": name, "error": "skipped: the secondary budget (%.0f s) was spent" % args.secondary_budget})
            continue
        t0 = time.time()
        try:
            res = fn()
        except Exception as e:   # noqa: BLE001  (an optional extra must not cost the line)
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        res = dict(leg=name, seconds=time.time() - t0, **res)
        legs.append(res)
    return legs


def main():
    ensure_built()
    args = build_parser().parse_args()
    if args.one_process:
        return run_one_process(args)
    if args.probe_child:
        return probe_child_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    # armed after the imports: the first `import torch` on a fresh box pages the image in for a minute or two
    dog = Watchdog(args.watchdog, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), args)
    R = RankRun(args, dog)
    rank, world, job, ctx, dev, ctl, comm, conn = R.rank, R.world, R.job, R.ctx, R.dev, R.ctl, R.comm, R.conn
    step, sync, placement_budget = R.step, R.sync, R.placement_budget

    # STAN_BENCH_TEST_HANG_RANK: test hook (tests/test_gpu_sharded.py): that rank never starts its steps
    if os.environ.get("STAN_BENCH_TEST_HANG_RANK", "") == str(rank):
        dog.stop()
        time.sleep(3600)
    for i in range(args.warmup):
        dog.touch("warm-up step %d" % (i + 1))
        step()
    dog.touch("barrier before the timed steps")
    sync()
    t0 = time.perf_counter()
    spmv_ms = spmv_n = 0.0
    spmv2_ms = spmv2_n = 0.0
    asm_ms = cg_ms = 0.0
    red_ms = red_n = halo_ms = halo_n = 0.0
    for i in range(args.steps):
        dog.touch("timed step %d" % (i + 1))
        rep, prof, info = step()
        dog.touch("timed step %d done" % (i + 1), step_done=True)
        red_ms += prof["comm_reduce_ms_total"]; red_n += prof["comm_reduce_calls"]
        halo_ms += prof["comm_halo_ms_total"]; halo_n += prof["comm_halo_calls"]
        spmv_ms += prof["spmv_ms_total"]
        spmv_n += prof["spmv_launches"]
        spmv2_ms += prof["spmv2_ms_total"]
        spmv2_n += prof["spmv2_launches"]
        asm_ms += prof["assemble_ms"]
        cg_ms += prof["cg_ms"]
    dog.touch("barrier after the timed steps")
    sync()
    dt = time.perf_counter() - t0
    per_rank = None
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

    # sanity of the timed work (rank 0): converged to eps, true residual through an
    # independent product is checked in tests; here the solver's own report
    # context for the roofline: what a plain read-only reduction reaches on THIS device
    dev_read = None
    if rank == 0:
        probe = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GiB
        probe.sum(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            probe.sum()
        e1.record(); torch.cuda.synchronize()
        dev_read = probe.numel() * 8 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del probe
    ok = rep["terminationtype"] == 1 and rep["rel_residual"] <= args.eps
    out = None
    if rank == 0:
        avg_ms = spmv_ms / max(spmv_n, 1)
        # HBM traffic per launch from the committed PMC passes (tools/pmc_run.sh), if one exists
        # for this workload; counters cannot be collected from inside this process
        traffic = traffic_source = None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_spmv*.json"))):
            d = json.load(open(f))
            if (d.get("workload_n") == args.n and world == 1 and
                    d.get("value_stream", 0) == prof["value_stream"]):
                traffic = d["traffic_bytes_per_launch"]
                traffic_source = os.path.relpath(f, ROOT)
        achieved = prof["spmv_bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # SURVEY section 8d prices the REDUCED system (fixed DOFs squeezed out); this library keeps
        # them as identity rows, so a launch moves slightly more.  Both
        # fractions are reported; the x-clamped cube's reduced block count is (3n-2)(3n+1)^2.
        frac_reduced = csr_equiv = None
        if args.etype == 2 and world == 1 and avg_ms > 0 and args.knockout == 0:
            blocks_red = (3 * args.n - 2) * (3 * args.n + 1) ** 2
            blk_bytes = 40 if args.mixed else 60 if args.fixed48 else 76
            bytes_red = blocks_red * blk_bytes + job.n_red * 16 + (job.n_red // 3) * 4
            frac_reduced = bytes_red / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            # the same product priced as scalar CSR (fp64 values, int32 columns): 12 nnz + 20 N
            csr_equiv = (12 * 9 * blocks_red + 20 * job.n_red) / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC,
            "value": job.n_dof * args.steps / dt,
            "unit": "DOF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # non-default library options this number depends on (DESIGN.md sections 3, 4)
            "merit_stop": False, "placement_tries": args.placement_tries, "placement_max_bytes": placement_budget,
            "dtype": dtype_text(args),
            "data": "synthetic",
            "config": {"workload": "%d^3 HEX8_G%d cube%s, %d DOF, clamp %s, PointLoad (0,0,50) on "
                                   "x=n; fp64 Jacobi-scaled CG to %.0e" %
                                   (args.n, args.etype,
                                    " with %.0f %% of its elements knocked out (irregular mesh)" % (100 * args.knockout)
                                    if args.knockout > 0 else "", job.n_dof,
                                    "x=0" if args.etype == 2 else "x=0, y=0, z=0", args.eps),
                       "cg_loop": "single-reduction (Chronopoulos-Gear)" if args.single_reduce
                                  else "classic (alglib lincg recurrences)",
                       "loop_kernel_launches_per_iteration":
                           prof["loop_kernel_launches"] / max(prof["loop_iterations_enqueued"], 1),
                       "n_dof": job.n_dof, "n_reduced": job.n_red,
                       "blocks_3x3_rank0": info["n_blocks"], "cg_iterations": rep["iterations"],
                       "termination_type": rep["terminationtype"],
                       "rel_residual": rep["rel_residual"], "converged": bool(ok),
                       # reduced-precision streams: rel_residual is the FP64 residual of the returned point (one product
                       # on the fp64 values); the loop's own recurrence, the passes and the fp64 products it took
                       "rel_residual_recurrence": prof["rel_residual_recurrence"], "refine_passes": prof["refine_passes"],
                       "fp64_products_per_step": prof["fp64_products"],
                       "assemble_ms": asm_ms / args.steps, "cg_ms": cg_ms / args.steps,
                       "matrix_format": "BSELL-64 3x3 blocks (%s values + block cols as 16-bit offsets from per-slot "
                                        "bases -- one, or two where a slice mixes row lengths -- in %.1f %% of the slots, int32 in the rest)" %
                                        ("fp32" if args.mixed else "48-bit fixed-point" if args.fixed48
                                         else "fp64", 100.0 * prof["col_slots_packed"] / max(info["n_slots"], 1)),
                       # SELL-C-sigma: slots streamed per structural block - 1 (padded slots are streamed like real ones)
                       "ell_padding": info["n_slots"] * 64.0 / max(info["n_blocks"], 1) - 1.0,
                       "sell_sigma": info["sell_sigma"], "repacked_streams": "folded rows" if prof["repacked_streams"] else "none",
                       "folded_slots_over_padded_slots": info["folded_slots_permille"] / 1000.0 if info["folded_slots_permille"] else None,
                       "parallelism": "rows sharded x%d" % world,
                       "transport": ("one rank" if world == 1 else
                                     "RCCL %s (ncclGetVersion %d), communicator of %d ranks, this = rank %d%s" %
                                     ("" if comm["rccl_version"] else "stand-in", comm["rccl_version"],
                                      comm["comm_ranks"], comm["comm_rank"], ", peer to peer" if comm["p2p"] else "")),
                       "elements_on_rank0": int(conn.shape[0]),
                       # the block pool keeps K's arrays between steps; with tries > 1 the first
                       # assembly picks the fastest-streaming of several hipMalloc blocks
                       "placement_tries": args.placement_tries,
                       "placement_search": {"candidates_timed": prof["placement_candidates"],
                                            "probe_ms_kept": prof["placement_ms_best"],
                                            "probe_ms_slowest": prof["placement_ms_worst"],
                                            "vectors_moved_instead": prof["placement_moved_vectors"] == 1,
                                            "product_vectors_moved": prof["placement_moved_vectors"] == 2},
                       # SURVEY section 8d assembly bytes: coords + connectivity read, K written once
                       "assembly_GBs": (conn.shape[0] * (192 + 32) + info["n_slots"] * 64 * 72)
                                       / (asm_ms / args.steps * 1e-3) / 1e9 if asm_ms > 0 else None,
                       # the same section's flop figure: the dense B'DB costs ~75 kflop per G2 element (an eighth per G1
                       # element); the kernels exploit B's sparsity and recompute an element once per incident row, so
                       # this is the rate of the WORK DEFINED, not of the instructions issued
                       "assembly_GFLOPs_dense_equivalent": conn.shape[0] * (75e3 if args.etype == 2 else 75e3 / 8)
                                                           / (asm_ms / args.steps * 1e-3) / 1e9 if asm_ms > 0 else None,
                       "device_read_GBs": dev_read},
            "roofline": {"bound": "hbm", "kernel": "k_spmv (BSELL-64 SpMV + fused p.Ap)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # counters cannot be read from inside this process: the figure is the
                         # committed PMC pass of the same workload, not a measurement of this run
                         "traffic_source": traffic_source,
                         # the committed counter traffic moved in this run's launch time (what the memory
                         # system really did; the fraction above prices algorithmic bytes only)
                         "traffic_rate_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic and avg_ms > 0 else None,
                         "frac_reduced_system_bytes": frac_reduced,
                         "csr_equivalent_GBs": csr_equiv,
                         # bytes of the format actually streamed (packed columns: 74 B per block, not 76);
                         # frac_reduced_system_bytes prices SURVEY section 8d's 76-B formula instead
                         "bytes_per_launch": prof["spmv_bytes"], "avg_launch_ms": avg_ms,
                         "launches": int(spmv_n),
                         # refresh iterations: A p and A x from one matrix pass (not in the average)
                         "two_product_launches": int(spmv2_n),
                         "two_product_avg_ms": (spmv2_ms / spmv2_n if spmv2_n else None)},
        }
        if per_rank is not None:
            fr = [p[0] / HBM_PEAK_GBS for p in per_rank]
            out["roofline"]["per_rank"] = {"frac_min": min(fr), "frac_max": max(fr), "frac": fr,
                                           "avg_launch_ms": [p[1] for p in per_rank]}
            # stream time per exchange (events around each, profiling only): RCCL launch -> sums available
            out["config"]["exchange"] = {"allreduce_us_per_call": [p[2] for p in per_rank],
                                         "halo_us_per_call": [p[3] for p in per_rank],
                                         "allreduces_per_step": red_n / args.steps, "halo_exchanges_per_step": halo_n / args.steps,
                                         "halo_block_rows": [int(p[4]) for p in per_rank],
                                         "owned_block_rows": [int(p[5]) for p in per_rank]}
            out["config"]["ranks"] = rank_devices
            out["config"]["distinct_devices"] = len(set(d["pci_bus_id"] for d in rank_devices))
        if not args.no_cpu and world == 1:
            dog.stop()   # the CPU sample is bounded by its size, not by the watchdog
            base, base_all = cpu_baseline(args.cpu_n, args.eps)
            out["cpu_baseline"] = base
            out["cpu_baseline_all_cores"] = base_all
        else:
            out["cpu_baseline"] = None
        # the CPU port on the WORKLOAD itself (not the in-run sample): a committed run of
        # tests/golden/make_bench_mode_golden.py on a GPU box's host (8 min at 148^3: --cpu-n 148 repeats it here)
        at_wl = cpu_at_workload(args.n) if args.knockout == 0 and args.etype == 2 else None
        if at_wl:
            out["cpu_baseline_at_workload"] = at_wl
            out["speedup_vs_cpu_at_workload"] = out["value"] / at_wl["value"]
        if not ok:   # a step that did not reach eps is not a step of this metric
            out["value"] = None
            out["error"] = "CG ended with type %d at %.3e (> eps %.0e): no DOF/s reported" % (
                rep["terminationtype"], rep["rel_residual"], args.eps)
        dog.touch("cpu baseline done")
    # N > 1: the loop forms and transports side by side on a capped solve, behind the measurement, in child processes:
    # whatever happens in there (a stall, a GPU fault, an abort inside RCCL or an IPC mapping), the line measured above is
    # printed (rc 0).
    line = json.dumps(out) if rank == 0 else None
    if world > 1 and not args.no_p2p_probe and not args.p2p and ok:
        dog.held_line, dog.optional, dog.enabled = line, True, True
        dog.bound = float(args.probe_watchdog) * (len(PROBE_LEGS) + 1) + 300.0   # (the children carry the real bound)
        if args.watchdog <= 0:   # the probe is bounded even when the run was not
            import threading
            threading.Thread(target=dog._run, daemon=True).start()
        probe = None
        try:
            probe = run_probe_children(args, R, dog)
        except Exception as e:   # noqa: BLE001  (an optional extra must not cost the line)
            sys.stderr.write("bench.py: rank %d: transport probe given up: %s\n" % (rank, e))
        if rank == 0 and probe is not None:
            out["config"]["p2p_probe"] = probe
            out["config"]["recommended_transport"] = probe["recommended"]
            line = json.dumps(out)
        dog.held_line = line
    # N = 1, the default workload: BASELINE.json's other single-GPU configurations behind the headline (child processes)
    if (world == 1 and rank == 0 and ok and not args.no_secondary and args.n == 148 and args.etype == 2 and args.knockout == 0
            and not args.mixed and not args.fixed48 and args.max_its == 0):
        dog.held_line, dog.optional, dog.enabled = line, True, True
        dog.bound = args.secondary_budget + 120.0
        try:
            # the context's pool gives its parked blocks back first: the legs are processes of their own on this GPU
            ctx.set_option(R.hip.OPT_POOL_MAX_BYTES, 0)
            out["secondary"] = secondary_legs(args, dog)
            for leg in out["secondary"]:   # counter traffic measured by this run replaces the committed figure
                if leg.get("traffic_bytes_per_launch") and "error" not in leg:
                    t_, ms_ = leg["traffic_bytes_per_launch"], out["roofline"]["avg_launch_ms"]
                    out["roofline"].update(traffic=t_, traffic_source="PMC passes of THIS run (secondary leg: rocprofv3 --pmc FETCH_SIZE / "
                                           "--pmc WRITE_SIZE over one step of the same workload in a child process; FETCH doubled)",
                                           traffic_rate_GBs=t_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None,
                                           traffic_over_algorithmic=t_ / out["roofline"]["bytes_per_launch"])
            line = json.dumps(out)
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench.py: secondary legs given up: %s\n" % e)
        dog.held_line = line
    if rank == 0:
        print(line, flush=True)
    # the measurement is out: a teardown that stalls (a peer that is gone) ends quietly with code 0, never a second line
    dog.held_line, dog.optional = None, True
    dog.touch("teardown")
    R.close()
    dog.stop()
    if rank == 0 and not ok:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
