"""bench_launch.py -- what surrounds bench.py's measurement when more than one GPU is involved (and what keeps a run from
hanging): the progress watchdog, the self-launcher of `python bench.py --gpus N`, the one-process form
(stan_hip_init_multi), and the transport probe {classic, single-reduction} x {RCCL, peer to peer} that runs in child
processes behind a multi-rank measurement.  bench.py holds the N = 1 path and the JSON line; bench_legs.py the CPU
baseline and the secondary single-GPU legs.  Nothing here is timed as part of `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
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
        self.enabled = False
        self._thread = None
        if bound_s > 0:
            self.arm()

    def arm(self):
        """(Re)start the checking thread: after `stop` (the unbounded CPU sample) or for a run that was started with
        --watchdog 0 but whose optional work must be bounded all the same."""
        import threading
        self.enabled = True
        self.t_last = time.time()
        if self._thread is None or not self._thread.is_alive():
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

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

def hold_line(dog, line):
    """once the headline is measured, optional work behind it (secondary legs, the transport probe)
    must not be able to cost it.  The finished line is (1) what the watchdog prints if that work stalls (`dog.held_line`),
    (2) written to a side file at once (STAN_BENCH_SIDE_FILE, default /tmp/stan_bench_line_<pid>.json) and (3) printed by
    a SIGTERM / SIGINT handler -- an outer `timeout` or a Ctrl-C ends the run with the measured line on stdout, code 0.
    Called again whenever the line grows (a leg attached)."""
    import signal
    dog.held_line = line
    if line is None:
        return
    path = os.environ.get("STAN_BENCH_SIDE_FILE") or "/tmp/stan_bench_line_%d.json" % os.getpid()
    try:
        with open(path + ".tmp", "w") as f:
            f.write(line + "\n")
        os.replace(path + ".tmp", path)
    except OSError:
        pass
    dog._side_file = path
    if getattr(dog, "_signals", False):
        return
    dog._signals = True

    def on_signal(signum, _frame):
        try:
            from bench_legs import kill_active_child
            kill_active_child()
        except Exception:   # noqa: BLE001
            pass
        try:
            if dog.held_line is not None:
                sys.stdout.write(dog.held_line + "\n")
                sys.stdout.flush()
            sys.stderr.write("bench.py: signal %d while optional work ran behind the measurement; the measured line stands\n" % signum)
        finally:
            os._exit(0)
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sg, on_signal)
        except (ValueError, OSError):   # not the main thread
            pass


def release_line(dog):
    """The line is on stdout: the default side file (not one the caller named) has served its purpose."""
    path = getattr(dog, "_side_file", None)
    if path and not os.environ.get("STAN_BENCH_SIDE_FILE"):
        try:
            os.remove(path)
        except OSError:
            pass


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
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH] + list(argv)
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


# ---- N > 1: the loop forms and transports side by side, in a process group of their own -------------------------------------
# the RCCL legs first: should a peer-to-peer leg stall or crash on a node (its IPC mappings have only ever met one GPU), the
# report of the two RCCL forms is already out (probe_child_main prints an interim result, the parent takes the last it sees)
PROBE_LEGS = (("classic_rccl", 0, 0), ("single_reduce_rccl", 1, 0), ("classic_p2p", 0, 1), ("single_reduce_p2p", 1, 1))


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
    missing = [name for name, _, _ in PROBE_LEGS if name not in legs]   # (an interim report: the legs that had run when it was printed)
    out = {"capped_at_iterations": capped_at, "legs": legs,
           "same_residual_bits_classic": (legs["classic_rccl"]["rel_residual"] == legs["classic_p2p"]["rel_residual"]
                                          if "classic_p2p" in legs else None),
           "agrees_with_classic_rccl": ok,
           "time_over_classic_rccl": {k: v["ms_per_iteration"] / base["ms_per_iteration"] for k, v in legs.items()},
           "fastest": best, "recommended": "%s (%s)" % (rec, opts[rec])}
    if missing:
        out["legs_not_run"] = missing
    return out


def probe_child_main(args, RankRun):
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
        if R.rank == 0 and name == "single_reduce_rccl":   # the interim report: whatever the peer-to-peer legs do, this much is out
            print(json.dumps({"probe_result": probe_report(legs, args.probe_its)}), flush=True)
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
            "--placement-tries", "1", "--pool-fraction", "0.2", "--steps", "1", "--warmup", "0", "--no-cpu"]
    if args.mixed:
        keep.append("--mixed")
    if args.fixed48:
        keep.append("--fixed48")
    cmd = [sys.executable, BENCH, "--probe-child"] + keep
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
        try:
            out, _ = proc.communicate(timeout=10.0)    # (what it had printed: the interim report of the RCCL legs, perhaps)
        except Exception:   # noqa: BLE001
            out = ""
        sys.stderr.write("bench.py: rank %d: probe child did not end within %.0f s and was killed\n" % (R.rank, bound))
    if proc.returncode not in (0, None):
        sys.stderr.write("bench.py: rank %d: probe child ended with code %s; the measured line stands\n" % (R.rank, proc.returncode))
    if R.rank != 0:
        return None
    res = None
    for ln in (out or "").splitlines():     # the LAST report counts: interim (RCCL legs) or final (all four)
        if ln.startswith("{") and '"probe_result"' in ln:
            try:
                res = json.loads(ln)["probe_result"]
            except ValueError:
                pass
    return res
