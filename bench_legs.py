"""bench_legs.py -- the legs bench.py attaches to its line, none of them inside the timed region: the CPU baseline (the
oracle on a bounded sample, in-process) and, behind an N = 1 headline, BASELINE.json's other single-GPU configurations as
CHILD PROCESSES with a bound each (`secondary`): 100^3, 200^3, 148^3 FIXED-48, the console driver end to end, the PMC
traffic passes, k_recover, and 400^3 (fp64, then FIXED-48 on the same resident model).  A leg that fails leaves
{"error": ...}; the headline and its exit code are untouched."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")
_active = []   # the child process a leg is waiting for (bench_launch.hold_line's signal handler ends exactly that group)


def kill_active_child():
    import signal
    for p in list(_active):
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass


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


# ---- N = 1: the other single-GPU configurations behind the headline, each in a process of its own ----------------------------
def _child_json(cmd, timeout, env=None, marker='"metric"', cwd=None):
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd or ROOT, env=env,
                            start_new_session=True)
    _active.append(proc)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return None, "timed out after %.0f s" % timeout
    finally:
        _active.remove(proc)
    for ln in out.splitlines():
        if ln.startswith("{") and marker in ln:
            try:
                return json.loads(ln), None
            except ValueError:
                pass
    return None, "rc %d, no line; stderr tail: %s" % (proc.returncode, (err or "")[-300:].replace("\n", " | "))


def _leg_summary(d):
    c, r = d["config"], d["roofline"]
    out = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "workload": c["workload"],
           "cg_iterations": c["cg_iterations"], "termination_type": c["termination_type"], "rel_residual": c["rel_residual"],
           "refine_passes": c.get("refine_passes"), "fp64_products_per_step": c.get("fp64_products_per_step"),
           "assemble_ms": c["assemble_ms"], "cg_ms": c["cg_ms"],
           "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_reduced_system_bytes",
                                              "stream_GBs", "frac_of_stream", "bytes_per_launch", "avg_launch_ms", "launches")},
           "speedup_vs_cpu_at_workload": d.get("speedup_vs_cpu_at_workload")}
    if d.get("error"):
        out["error"] = d["error"]
    return out


def _bench_leg(extra, timeout, steps=2, warmup=1):
    d, why = _child_json([sys.executable, BENCH, "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu",
                          "--no-secondary"] + extra, timeout)
    return _leg_summary(d) if d is not None else {"error": why}


def _host_mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _n400_leg(timeout):
    """BASELINE.json configs[4]'s size: the 400^3 cube (193 M DOF, 1.75 G blocks, 126 GB of fp64 values),
    HEX8_G2, `--steps 1 --warmup 0` -- ONE cold step: the placement search and the first allocations are inside it --
    first on the fp64 stream, then on the FIXED-48 stream of the same resident model (one host set-up, ~90 s of mesh and
    AssignDOF, for both).  Returns two entries: DOF/s, SpMV fraction, refinement passes and the FP64 residual of each.
    (The config as NAMED -- fp32 matrix + HEX8_G1 -- is ill-posed at this size, profiles/r02/CONFIG5.md; its halves are
    tests/test_gpu_configs.py.)"""
    avail_gb = _host_mem_available_gb()
    if avail_gb < 48:   # (the child's mesh, DOF tables and incidence lists take ~20 GB of host memory: never drive a box out of memory)
        return [{"leg": "config 5's size: 400^3", "error": "skipped: %.0f GB of host memory available, the 400^3 set-up wants 48" % avail_gb}]
    d, why = _child_json([sys.executable, BENCH, "--gpus", "1", "--size", "400", "--steps", "1", "--warmup", "0", "--no-cpu",
                          "--no-secondary", "--then-fixed48", "--watchdog", str(int(timeout))], timeout)
    if d is None:
        return [{"leg": "config 5's size: 400^3 fp64", "error": why}]
    legs = [dict(leg="config 5's size: 400^3 fp64 (one cold step)", **_leg_summary(d))]
    if "then_fixed48" in d:
        x = dict(d["then_fixed48"])
        x.setdefault("error", None)
        legs.append(dict(leg="400^3, FIXED-48 value stream (same resident model, one step)", **_leg_summary(x)))
        if legs[-1].get("error") is None:
            legs[-1].pop("error", None)
    return legs


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
                   sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--no-cpu", "--no-secondary"]
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


def secondary_legs(args, dog, attach=None):
    """BASELINE.json's other single-GPU configurations, measured by the driver's own run:
    each leg is a fresh child process with a bound of its own; a leg that fails leaves {"error": ...} and the headline
    untouched.  `attach(leg)` is called as each leg ends (bench.py keeps its held line current); returns the list."""
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
                              marker='"kernel"'))),
            # last and longest: ~90 s of host set-up + two solves of ~85 s; skipped unless ~5 min of the budget are left
            ("config 5's size: 400^3, fp64 then FIXED-48", lambda: _n400_leg(min(420.0, left())))]
    for name, fn in plan:
        dog.touch("secondary leg: " + name)
        need = 300.0 if name.startswith("config 5") else 0.0
        if time.time() + need > t_end:
            res = [{"leg": name, "error": "skipped: %.0f s of the secondary budget (%.0f s) left" % (max(0.0, t_end - time.time()), args.secondary_budget)}]
        else:
            t0 = time.time()
            try:
                res = fn()
            except Exception as e:   # noqa: BLE001  (an optional extra must not cost the line)
                res = {"error": "%s: %s" % (type(e).__name__, e)}
            res = [dict(leg=name, seconds=time.time() - t0, **res)] if isinstance(res, dict) else \
                  [dict(r, seconds=time.time() - t0) for r in res]
        for r in res:
            legs.append(r)
            if attach:
                attach(r)
    return legs
