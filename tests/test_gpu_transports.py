"""How the ranks of a sharded solve exchange: peer to peer (mailboxes + arrival counters, STAN_OPT_COMM_P2P) against the
RCCL-shaped path bit for bit, the hardware-queue requirement when ranks share a device, a failing rank, the result
segments of the one-process handle, the console driver on two ranks peer to peer.  (The reference is one process on
one CPU: Solver.cs:18-69; SURVEY.md section 8e is the design these follow.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from stan_amd import problem
from stan_amd.cube import cube_mesh, revolved_mesh
from tests.conftest import fake_rccl_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
U_TOL = 1e-6
K_TOL = 1e-13
OPT_ASSEMBLY_MODE = 5
OPT_FOLD = 19
OPT_SELL_SIGMA, OPT_MERIT = 17, 1


def _p2p_run(tmp_path, spec, nranks, wait_mode=None, fake_mode="sync"):
    out = str(tmp_path / "p2p.npz")
    env = dict(os.environ, STAN_RCCL_LIB=FAKE, **fake_rccl_env(fake_mode))
    env["GPU_MAX_HW_QUEUES"] = str(2 * nranks + 4)
    if wait_mode is not None:
        env["STAN_P2P_WAIT_MODE"] = str(wait_mode)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), spec, str(nranks), out],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0 and "P2P_WORKER_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("spec,nranks,wait_mode,fake_mode",
                         [("12", 2, None, "sync"), ("12", 3, None, "sync"), ("perf:10:0.3", 4, None, "sync"), ("3", 4, None, "sync"),
                          ("12", 3, 0, "sync"), ("12", 3, 2, "sync"), ("perf:10:0.3", 4, 2, "sync"),
                          ("12", 3, None, "async"), ("perf:10:0.3", 4, None, "async"), ("3", 4, None, "async")])
def test_peer_to_peer_exchanges_give_the_bits_of_the_rccl_path(built_libs, tmp_path, spec, nranks, wait_mode, fake_mode):
    """STAN_OPT_COMM_P2P: the sharded CG's reductions and halo exchanges without one collective launch
    (mailboxes + arrival counters + stream waits, p2p.hip).  The partials are added in rank order, like
    the stand-in transport's all-reduce: U, the iteration count and the termination code are IDENTICAL,
    classic and single-reduction loop, fp64 and FIXED-48 stream, a MaxIts stop inside a refresh cycle,
    folded and unfolded reductions; "3" on 4 ranks = three ranks own no rows.  wait_mode: how a stream waits for
    an arrival count (STAN_P2P_WAIT_MODE): default a one-wave polling kernel, 0 hipStreamWaitValue64, 2 the
    reductions are polled by the consuming kernel itself (no wait launch for them at all).  fake_mode "async": the
    RCCL-shaped legs run over the stream-ordered stand-in -- the same bits again."""
    d = _p2p_run(tmp_path, spec, nranks, wait_mode, fake_mode)
    for loop in ("classic", "sr"):
        for prec in ("f64", "fx48", "cap"):
            a, b = "rccl_%s_%s" % (loop, prec), "p2p_%s_%s" % (loop, prec)
            assert np.array_equal(d["rep_" + a], d["rep_" + b]), (a, d["rep_" + a], d["rep_" + b])
            assert np.array_equal(d["U_" + a], d["U_" + b]), a
            assert np.array_equal(d["U_" + a], d["U_rccl2_%s_%s" % (loop, prec)])      # and again: the counters keep counting
            assert np.array_equal(d["U_" + b], d["U_p2p2_%s_%s" % (loop, prec)])
            coll, waits, launches, its = (int(v) for v in d["coll_" + b][:4])
            assert coll == 0 and waits > 0, (b, coll, waits)
            rc, rw = (int(v) for v in d["coll_" + a][:2])
            assert rc > 0 and rw == 0
            if prec != "cap" and its > 20:
                # classic: 2 reduction waits (+ 1 halo wait on a rank with neighbours) per iteration; single reduction: 1 (+ 1)
                per_it = waits / its
                lo, hi = (1.9, 3.4) if loop == "classic" else (0.95, 2.4)
                if wait_mode == 2:     # only the halo waits are launches (rank 0 has one neighbour: ~1.1 per iteration)
                    lo, hi = 0.9, 1.4
                assert lo <= per_it <= hi, (b, per_it)
        assert int(d["rep_rccl_%s_cap" % loop][0]) == 5 and int(d["rep_rccl_%s_cap" % loop][1]) == 37
        assert int(d["rep_rccl_%s_f64" % loop][0]) in (1, 7)
    for fold in (1, 0):
        assert np.array_equal(d["U_rccl_nomerit_fold%d" % fold], d["U_p2p_nomerit_fold%d" % fold])
        assert np.array_equal(d["rep_rccl_nomerit_fold%d" % fold], d["rep_p2p_nomerit_fold%d" % fold])
    assert np.array_equal(d["U_p2p_nomerit_fold0"], d["U_p2p_nomerit_fold1"])


def test_peer_to_peer_needs_a_hardware_queue_per_stream_when_ranks_share_a_device(built_libs):
    """Two ranks on ONE device share its hardware queues; a stream wait blocks the queue it sits in
    (profiles/r03/waitvalue_probe_default_hw_queues.txt: deadlock).  The option is refused with the reason
    instead of hanging; on distinct devices nothing is shared."""
    code = r'''
import sys
sys.path.insert(0, %r)
import torch
from stan_amd import hip
ctx = hip.Context(devices=[0, 0])
try:
    ctx.set_option(hip.OPT_COMM_P2P, 1)
    print("NOERROR")
except hip.StanHipError as e:
    print("ERR", e.code, "GPU_MAX_HW_QUEUES" in str(e))
ctx.set_option(hip.OPT_COMM_P2P, 0)
one = hip.Context(0)
try:
    one.set_option(hip.OPT_COMM_P2P, 1)
    print("NOERROR")
except hip.StanHipError as e:
    print("ERR1", e.code)
one.close(); ctx.close()
''' % ROOT
    env = dict(os.environ, STAN_RCCL_LIB=FAKE)
    env.pop("GPU_MAX_HW_QUEUES", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    assert "ERR -8 True" in p.stdout and "ERR1 -8" in p.stdout, p.stdout


def _run_script(code, env, timeout):
    """A child python process whose output survives a hang: on timeout the process is killed and what it had
    printed so far is shown (the scripts print with flush)."""
    p = subprocess.Popen([sys.executable, "-u", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=env, cwd=ROOT)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        raise AssertionError("child timed out after %d s; stdout so far:\n%s\nstderr:\n%s" % (timeout, out[-3000:], err[-3000:]))
    return p.returncode, out, err


def test_a_failing_rank_does_not_hang_the_peer_to_peer_loop(built_libs):
    """The peers of a rank that fails are blocked in stream waits / the host barrier of the publication, not
    in RCCL: the group releases every wait (the arrival counters jump), wakes the barrier, the surviving
    loops see the flag at their next poll."""
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from stan_amd import hip, problem
job = problem.cube_job(10)
ctx = hip.Context(devices=[0, 0, 0])
print("CONTEXT", flush=True)
ctx.set_option(hip.OPT_COMM_P2P, 1)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
print("ASSEMBLED", flush=True)
U, rep = K.cg_solve(job.F, 1e-6)
print("SOLVED", rep, flush=True)
assert rep["terminationtype"] == 1
os.environ["STAN_TEST_FAIL_RANK"] = "1"
t0 = time.time()
try:
    K.cg_solve(job.F, 1e-8)
    print("NOERROR", flush=True)
except hip.StanHipError as e:
    print("ERR1", e.code, "rank 1" in str(e), "%%.1f" %% (time.time() - t0), flush=True)
del os.environ["STAN_TEST_FAIL_RANK"]
try:
    K.cg_solve(job.F, 1e-8)
    print("NOERROR", flush=True)
except hip.StanHipError as e:
    print("ERR2", e.code, "aborted" in str(e), flush=True)
K.free()
print("FREED", flush=True)
ctx.close()
print("CLOSED", flush=True)
''' % ROOT
    env = dict(os.environ, STAN_RCCL_LIB=FAKE, GPU_MAX_HW_QUEUES="12")
    rc, out, err = _run_script(code, env, 150)
    assert rc == 0, out[-2000:] + err[-3000:]
    l1 = [l for l in out.splitlines() if l.startswith("ERR1")][0].split()
    assert l1[1] == "-1" and l1[2] == "True" and float(l1[3]) < 60.0
    l2 = [l for l in out.splitlines() if l.startswith("ERR2")][0].split()
    assert l2[1] == "-7" and l2[2] == "True"
    assert "CLOSED" in out


def test_group_solve_leaves_every_rank_its_own_segment(built_libs, oracle, tmp_path):
    """The one-process handle gathers nothing: each rank uploads its entries of F and copies its entries of
    U into the caller's buffer (stan_matrix::u0/u1); the assembled vector is the oracle's."""
    out = str(tmp_path / "multi.npz")
    env = dict(os.environ, STAN_RCCL_LIB=FAKE)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_worker.py"), "9", "3", out],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    d = np.load(out)
    job = problem.cube_job(9, jitter=0.05)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    Uo, rep = oracle.cg(A, job.F, 1e-6)
    assert np.abs(d["U"] - Uo).max() <= 1e-4 * np.abs(Uo).max() and np.all(d["U"][job.F != 0] != 0)


def test_console_driver_on_two_ranks_peer_to_peer(built_libs, tmp_path):
    """stan_solver --devices 0,0 --p2p: the reference's console entry point (Solver.cs:18-69) driving two ranks
    whose CG exchanges go peer to peer; the result file is byte-identical to the RCCL-path run's (rank-ordered
    sums on both transports), and --p2p is refused with the reason when the ranks share a device without a
    hardware queue per stream."""
    from stan_amd import host
    from stan_amd.cube import cube_bcs, cube_mesh
    exe = os.path.join(ROOT, "stan_amd", "bin", "stan_solver")
    n = 8
    xyz, conn = cube_mesh(n, jitter=0.1)
    out_bytes = {}
    for mode in ("rccl", "p2p"):
        d = host.Db()
        ne = conn.shape[0]
        d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
        d.add_material(1, "Steel", 210000.0, 0.3)
        d.assign_part(1, 1, "HEX8_G2")
        spc, ld, f = cube_bcs(n)
        d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
        d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
        d.set_analysis(tol=1e-9)
        path = str(tmp_path / (mode + ".STdb"))
        d.write_stdb(path)
        env = dict(os.environ, STAN_RCCL_LIB=FAKE, GPU_MAX_HW_QUEUES="8")
        args = [exe, "--devices", "0,0", "--json"] + (["--p2p"] if mode == "p2p" else []) + [path]
        out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "NORMAL" in out.stdout
        out_bytes[mode] = open(path, "rb").read()
    assert out_bytes["rccl"] == out_bytes["p2p"]
    env = dict(os.environ, STAN_RCCL_LIB=FAKE)
    env.pop("GPU_MAX_HW_QUEUES", None)
    out = subprocess.run([exe, "--devices", "0,0", "--p2p", str(tmp_path / "p2p.STdb")], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "GPU_MAX_HW_QUEUES" in (out.stdout + out.stderr)
