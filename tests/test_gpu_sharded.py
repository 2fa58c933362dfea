"""The sharded path END TO END on the single GPU of the test box: several ranks (processes)
share GPU 0, RCCL is replaced by tests/fake_rccl (shared memory + host-staged copies, because
real RCCL refuses two ranks on one device).  Everything else is the product: partition, shard
assembly, halo plan, the CG loop with its exchanges, the two-stream overlap, the result gather,
and bench.py's N > 1 code path."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

from stan_amd import problem
from tests.conftest import fake_rccl_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
ASYNC_BANNER = "fake_rccl: asynchronous mode"


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _torchrun(nproc, script_args, env_extra, timeout=300, attempts=2):
    """Several ranks sharing ONE GPU through the shared-memory RCCL stand-in (which gives up after
    60 s of waiting with a dump of its counters).  Many processes time-slicing one device can
    be slow for reasons that have nothing to do with the code under test (see the docstring of
    test_sharded_solve_matches_oracle), so a timed-out attempt is killed as a process group,
    its output shown, and repeated once."""
    for attempt in range(1, attempts + 1):
        env = dict(os.environ, STAN_RCCL_LIB=FAKE, **env_extra)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_port())] + script_args
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                                cwd=ROOT, start_new_session=True)   # own process group: launcher + ranks
        try:
            out, err = proc.communicate(timeout=timeout)
            return subprocess.CompletedProcess(cmd, proc.returncode, out, err)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)                      # exactly the group started above
            out, err = proc.communicate()
            print("sharded run timed out (attempt %d):\n%s" % (attempt, (err or "")[-3000:]))
            if attempt == attempts:
                raise


@pytest.mark.parametrize("world,overlap,spec", [(2, 1, "12"), (3, 1, "12"), (3, 0, "12"),
                                                (4, 1, "fuzz:124"), (4, 1, "fuzz:101"), (3, 1, "rev:3")])
def test_sharded_solve_matches_oracle(built_libs, oracle, tmp_path, world, overlap, spec, fake_mode):
    """Both modes of the stand-in (round 6): "sync" drains every stream on the host, "async" enqueues the exchanges on
    the caller's stream the way RCCL does, so the two-stream overlap really overlaps.
    fuzz:124 = 383 shuffled nodes on 4 ranks (1, 3, 2, 2 neighbours); fuzz:101 = 191 nodes on
    4 ranks, the first of which owns no rows.  More processes than that on the one GPU next to the
    pytest process's own context make every kernel launch crawl (device time-slicing), so the
    6-rank (4 neighbours) and 7-rank (four empty ranks) runs of the same jobs live in
    tools/shard_loop.sh, where they take 4 s each."""
    assert os.path.exists(FAKE), "run __graft_entry__.build()"
    out = _torchrun(world, [os.path.join(ROOT, "tests", "sharded_worker.py"), spec, str(tmp_path), str(overlap)],
                    fake_rccl_env(fake_mode))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (ASYNC_BANNER in out.stderr) == (fake_mode == "async")
    # (the stand-in says where its mailboxes are: the receivers' device memory through HIP IPC -- what it does on this pool --
    # or, where no IPC mapping can be had, host memory; both forms give the same bits: test_the_two_modes_...)
    assert fake_mode != "async" or "mailboxes in " in out.stderr
    if spec.startswith("fuzz:"):
        from tests import fuzz
        job = fuzz.random_job(int(spec[5:]))
    elif spec.startswith("rev:"):     # round 4: 72 sectors, shuffled wire order: collapsed hexes, 288 incidences on the axis
        from tests import fuzz
        job = fuzz.random_revolved_job(int(spec[4:]))
    else:
        job = problem.cube_job(int(spec), jitter=0.05)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    Uo, rep = oracle.cg(A, job.F, 1e-6)   # above the type-7 floor (1.7e-7 here)
    Ux, _ = oracle.cg(A, job.F, 1e-12)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    rows = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(d["term"]) == rep["terminationtype"] == 1
        assert abs(int(d["its"]) - rep["iterations"]) <= max(3, rep["iterations"] // 20)
        assert np.abs(d["U"] - Uo).max() <= 1e-4 * np.abs(Uo).max()      # two eps = 1e-6 solves
        assert np.abs(d["Um"] - Ux).max() <= 1e-2 * np.abs(Ux).max()     # fp32 matrix, eps 1e-5
        assert np.abs(d["Ux"] - d["U"]).max() <= 1e-6 * np.abs(Uo).max()  # FIXED-48 stream, same eps (1e-6): both within kappa eps
        assert abs(int(d["its_x"]) - int(d["its"])) <= 1
        assert np.array_equal(d["U"], r0["U"])                            # every rank gets the same U
        rows.append(d["rows"])
    assert rows[0][0] == 0 and rows[-1][1] == job.xyz.shape[0]
    assert all(r[2] > 0 for r in rows if r[1] > r[0])     # every rank that owns rows has a halo


@pytest.mark.parametrize("n,world,p2p,mode", [(100, 2, False, "sync"), (100, 2, False, "async"), (100, 4, False, "async"),
                                              (100, 2, True, "sync"), (148, 2, False, "sync"), (148, 2, False, "async"),
                                              (148, 3, True, "async")])
def test_sharded_bench_mode_against_the_oracle_fixture(built_libs, tmp_path, n, world, p2p, mode):
    """BASELINE config 4 in miniature: the rows of the 100^3 cube (config 2's size, 3 M DOF) and of the 148^3 cube (the
    headline's 10 M DOF) partitioned over 2 / 3 / 4 ranks -- here all on the one GPU, over the test transport, RCCL-shaped
    exchanges and peer-to-peer mailboxes -- in bench mode (merit stop off, 1e-8) against the ORACLE's committed answer
    (tests/golden/bench_mode_<n>.npz): iterations within 2,
    max |dU| / max |U| <= 1e-9 at the 4096 sampled DOFs, the norms; every rank returns the same bits.  mode: the
    stand-in's synchronous or stream-ordered form (the peer-to-peer legs use it for set-up and the gather only)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_mode_%d.npz" % n))
    args = [os.path.join(ROOT, "tests", "sharded_worker.py"), "bench:%d" % n, str(tmp_path), "1"] + (["p2p"] if p2p else [])
    out = _torchrun(world, args, dict(fake_rccl_env(mode), **({"GPU_MAX_HW_QUEUES": str(2 * world + 4)} if p2p else {})))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    um = float(g["u_max"])
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(d["term"]) == int(g["terminationtype"]) == 1
        assert abs(int(d["its"]) - int(g["iterations"])) <= 2, (int(d["its"]), int(g["iterations"]))
        assert d["U"].shape[0] == int(g["n_red"])
        assert np.abs(d["U"][g["idx"]] - g["U"]).max() <= 1e-9 * um
        assert abs(np.sqrt(d["U"] @ d["U"]) - float(g["u_l2"])) <= 1e-9 * float(g["u_l2"])
        assert np.array_equal(d["U"], r0["U"])
        assert d["rows"][2] > 0                                  # a halo on every rank


@pytest.mark.parametrize("world,spec", [(3, "12"), (4, "fuzz:124")])
def test_the_two_modes_of_the_stand_in_give_the_same_bits(built_libs, tmp_path, world, spec):
    """the sharded solve over the synchronous stand-in and over the stream-ordered one -- fp64,
    fp32 copy, FIXED-48, the refinement passes of sharded_worker.py -- returns IDENTICAL bits on every rank: both add
    the ranks' partials in rank order, and the product orders its streams itself instead of leaning on the host
    drains of the synchronous mode."""
    res = {}
    # "async-late": every message's data lands 300 us after it was announced (FAKE_RCCL_ASYNC_DELAY_US) -- a kernel that
    # read its halo without waiting for the exchange on its stream would read the previous iteration's
    # "async-hostboxes": the fallback of a host without IPC mappings (FAKE_RCCL_ASYNC_HOST_BOXES: mailboxes in shared host memory)
    modes = ("sync", "async", "async-late") + (("async-hostboxes",) if world == 3 else ())
    for mode in modes:
        d = tmp_path / mode
        d.mkdir()
        env = dict(fake_rccl_env(mode.split("-")[0]), **({"FAKE_RCCL_ASYNC_DELAY_US": "300"} if mode == "async-late" else
                                                         {"FAKE_RCCL_ASYNC_HOST_BOXES": "1"} if mode == "async-hostboxes" else {}))
        out = _torchrun(world, [os.path.join(ROOT, "tests", "sharded_worker.py"), spec, str(d), "1"], env)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        assert (ASYNC_BANNER in out.stderr) == (mode != "sync")
        res[mode] = [np.load(str(d / ("rank%d.npz" % r))) for r in range(world)]
    for other in modes[1:]:
        for a, b in zip(res["sync"], res[other]):
            assert int(a["its"]) == int(b["its"]) and int(a["its_x"]) == int(b["its_x"]) and int(a["term"]) == int(b["term"])
            for k in ("U", "Um", "Ux"):
                assert np.array_equal(a[k], b[k]), (other, k)


def _broken_library_run(tmp_path, tag, n, env_extra):
    """bench:<n> on 2 ranks with the library that lacks the stream wait; True = it still reproduces the oracle fixture."""
    broken = os.path.join(ROOT, "stan_amd", "csrc", "build_broken", "libstan_hip_no_overlap_wait.so")
    assert os.path.exists(broken), "make -C stan_amd/csrc broken"
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_mode_%d.npz" % n))
    um = float(g["u_max"])
    d = tmp_path / tag
    d.mkdir()
    try:     # (a loop that lost its way ends at 3000 iterations with type 5; the fixture runs take 900 / 1334)
        out = _torchrun(2, [os.path.join(ROOT, "tests", "sharded_worker.py"), "bench:%d" % n, str(d), "1"],
                        dict(env_extra, STAN_HIP_LIB=broken, SHARDED_WORKER_MAXITS="3000"), timeout=240, attempts=1)
        ok = out.returncode == 0
    except subprocess.TimeoutExpired:
        ok = False
    if ok:
        for r in range(2):
            x = np.load(str(d / ("rank%d.npz" % r)))
            ok = ok and int(x["term"]) == 1 and abs(int(x["its"]) - int(g["iterations"])) <= 2 \
                and bool(np.abs(x["U"][g["idx"]] - g["U"]).max() <= 1e-9 * um)
    print("library without the stream wait, %d^3 on 2 ranks, %s: %s" % (n, tag, "passes the fixture" if ok else "CAUGHT"))
    return ok


def test_a_serialising_stand_in_hides_a_missing_stream_wait_and_the_stream_ordered_one_does_not(built_libs, tmp_path):
    """VERDICT r05 item 1, shown on a DELIBERATELY BROKEN library (stan_amd/csrc/lab/drop_overlap_wait.patch ->
    build_broken/libstan_hip_no_overlap_wait.so: cg.hip's boundary product and the vector kernels behind it no longer
    wait for the interior product on the side stream -- `hipStreamWaitEvent(st_, ev_b)` dropped).
      * Over a transport that drains the DEVICE at every call (FAKE_RCCL_SYNC_DEVICE=1: the fully serialising stand-in)
        the interior product has always finished when the exchange returns: the broken library reproduces the oracle
        fixture like the good one.  A suite that only ever ran over such a transport would never see the bug.
      * Over the stream-ordered stand-in nothing orders the two streams but the product itself: at 148^3 on 2 ranks the
        interior product (~0.5 ms) outlasts the enqueued exchange several times over, the boundary product and r' = r - a v
        overtake it, and the solve no longer reproduces the fixture.
    (The default synchronous stand-in drains only the stream it is given, so the side stream's product overlaps there
    too, with the host's timing: tools/lab/explore_broken_library.py ran this library at 100^3 and 148^3 under the three
    transports, twice each -- device-draining: passes 4 of 4; synchronous: caught 4 of 4; stream-ordered: caught 2 of 2 at
    148^3 and 0 of 2 at 100^3, where its exchange takes as long as the interior product:
    profiles/r06/broken_library_under_the_three_transports.txt.)  The GOOD library passes under
    every mode (test_sharded_bench_mode_against_the_oracle_fixture, test_the_two_modes_of_the_stand_in_give_the_same_bits)."""
    hidden = _broken_library_run(tmp_path, "device-draining", 148, dict(fake_rccl_env("sync"), FAKE_RCCL_SYNC_DEVICE="1"))
    assert hidden, "a transport that drains the device was expected to hide the missing wait"
    # exposing a race is a matter of timing by nature (7 of 7 runs at 148^3 in round 6, 0 of 4 at 100^3): up to three runs; a box on
    # which the enqueued exchange outlasts the interior product every time is reported, not failed
    for attempt in (1, 2, 3):
        if not _broken_library_run(tmp_path, "stream-ordered-%d" % attempt, 148, fake_rccl_env("async")):
            return
    pytest.skip("the stream-ordered stand-in did not expose the missing wait in three runs on this box (timing)")


@pytest.mark.parametrize("world,spec", [(2, "12"), (3, "12"), (4, "fuzz:124")])
def test_sharded_solve_peer_to_peer_between_processes(built_libs, tmp_path, world, spec):
    """STAN_OPT_COMM_P2P with one PROCESS per rank (what bench.py --gpus N --p2p runs under the launcher): the
    ranks map each other's mailboxes, arrival counters and gather vectors through HIP IPC handles that travel
    over the communicator they already have; the loop then makes no RCCL call.  Same bits as the RCCL path
    (rank-ordered sums on both sides), every rank the same U."""
    a, b = tmp_path / "rccl", tmp_path / "p2p"
    a.mkdir(); b.mkdir()
    out = _torchrun(world, [os.path.join(ROOT, "tests", "sharded_worker.py"), spec, str(a), "1"], {})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    out = _torchrun(world, [os.path.join(ROOT, "tests", "sharded_worker.py"), spec, str(b), "1", "p2p"], {})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for r in range(world):
        da, db = np.load(str(a / ("rank%d.npz" % r))), np.load(str(b / ("rank%d.npz" % r)))
        assert int(da["its"]) == int(db["its"]) and int(da["term"]) == int(db["term"]) == 1
        for k in ("U", "Um", "Ux"):
            assert np.array_equal(da[k], db[k]), (r, k)


def test_bench_multi_rank_code_path(built_libs):
    """bench.py --gpus 2 (gloo control plane, both ranks on GPU 0): one JSON line, converged."""
    out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--size", "16", "--no-cpu", "--no-p2p-probe"],
                    {"STAN_BENCH_BACKEND": "gloo", "STAN_BENCH_DEVICE": "0"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["converged"] and d["value"] > 0
    assert d["config"]["parallelism"] == "rows sharded x2" and d["roofline"]["launches"] > 0
    # round 3: a multi-GPU line explains itself -- transport, per-rank SpMV rates, exchange times per call
    assert "stand-in" in d["config"]["transport"] and "communicator of 2 ranks" in d["config"]["transport"]
    assert "libfake_rccl.so" in d["config"]["transport"]      # round 6: the line names the FILE the RCCL entry points came from
    pr = d["roofline"]["per_rank"]
    assert len(pr["frac"]) == 2 and 0 < pr["frac_min"] <= pr["frac_max"]
    ex = d["config"]["exchange"]
    assert len(ex["allreduce_us_per_call"]) == 2 and min(ex["allreduce_us_per_call"]) > 0
    assert min(ex["halo_us_per_call"]) > 0 and ex["allreduces_per_step"] > 2 * d["config"]["cg_iterations"] * 0.9
    assert sum(ex["owned_block_rows"]) == 17 ** 3 and min(ex["halo_block_rows"]) > 0
    assert d["roofline"]["two_product_launches"] >= d["config"]["cg_iterations"] // 10


def test_bench_multi_rank_peer_to_peer(built_libs):
    """bench.py --gpus 2 --p2p: the rank processes exchange peer to peer (HIP IPC); the line says so, the loop
    enqueued no collective, and the result is the RCCL-path line's (same iterations, converged)."""
    lines = {}
    for flag in ([], ["--p2p"]):
        out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                            "--size", "14", "--no-cpu", "--no-p2p-probe"] + flag,
                        {"STAN_BENCH_BACKEND": "gloo", "STAN_BENCH_DEVICE": "0"})
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        lines[bool(flag)] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    a, b = lines[False], lines[True]
    assert "peer to peer" in b["config"]["transport"] and "peer to peer" not in a["config"]["transport"]
    assert a["config"]["cg_iterations"] == b["config"]["cg_iterations"] and b["config"]["converged"]
    assert a["config"]["rel_residual"] == b["config"]["rel_residual"]          # rank-ordered sums on both transports
    assert min(b["config"]["exchange"]["allreduce_us_per_call"]) > 0


def test_bench_watchdog_ends_a_stuck_multi_rank_run(built_libs):
    """First contact with an 8-GPU node must not hang the driver: rank 1 never starts its steps (test hook),
    rank 0 blocks in the first exchange; the watchdog prints one JSON error line and the job exits non-zero
    well inside a minute -- by os._exit from a thread, never by re-executing a GPU process."""
    import time
    t0 = time.time()
    out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--size", "10", "--no-cpu", "--watchdog", "12"],
                    {"STAN_BENCH_BACKEND": "gloo", "STAN_BENCH_DEVICE": "0", "STAN_BENCH_TEST_HANG_RANK": "1"})
    took = time.time() - t0
    assert out.returncode != 0, out.stdout[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) >= 1, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and "watchdog" in d["error"] and d["watchdog"]["rank"] == 0
    assert "warm-up" in d["watchdog"]["phase"] and d["n_gpus"] == 2
    assert took < 100, took     # launcher start + torch import + 12 s bound + teardown


def _bench_no_launcher(script_args, env_extra, timeout=420):
    """`python bench.py --gpus N ...` typed the way the driver types it: NO launcher around it."""
    env = dict(os.environ, STAN_RCCL_LIB=FAKE, STAN_BENCH_BACKEND="gloo", STAN_BENCH_DEVICE="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + script_args
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)     # (the launcher's children sit in a group of their own: it ends them)
        out, err = proc.communicate()
        raise AssertionError("bench.py did not end:\n" + (err or "")[-3000:])
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def test_bench_gpus_2_without_a_launcher(built_libs):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its two rank processes itself
    (fresh children, the parent never touches the GPU), relays ONE JSON line and the exit code; the line carries the
    capped comparison of the two transports and says which one it would select."""
    out = _bench_no_launcher(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "16", "--no-cpu",
                              "--probe-its", "40"], {})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["converged"] and d["value"] > 0
    assert d["config"]["parallelism"] == "rows sharded x2"
    pr = d["config"]["p2p_probe"]
    legs = pr["legs"]
    assert pr["capped_at_iterations"] == 40 and sorted(legs) == ["classic_p2p", "classic_rccl", "single_reduce_p2p", "single_reduce_rccl"]
    assert all(v["iterations"] == 40 and v["ms_per_iteration"] > 0 and v["every_rank_same_residual_bits"] for v in legs.values())
    assert pr["same_residual_bits_classic"]                           # (the test transport adds in rank order, like peer to peer)
    assert all(pr["agrees_with_classic_rccl"].values())
    assert legs["classic_rccl"]["collectives_per_iteration"] >= 2 and legs["classic_p2p"]["collectives_per_iteration"] == 0
    assert 1 <= legs["single_reduce_rccl"]["collectives_per_iteration"] < 1.5 and legs["single_reduce_p2p"]["collectives_per_iteration"] == 0
    assert legs["classic_p2p"]["stream_waits_per_iteration"] >= 2 and legs["classic_rccl"]["stream_waits_per_iteration"] == 0
    assert d["config"]["recommended_transport"] == pr["recommended"]
    assert pr["recommended"].split()[0] in legs and pr["fastest"] in legs
    # round 5: the line says which physical device every rank drove (here: both on the one GPU of the test box)
    ranks = d["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["comm_ranks"] == 2 and r["pci_bus_id"] for r in ranks)
    assert d["config"]["distinct_devices"] == 1 and ranks[0]["pid"] != ranks[1]["pid"]


def test_bench_probe_that_crashes_leaves_the_line_alone(built_libs):
    """a HARD failure inside the probe -- here rank 1's probe process aborts (test hook) in front of its
    first peer-to-peer leg, as a GPU fault or an abort inside RCCL / an IPC mapping would -- must not cost the measurement:
    the probe runs in child processes, the measured ranks print their line unchanged and end with code 0."""
    out = _bench_no_launcher(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "12", "--no-cpu",
                              "--probe-its", "20", "--probe-watchdog", "10"], {"STAN_BENCH_TEST_CRASH_PROBE": "1"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["converged"] and d["value"] > 0 and "error" not in d
    # round 6: the RCCL legs run first and their report is out before the peer-to-peer legs start: it survives the crash
    pr = d["config"]["p2p_probe"]
    assert sorted(pr["legs"]) == ["classic_rccl", "single_reduce_rccl"] and pr["legs_not_run"] == ["classic_p2p", "single_reduce_p2p"]
    assert pr["recommended"].split()[0] in pr["legs"] and pr["same_residual_bits_classic"] is None


def test_bench_probe_that_stalls_leaves_the_line_alone(built_libs):
    """The transport probe is an optional extra behind the measurement: rank 1 never enters its peer-to-peer leg
    (test hook), rank 0 blocks in the collective set-up; the probe's own watchdog prints the measured line
    unchanged and the job ends with code 0."""
    out = _bench_no_launcher(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "12", "--no-cpu",
                              "--probe-its", "20", "--probe-watchdog", "8"], {"STAN_BENCH_TEST_HANG_PROBE": "1"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["converged"] and d["value"] > 0 and "error" not in d
    assert sorted(d["config"]["p2p_probe"]["legs"]) == ["classic_rccl", "single_reduce_rccl"]     # (the interim report)


def test_bench_one_process_mode(built_libs):
    """--one-process: the handle of stan_hip_init_multi (what the reference's single process would use) timed by
    bench.py: two ranks of ONE process on GPU 0 over the test transport, same answer as the one-rank handle."""
    res = {}
    for n in (1, 2):
        out = _bench_no_launcher(["--gpus", str(n), "--one-process", "--steps", "1", "--warmup", "1", "--size", "14",
                                  "--no-cpu"], {})
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        res[n] = json.loads(lines[0])
        assert res[n]["n_gpus"] == n and res[n]["config"]["converged"] and res[n]["value"] > 0
        assert "ONE process" in res[n]["config"]["process_model"]
    assert abs(res[1]["config"]["cg_iterations"] - res[2]["config"]["cg_iterations"]) <= 1
    assert abs(res[1]["config"]["u_max"] - res[2]["config"]["u_max"]) <= 1e-7 * res[1]["config"]["u_max"]


def test_peer_to_peer_survivor_releases_itself_when_its_peer_dies(built_libs):
    """ADVICE r03 (p2p.hip:70): process-per-GPU peer to peer, the peer dies after publishing its vectors; the
    survivor's device-side wait would spin forever.  The host loop's bounded wait releases this rank's own counters
    after STAN_P2P_STALL_S without progress and the solve returns STAN_E_COMM (see tests/p2p_peer_dies_worker.py)."""
    import time
    t0 = time.time()
    out = _torchrun(2, [os.path.join(ROOT, "tests", "p2p_peer_dies_worker.py")],
                    {"STAN_P2P_STALL_S": "4", "FAKE_RCCL_EXIT_AFTER_BCAST_GROUPS": "1:4"})
    took = time.time() - t0
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "SURVIVOR code -7" in out.stdout and "released" in out.stdout, out.stdout[-2000:]
    assert "leaves after broadcast group 4" in out.stderr
    assert took < 120, took
