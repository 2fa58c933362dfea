"""bench.py as its own launcher, on a host WITHOUT a GPU: `python bench.py --gpus 2` must still end by itself with
one JSON line (an error line: the ranks cannot start) and a non-zero exit code -- a first contact with a
multi-GPU node yields a line, never a hang or a bare SystemExit (the working case is in
tests/test_gpu_sharded.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_without_launcher_and_without_gpu_ends_with_an_error_line(built_libs):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_sharded.py covers the working launcher")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "6", "--steps", "1",
                          "--warmup", "1", "--no-cpu"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "launcher" in d["error"]


def test_cpu_at_workload_lookup():
    """The line's cpu_baseline_at_workload comes from a committed run of the CPU port on the workload itself."""
    import bench_legs
    e = bench_legs.cpu_at_workload(148)
    if e is None:
        pytest.skip("no profiles/r*/cpu_at_workload.json yet")
    assert e["unit"] == "DOF/s" and e["cores"] >= 1 and e["kind"] == "port" and e["seconds"] > 60
    assert e["n_dof"] == 9923847 and e["source"].startswith("profiles/")


def test_secondary_legs_never_cost_the_headline(monkeypatch):
    """bench.py's `secondary` legs are child processes behind the measured line: a leg that times out,
    crashes or raises leaves {"error": ...}; a spent budget skips the rest; a good leg is reduced to its key figures."""
    import types
    import bench_legs as bench

    class Dog:
        def touch(self, *a, **k):
            pass
    good = json.loads(open(os.path.join(ROOT, "profiles", "r05", "bench_default_flags_r05_final.json")).read().strip().splitlines()[-1])
    calls = []

    def fake_child(cmd, timeout, env=None, marker='"metric"'):
        calls.append((cmd, timeout))
        if "--size" in cmd and cmd[cmd.index("--size") + 1] == "400":      # fp64 leg fine, its FIXED-48 half missed eps
            bad = json.loads(json.dumps(good))
            bad.update(value=None, error="CG ended with type 7 at 2.1e-08 (> eps 1e-08): no DOF/s reported")
            return dict(good, then_fixed48={k: bad[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "roofline", "error")}), None
        if "--size" in cmd and cmd[cmd.index("--size") + 1] == "100":
            return good, None
        if "--size" in cmd and cmd[cmd.index("--size") + 1] == "200":
            return None, "timed out after %.0f s" % timeout
        if "--fixed48" in cmd:
            raise RuntimeError("boom")
        return None, "rc -11, no line; stderr tail: Segmentation fault"
    monkeypatch.setattr(bench, "_child_json", fake_child)
    monkeypatch.setattr(bench, "_console_leg", lambda n, timeout: {"error": "no GPU here"})
    monkeypatch.setattr(bench, "_pmc_leg", lambda timeout: {"error": "no profiler here"})
    monkeypatch.setattr(bench, "_host_mem_available_gb", lambda: 200.0)
    args = types.SimpleNamespace(secondary_budget=600.0)
    seen = []
    legs = bench.secondary_legs(args, Dog(), seen.append)
    assert seen == legs                      # every leg is handed to the caller as it ends (the held line grows)
    assert [l["leg"].split(":")[0] for l in legs][:2] == ["config 2", "config 3"] and len(legs) == 8
    # round 6: 400^3 from ONE child (fp64, then FIXED-48 on the same resident model) = two entries; a half that missed
    # eps keeps its error and has no value
    assert "400^3 fp64" in legs[6]["leg"] and legs[6]["value"] == good["value"] and "error" not in legs[6]
    assert "FIXED-48" in legs[7]["leg"] and legs[7]["value"] is None and "type 7" in legs[7]["error"]
    cmd400 = [c for c, _ in calls if "400" in c][0]
    assert cmd400[cmd400.index("--steps") + 1] == "1" and cmd400[cmd400.index("--warmup") + 1] == "0" and "--then-fixed48" in cmd400
    assert legs[0]["value"] == good["value"] and legs[0]["roofline"]["frac"] == good["roofline"]["frac"] and "error" not in legs[0]
    assert "timed out" in legs[1]["error"] and "boom" in legs[2]["error"] and legs[3]["error"] == "no GPU here"
    assert legs[4]["error"] == "no profiler here" and "Segmentation fault" in legs[5]["error"]
    assert all(t <= 420.0 for _, t in calls)
    # the long leg is only started with ~5 min of the budget left
    calls.clear()
    legs = bench.secondary_legs(types.SimpleNamespace(secondary_budget=200.0), Dog())
    assert not [c for c, _ in calls if "400" in c] and "skipped" in legs[-1]["error"] and len(legs) == 7
    # a host without the memory for the 400^3 set-up: that leg is skipped, never attempted
    calls.clear()
    monkeypatch.setattr(bench, "_host_mem_available_gb", lambda: 31.0)
    legs = bench.secondary_legs(types.SimpleNamespace(secondary_budget=600.0), Dog())
    assert not [c for c, _ in calls if "400" in c] and "host memory" in legs[-1]["error"]
    # a spent budget: nothing is started any more
    calls.clear()
    legs = bench.secondary_legs(types.SimpleNamespace(secondary_budget=-1.0), Dog())
    assert not calls and all("skipped" in l["error"] for l in legs)


def test_probe_recommendation_goes_by_tolerance_not_by_bits():
    """with N >= 3 RCCL's summation order is not rank order, the residual bits of the RCCL and the
    peer-to-peer leg differ after 200 iterations -- that must not stop a faster transport from being recommended."""
    import bench_launch as bench

    def leg(ms, its=200, res=1.4766136689192024):
        return {"ms_per_iteration": ms, "iterations": its, "rel_residual": res}
    legs = {"classic_rccl": leg(0.190), "classic_p2p": leg(0.150, res=1.4766136689192024 * (1 + 3e-13)),
            "single_reduce_rccl": leg(0.170, res=1.47661344), "single_reduce_p2p": leg(0.160, res=1.47661344)}
    r = bench.probe_report(legs, 200)
    assert not r["same_residual_bits_classic"] and all(r["agrees_with_classic_rccl"].values())
    assert r["fastest"] == "classic_p2p" and r["recommended"].startswith("classic_p2p (STAN_OPT_COMM_P2P=1)")
    # a leg that is fastest but does not agree (another iteration count) is not recommended
    legs["classic_p2p"] = leg(0.150, its=199)
    r = bench.probe_report(legs, 200)
    assert not r["agrees_with_classic_rccl"]["classic_p2p"] and r["fastest"] == "classic_p2p"
    assert r["recommended"].startswith("single_reduce_p2p")          # the fastest leg that agrees
    # an interim report (the RCCL legs only: what is out before the peer-to-peer legs start on a node)
    r = bench.probe_report({k: v for k, v in legs.items() if k.endswith("rccl")}, 200)
    assert r["legs_not_run"] == ["classic_p2p", "single_reduce_p2p"] and r["same_residual_bits_classic"] is None
    assert r["recommended"].startswith("single_reduce_rccl") and sorted(r["agrees_with_classic_rccl"]) == ["classic_rccl", "single_reduce_rccl"]
    # nothing is 3 % faster than the defaults: the defaults stay
    legs = {k: leg(0.190 if k == "classic_rccl" else 0.188) for k in legs}
    assert bench.probe_report(legs, 200)["recommended"].startswith("classic_rccl (library defaults)")


def test_pmc_per_launch_counts_only_launches_that_did_work(tmp_path):
    """bench.py's PMC leg averages a counter over the launches of k_spmv that did work: the launches queued behind a converged
    solve return at once (a few KiB) and must not pull the average down; other kernels and counters are ignored."""
    import bench_legs as bench
    f = tmp_path / "pmc_counter_collection.csv"
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    rows += ['"void (anonymous namespace)::k_spmv<double, 1, 9>(int, long)",FETCH_SIZE,3338000.0'] * 10
    rows += ['"void (anonymous namespace)::k_spmv<double, 1, 9>(int, long)",FETCH_SIZE,12.0'] * 4          # early exits
    rows += ['"void (anonymous namespace)::k_spmv<double, 1, 9>(int, long)",WRITE_SIZE,81000.0'] * 3
    rows += ['"void (anonymous namespace)::k_spmv2<double>(int, long)",FETCH_SIZE,3450000.0'] * 2
    f.write_text("\n".join(rows) + "\n")
    avg, work, total = bench.pmc_per_launch([str(f)], "FETCH_SIZE")
    assert (avg, work, total) == (3338000.0, 10, 14)
    assert bench.pmc_per_launch([str(f)], "WRITE_SIZE") == (81000.0, 3, 3)
    assert bench.pmc_per_launch([str(f)], "TCC_HIT") is None


def test_bench_py_stays_small_and_the_split_modules_import():
    """bench.py = the N = 1 path + the JSON line, under 500 lines; launcher / watchdog / probes in
    bench_launch.py, CPU baseline and secondary legs in bench_legs.py."""
    assert len(open(os.path.join(ROOT, "bench.py")).read().splitlines()) < 500
    import bench
    import bench_launch
    import bench_legs
    assert bench.METRIC == bench_launch.METRIC and callable(bench.measure) and callable(bench_legs.secondary_legs)
    ap = bench.build_parser()
    a = ap.parse_args([])
    assert (a.gpus, a.n, a.eps, a.etype) == (1, 148, 1e-8, 2) and a.secondary_budget >= 600


def test_a_signal_behind_the_measurement_prints_the_held_line(tmp_path):
    """the headline must survive the optional work behind it.  bench_launch.hold_line writes the
    finished line to a side file at once and installs SIGTERM / SIGINT handlers that print it and leave with code 0 --
    what an outer `timeout` does to a run that is still in its secondary legs."""
    import signal
    import time
    side = str(tmp_path / "line.json")
    code = r'''
import sys, time
sys.path.insert(0, %r)
import bench_launch as BL
class A: steps = 1; warmup = 0
dog = BL.Watchdog(0, 0, 1, A())
BL.hold_line(dog, '{"metric": "m", "value": 1.0}')
BL.hold_line(dog, '{"metric": "m", "value": 1.0, "secondary": [1]}')
print("READY", flush=True)
time.sleep(60)
''' % ROOT
    env = dict(os.environ, STAN_BENCH_SIDE_FILE=side)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p.stdout.readline().strip() == "READY"
    assert json.loads(open(side).read())["secondary"] == [1]
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=20)
    assert p.returncode == 0 and json.loads(out.strip().splitlines()[-1])["secondary"] == [1]
    assert "the measured line stands" in err
