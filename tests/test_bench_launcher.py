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
    import bench
    e = bench.cpu_at_workload(148)
    if e is None:
        pytest.skip("no profiles/r*/cpu_at_workload.json yet")
    assert e["unit"] == "DOF/s" and e["cores"] >= 1 and e["kind"] == "port" and e["seconds"] > 60
    assert e["n_dof"] == 9923847 and e["source"].startswith("profiles/")
