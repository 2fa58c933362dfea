"""The C-ABI libraries load on a CPU-only host and export every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_hip_library_exports(built_libs):
    from stan_amd import hip
    lib = hip.load()
    names = _declared("stan_hip.h", "stan_hip_")
    assert set(names) == set(hip.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_product_library_carries_no_lab_code(built_libs):
    """The A/B kernel variants of round 1 (one of them, variant 8, computes wrong numbers on purpose)
    and the scalar-CSR comparison kernel live in the lab build only (make -C stan_amd/csrc lab)."""
    from stan_amd import hip
    lib = hip.load()
    for n in hip.LAB_EXPORTS:
        assert not hasattr(lib, n), n
    api = open(os.path.join(ROOT, "stan_amd", "csrc", "api.hip")).read()
    assert "value == -1 || value == 0 || value == 9 || value == 12 || value == 20)" in api
    # round 5: the product SOURCES carry no lab switch either -- ablations, A/B kernel variants and compile-time policy
    # macros are a patch applied to copies (stan_amd/csrc/lab/lab_hooks.patch, `make lab`)
    csrc = os.path.join(ROOT, "stan_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".inc", ".h")):
            src = open(os.path.join(csrc, f)).read()
            for word in ("STAN_LAB", "STAN_ABL", "STAN_VEC_NT", "STAN_Y_NT", "STAN_VLD_RO_NT", "STAN_VLD_RMW_NT", '#include "lab/'):
                assert word not in src, (f, word)


def test_host_library_exports(built_libs):
    from stan_amd import host
    lib = host.load()
    names = _declared("stan_host.h", "stan_host_")
    for n in names:
        assert hasattr(lib, n), n
    assert set(host.EXPORTS) <= set(names)


def test_no_cpu_fallback(built_libs):
    """Without a GPU the product path must fail loudly, not fall back."""
    import torch
    from stan_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.StanHipError) as ei:
        hip.Context(0)
    assert ei.value.code == hip.E_HIP


def test_product_does_not_import_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "stan_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "pyoracle" not in src and "stan_oracle" not in src, f  # test guards itself


def test_plain_c_consumer_builds_and_fails_loudly_without_a_gpu(built_libs):
    """tests/c_abi/consumer.c is compiled with -std=c99 -pedantic -Werror against include/ only
    (the headers are C, not C++-only) and drives the host steps; without a GPU the device entry
    point refuses with a message instead of computing anything on the CPU."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "tests", "c_abi", "consumer")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_parity.py::test_plain_c_consumer_end_to_end")
    out = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, out.stdout + out.stderr
    assert "nodes 27 elements 8 nDOF 81 fixed 27 N 54" in out.stdout     # SURVEY.md Appendix E, n = 2
    assert "stan_hip_init failed" in out.stderr


@pytest.mark.parametrize("host_has_rccl", [True, False])
def test_one_rccl_per_process(built_libs, host_has_rccl):
    """a host that has already mapped RCCL (bench.py and every torch host:
    libtorch_hip.so NEEDs its bundled librccl.so) must not get a second build next to it -- comm.hip asks with
    RTLD_NOLOAD first.  A host without RCCL gets exactly one fresh load.  (ncclGetUniqueId needs no GPU.)"""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
if %d:
    import torch
    from stan_amd import hip
    lib = hip.load()
else:      # a host without torch (the C# shim, stan_solver): the bare library
    lib = C.CDLL(os.path.join(%r, "stan_amd", "lib", "libstan_hip.so"))
before = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
buf = C.create_string_buffer(128)
rc = lib.stan_hip_comm_unique_id(buf)   # (ROCm 7.2's own RCCL wants a device even for this; the bundled one does not)
assert rc == 0 or not %d
after = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
print("BEFORE", len(before), "AFTER", len(after), " ".join(after))
''' % (ROOT, int(host_has_rccl), ROOT, int(host_has_rccl))
    env = dict(os.environ)
    env.pop("STAN_RCCL_LIB", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    w = [l for l in p.stdout.splitlines() if l.startswith("BEFORE")][0].split()
    assert int(w[1]) == (1 if host_has_rccl else 0) and int(w[3]) == 1, p.stdout
    assert ("torch" in w[4]) == host_has_rccl


def test_lab_patches_still_apply(tmp_path):
    """stan_amd/csrc/lab/lab_hooks.patch (the lab build) and drop_overlap_wait.patch (the broken build of
    tests/test_gpu_sharded.py) are diffs against the PRODUCT sources and must track every product edit: `patch --dry-run` on
    copies, so that a product change that breaks `make lab` / `make broken` fails here, in the CPU suite, not on the GPU box."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "stan_amd", "csrc")
    work = tmp_path / "src"
    work.mkdir()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".inc", ".h")):
            shutil.copy(os.path.join(csrc, f), str(work / f))
    for patch in ("lab_hooks.patch", "drop_overlap_wait.patch"):
        p = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(csrc, "lab", patch)], cwd=str(work),
                           capture_output=True, text=True)
        assert p.returncode == 0 and "FAILED" not in p.stdout, patch + ":\n" + p.stdout[-1500:]
