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
