"""CPU: libvqe_hip.so loads and exports every symbol include/vqe_hip.h declares; the
host-only COBYLA entry points reproduce scipy's Fortran COBYLA traces bit for bit; the
product never imports the oracle."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for header in ("vqe_hip.h", "vqe_env.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(vqe_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import _lib
    lib = C.CDLL(tq.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def _f(name):
    def rosen(x):
        return float(sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))

    def quad(x):
        a = np.arange(1, len(x) + 1, dtype=float)
        return float(np.sum(a * (x - 0.3 * a) ** 2) + 0.5 * x[0] * x[-1])

    def trig(x):
        k = np.arange(1, len(x) + 1, dtype=float)
        return float(-np.sum(np.cos(x - 0.1 * k)) + 0.25 * np.sum(np.sin(x[:-1] * x[1:])))
    return {"rosen": rosen, "quad": quad, "trig": trig}[name]


def test_host_cobyla_reproduces_scipy_traces():
    """scipy 1.15.3 Fortran COBYLA (the optimiser the reference calls) vs vqe_cobyla_*: every
    trial point identical, bit for bit, including the returned x (last trial point on normal
    termination, best vertex on maxfun)."""
    import tensorrl_qas_amd as tq
    gold = json.load(open(os.path.join(GOLDEN, "cobyla_scipy.json")))
    for c in gold["cases"]:
        fun = _f(c["fun"])
        opt = tq.HostCobyla(c["x0"], 1.0, 1e-4, c["maxiter"])
        evals = []
        while True:
            x = opt.ask()
            if x is None:
                break
            evals.append(x)
            opt.tell(fun(x))
        x, f, nfev, status = opt.result()
        assert nfev == c["nfev"] == len(evals)
        assert np.array_equal(np.array(evals), np.array(c["evals"])), (c["fun"], c["n"])
        assert np.array_equal(x, np.array(c["x"])) and f == c["f"]
        assert status == (2 if nfev >= c["maxiter"] else 1)
    opt = tq.HostCobyla(np.zeros(0))
    assert opt.ask() is not None
    opt.tell(1.25)
    assert opt.ask() is None and opt.result()[2] == gold["empty"]["nfev"] == 1


def test_host_cobyla_against_live_scipy():
    from scipy.optimize import minimize
    import tensorrl_qas_amd as tq
    rng = np.random.default_rng(3)
    for n in (1, 3, 7, 16):
        x0 = rng.uniform(-2, 2, n)
        fun = _f("trig") if n > 1 else _f("quad")
        r = minimize(fun, x0, method="COBYLA", options={"maxiter": 400})
        x, f, nfev, _ = tq.HostCobyla(x0, 1.0, 1e-4, 400).minimize(fun)
        assert nfev == r.nfev and np.array_equal(x, r.x)


def test_no_gpu_means_loud_failure():
    import torch
    import tensorrl_qas_amd as tq
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(tq.VQEError):
        tq.VQEEngine(4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tensorrl-qas_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "vqe_oracle" not in text and "c_oracle" not in text and "oracle/" not in text, fn


def test_device_code_has_no_function_calls(tmp_path):
    """Every device helper must be inlined into its kernel: a real call (s_swappc) loses the LDS
    address spaces, spills the caller's registers, and under a waves-per-SIMD register cap it
    once produced a kernel that faulted on the GPU (n = 11 with many parameters)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "tensorrl-qas_amd", "csrc")
    out = tmp_path / "dev.s"
    flags = None
    for line in open(os.path.join(csrc, "Makefile")):
        if line.startswith("FLAGS :="):
            flags = line.split(":=", 1)[1].strip().rstrip("\\").split()
        elif flags is not None and line.startswith(" ") and "-mllvm" in line:
            flags += line.strip().split()
    flags = [f for f in flags if f not in ("-shared", "-fPIC")]
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags]
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S",
                    os.path.join(csrc, "vqe_api.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    assert "k_lds_minimizeILi12" in text
    assert "s_swappc" not in text and "s_call" not in text
