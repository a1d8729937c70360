"""The device COBYLA loop, trajectory level.

``cobyla_m0.h`` is one source with three execution contexts: HostCtx (pinned BIT FOR BIT against scipy
1.15.3's Fortran COBYLA - the optimiser the reference calls at environment_qulacs_TN_notin_agent.py:478 - by
tests/test_abi.py), and the wave / workgroup contexts of the fused kernel, which differ from it only in
the ORDER of the optimiser's own sums (DPP reduction trees, two lanes per row, closed-form trust-region
step).  tests/cpp/cobyla_wave_emulation.cpp plays the 64 lanes of the device's wavefront with 64 host
threads - same header, same trees lane for lane.  Told the device's function values it must reproduce
the device's trial points bit for bit:

* CPU (``not gpu``): on traces recorded on an MI355X (tests/golden/cobyla_device_traces/, written by
  tools/dump_cobyla_traces.py through vqe_batch_set_trace);
* GPU: on traces taken live from the library under test.

So the device loop is exactly "cobyla_m0.h with another summation order".  Where that order decides -
COBYLA compares quantities that are equal in exact arithmetic while the simplex is still the regular
initial one (e.g. |simi_j . dx| against 1) - the device and scipy pick different, equally valid vertices
and the runs part; tests/test_configs_gpu.py::test_device_cobyla_trajectory measures for how long they
coincide."""
import glob
import os
import subprocess
import sys

import pytest

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "cobyla_wave_emulation.cpp")


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    exe = tmp_path_factory.mktemp("emu") / "cobyla_wave_emulation"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-I",
                    os.path.join(ROOT, "tensorrl-qas_amd", "csrc"), SRC, "-o", str(exe)], check=True)
    return str(exe)


FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "cobyla_device_traces", "*.txt")))


@pytest.mark.parametrize("trace", FIXTURES, ids=[os.path.basename(f) for f in FIXTURES])
def test_wave_emulation_reproduces_recorded_device_traces(emulator, trace):
    r = subprocess.run([emulator, trace], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical to the device over the whole trace" in r.stdout, r.stdout + r.stderr


def test_fixtures_present():
    assert len(FIXTURES) >= 5


@pytest.mark.gpu
def test_wave_emulation_reproduces_live_device_traces(emulator, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_cobyla_traces as dct
    for case in dct.CASES:
        th, ft, xt, nf = dct.device_trace(*case)
        path = str(tmp_path / dct.case_name(*case))
        dct.write_trace(path, th, ft, xt, nf)
        r = subprocess.run([emulator, path], capture_output=True, text=True, timeout=600)
        print(r.stdout.strip())
        assert r.returncode == 0 and "bit-identical to the device over the whole trace" in r.stdout, r.stdout + r.stderr
