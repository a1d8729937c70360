#!/bin/bash
# one-size experiment library: live device traces of the >64-variable one-wave optimiser against the CPU emulation
set -e
export VQE_HIP_LIB=$PWD/tools/libvqe_hip_n8_tile.so
g++ -O2 -std=c++17 -ffp-contract=off -pthread -I tensorrl-qas_amd/csrc tests/cpp/cobyla_wave_emulation.cpp -o /tmp/emu
python - <<'PY'
import sys, subprocess
sys.path.insert(0, "tools")
import dump_cobyla_traces as dct
for case in dct.CASES:
    if case[0] != 8: continue
    th, ft, xt, nf = dct.device_trace(*case)
    path = "/tmp/" + dct.case_name(*case)
    dct.write_trace(path, th, ft, xt, nf)
    r = subprocess.run(["/tmp/emu", path], capture_output=True, text=True)
    print(case, nf, r.returncode, r.stdout.strip()[-200:])
PY
