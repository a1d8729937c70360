#!/bin/bash
# Kernel experiments: libvqe_hip with the LDS-resident kernels of ONE size only (seconds instead of minutes to build).
#   tools/build_only_n.sh 12 [extra hipcc flags]  ->  tools/libvqe_hip_n12.so   (use it with VQE_HIP_LIB=...)
N=${1:-12}; shift
cd "$(dirname "$0")/../tensorrl-qas_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wall -Wno-unused-function \
  -mllvm -structurizecfg-skip-uniform-regions -DVQE_ONLY_N=$N "$@" vqe_api.hip vec_env.cpp -o ../../tools/libvqe_hip_n$N.so
