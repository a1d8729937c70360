"""ctypes binding of libvqe_hip.so (include/vqe_hip.h).  No fallback: if the shared library
is missing or no HIP device is present the engine raises."""
from __future__ import annotations

import ctypes as C
import os

# Load order matters: PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so) while
# libvqe_hip.so is linked against the system one (/opt/rocm/lib).  With torch imported FIRST both sides
# share one initialised runtime; the other way round torch finds "No HIP GPUs" later
# (tools/probe_torch_after.py).  PyTorch is this package's plumbing for device tensors and
# torch.distributed anyway, so import it before the library whenever it is installed.
try:
    import torch  # noqa: F401
except ImportError:      # the engine itself needs only the HIP runtime
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQE_HIP_LIB") or os.path.join(_HERE, "libvqe_hip.so")  # override for kernel experiments

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u64p = C.POINTER(C.c_uint64)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/vqe_hip.h one to one
SIGNATURES = {
    "vqe_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(vp)]),
    "vqe_destroy": (None, [vp]),
    "vqe_last_error": (C.c_char_p, [vp]),
    "vqe_set_stream": (C.c_int, [vp, vp]),
    "vqe_get_stream": (C.c_int, [vp, C.POINTER(vp)]),
    "vqe_sync": (C.c_int, [vp]),
    "vqe_device_info": (C.c_int, [vp, c_i64p]),
    "vqe_set_init_state": (C.c_int, [vp, c_f64p]),
    "vqe_set_hamiltonian_pauli": (C.c_int, [vp, C.c_int, c_u64p, c_u64p, c_f64p]),
    "vqe_set_hamiltonian_dense": (C.c_int, [vp, c_f64p, C.c_double]),
    "vqe_hamiltonian_terms": (C.c_int, [vp, c_i32p, c_i32p]),
    "vqe_hamiltonian_layout": (C.c_int, [vp, c_i32p]),
    "vqe_comm_unique_id": (C.c_int, [vp]),
    "vqe_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "vqe_comm_allreduce_energy": (C.c_int, [vp]),
    "vqe_comm_destroy": (C.c_int, [vp]),
    "vqe_set_noise_mode": (C.c_int, [vp, C.c_int]),
    "vqe_noise_mode_info": (C.c_int, [vp, c_i32p]),
    "vqe_dm_plan": (C.c_int, [C.c_int, C.c_int, c_i32p, c_i32p, c_i32p, c_i32p, c_f64p, C.c_double, C.c_double, C.c_int, c_i32p,
                              c_i32p, c_f64p]),
    "vqe_set_init_state_dev": (C.c_int, [vp, vp]),
    "vqe_get_state_dev": (C.c_int, [vp, c_f64p, vp]),
    "vqe_set_term_shard": (C.c_int, [vp, C.c_int, C.c_int]),
    "vqe_set_amplitude_shard": (C.c_int, [vp, C.c_int, C.c_int]),
    "vqe_term_owner": (C.c_int, [C.c_int, C.c_int, c_u64p, C.c_int, c_i32p]),
    "vqe_set_noise": (C.c_int, [vp, C.c_double, C.c_double, C.c_uint64]),
    "vqe_set_shot_noise": (C.c_int, [vp, C.c_double, C.c_uint64]),
    "vqe_set_circuit": (C.c_int, [vp, C.c_int, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int]),
    "vqe_energy": (C.c_int, [vp, c_f64p, c_f64p]),
    "vqe_energy_batch": (C.c_int, [vp, C.c_int, c_f64p, c_f64p]),
    "vqe_get_state": (C.c_int, [vp, c_f64p, c_f64p]),
    "vqe_minimize_cobyla": (C.c_int, [vp, c_f64p, C.c_double, C.c_double, C.c_int, c_f64p, c_f64p, c_i32p]),
    "vqe_batch_load": (C.c_int, [vp, C.c_int, c_i64p, c_i32p, c_i32p, c_i32p, c_i32p, c_i64p, c_f64p]),
    "vqe_batch_run_energy": (C.c_int, [vp]),
    "vqe_batch_run_reduction": (C.c_int, [vp]),
    "vqe_batch_run_minimize": (C.c_int, [vp, C.c_double, C.c_double, C.c_int]),
    "vqe_batch_set_new_gate": (C.c_int, [vp, c_i32p]),
    "vqe_batch_run_env_step": (C.c_int, [vp, C.c_double, C.c_double, C.c_int]),
    "vqe_batch_fetch": (C.c_int, [vp, c_f64p, c_f64p, c_i32p]),
    "vqe_batch_fetch_xopt": (C.c_int, [vp, c_f64p]),
    "vqe_batch_energy_devptr": (C.c_int, [vp, C.POINTER(vp)]),
    "vqe_batch_copy_energy": (C.c_int, [vp, vp]),
    "vqe_batch_set_trace": (C.c_int, [vp, C.c_int]),
    "vqe_batch_fetch_trace": (C.c_int, [vp, C.c_int, c_f64p, c_i32p, c_i32p]),
    "vqe_debug_counters": (C.c_int, [vp, c_u64p]),
    "vqe_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "vqe_cobyla_create": (C.c_int, [C.c_int, c_f64p, C.c_double, C.c_double, C.c_int, C.POINTER(vp)]),
    "vqe_cobyla_ask": (C.c_int, [vp, c_f64p]),
    "vqe_cobyla_tell": (C.c_int, [vp, C.c_double]),
    "vqe_cobyla_result": (C.c_int, [vp, c_f64p, c_f64p, c_i32p, c_i32p]),
    "vqe_cobyla_destroy": (None, [vp]),
}



class VecEnvConfig(C.Structure):
    """vqe_vecenv_config_t of include/vqe_env.h"""
    _fields_ = [("n_qubits", C.c_int32), ("num_layers", C.c_int32), ("num_envs", C.c_int32), ("layer_offset", C.c_int32),
                ("noisy", C.c_int32), ("num_layers_termination", C.c_int32), ("maxfun", C.c_int32),
                ("min_eig", C.c_double), ("accept_err", C.c_double),
                ("n_thresholds", C.c_int32), ("thresholds", c_f64p), ("switch_episodes", c_i64p),
                ("n_init_gates", C.c_int32), ("init_layer", c_i32p), ("init_kind", c_i32p), ("init_q0", c_i32p),
                ("init_q1", c_i32p), ("init_angle", C.POINTER(C.c_float)), ("init_energy", C.c_double),
                ("n_actions", C.c_int32), ("action_table", c_i32p)]


# include/vqe_env.h (same shared library)
SIGNATURES.update({
    "vqe_vecenv_create": (C.c_int, [C.POINTER(VecEnvConfig), vp, C.POINTER(vp)]),
    "vqe_vecenv_destroy": (None, [vp]),
    "vqe_vecenv_last_error": (C.c_char_p, [vp]),
    "vqe_vecenv_reset": (C.c_int, [vp, C.c_int32, c_i32p, c_i32p]),
    "vqe_vecenv_illegal_actions": (C.c_int, [vp, c_i32p]),
    "vqe_vecenv_step_begin": (C.c_int, [vp, c_i32p]),
    "vqe_vecenv_step_end": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), c_i32p, c_i64p]),
    "vqe_vecenv_get": (C.c_int, [vp, C.c_int, c_f64p]),
    "vqe_vecenv_state": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_float)]),
    "vqe_vecenv_moments": (C.c_int, [vp, C.c_int32, c_i32p, c_i32p]),
    "vqe_vecenv_actions": (C.c_int, [vp, C.c_int32, c_i32p, c_i32p]),
    "vqe_vecenv_opt_ang": (C.c_int, [vp, C.c_int32, c_f64p, c_i32p]),
    "vqe_vecenv_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
})

# include/mps2qc_hip.h (libmps2qc_hip.so: the offline MPS -> PQC fit)
MPS2QC_LIB_PATH = os.environ.get("MPS2QC_HIP_LIB") or os.path.join(_HERE, "libmps2qc_hip.so")
MPS2QC_SIGNATURES = {
    "mps2qc_brickwork_sites": (C.c_int, [C.c_int, C.c_int, c_i32p, C.c_int]),
    "mps2qc_fit_brickwork": (C.c_int, [C.c_int, C.c_int, C.c_int, c_i32p, C.c_int, c_f64p, C.c_int, c_f64p,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.c_int, C.c_double, C.c_double, C.c_int,
                                       c_f64p, c_f64p, c_f64p, c_f64p, c_i32p, c_f64p, c_f64p,
                                       C.POINTER(C.c_float)]),
    "mps2qc_fit_brickwork_stream": (C.c_int, [C.c_int, C.c_int, C.c_int, c_i32p, C.c_int, c_f64p, C.c_int, c_f64p,
                                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                              C.c_int, C.c_double, C.c_double,
                                              c_f64p, c_f64p, c_f64p, c_f64p, c_i32p, c_f64p, c_f64p,
                                              C.POINTER(C.c_float)]),
    "mps2qc_last_error": (C.c_char_p, []),
}

_lib = None
_lib_mps2qc = None


class VQEError(RuntimeError):
    pass


def load():
    """Load libvqe_hip.so once; raises VQEError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VQEError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def load_mps2qc():
    """Load libmps2qc_hip.so once; raises VQEError when it has not been built."""
    global _lib_mps2qc
    if _lib_mps2qc is None:
        if not os.path.exists(MPS2QC_LIB_PATH):
            raise VQEError(f"{MPS2QC_LIB_PATH} not found: build it with __graft_entry__.build(); no CPU fallback.")
        lib = C.CDLL(MPS2QC_LIB_PATH)
        for name, (res, args) in MPS2QC_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib_mps2qc = lib
    return _lib_mps2qc
