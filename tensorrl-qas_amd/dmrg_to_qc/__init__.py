"""Host mirror of the reference's offline ``dmrg-to-qc`` block for the part that runs on the
GPU: fitting a brickwork circuit of SU(4) gates to a target MPS with Riemannian Adam
(``mps2qc.mps_to_qc``, ``stiefel_opt.StiefelAdam``, ``tnqc_ansatze.brickwork_ansatz``).  The
arithmetic lives in ``libmps2qc_hip.so`` (csrc/mps2qc_fit.hip, C ABI include/mps2qc_hip.h)."""
from . import mps2qc, stiefel_opt, su4_to_qasm, tnqc_ansatze  # noqa: F401
from .mps2qc import mps_to_qc, rand_uni  # noqa: F401
from .stiefel_opt import StiefelAdam, BrickworkOverlap  # noqa: F401
from .tnqc_ansatze import brickwork_ansatz, closest_unitary  # noqa: F401
from .su4_to_qasm import brickwork_to_qasm, decompose_su4  # noqa: F401
