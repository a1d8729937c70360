"""Multi-GPU use of the engine: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in CPU tests).

Two ways the path shards:
* parallel environments / RL seeds - replicas, no data-path collective (``env_shard``);
* Pauli-term sharding of <psi|H|psi> - every rank applies the same circuit to its own copy
  of the state and evaluates a disjoint set of X-mask groups; ONE all-reduce of the partial
  energies (8 bytes per evaluation) completes each batch of evaluations.
The reference has no counterpart (single process)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .engine import HostCobyla


def dist_env():
    """(rank, world_size, local_rank) from the launcher's environment (torchrun)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def env_shard(num_envs_total: int, rank: int, world: int) -> range:
    """Contiguous block of environment ids owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(num_envs_total, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def term_owner(n_qubits: int, xmask, world: int) -> np.ndarray:
    """Rank owning each Pauli term under vqe_set_term_shard(rank, world) (host only)."""
    x = np.ascontiguousarray(xmask, np.uint64)
    out = np.zeros(x.size, np.int32)
    rc = _lib.load().vqe_term_owner(int(n_qubits), int(x.size), x.ctypes.data_as(_lib.c_u64p), int(world),
                                    out.ctypes.data_as(_lib.c_i32p))
    if rc:
        raise _lib.VQEError(f"vqe_term_owner failed ({rc})")
    return out


def allreduce_sum(values, group=None):
    """Sum a float64 tensor (or array) of partial energies over all ranks, in place.

    Device tensors go to RCCL as they are (backend "nccl"); host values under RCCL travel through this rank's
    GPU.  Under gloo (CPU tests, the two-ranks-on-one-GPU test) a device tensor is summed through a host copy:
    gloo's own device path (pinned staging buffers filled by its worker threads) is never entered."""
    import torch
    import torch.distributed as dist
    t = values if isinstance(values, torch.Tensor) else torch.as_tensor(np.asarray(values, np.float64))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        nccl = dist.get_backend(group) == "nccl"
        if t.is_cuda != nccl:
            d = t.to(torch.device("cuda", torch.cuda.current_device())) if nccl else t.cpu()
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
            t.copy_(d)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sharded_minimize(partial_energy, x0, rhobeg=1.0, rhoend=1e-4, maxfun=1000, group=None):
    """COBYLA where every evaluation is ``allreduce_sum(partial_energy(x))``.  All ranks run
    the same ask/tell sequence on bit-identical reduced energies, so they stay in lock-step
    without broadcasting x.  ``partial_energy(x) -> float`` (this rank's share)."""
    opt = HostCobyla(x0, rhobeg, rhoend, maxfun)
    while True:
        x = opt.ask()
        if x is None:
            break
        e = allreduce_sum(np.array([partial_energy(x)], np.float64), group)
        opt.tell(float(e[0]))
    return opt.result()


class TermShardedEngine:
    """An engine that owns ``rank``'s share of the Pauli terms; ``energies()`` returns the
    all-reduced energies of the resident batch as a device tensor."""

    def __init__(self, engine, rank: int, world: int, device, by_amplitude: bool | None = None):
        import torch
        self.engine, self.rank, self.world = engine, rank, world
        self.device = torch.device(device)
        # n >= 14: slice the amplitude range (1/world of the sweep per rank, perfectly
        # balanced); n <= 13: a workgroup holds the whole state, so split the X-mask groups
        if by_amplitude is None:
            by_amplitude = not engine.device_info()["lds_path"]
        if by_amplitude:
            engine.set_amplitude_shard(rank, world)
        else:
            engine.set_term_shard(rank, world)
        # ONE stream for the engine's launches, the device-to-device copy and the collective: a torch stream of
        # this object's own that the handle is told to issue on (vqe_set_stream), so inside energies() plain stream
        # order is all the ordering there is; the caller's current stream is ordered against it once on the way
        # in and once on the way out, with torch's own wait_stream on torch's own streams.  (torch's DEFAULT stream
        # cannot be handed to the handle: its pointer is NULL, which vqe_set_stream reads as "your own stream" -
        # a non-blocking one that the default stream does not order against.  Round 2 wrapped the handle's stream
        # in torch.cuda.ExternalStream instead; that code is gone.)
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.Stream(device=self.device)
        engine.set_stream(self._stream.cuda_stream)
        self._buf = None

    def close(self):
        """Give the handle its own stream back (after draining this object's) and drop the result buffer."""
        if self.engine is not None:
            self._stream.synchronize()
            self.engine.set_stream(None)
            self.engine, self._buf = None, None

    def energies(self, batch: int):
        import torch
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)             # whatever the caller still does with the previous result
        with torch.cuda.stream(self._stream):
            if self._buf is None or self._buf.numel() != batch:
                self._buf = torch.zeros(batch, dtype=torch.float64, device=self.device)
            self.engine.batch_run_energy()
            self.engine.batch_copy_energy(self._buf.data_ptr())
            out = allreduce_sum(self._buf)
        cur.wait_stream(self._stream)             # the caller reads the sum on its own stream
        out.record_stream(cur)
        return out

    def partial_energies(self):
        """This rank's share of the energies of the resident batch (host array, not reduced)."""
        self.engine.batch_run_energy()
        return self.engine.batch_fetch(want_x=False)[1]

    def minimize(self, circuit, x0, rhobeg=1.0, rhoend=1e-4, maxfun=1000):
        """Lock-step COBYLA over all ranks on the all-reduced energy (``sharded_minimize`` does
        the reduction: ``partial`` must return this rank's share only)."""
        def partial(x):
            self.engine.batch_load([circuit], [x])
            return float(self.partial_energies()[0])
        return sharded_minimize(partial, x0, rhobeg, rhoend, maxfun)


def fit_restarts_sharded(fit_shard, n_restarts_total: int, group=None):
    """Random restarts of the MPS -> PQC fit spread over the ranks (independent units, replicas
    only): every rank fits its contiguous share of the restart ids and ONE all-gather of the
    best losses (8 bytes per rank) picks the winner, whose gates are broadcast.

    ``fit_shard(restart_ids: range) -> (best_val: float, gates: ndarray[G,4,4] complex)`` runs
    this rank's restarts (on the GPU: ``dmrg_to_qc.mps_to_qc(..., n_restarts=len(ids))`` with a
    generator seeded per restart id, so the result does not depend on the world size).
    Returns ``(best_val, gates, owner_rank)``, identical on all ranks; ties go to the lowest rank."""
    import torch
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    ids = env_shard(n_restarts_total, rank, world)
    val, gates = fit_shard(ids) if len(ids) else (float("inf"), None)
    if world == 1:
        return float(val), gates, 0
    # RCCL moves device memory only: the few bytes exchanged live on this rank's GPU under "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    vals = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(vals, torch.tensor([float(val)], dtype=torch.float64, device=dev), group=group)
    owner = int(np.argmin([float(v[0]) for v in vals]))
    shape = torch.tensor(list(np.shape(gates)) if rank == owner else [0, 0, 0], dtype=torch.int64, device=dev)
    dist.broadcast(shape, src=owner, group=group)
    buf = torch.zeros(tuple(int(s) for s in shape.tolist()) + (2,), dtype=torch.float64, device=dev)
    if rank == owner:
        buf.copy_(torch.view_as_real(torch.as_tensor(np.ascontiguousarray(gates, np.complex128))))
    dist.broadcast(buf, src=owner, group=group)
    return float(vals[owner][0]), torch.view_as_complex(buf.cpu()).numpy().copy(), owner
