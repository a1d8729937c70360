"""Host-thread placement.

A GPU queue is fed by the launching thread TOGETHER WITH the HIP / ROCr runtime's helper threads (signal handlers,
the graph executor).  On the two-socket, sixteen-L3-domain hosts of the MI355X pool the scheduler spreads those threads
over the machine, and every hand-over between two of them that crosses an L3 domain costs tens of microseconds: a loop
of ~10 us launches (the streaming MPS -> PQC fit: 3300 launches per fit) then runs at 17.6 ... 100 ms from process to
process, and always at 17.6 ms when the process is confined to one L3 domain.  Long kernels do not notice.

``bind_host_threads()`` confines the CALLING thread - and with it every thread created afterwards, which is why it has
to run before the first HIP call - to the cores of one L3 domain (one CCD).  Thread pools that exist already (the BLAS
pool numpy started at import) keep their placement.  ``VQE_CPU_BIND=0`` switches it off."""
import os


def l3_domains(cpus=None):
    """The L3 domains (lists of logical CPUs) among ``cpus`` (default: the CPUs this thread may run on), in CPU order."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    dom = {}
    for c in cpus:
        try:
            key = open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read().strip()
        except OSError:
            return []
        dom.setdefault(key, []).append(c)
    return sorted(dom.values(), key=lambda v: v[0])


def bind_host_threads(local_rank=0, ranks_per_node=8):
    """Confine the calling thread (and the threads it creates from now on) to one L3 domain: a single process takes the
    domain it is running on; under a launcher (LOCAL_RANK set) rank r of a node takes domain r * (domains //
    ranks_per_node), so that the ranks of a node spread over its sockets the way its GPUs do.
    Returns the CPU list, or None when nothing was done (switched off, one domain only, no sysfs)."""
    if os.environ.get("VQE_CPU_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    doms = l3_domains()
    if len(doms) <= 1:
        return None
    # start from the domain the scheduler put this thread on (on a shared host the first domain is everybody's first
    # choice), ranks of one node a fixed stride apart
    here = 0
    if hasattr(os, "sched_getcpu"):
        cpu = os.sched_getcpu()
        here = next((i for i, d in enumerate(doms) if cpu in d), 0)
    stride = max(1, len(doms) // max(1, ranks_per_node))
    first = here if ranks_per_node <= 1 or "LOCAL_RANK" not in os.environ else 0
    cpus = doms[(first + local_rank * stride) % len(doms)]
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus
