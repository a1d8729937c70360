# where does an LDS-path evaluation spend its time?  (energy-only kernel, large batch)
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq
n=12; H=tq.hamiltonian.synthetic_lih12(); psi0=tq.hamiltonian.brickwork_state(n,12)
B=8192
def run(label, ham, G, p_cnot=0.5, kinds=None):
    eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    rng=np.random.default_rng(1)
    cs=[];ts=[]
    for b in range(B):
        c,t=tq.circuits.random_circuit(n,G,rng,p_cnot)
        if kinds is not None:
            c.kind[c.kind>0]=kinds
        cs.append(c); ts.append(t)
    eng.batch_load(cs,ts)
    eng.batch_run_energy(); eng.sync()
    ms=[]
    for _ in range(3):
        eng.batch_run_energy(); eng.sync(); ms.append(eng.last_kernel_ms())
    m=min(ms); P=np.mean([c.n_params for c in cs])
    print(f"{label:34s} G={G:3d} P={P:5.1f}  {m:7.3f} ms  {B/m*1e3/1e6:6.2f} M evals/s  {m*1e3/B*512:7.2f} us/eval/WG-slot", flush=True)
full=(H.xmask,H.zmask,H.coeff)
one=(H.xmask[:1]*0,H.zmask[:1],H.coeff[:1])
diag=(H.xmask[H.xmask==0],H.zmask[H.xmask==0],H.coeff[H.xmask==0])
run("no gates, 1 term", one, 0)
run("no gates, diagonal group (79 terms)", diag, 0)
run("no gates, full H (91 groups)", full, 0)
run("64 gates all CNOT, 1 term", one, 64, 1.0)
run("32 RZ, 1 term", one, 32, 0.0, 3)
run("32 RX, 1 term", one, 32, 0.0, 1)
run("32 RY, 1 term", one, 32, 0.0, 2)
run("64 gates mixed, 1 term", one, 64)
run("64 gates mixed, full H", full, 64)
