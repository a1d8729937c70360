import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq
n=12; H=tq.hamiltonian.synthetic_lih12(); psi0=tq.hamiltonian.brickwork_state(n,12)
B=8192
eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask,H.zmask,H.coeff)
cs=[tq.Circuit.empty() for _ in range(B)]; ts=[np.zeros(0)]*B
eng.batch_load(cs,ts)
eng.batch_run_energy(); eng.sync()
ms=[]
for _ in range(3):
    eng.batch_run_energy(); eng.sync(); ms.append(eng.last_kernel_ms())
m=min(ms)
print(f"{sys.argv[1] if len(sys.argv)>1 else ''}: no gates, full H: {m:.3f} ms  {m*1e3/B*512:.2f} us/eval/WG-slot  {(m*1e3/B*512-7.3)/91:.3f} us/group", flush=True)
