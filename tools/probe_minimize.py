# breakdown of the fused env-step kernel: vary Hamiltonian size / circuit content
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq
import bench
n=12; H=tq.hamiltonian.synthetic_lih12(); psi0=tq.hamiltonian.brickwork_state(n,12)
B=2048
def run(label, ham, G, maxfun=1000, kinds=None):
    eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    b=bench.make_batch(tq,n,B,G,1000)
    if kinds is not None:
        b["kind"][b["kind"]>0]=kinds
    eng.batch_load_flat(b["gate_off"],b["kind"],b["q0"],b["q1"],b["pidx"],b["par_off"],b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    eng.batch_run_env_step(1.0,1e-4,maxfun); eng.sync()
    eng.batch_run_env_step(1.0,1e-4,maxfun); eng.sync(); ms=eng.last_kernel_ms()
    _,f,nfev=eng.batch_fetch(want_x=False)
    ev=nfev.sum()+B
    print(f"{label:40s} G={G:3d} {ms:8.1f} ms  mean nfev={nfev.mean():6.1f}  {ev/ms/1e3:6.2f} M evals/s  {ms*1e3/ (ev/512):7.2f} us/eval/WG-slot", flush=True)
full=(H.xmask,H.zmask,H.coeff)
one=(H.xmask[:1]*0,H.zmask[:1],H.coeff[:1])
run("full H, mixed gates", full, 64)
run("1-term H, mixed gates (gates+COBYLA)", one, 64)
run("1-term H, all RZ", one, 64, kinds=3)
run("full H, G=16", full, 16)
run("1-term H, G=16", one, 16)
