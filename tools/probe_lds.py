# quick perf probe (not the bench): 12q, synthetic LiH-like H, B random circuits
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq
n=12; H=tq.hamiltonian.synthetic_lih12(); psi0=tq.hamiltonian.brickwork_state(n,12)
eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask,H.zmask,H.coeff)
for G,B in ((64,2048),(32,2048),(110,1024)):
    rng=np.random.default_rng(G)
    cs=[];ts=[]
    for b in range(B):
        c,t=tq.circuits.random_circuit(n,G,rng); cs.append(c); ts.append(t.astype(np.float32).astype(float))
    eng.batch_load(cs,ts)
    eng.batch_run_energy(); eng.sync(); 
    t0=time.time(); eng.batch_run_energy(); eng.sync(); dt=time.time()-t0
    P=np.mean([c.n_params for c in cs])
    print(f"G={G} B={B} meanP={P:.1f} energy: {dt*1e3:.2f} ms  {B/dt:.0f} evals/s  kernel_ms={eng.last_kernel_ms():.2f}", flush=True)
    eng.batch_set_new_gate(np.full(B, G-1, np.int32))
    t0=time.time(); eng.batch_run_env_step(1.0,1e-4,1000); eng.sync(); dt=time.time()-t0
    x,f,nfev=eng.batch_fetch()
    print(f"   env_step: {dt*1e3:.1f} ms  {B/dt:.0f} steps/s  mean nfev={nfev.mean():.1f} max={nfev.max()} evals/s={nfev.sum()/dt:.0f} kernel_ms={eng.last_kernel_ms():.1f}", flush=True)
