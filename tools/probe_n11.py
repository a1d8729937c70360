# occupancy experiment at n = 11 (32 KiB of state per workgroup): same kernel, launch bounds 2 vs 4 waves/SIMD
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n=11; rng=np.random.default_rng(5)
H=tq.hamiltonian.synthetic_lih12()
keep=(H.xmask < 2048) & (H.zmask < 2048)
xs,zs,cs=H.xmask[keep],H.zmask[keep],H.coeff[keep]
psi0=tq.hamiltonian.brickwork_state(n,11)
eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(xs,zs,cs)
B=4096; G=64
b=bench.make_batch(tq,n,B,G,1000)
eng.batch_load_flat(b["gate_off"],b["kind"],b["q0"],b["q1"],b["pidx"],b["par_off"],b["theta"])
eng.batch_set_new_gate(b["new_gate"])
eng.batch_run_env_step(1.0,1e-4,1000); eng.sync()
eng.batch_run_env_step(1.0,1e-4,1000); eng.sync(); ms=eng.last_kernel_ms()
_,f,nfev=eng.batch_fetch(want_x=False)
print(f"{sys.argv[1]}: n=11 terms={len(cs)} groups={len(set(xs.tolist()))} B={B}: {ms:.1f} ms  {(nfev.sum()+B)/ms/1e3:.2f} M evals/s  wg/cu={eng.device_info()['wg_per_cu']}", flush=True)
