# phase times inside the fused env-step kernel (diagnostic -DVQE_STAMPS build)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n=12; H=tq.hamiltonian.synthetic_lih12(); psi0=tq.hamiltonian.brickwork_state(n,12)
for G in (64,):
    eng=tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask,H.zmask,H.coeff)
    B=2048
    b=bench.make_batch(tq,n,B,G,1000)
    eng.batch_load_flat(b["gate_off"],b["kind"],b["q0"],b["q1"],b["pidx"],b["par_off"],b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    eng.batch_run_env_step(1.0,1e-4,1000); eng.sync(); ms=eng.last_kernel_ms()
    c=eng.debug_counters().astype(float)
    ev=c[0]; tot=(c[1]+c[2]+c[3])
    print(f"G={G} P~{G//2}: kernel {ms:.1f} ms, evals {int(ev)}; per eval (100 MHz ticks->us?) circuit {c[1]/ev:.0f}  energy {c[2]/ev:.0f}  tell {c[3]/ev:.0f} ticks; shares {c[1]/tot:.2f}/{c[2]/tot:.2f}/{c[3]/tot:.2f}; circuit: init {c[5]/ev:.0f} relayouts {c[6]/ev:.0f} scatter {c[7]/ev:.0f}", flush=True)
