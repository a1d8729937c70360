import sys; sys.path.insert(0,'/root/repo')
import tensorrl_qas_amd as tq, numpy as np
ham = tq.hamiltonian.synthetic_lih12()
eng = tq.VQEEngine(12); eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
lay = eng.hamiltonian_layout(); print(lay)
