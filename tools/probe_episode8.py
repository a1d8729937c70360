# the reference's most common size through the environment API: TensorRL_fixed/H2O8q_TNbond2 (shipped Hamiltonian and init
# circuit, 20 steps per episode) through VecCircuitEnv with the compiled host loop, uniformly random legal actions
import sys, time, json, tempfile, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
from helpers import make_data_root, reference_config
from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
from tensorrl_qas_amd.environments.vec_env import VecCircuitEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
conf = reference_config("TensorRL_fixed/H2O8q_TNbond2", make_data_root(tempfile.mkdtemp()))
half = B // 2
vecs = [VecCircuitEnv(CircuitEnv, conf, torch.device("cuda:0"), half, seed=s, native=True) for s in (0, 1)]
tdict = vecs[0]._proto._actions_table
table = np.array([tdict[i] for i in range(len(tdict))], np.int32)
rng = np.random.default_rng(7)
for v in vecs: v.reset()
n_steps = vecs[0]._proto.num_layers_termination
def choose(vec):
    ill = vec.illegal_actions_array()
    a = rng.integers(0, table.shape[0], vec.num_envs)
    bad = (ill == a[:, None]).any(axis=1)
    while bad.any():
        a[bad] = rng.integers(0, table.shape[0], int(bad.sum()))
        bad = (ill == a[:, None]).any(axis=1)
    return table[a]
steps = 0; nfev = 0.0; t_gpu = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
vecs[0].step_async(choose(vecs[0]))
for it in range(n_steps):
    vecs[1].step_async(choose(vecs[1]))
    vecs[0].step_wait(); t_gpu += vecs[0].engine.last_kernel_ms() * 1e-3; steps += half; nfev += float(np.sum(vecs[0].nfev))
    if it + 1 < n_steps: vecs[0].step_async(choose(vecs[0]))
    vecs[1].step_wait(); t_gpu += vecs[1].engine.last_kernel_ms() * 1e-3; steps += half; nfev += float(np.sum(vecs[1].nfev))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
err = np.concatenate([v.errors for v in vecs])
print(json.dumps({"config": "TensorRL_fixed/H2O8q_TNbond2", "envs": B, "steps_per_episode": n_steps, "env_steps": steps,
                  "env_steps_per_s_wall": steps / dt, "env_steps_per_s_device": steps / t_gpu, "mean_nfev": nfev / steps,
                  "final_error_mean": float(err.mean()), "final_error_min": float(err.min())}))
