"""The named configuration (TensorRL_fixed/LIH12q_TNbond2 episodes through VecCircuitEnv) driven by
several host processes that share ONE GPU - parallel RL seeds the way a user would run them: the
per-environment Python bookkeeping (state tensors, illegal actions, rewards) is what limits a single
process, the device is not.  The parent never touches the GPU (workers are
spawned interpreters).  usage: python tools/probe_episode_procs.py [procs=4] [envs_per_proc=2048] [steps=110]"""
import json
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(i, procs, envs, steps, barrier, q):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import tensorrl_qas_amd as tq
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    q.put(bench.episode_aux(tq, torch, 0, envs, steps, barrier=barrier, seed0=2 * i))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    envs = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 110
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(i, procs, envs, steps, barrier, q)) for i in range(procs)]
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    total = sum(r["env_steps"] for r in res)
    wall = max(r["t_end"] for r in res) - min(r["t_start"] for r in res)
    print(json.dumps({"workload": f"{procs} host processes x ({res[0]['workload']}) on one GPU",
                      "env_steps_per_s_wall": total / wall, "env_steps": total, "wall_s": wall,
                      "per_process_wall": [r["env_steps_per_s_wall"] for r in res],
                      "mean_nfev_per_step": sum(r["mean_nfev_per_step"] for r in res) / procs}), flush=True)
