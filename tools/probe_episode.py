# the bench's episode auxiliary at a chosen number of environments
import sys, json
sys.path.insert(0, '/root/repo')
import torch, tensorrl_qas_amd as tq, bench
print(json.dumps(bench.episode_aux(tq, torch, 0, int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 110)), flush=True)
