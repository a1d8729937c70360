import sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import torch, tensorrl_qas_amd as tq, bench
bench.episode_aux(tq, torch, 0, 64, 5)
pr = cProfile.Profile(); pr.enable()
r = bench.episode_aux(tq, torch, 0, 256, 40)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:4500]); print(r)
