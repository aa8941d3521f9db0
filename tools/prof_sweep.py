import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_sweep
parity_sweep.run(1, True, 900, n=2000, log=lambda s: None)   # warm
pr = cProfile.Profile(); pr.enable()
parity_sweep.run(2, True, 500, n=2000, log=print)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
