"""Round 6 experiment: the density-matrix showcase (tools/bench_density.py) with the pass planner's wide search (what states of
>= 2^31 amplitudes get by default) and a wider one.  usage: python tools/experiments/r06_density_planner.py default|wide|wider [bench_density args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepquantum_amd as dq  # noqa: E402

mode = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
if mode == 'wide':
    dq.executor.CONFIG['plan_big_amps'] = 1 << 28
if mode == 'wider':
    dq.executor.CONFIG.update(plan_width=16, plan_branch=4, plan_restarts=12)
runpy.run_path(os.path.join(ROOT, 'tools', 'bench_density.py'), run_name='__main__')
