"""Experiment: is the first (random-plane) iteration bound by the L2 working set of ALL source views?
Runs iteration 0 of config C with k = 1, 2, 5, 10 selected views and prints sweep ms per view."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gipuma_amd import synth
from gipuma_amd.problem import Session

dev = torch.device("cuda:0")
gs, info = synth.build_problem("C", ref_view=14, device=dev, keep_on_device=True, iterations=1)
for k in (10, 5, 2, 1):
    gs.desc.n_selected = k
    with Session(gs) as s:
        s.solve()
        t = s.solve()
    print("n_selected=%2d  sweeps %.2f ms (2 launches)  -> %.3f ms per view per launch" %
          (k, t.ms_sweeps, t.ms_sweeps / 2 / k))
