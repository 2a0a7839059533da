"""Debug aid: where the fused plane-keyed colour launches differ from the oracle (per half-sweep count, per-tile map)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GIPUMA_HIP_EXPERIMENTS"] = "1"
os.environ.setdefault("GIPUMA_HIP_ET_FORCE", "1")
os.environ.update({"GIPUMA_HIP_PUSH_LAUNCHES": "0", "GIPUMA_HIP_GROUP_FROM": "0",
                   "GIPUMA_HIP_GROUP_FUSED": sys.argv[1] if len(sys.argv) > 1 else "1"})
from gipuma_amd import synth
from gipuma_amd.problem import runcuda
from tests.oracle_lib import OracleState

def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

for it in (1,):
    gs, _ = synth.build_problem(synth.tiny_config(cols=150, rows=100, n_src=4, blocksize=15, iterations=it, n_best=3), colour=True)
    o = OracleState(gs).run()
    for rep in range(2):
        a = runcuda(gs)
        bad = (bits(a[0]) != bits(o[0])).any(-1)
        badc = bits(a[1]) != bits(o[1])
        print("iterations %d run %d: %d pixels differ in norm4, %d in cost" % (it, rep, bad.sum(), badc.sum()))
        if bad.any() and rep == 0:
            rows, cols = bad.shape
            for ty in range(0, rows, 16):
                print(" ".join("%3d" % bad[ty:ty + 16, tx:tx + 32].sum() for tx in range(0, cols, 32)))
            ys, xs = np.nonzero(bad)
            print("first:", list(zip(ys[:8].tolist(), xs[:8].tolist())), "parity (x+y)&1:", np.bincount((ys + xs) & 1, minlength=2).tolist())
