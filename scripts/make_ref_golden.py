#!/usr/bin/env python3
"""Generate tests/golden/ref_tiny64.npz: inputs and outputs of the REFERENCE's own device code
(oracle/_ref/libgipuma_ref.so = /root/reference/gipuma.cu compiled for the CPU, see
oracle/ref_shim/) on one small problem, so the oracle stays pinned where the reference tree is
not available.

    make -C oracle && python scripts/make_ref_golden.py

Contents: the four 64x64 8-bit images, the camera blocks and parameters exactly as handed to
runcuda(), and the reference's state after init (planes + cost), after the first black sweep,
and after the whole run (2 iterations + gipuma_compute_disp).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gipuma_amd import abi, synth  # noqa: E402
from tests.ref_lib import RefState  # noqa: E402


FIXTURES = {
    # name: tiny_config arguments
    "ref_tiny64": dict(cols=64, rows=64, n_src=3, blocksize=7, iterations=2),
    # the shipped fast-path geometry: box 15 (8x8 samples), best-3 of 4 views (the reference needs sizes that are multiples of 32)
    "ref_box15": dict(cols=96, rows=64, n_src=4, blocksize=15, iterations=2, n_best=3),
}


def make(name, cfg):
    gs, info = synth.build_problem(synth.tiny_config(**cfg))
    r = RefState(gs, tex_mode=0)
    r.init_planes()
    init_n4, init_c = r.get_state()
    r.sweep(0, abi.BLACK)
    b_n4, b_c = r.get_state()
    r.sweep(0, abi.RED)
    r.sweep(1, abi.BLACK)
    r.sweep(1, abi.RED)
    pre_n4, pre_c = r.get_state()
    r.finalize()
    fin_n4, fin_c = r.get_state()
    cams = np.frombuffer(bytes(gs.cameras.c_array), dtype=np.float32).reshape(gs.cameras.n, -1).copy()
    p = gs.desc.params
    params = dict(box_hsize=p.box_hsize, box_vsize=p.box_vsize, iterations=p.iterations, n_best=p.n_best,
                  cost_comb=p.cost_comb, alpha=p.alpha, tau_color=p.tau_color, tau_gradient=p.tau_gradient,
                  gamma=p.gamma, min_disparity=p.min_disparity, max_disparity=p.max_disparity,
                  good_factor=p.good_factor)
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(
        out, images=np.stack(gs.images).astype(np.uint8), cameras=cams,
        selected=np.array(gs.selected, dtype=np.int32), seed=np.uint32(gs.desc.seed),
        param_names=np.array(list(params.keys())), param_values=np.array(list(params.values()), dtype=np.float64),
        init_norm4=init_n4, init_cost=init_c, black0_norm4=b_n4, black0_cost=b_c,
        presweep_final_norm4=pre_n4, final_norm4=fin_n4, final_cost=fin_c)
    print("wrote", out, os.path.getsize(out), "bytes")


def main():
    for name, cfg in FIXTURES.items():
        make(name, cfg)


if __name__ == "__main__":
    main()
