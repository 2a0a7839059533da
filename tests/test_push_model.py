"""CPU check of the address arithmetic of gipuma_amd/csrc/pm_push.h: scripts/exp/push_model.py restates
every index formula of the push kernel (stencil points per step, sample-buffer slots, chain reads,
checkerboard-compressed reference tile) and compares them with the plain definition -- window samples
q = p + (2i-7, 2j-7) of the eight consumers of a producer (reference gipuma.cu:633-676, 1437-1462)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", "exp", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_push_index_model():
    _load("push_model").main_all()


def test_push_sample_buffer_layout_is_at_most_two_way_conflicted():
    """the strides shipped in PushLayout<15> (dstride = 2*104 + 2, horizontal family at 104, tile rows of 29)
    keep every chain read at <= 2 distinct addresses per LDS bank (ds_read_b32: 32 lanes x 32 banks)"""
    m = _load("push_banks")
    mean, worst = m.score_dis(2 * m.NF + 2, m.NF)
    assert worst <= 2, (mean, worst)
    mean, worst = m.score_ipl(29)
    assert worst <= 2, (mean, worst)
