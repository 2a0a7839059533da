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


def test_push_sample_buffer_layout_is_conflict_free():
    """the strides shipped in PushLayout<15> (rows of the two stencil families 9 / 17 words apart, the
    horizontal family at word 124, groups 264 words apart, I-plane rows 41 words apart) keep every read of
    the chain phase conflict-free (ds_read_b32: 32 lanes x 32 banks); the dense layout is 2-way"""
    m = _load("push_banks")
    assert m.score_dis(264, 124, 9, 17) == (1.0, 1)
    assert m.score_ipl(41) == (1.0, 1)
    assert m.score_dis(2 * m.NF + 2, m.NF)[1] == 2 and m.score_ipl(29)[1] == 2
