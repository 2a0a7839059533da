"""The C-ABI library: loads, exports every symbol include/gipuma_hip.h declares, mirrors the
header's struct layouts, validates descriptors, and fails LOUDLY (no CPU fallback) without a GPU.
No compute calls here -- those are the gpu-marked parity tests."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from gipuma_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gipuma_hip.h")


def test_library_loads_and_exports_every_declared_symbol():
    lib = abi.load_library()
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(gipuma_hip_[a-z_]+)\s*\(", src))
    bound = {name for name, _, _ in abi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gipuma_hip_version() == abi.ABI_VERSION


def test_ctypes_structs_match_the_header_layout():
    """compile a tiny C program that prints sizeof/offsetof of the header's structs"""
    fields = {
        "gipuma_hip_camera": ["K", "K_inv", "R", "t", "M_inv", "P_col34", "C", "R_orig_inv", "fx",
                              "fy", "f", "alpha", "baseline", "depth_min", "depth_max"],
        "gipuma_hip_params": ["box_hsize", "box_vsize", "iterations", "n_best", "cost_comb", "alpha",
                              "tau_color", "tau_gradient", "gamma", "min_disparity", "max_disparity",
                              "good_factor"],
        "gipuma_hip_desc": ["abi_version", "rows", "cols", "channels", "pitch", "n_images", "images",
                            "cameras", "n_selected", "selected", "params", "seed", "device_id",
                            "stream", "flags"],
        "gipuma_hip_timing": ["ms_init", "ms_sweeps", "ms_finalize", "ms_total", "n_sweep_launches",
                              "ms_sweep_avg"],
    }
    py = {"gipuma_hip_camera": abi.Camera, "gipuma_hip_params": abi.Params,
          "gipuma_hip_desc": abi.Desc, "gipuma_hip_timing": abi.Timing}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, 'int main(void){']
    for s, fs in fields.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for f in fs:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    lines.append('return 0;}')
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "l.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(td, "l")
        subprocess.check_call(["gcc", "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l)
    for s, fs in fields.items():
        assert int(got[s]) == C.sizeof(py[s]), s
        assert [f for f, _ in py[s]._fields_] == fs
        for f in fs:
            assert int(got["%s.%s" % (s, f)]) == getattr(py[s], f).offset, (s, f)


def test_create_validates_the_descriptor():
    lib = abi.load_library()
    gs, _ = synth.build_problem(synth.tiny_config())
    h = C.c_void_p()

    def rc_with(**kw):
        d = abi.Desc()
        C.memmove(C.byref(d), C.byref(gs.desc), C.sizeof(d))
        for k, v in kw.items():
            if k.startswith("p_"):
                setattr(d.params, k[2:], v)
            else:
                setattr(d, k, v)
        return lib.gipuma_hip_create(C.byref(d), C.byref(h)), lib.gipuma_hip_last_error()

    assert rc_with(abi_version=99)[0] == -1
    assert rc_with(rows=0)[0] == -1
    assert rc_with(pitch=3)[0] == -1
    assert rc_with(channels=3)[0] == -4           # 1 = gray (T=float) or 4 = colour (T=float4) only
    assert rc_with(channels=4)[0] == -1           # colour needs pitch >= 4*cols
    assert rc_with(n_selected=33)[0] == -1        # costVector[32], gipuma.cu:736
    rc, msg = rc_with(p_box_hsize=8)
    assert rc == -1 and b"odd" in msg              # main.cpp:269-276
    assert rc_with(p_box_hsize=51, p_box_vsize=51)[0] == -4
    assert lib.gipuma_hip_create(None, C.byref(h)) == -1
    assert lib.gipuma_hip_destroy(None) == 0


def test_no_gpu_means_a_loud_error_not_a_fallback():
    """on a box without a HIP device every compute entry point refuses; nothing is computed on
    the CPU.  (On the GPU box this test checks that a device IS found.)"""
    lib = abi.load_library()
    gs, _ = synth.build_problem(synth.tiny_config())
    h = C.c_void_p()
    rc = lib.gipuma_hip_create(C.byref(gs.desc), C.byref(h))
    if lib.gipuma_hip_device_count() == 0:
        assert rc == -3 and b"no CPU fallback" in lib.gipuma_hip_last_error()
        from gipuma_amd.problem import runcuda
        with pytest.raises(abi.GipumaHipError):
            runcuda(gs)
    else:
        assert rc == 0
        lib.gipuma_hip_destroy(h)


def test_missing_library_raises():
    with pytest.raises(abi.GipumaHipError):
        abi.load_library(os.path.join(ROOT, "gipuma_amd", "csrc", "does_not_exist.so"))


def test_product_never_touches_the_oracle():
    """nothing under gipuma_amd/ (or the C-ABI sources) may import, include or link oracle/"""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "gipuma_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle", txt, re.I):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    so = os.path.join(ROOT, "gipuma_amd", "csrc", "libgipuma_hip.so")
    if os.path.exists(so):
        needed = subprocess.check_output(["readelf", "-d", so]).decode()
        assert "oracle" not in needed


def test_flag_constants_match_the_header_and_the_other_flavours_are_not_exported():
    """the bindings' GIPUMA_HIP_FLAG_* values are the header's; the library's dynamic symbol table holds the declared C-ABI
    only -- the entry points of the tolerance-judged and reference-order flavours (gipuma_hipf_*, gipuma_hipl_*, reached
    through GIPUMA_HIP_FLAG_FAST / _LITERAL) have hidden visibility"""
    src = open(HEADER).read()
    flags = {k: int(v) for k, v in re.findall(r"#define\s+GIPUMA_HIP_FLAG_([A-Z_]+)\s+(\d+)u", src)}
    assert flags == {"IMAGES_ON_DEVICE": abi.FLAG_IMAGES_ON_DEVICE, "UNFUSED": abi.FLAG_UNFUSED,
                     "CACHE_IMAGES": abi.FLAG_CACHE_IMAGES, "FAST": abi.FLAG_FAST, "LITERAL": abi.FLAG_LITERAL}
    assert len(set(flags.values())) == len(flags) and all(v & (v - 1) == 0 for v in flags.values())
    out = subprocess.check_output(["nm", "-D", "--defined-only", abi.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    c_abi = {e for e in exported if e.startswith("gipuma_hip")}
    assert c_abi == {name for name, _, _ in abi.SYMBOLS}, c_abi ^ {name for name, _, _ in abi.SYMBOLS}
