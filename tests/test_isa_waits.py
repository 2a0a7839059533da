"""Compile-time properties of the gfx950 code object that cost measurable time when they were violated (DESIGN.md 5,
round 4) -- checked on the assembly hipcc emits, so no GPU is needed:

  * the strip loop of the plane-keyed kernels keeps the other register set's window load in flight: the `vmcnt`
    operands of its waits are 1 (gray: two loads per iteration) / 3 (colour: six), never 0.  With two specialised
    copies of the loop the compiler drained both loads at the top of every iteration of one of them (config C +3.5 %);
  * no `flat_*` memory instruction: the pointers held in the Problem block are global-address-space pointers
    (DevPtr, pm_core.h), and a flat access counts on lgkmcnt as well as vmcnt.
"""
import hashlib
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gipuma_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]


def _compile_to_assembly(tu):
    """assembly of one translation unit, cached per source state under /tmp.  The file appears under its final name only
    when the compile has finished and the text is complete (an interrupted or concurrent run -- pytest-xdist -- must not
    leave a truncated cache entry that later runs parse)."""
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    key = hashlib.sha256(b"".join(open(os.path.join(CSRC, f), "rb").read() for f in srcs)).hexdigest()[:16]
    out = os.path.join("/tmp", "gipuma_hip_%s_%s.s" % (key, tu.replace(".hip", "")))
    if not os.path.exists(out):
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(out) + ".", suffix=".part", dir="/tmp")
        os.close(fd)
        try:
            subprocess.run([HIPCC] + FLAGS + ["-S", "--offload-device-only", "-o", tmp, tu], cwd=CSRC,
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert "s_endpgm" in open(tmp).read(), "hipcc left an incomplete assembly file"
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    text = open(out).read()
    assert "s_endpgm" in text, "cached assembly %s is incomplete: delete it" % out
    return text.split("\n")


@pytest.fixture(scope="module")
def asm():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    return _compile_to_assembly("gipuma_hip.hip")


def kernel(asm, prefix):
    start = next(i for i, l in enumerate(asm) if l.startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    return asm[start:end]


def loops(body):
    label = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i:
            out.append((label[m.group(1)], i))
    return out


@pytest.mark.parametrize("prefix,loads,depth", [
    ("_ZN2pm18sweep_group_kernelILi11ELi1E", 2, 1), ("_ZN2pm18sweep_group_kernelILi15ELi1E", 2, 1),
    ("_ZN2pm18sweep_group_kernelILi25ELi1E", 2, 1), ("_ZN2pm12group_kernelILi15ELi1E", 2, 1),
    ("_ZN2pm12group_kernelILi15ELi4E", 6, 3)])
def test_strip_loop_keeps_a_window_load_in_flight(asm, prefix, loads, depth):
    k = kernel(asm, prefix)
    found = []
    for a, b in loops(k):
        body = k[a:b + 1]
        n_loads = sum("global_load_dwordx4" in l for l in body)
        # the strip loop: both reciprocals (one loop serves them), the window loads of its two register sets, and the
        # dis values written to the wavefront's LDS buffer (which tells it from the refinement stage's loops)
        if (n_loads == loads and any("v_div_scale" in l for l in body) and any("v_rcp_f32" in l for l in body) and
                any("ds_write_b32" in l for l in body)):
            waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in body if "s_waitcnt" in l and "vmcnt" in l]
            found.append((b - a, waits))
    assert found, "strip loop not recognised in %s: update this test to the loop's new shape" % prefix
    length, waits = max(found)  # (the whole iteration, not a sub-cycle of its branches)
    assert len(waits) >= 2 and min(waits) >= depth, "strip loop of %s waits vmcnt%s" % (prefix, waits)


def test_no_flat_memory_instruction(asm):
    bad = [l.strip() for l in asm if re.match(r"\s+flat_(load|store|atomic)", l)]
    assert not bad, "%d flat accesses, e.g. %s" % (len(bad), bad[:3])


@pytest.mark.parametrize("prefix,max_scratch", [
    ("_ZN2pm11push_kernelILi15EE", 0), ("_ZN2pm11push_kernelILi19EE", 0), ("_ZN2pm11push_kernelILi25EE", 96),
    ("_ZN2pm17sweep_cols_kernelILi15ELb1ELi1EE", 0), ("_ZN2pm17sweep_cols_kernelILi25ELb1ELi1EE", 0),
    ("_ZN2pm16init_cols_kernelILi15ELb1ELi1EE", 0), ("_ZN2pm12group_kernelILi15ELi4EE", 0)])
def test_first_half_sweep_kernels_do_not_spill(asm, prefix, max_scratch):
    """the kernels of the random-plane phase (push, column-per-lane, init) and the colour plane-keyed kernel hold everything
    in registers: no scratch access and no scalar register spilled into a vector lane anywhere in their code.  (The fused
    kernels -- sweep_kernel, sweep_group_kernel -- do spill: ~100 scalars into lanes and a few registers to scratch, outside
    their sample loops)  One exception since round 6: push_kernel<25>, whose 60-step stencil is fully
    unrolled (config D 4 % faster than the rolled loop, which spilled nothing): 256 registers and 76 spill accesses per
    ~12 000 instructions of a view -- bounded here so that it does not grow unnoticed."""
    k = kernel(asm, prefix)
    assert len([l for l in k if re.match(r"\s+scratch_", l)]) <= max_scratch, prefix
    assert not [l for l in k if "v_writelane_b32" in l or "v_readlane_b32" in l], prefix


@pytest.mark.parametrize("prefix", ["_ZN2pm18sweep_group_kernelILi15ELi1E", "_ZN2pm18sweep_group_kernelILi19ELi1E",
                                    "_ZN2pm18sweep_group_kernelILi25ELi1E", "_ZN2pm18sweep_group_kernelILi11ELi1E"])
def test_fused_kernel_spills_stay_out_of_its_sample_loops(asm, prefix):
    """Round 5's review counted 263 scalar-spill lane accesses and 90 scratch accesses in sweep_group_kernel<15, 1> and asked
    for none of them in the strip loop.  Where they are: every INNERMOST loop that does sample work -- the strips, the chains,
    the refinement's bounded and exact chains, the prefilter (recognised by the byte -> float conversions of window texels, or LDS-fed chains) -- holds everything
    in registers: no scratch access, no v_writelane / v_readlane.  The spills sit in the straight-line code between the
    stages and in the per-candidate set-up of the refinement (homography operands, parameters), executed once per batch or
    candidate, not per sample (95 .. 105 v_writelane per kernel)."""
    k = kernel(asm, prefix)
    ls = sorted(set(loops(k)))
    hot = 0
    for a, b in ls:
        if any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in ls):
            continue  # not innermost
        body = k[a:b + 1]
        # sample work: the byte -> float conversions of a window's texels (strips, refinement chains, prefilter), or a chain
        # over the wavefront's dis buffer (LDS reads feeding >= 60 VALU instructions, no global load)
        samples = any("v_cvt_f32_ubyte" in l for l in body)
        chain = (sum(1 for l in body if re.match(r"\s+v_", l)) >= 60 and sum("ds_read" in l for l in body) >= 8 and
                 not any("global_load" in l for l in body))
        if not (samples or chain):
            continue
        hot += 1
        bad = [l.strip() for l in body if re.match(r"\s+scratch_", l) or "v_writelane_b32" in l or "v_readlane_b32" in l]
        assert not bad, "%s: sample loop at +%d (%d instructions) spills: %s" % (prefix, a, b - a, bad[:4])
    assert hot >= 4, "sample loops not recognised in %s" % prefix
    assert sum("v_writelane_b32" in l for l in k) <= 128, prefix  # (95 .. 105 today, all outside the sample loops)
