#!/usr/bin/env python3
"""Registers, scratch and code size of every kernel of libgipuma_hip.so: compiles gipuma_hip.hip to
assembly under /tmp (same flags as __graft_entry__.build) and prints the .amdhsa metadata per kernel.
usage: kernel_regs.py [substring ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPCC, HIP_FLAGS  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
out = "/tmp/gipuma_hip_gfx950.s"
flags = [f for f in HIP_FLAGS if f not in ("-shared", "-fPIC")]
if not os.environ.get("KR_REUSE"):
    subprocess.check_call([HIPCC] + flags + extra + ["--cuda-device-only", "-S", "-o", out,
                                                       os.path.join(ROOT, "gipuma_amd", "csrc", "gipuma_hip.hip")])
s = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if args and not any(a in dn for a in args):
        continue
    g = lambda k: re.search(k + r"\s+(\S+)", body).group(1)  # noqa: E731
    fn = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
    ninstr = sum(1 for ln in fn.group(1).split("\n") if re.match(r"\s+[sv]_|\s+ds_|\s+global_|\s+buffer_|\s+flat_|\s+scratch_", ln)) if fn else -1
    print("%-100s vgpr %3s sgpr %3s scratch %4s instr %6d" % (dn[:100], g(".amdhsa_next_free_vgpr"), g(".amdhsa_next_free_sgpr"),
                                                              g(".amdhsa_private_segment_fixed_size"), ninstr))


def scratch_report(sub):
    """scratch (spill) instructions of the kernels matching `sub`, grouped by their innermost loop"""
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if sub not in dn:
            continue
        labels, loops = {}, []
        for i, ln in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if mm:
                labels[mm.group(1)] = i
        for i, ln in enumerate(body):
            mm = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loops.append((labels[mm.group(1)], i))
        per = {}
        for i, ln in enumerate(body):
            if "scratch_" in ln:
                inl = [(a, b) for a, b in loops if a <= i <= b]
                key = min(inl, key=lambda t: t[1] - t[0]) if inl else None
                per[key] = per.get(key, 0) + 1
        print(dn[:80], {("loop@%d len %d" % (k[0], k[1] - k[0]) if k else "straight"): v for k, v in per.items()})


if os.environ.get("KR_SCRATCH"):
    scratch_report(os.environ["KR_SCRATCH"])
