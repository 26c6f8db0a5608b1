#!/usr/bin/env python3
"""CPU only: SGPR spill traffic of the specialised bounce kernels as hipcc builds them -- v_readlane / v_writelane (scalar values parked in
vector-register lanes: each restore is a VECTOR instruction, plus wait states) inside and outside the tile loop, next to the vector
instructions there.  The tile loop = the longest backward branch of the kernel.   python tools/spill_report.py [EXTRA flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mygpuraytracer_amd", "csrc")
OPT = "-Os -fno-unroll-loops -fno-slp-vectorize -mllvm -disable-machine-licm".split()
KERNELS = {"k_bounce<false,0,FAST>": "_ZN12_GLOBAL__N_18k_bounceILb0ELi0ELb1EEEvNS_12BounceParamsE", "k_bounce<true,0,FAST>": "_ZN12_GLOBAL__N_18k_bounceILb1ELi0ELb1EEEvNS_12BounceParamsE",
           "k_bounce<false,1,FAST>": "_ZN12_GLOBAL__N_18k_bounceILb0ELi1ELb1EEEvNS_12BounceParamsE", "k_bounce<true,1,FAST>": "_ZN12_GLOBAL__N_18k_bounceILb1ELi1ELb1EEEvNS_12BounceParamsE"}
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", *OPT, "-fPIC", "-ffp-contract=off", *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage",
                        "--cuda-device-only", "-S", "pt_engine.hip", "-o", out], cwd=CSRC, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    text = open(out).read().splitlines()
    res = {}
    cur = None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m: cur = m.group(1); res[cur] = {}
        m = re.search(r"remark:\s+(VGPRs|TotalSGPRs|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", ln)
        if m and cur: res[cur][m.group(1)] = int(m.group(2))
    for name, sym in KERNELS.items():
        try:
            a = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
        except StopIteration:
            continue
        b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
        body = text[a:b]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        back = [(labels[t], i) for i, l in enumerate(body) for m in [re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if m for t in [m.group(1)] if t in labels and labels[t] < i]
        la, lb = max(back, key=lambda x: x[1] - x[0])
        cnt = lambda seg, pat: sum(1 for l in seg if re.match(r"\s+" + pat, l))
        loop, rest = body[la:lb], body[:la] + body[lb:]
        u = res.get(sym, {})
        print("%-24s vgpr %3s sgpr %3s spilled %3s scratch %s occ %s | tile loop: valu %4d readlane %3d writelane %3d s_load %3d | outside: valu %4d readlane %3d writelane %3d" % (
            name, u.get("VGPRs"), u.get("TotalSGPRs"), u.get("SGPRs Spill"), u.get("ScratchSize [bytes/lane]"), u.get("Occupancy [waves/SIMD]"),
            cnt(loop, "v_"), cnt(loop, "v_readlane"), cnt(loop, "v_writelane"), cnt(loop, "s_load"), cnt(rest, "v_"), cnt(rest, "v_readlane"), cnt(rest, "v_writelane")))
