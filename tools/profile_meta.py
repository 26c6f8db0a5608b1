"""Shared by the profile collectors (collect_profiles.py, collect_sq.py, collect_sq_c5.py) and bench.py: kernel labels from
rocprofv3's kernel names, the hash of the kernel sources a profile was taken with, and the description of the profiled command
that the profiling scripts leave next to their outputs."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("pt_engine.hip", "pt_device.h", "pt_bvh.h", "pt_arith.hip", "Makefile")


def source_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "mygpuraytracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def kernel_label(name):
    """'k_bounce' / 'k_bounce<first>' for the unsplit kernel (MODE 0); the halves of the split bounce apart: 'k_bounce pass1',
    'k_bounce<first> pass1', '... pass2'; k_mesh / k_mesh<first>; k_gather.  Both spellings rocprofv3 has used (demangled
    k_bounce<false, 1, true>, mangled k_bounceILb0ELi1ELb1E)."""
    m = re.search(r"k_bounce<\s*(true|false)\s*,\s*(\d)", name) or re.search(r"k_bounceILb([01])ELi(\d)", name)
    if m:
        first = m.group(1) in ("true", "1")
        mode = int(m.group(2))
        return ("k_bounce<first>" if first else "k_bounce") + ("" if mode == 0 else " pass%d" % mode)
    m = re.search(r"k_mesh<\s*(true|false)", name) or re.search(r"k_meshILb([01])", name)
    if m:
        return "k_mesh<first>" if m.group(1) in ("true", "1") else "k_mesh"
    m = re.search(r"k_finish<\s*(true|false)", name) or re.search(r"k_finishILb([01])", name)
    if m:
        return "k_finish<first>" if m.group(1) in ("true", "1") else "k_finish"
    for k in ("k_gather", "k_onepass"):
        if k in name:
            return k
    return None


def how_of(out_dir):
    """<out_dir>_how.txt as the profiling script wrote it, or a statement that it is missing (never a guess)."""
    p = out_dir.rstrip("/") + "_how.txt"
    if os.path.exists(p):
        return open(p).read().strip()
    return "UNRECORDED command (no %s next to the counter files)" % os.path.basename(p)
