"""CPU: what the compiler made of the kernels (hipcc cross-compiles gfx950 without a GPU; `make resource-usage` =
-Rpass-analysis=kernel-resource-usage on pt_engine.hip).  The specialised bounce kernels are launched as seven workgroups per CU
and compiled for seven waves per SIMD: a change that pushes them over 72 registers would not fail any parity
test, it would spill -- this test is where that shows."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def usage():
    if not shutil.which("hipcc"):
        pytest.skip("hipcc is not on PATH: nothing to compile the kernels with")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "mygpuraytracer_amd", "csrc"), "resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in (r.stdout + r.stderr).splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("waves", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def _bounce(usage, first, mode, fast):
    tag = "k_bounceILb%dELi%dELb%dE" % (first, mode, fast)
    hits = [v for k, v in usage.items() if tag in k]
    assert len(hits) == 1, (tag, list(usage))
    return hits[0]


def test_no_kernel_of_the_path_spills(usage):
    names = [k for k in usage if "k_bounce" in k or "k_mesh" in k or "k_gather" in k]
    assert len(names) >= 14                      # 12 k_bounce variants + k_mesh + k_gather (k_move is gone: round 3)
    for k in names:
        assert usage[k]["scratch"] == 0, (k, usage[k])


def test_specialised_bounce_kernels_fit_seven_waves(usage):
    # only what the launch configuration relies on (enqueue_batch launches these as 7 workgroups per CU = 7 waves per SIMD; the
    # split bounce's halves and k_mesh as what their launch bounds say) -- not the register counts of one compiler version
    for first in (0, 1):
        u = _bounce(usage, first, 0, 1)
        assert u["waves"] >= 7, u
        assert _bounce(usage, first, 1, 1)["waves"] >= 4
        assert _bounce(usage, first, 2, 1)["waves"] >= 5
    mesh = [v for k, v in usage.items() if "k_mesh" in k][0]
    assert mesh["waves"] >= 5, mesh
