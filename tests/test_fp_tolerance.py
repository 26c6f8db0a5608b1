"""CPU: the fp32 tolerance north_star asks for, STATED.  HIP == oracle is bit-exact (tests/test_gpu_parity.py), and the oracle ==
the reference's own headers built host-only, bit for bit (test_oracle_pin.py) -- but the reference's real build is nvcc's: FMA
contraction on, CUDA's libm.  No CPU (or HIP) build reproduces that bit for bit, and because the shading RNG is seeded by
stream position (src/pathtrace.cu:373) a single last-bit flip of a hit/miss re-seeds the rest of a bounce.  So the tolerance
between "this implementation" and "the CUDA build" can only be the tolerance between two legal arithmetics of the same source,
and it is statistical.  Here the oracle is built both ways (tests/fp_tolerance.py) and the bound is asserted:

  * iteration 1: the rays entering bounce 1 agree to within 0.1 % (the camera rays' discrete decisions almost all agree); the
    later per-bounce counts differ like two samples of the same process (within 5 sigma of counting noise), because a flipped
    decision re-seeds the paths behind it;
  * at 1 spp fewer than 8 % of the pixels differ at all (measured: C4 1.8 % -- the figure SURVEY 7 found for contraction
    alone --, C3 0 %, C2 4.8 %: with antialiasing off every primary ray meets the geometry exactly on a pixel centre);
  * at 16 spp the frame means differ by < 2 % per channel and by < 4 standard errors of the Monte-Carlo estimate, and the
    per-pixel RMS difference stays below 1.5 x the per-pixel Monte-Carlo noise (two INDEPENDENT renders would sit at 1.41,
    which is where C2 ends up: 1.39; C3 and C4 stay at 0.26-0.32 because most paths never meet a flipped decision):
    the builds are different samples of the same estimator, not different estimators.
The 64-spp figures for C2-C4 are in profiles/fp_tolerance_round2.json (tools/fp_tolerance_report.py)."""
import numpy as np
import pytest

import fp_tolerance


@pytest.mark.parametrize("config", ["C2", "C3", "C4"])
def test_contracted_arithmetic_stays_within_the_stated_tolerance(product, config):
    if not fp_tolerance.cpu_has_fma():
        pytest.skip("this CPU has no FMA: the contracted build cannot run here")
    r = fp_tolerance.measure(product, config, spp_marks=(1, 16))
    ca, cb = r["rays_per_bounce_iter1"]["A"], r["rays_per_bounce_iter1"]["B"]
    assert ca[0] == cb[0] and len(ca) == len(cb)
    assert abs(ca[1] - cb[1]) <= 0.001 * ca[1] + 2, (ca, cb)           # who survives the camera ray's hit: hardly any flips
    for a, b in zip(ca[2:], cb[2:]):                                     # later: re-seeded paths, i.e. sampling noise only
        assert abs(a - b) <= 5.0 * np.sqrt(a + b) + 2, (ca, cb)
    one, many = r["spp"][1], r["spp"][16]
    assert one["flipped_pixel_fraction"] < 0.08, one
    assert max(many["frame_mean_relative_difference"]) < 0.02, many
    assert max(many["frame_mean_difference_in_standard_errors"]) < 4.0, many
    assert many["pixel_rms_difference_over_mc_noise"] < 1.5, many
    assert np.isfinite(many["max_abs_pixel_difference"])


def test_every_golden_fixture_states_its_provenance():
    """tests/golden/make_golden.py labels each fixture: produced by CALLING the reference's own function ("direct") or by the
    restated kernel loop of oracle/ref_driver.cpp ("restated") -- so that nobody reads a frame-level fixture as the reference's."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for f in os.listdir(here):
        if f.endswith(".npz"):
            keys = [k for k in m.PROVENANCE if f.startswith(k)]
            assert keys, f
            assert m.PROVENANCE[max(keys, key=len)] in ("direct", "restated", "direct+restated")
    assert all(m.PROVENANCE[k] == "restated" for k in m.PROVENANCE if k.startswith("render_"))
