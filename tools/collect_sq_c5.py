#!/usr/bin/env python3
"""Condenses tools/pmc_c5.sh's counter pass into profiles/<tag>_c5_sq_counters.json: per kernel of the split bounce, means per launch
and the derived figures (VALU lane utilisation, VALU wave-instructions per launch, share of wave time issuing / waiting).

    python tools/collect_sq_c5.py TAG
"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_meta import kernel_label, source_sha16, how_of


def main(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    # (gpurun_out/ keeps earlier calls' files too -- other process ids in the names: only the newest pass counts)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_c5_%s_a" % tag, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            lab = kernel_label(r["Kernel_Name"])
            if lab:
                acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["_launches_sampled"] = min(len(v) for v in cs.values())
        g = m.get
        d = {}
        if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
            # time the chip needs to ISSUE these instructions with every SIMD busy: 1024 SIMDs, one wave64 VALU instruction per 4 cycles, 2.4 GHz
            d["valu_issue_floor_us"] = g("SQ_INSTS_VALU") * 4 / (1024 * 2.4e9) * 1e6
        if g("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if g(c):
                    d["share_of_wave_cycles_" + c] = g(c) / g("SQ_WAVE_CYCLES")
        m["_derived"] = d
        out[k] = m
    out["_how"] = how_of(os.path.join(ROOT, "gpurun_out", "pmc_c5_%s" % tag)) + "; means per launch."
    out["_source_sha16"] = source_sha16()
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_c5_sq_counters.json" % tag), "w"), indent=1, sort_keys=True)
    for k, m in sorted(out.items()):
        if not k.startswith("_"):
            print(k, m["_launches_sampled"], json.dumps({a: round(b, 4) for a, b in m["_derived"].items()}), "VALU insts %.3e" % m.get("SQ_INSTS_VALU", 0))


if __name__ == "__main__":
    main(sys.argv[1])
