#!/usr/bin/env python3
"""Condenses the counter passes of tools/pmc_sq.sh into profiles/<tag>_sq_counters.json: per kernel, the mean of every counter
per launch plus a few derived figures (VALU lane utilisation, share of wave time issuing / waiting, instruction-cache hit rate).

    python tools/collect_sq.py TAG
"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_meta import kernel_label, source_sha16, how_of


def main(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_?" % tag))):
        # (gpurun_out/ keeps earlier calls' files too -- other process ids in the names: only the newest pass counts)
        for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                lab = kernel_label(r["Kernel_Name"])
                if lab:
                    acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["_launches_sampled"] = min(len(v) for v in cs.values())
        d = {}
        g = m.get
        if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
            d["quad_cycles_per_valu_instruction"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
        if g("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if g(c):
                    d["share_of_wave_cycles_" + c] = g(c) / g("SQ_WAVE_CYCLES")
            if g("SQ_BUSY_CYCLES"):
                d["mean_waves_per_simd"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES") / 4.0 if False else None
        if g("SQC_ICACHE_REQ"):
            d["icache_hit_rate"] = g("SQC_ICACHE_HITS", 0.0) / g("SQC_ICACHE_REQ")
            d["icache_misses_per_launch"] = g("SQC_ICACHE_MISSES", 0.0)
        m["_derived"] = {a: b for a, b in d.items() if b is not None}
        out[k] = m
    # rays per launch of the dominant kernel in the profiled command (bench.py scales the instruction counts to its own launches by it)
    try:
        line = json.loads(open(os.path.join(ROOT, "gpurun_out", "pmc_%s_a.json" % tag)).read().strip().splitlines()[-1])
        out["_units_per_launch"] = {line["roofline"]["kernel"]: line["roofline"]["units_per_launch"]}
    except Exception:
        pass
    out["_how"] = how_of(os.path.join(ROOT, "gpurun_out", "pmc_%s" % tag)) + ("; means per launch. SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_INST_* count quad-cycles summed over waves.")
    out["_source_sha16"] = source_sha16()      # of the kernel sources this was collected with (bench.py warns when they have changed since)
    p = os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag)
    json.dump(out, open(p, "w"), indent=1, sort_keys=True)
    # what bench.py reads: sq_latest.json (the exact level), or -- `python tools/collect_sq.py TAG --as sq_arith1.json` -- the file of
    # another arithmetic level's passes (PMC_EXTRA_ARGS="--arith 1" bash tools/pmc_sq.sh TAG)
    latest = sys.argv[sys.argv.index("--as") + 1] if "--as" in sys.argv else "sq_latest.json"
    json.dump(out, open(os.path.join(ROOT, "profiles", latest), "w"), indent=1, sort_keys=True)
    for k in ("k_bounce", "k_bounce<first>", "k_move"):
        if k in out:
            print(k, json.dumps(out[k]["_derived"]), {c: round(v) for c, v in out[k].items() if not c.startswith("_")})


if __name__ == "__main__":
    main(sys.argv[1])
