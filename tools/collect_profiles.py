#!/usr/bin/env python3
"""Turns the rocprofv3 outputs that a profiling gpurun call left under gpurun_out/ into the committed summaries under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_{fetch,write}_size.csv and
traffic_<tag>.json (HBM bytes per launch per kernel = mean FETCH_SIZE * 1024 * 2 + mean WRITE_SIZE * 1024; FETCH_SIZE is
in KiB and reads half of the bytes on gfx950, MI355X_MICROARCH.md "HBM").

    python tools/collect_profiles.py TAG gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_meta import kernel_label, source_sha16, how_of


def main(tag, d_stats, d_fetch, d_write):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun_out/ keeps earlier calls' files too
    shutil.copy(newest(os.path.join(d_stats, "*", "*_kernel_stats.csv")), os.path.join(out, "%s_kernel_stats.csv" % tag))
    d_stats2 = d_stats.rstrip("/") + "2"           # the default command (three launch sets in flight), when profiled as well
    if glob.glob(os.path.join(d_stats2, "*", "*_kernel_stats.csv")):
        shutil.copy(newest(os.path.join(d_stats2, "*", "*_kernel_stats.csv")), os.path.join(out, "%s_kernel_stats_default_3lanes.csv" % tag))
    acc = {}
    for kind, d in (("fetch", d_fetch), ("write", d_write)):
        f = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        shutil.copy(f, os.path.join(out, "%s_pmc_%s_size.csv" % (tag, kind)))
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            lab = kernel_label(r["Kernel_Name"])
            if lab:
                a[lab].append(float(r["Counter_Value"]))
        acc[kind] = a
    res, detail = {}, {}
    for k in sorted(set(acc["fetch"]) | set(acc["write"])):
        f, w = acc["fetch"].get(k), acc["write"].get(k)
        if not f or not w:
            continue
        fb, wb = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
        res[k] = fb + wb
        detail[k] = dict(launches_sampled=len(f), fetch_size_kib_mean=sum(f) / len(f), write_size_kib_mean=sum(w) / len(w),
                         hbm_read_bytes_per_launch=fb, hbm_write_bytes_per_launch=wb)
    res["_detail"] = detail
    # rays per launch of the dominant kernel in the profiled command (bench.py scales the bytes to its own launches by it)
    fb = os.path.join(os.path.dirname(d_fetch.rstrip("/")), "prof_fetch_bench.json")
    if os.path.exists(fb) and "c5" not in tag:         # (that file belongs to the C4 bench passes)
        try:
            line = json.loads(open(fb).read().strip().splitlines()[-1])
            res["_units_per_launch"] = {line["roofline"]["kernel"]: line["roofline"]["units_per_launch"]}
        except Exception:
            pass
    # what was profiled, as the profiling script itself recorded it next to its outputs (profile_round.sh / profile_c5.sh write
    # <dir>_how.txt: the command after `--`, what a launch covers); never a fixed text -- round 2 and round 3 both shipped a C5
    # traffic file that described the C4 command
    res["_how"] = how_of(d_fetch) + ("; bytes per launch = mean(FETCH_SIZE)*1024*2 + mean(WRITE_SIZE)*1024; the factor 2 is the gfx950 "
                   "FETCH_SIZE correction (checked in round 1 on the then k_move: ~77 MB of dword-per-lane reads expected, counter 40.5 MB; WRITE_SIZE "
                   "checked on torch's 24.9 MB fill = 24300 KiB).")
    res["_iterations_per_launch"] = 12       # the default batch at 1080p, which the profiled command runs with (bench.py: physical_frac_wall)
    res["_source_sha16"] = source_sha16()      # of the kernel sources this was collected with (bench.py warns when they have changed since)
    json.dump(res, open(os.path.join(out, "traffic_%s.json" % tag), "w"), indent=1)
    if "c5" not in tag:                                # what bench.py reads
        json.dump(res, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
