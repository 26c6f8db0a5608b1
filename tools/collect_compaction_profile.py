#!/usr/bin/env python3
"""profiles/<tag>_compaction_{kernel_stats,pmc_fetch_size,pmc_write_size}.csv and profiles/traffic_<tag>_compaction.json
from what tools/profile_compaction.sh left under gpurun_out/ (HBM bytes per launch = mean FETCH_SIZE * 1024 * 2 + mean
WRITE_SIZE * 1024, the gfx950 correction of MI355X_MICROARCH.md, next to the algorithmic bytes of the same launch).

    python tools/collect_compaction_profile.py TAG [N]
"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
G = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles")
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(newest(os.path.join(G, "sc_stats", "*", "*_kernel_stats.csv")), os.path.join(out, "%s_compaction_kernel_stats.csv" % tag))
acc = {}
for kind in ("fetch", "write"):
    f = newest(os.path.join(G, "sc_" + kind, "*", "*_counter_collection.csv"))
    shutil.copy(f, os.path.join(out, "%s_compaction_pmc_%s_size.csv" % (tag, kind)))
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_onepass" in name:
            a["compact" if "k_onepass<true" in name else "scan"].append(float(r["Counter_Value"]))
    acc[kind] = a
bw = [json.loads(l) for l in open(os.path.join(G, "sc_bw.json")) if l.startswith("{")]
kept = next(r["kept"] for r in bw if r["n"] == n)
res = {}
for k, algo in (("scan", 8 * n), ("compact", 4 * n + 4 * kept)):
    f, w = acc["fetch"][k], acc["write"][k]
    fb, wb = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
    res[k] = dict(launches_sampled=len(f), hbm_read_bytes_per_launch=fb, hbm_write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb,
                  algorithmic_bytes_per_launch=algo, ratio=(fb + wb) / algo)
res["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python3 tools/gpu_compaction_bw.py %d; "
               "n = %d ints, %d survive" % (n, n, kept))
res["_rates_under_kernel_trace"] = bw
json.dump(res, open(os.path.join(out, "traffic_%s_compaction.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
