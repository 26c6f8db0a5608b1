#!/usr/bin/env python3
"""How well the launch sets overlap: from a rocprofv3 --kernel-trace CSV (last burst of launches), the share of the span during which
0 / 1 / 2 / 3+ kernels were running, per-kernel-kind total duration, and the longest gaps with nothing running.
    python tools/timeline_concurrency.py DIR"""
import collections, csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))), key=lambda x: x[0])
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-8:]) > 20_000_000: bursts.append(cur); cur = [r]
    else: cur.append(r)
bursts.append(cur)
b = max(bursts, key=len)
ev = []
for st, en, name, q in b:
    ev.append((st, 1)); ev.append((en, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
share = collections.Counter(); n = 0; last = t0; gaps = []
for t, d in ev:
    if t > last:
        share[min(n, 3)] += t - last
        if n == 0: gaps.append(t - last)
    n += d; last = t
span = t1 - t0
print("burst: %d kernels, span %.2f ms" % (len(b), span / 1e6))
for k in range(4): print("  %s kernels running: %5.1f %%" % (("3+" if k == 3 else str(k)), 100.0 * share[k] / span))
print("  idle gaps: %d, longest %.1f us, total %.1f us" % (len(gaps), max(gaps or [0]) / 1e3, sum(gaps) / 1e3))
kind = collections.Counter(); cnt = collections.Counter()
for st, en, name, q in b:
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    kind[short] += en - st; cnt[short] += 1
for k, v in kind.most_common(12): print("  %-28s %4d launches, sum of durations %8.2f ms = %4.1f %% of the span" % (k, cnt[k], v / 1e6, 100.0 * v / span))
