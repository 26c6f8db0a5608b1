#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST burst of launches in a rocprofv3 --kernel-trace CSV (bursts are separated by >= 20 ms)."""
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0) for r in csv.DictReader(open(f))), key=lambda x: x[0])
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 20_000_000: bursts.append(cur); cur = [r]
    else: cur.append(r)
bursts.append(cur)
b = bursts[-1]
t0 = b[0][0]
busy = 0; last_end = t0
for st, en, name, q, g in b:
    short = name.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:24]
    print("%8.1f %8.1f us  dur %7.1f  q%s grid %7d  %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, q, g, short))
    if en > last_end: busy += en - max(st, last_end); last_end = en
print("burst: %d kernels, span %.1f us, covered by >=1 kernel %.1f us" % (len(b), (b[-1][1] - t0) / 1e3, busy / 1e3))
