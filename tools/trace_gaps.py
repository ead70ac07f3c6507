"""Kernel-trace analysis: for the LAST occurrence of a phase delimited by two kernel-name substrings, list the kernels,
their durations and the idle gaps between them.  usage: trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
busy = 0
prev_end = None
agg = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oak::", "")
    gap = (s - prev_end) if prev_end is not None else 0
    a = agg.setdefault(name, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    busy += e - s
    prev_end = max(e, prev_end or e)
span = prev_end - t0
print(f"span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {100*(1-busy/span):.1f} %")
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"{k[:50]:50s} n={c:4d} dur={d/1e3:8.1f} us  gap_before={g/1e3:8.1f} us")
