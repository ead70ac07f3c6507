"""List the kernels of the last evaluation's tail (from the last syrk_reduce on) in a rocprofv3 kernel trace:
python tools/tail_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "syrk_reduce" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev = None
agg = {}
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oak::", "")[:44]
    if "-v" in sys.argv:
        print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {name}  grid={r.get('Grid_Size_X', '')}")
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev = e
print(f"tail span {(prev - t0) / 1e3:.1f} us")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:46s} n={c:4d}  {d:8.1f} us")
