"""Timeline of the LAST evaluation in a rocprofv3 kernel trace: every kernel with start / end relative to the step's first
kernel, one line each, streams told apart by queue id.  python tools/step_timeline.py <kernel_trace.csv> [min_dur_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mind = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
# the last step starts at the last featurize_tile_kernel / featurize kernel that is preceded by a gap > 50 us
starts = [i for i, r in enumerate(rows) if i > 0 and int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in rows[max(0, i - 8):i]) > 50_000]
i0 = starts[-1] if starts else 0
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) / 1e3 < mind:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oak::", "")[:40]
    print(f"q{r.get('Queue_Id', '?'):>3s} {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {name}  grid={r.get('Grid_Size_X', '')}")
