"""Aggregate rocprofv3 outputs written by tools/profile_headline.sh: per-kernel average duration (kernel trace) and
per-launch counter values (one --pmc pass per counter set; per kernel only the launches with the largest grid).  FETCH_SIZE is reported in KB by the tool and doubled here
(gfx950 counts 128-byte requests as 64 B, MI355X_MICROARCH.md); WRITE_SIZE is in KB."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {"kernels": {}, "counters_per_launch": {}}
for f in glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        out["kernels"][row["Name"].split("(")[0]] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                                    "pct": float(row["Percentage"])}
for d in glob.glob(os.path.join(root, "pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        rows = list(csv.DictReader(open(f)))
        biggest = defaultdict(int)                      # per kernel: only the launches with the largest grid (the N-sized ones)
        for row in rows:
            k = row["Kernel_Name"].split("(")[0]
            biggest[k] = max(biggest[k], int(row["Grid_Size"]))
        for row in rows:
            k = row["Kernel_Name"].split("(")[0]
            if int(row["Grid_Size"]) != biggest[k]:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
        for k, cs in acc.items():
            for c, v in cs.items():
                n = max(1, len(disp[(k, c)]))
                v = v / n
                if c == "FETCH_SIZE":
                    v = v * 1024.0 * 2.0
                elif c == "WRITE_SIZE":
                    v = v * 1024.0
                out["counters_per_launch"].setdefault(k, {})[c + ("_bytes" if c.endswith("_SIZE") else "")] = v
json.dump(out, sys.stdout, indent=1, sort_keys=True)
