#!/usr/bin/env python3
"""show_timeline.py DIR [N] -- the last N kernel dispatches of a rocprofv3 --kernel-trace run: start and end in microseconds
relative to the first of them, duration, grid, name."""
import csv
import glob
import os
import re
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ptmi::(anonymous namespace)::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f  %8.1f us  grid %-8s queue %-3s %s" % (s, e, e - s, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Queue_Id", "?"), name[:60]))
