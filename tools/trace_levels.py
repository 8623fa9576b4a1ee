#!/usr/bin/env python3
"""trace_levels.py DIR -- per-launch durations of the last render call in a rocprofv3 kernel trace of the stream form
(rocprofv3 --kernel-trace -d DIR --output-format csv -- python3 bench.py ... --streams-form stream)."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "streams_primary" in r["Kernel_Name"]]
s = starts[-1]
prev_end = None
total = 0.0
line = []
for r in rows[s:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = "L" + n[n.find("<"):n.find(">") + 1].replace("true", "1").replace("false", "0").replace(", ", "") if "level" in n else \
        ("primary" if "primary" in n else ("seed" if "update_seed" in n else ("fill" if "fill" in n else ("copy" if "copy" in n else n[:20]))))
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    line.append("%s %.0f%s" % (short, (en - st) / 1e3, (" (+%.0f)" % gap) if gap > 3 else ""))
    if short == "seed":
        print(" | ".join(line)); line = []
    prev_end = en
    total += (en - st) / 1e3
print(" | ".join(line))
print("kernels %.0f us, span %.0f us" % (total, (int(rows[-1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3))
