"""Prints the last dispatches of a rocprofv3 kernel trace as a timeline (µs relative to the first one shown): which
kernels overlap, on which queue.  usage: trace_timeline.py <kernel_trace.csv> [n_dispatches]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - t0) / 1e3:10.1f}  q{r.get('Queue_Id', '?'):>3} "
          f"grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}x{r.get('Grid_Size_Y', '1'):<2} {name}")
