"""Summarises rocprofv3 CSV output (kernel trace / PMC counter collection) into profiles/.

usage: python tools/rocprof_summary.py <round-tag> <kernel_trace.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv> [sequences rows]]
(sequences / rows = the workload of the profiled k_scan_sliced launches, recorded so that bench.py can match them)
Writes profiles/<tag>_kernel_summary.md and, with counters, profiles/pmc_traffic.json — the file bench.py
reads for roofline.traffic.  Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is; separate --pmc passes.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    groups = collections.defaultdict(list)
    for row in csv.DictReader(open(trace)):
        key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row.get("Grid_Size", 0)))
        groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    lines = [f"# {tag}: per-kernel summary of `rocprofv3 --kernel-trace` (durations in µs, grouped by kernel and grid size)", "",
             "| kernel | grid (threads) | calls | avg | median | min | max |", "|---|---|---|---|---|---|---|"]
    for (name, grid), durations in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| {name} | {grid} | {len(durations)} | {sum(durations) / len(durations) / 1e3:.1f} | "
                     f"{sorted(durations)[len(durations) // 2] / 1e3:.1f} | {min(durations) / 1e3:.1f} | {max(durations) / 1e3:.1f} |")
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_summary.md")
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)

    if len(sys.argv) >= 5:
        def counter(path, wanted):
            values = collections.defaultdict(list)
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] == wanted:
                    values[(short(row["Kernel_Name"]), int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
            return {key: sum(v) / len(v) for key, v in values.items()}

        fetch = counter(sys.argv[3], "FETCH_SIZE")
        write = counter(sys.argv[4], "WRITE_SIZE")
        traffic = {}
        for (name, grid), kib in fetch.items():
            if name.startswith("k_"):
                read_bytes = 2.0 * kib * 1024.0          # gfx950: FETCH_SIZE counts half of a wide streaming read
                write_bytes = write.get((name, grid), 0.0) * 1024.0
                traffic[f"{name}@{grid}"] = {
                    "fetch_size_kib_raw": kib, "write_size_kib_raw": write.get((name, grid), 0.0),
                    "hbm_read_bytes": read_bytes, "hbm_write_bytes": write_bytes, "hbm_bytes": read_bytes + write_bytes,
                }
        if len(sys.argv) >= 7:
            for key, entry in traffic.items():
                if key.startswith("k_scan_sliced"):
                    entry["sequences"], entry["rows"] = int(sys.argv[5]), int(sys.argv[6])
        doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {tag}",
               "correction": "bytes = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024 (MI355X_MICROARCH.md §HBM)",
               "kernels": traffic}
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        json.dump(doc, open(path, "w"), indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
