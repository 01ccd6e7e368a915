"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py, for the launches that bench line
lists in roofline.launches_per_scan: entries `kernel@blocks` with `sequences` and `bytes` in profiles/pmc_traffic.json, which
bench.py reads for roofline.traffic.  Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it
is doubled; WRITE_SIZE is taken as is; separate --pmc passes.

usage: python tools/pmc_entries.py <tag> <bench line of the FETCH pass .json> <fetch counter_collection.csv> <write counter_collection.csv>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counter(path, wanted):
    values = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == wanted:
            group = int(row.get("Workgroup_Size", 0) or 0)
            blocks = int(row["Grid_Size"]) // group if group else 0
            values[(short(row["Kernel_Name"]), blocks)].append(float(row["Counter_Value"]))
    return {key: (sum(v) / len(v), len(v)) for key, v in values.items()}


def main():
    tag, bench_path, fetch_path, write_path = sys.argv[1:5]
    line = json.loads(open(bench_path).read().strip().splitlines()[-1])
    sequences = line["config"]["sequences"]
    fetch, write = counter(fetch_path, "FETCH_SIZE"), counter(write_path, "WRITE_SIZE")
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path))
    rows = []
    for launch in line["roofline"]["launches_per_scan"]:
        # the timing log counts blocks over all grid dimensions; the profiler's Grid_Size is threads over all dimensions
        candidates = [(key, value) for key, value in fetch.items() if key[0] == launch["kernel"]]
        match = [kv for kv in candidates if kv[0][1] == launch["blocks"]] or candidates
        if not match:
            continue
        (name, blocks), (kib, launches) = max(match, key=lambda kv: kv[1][1])
        write_kib = write.get((name, blocks), (0.0, 0))[0]
        entry = {"fetch_size_kib_raw": kib, "write_size_kib_raw": write_kib, "hbm_read_bytes": 2.0 * kib * 1024.0,
                 "hbm_write_bytes": write_kib * 1024.0, "hbm_bytes": 2.0 * kib * 1024.0 + write_kib * 1024.0, "launches": launches,
                 "sequences": sequences, "bytes": launch["bytes"], "plane_rows": launch["plane_rows"], "round": tag}
        doc["kernels"][f"{launch['kernel']}@{launch['blocks']}"] = entry
        rows.append((launch["kernel"], launch["blocks"], launch["bytes"], entry["hbm_bytes"], entry["hbm_read_bytes"], entry["hbm_write_bytes"], launches))
    doc["source"] = (doc.get("source", "") + f"; {tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --no-also "
                     "--no-cpu-baseline --steps 3 --warmup 1 (tools/profile_round3.sh, tools/pmc_entries.py)")
    json.dump(doc, open(path, "w"), indent=1)
    out = [f"# {tag}: HBM traffic per launch of the headline scan (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes)", "",
           "bytes = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024 (gfx950 correction of the microarchitecture guide); `has to read` = the bytes the",
           "launch's timing record states (plane rows x row bytes + filter; 4 B per key + a 16 KiB filter slice per block; 12 B per run; 8 B per sparse key).", "",
           "| kernel | blocks | has to read | HBM traffic | traffic / bytes | read | written | launches |", "|---|---|---|---|---|---|---|---|"]
    for name, blocks, need, traffic, read, written, launches in rows:
        out.append(f"| {name} | {blocks} | {need / 1e6:.1f} MB | {traffic / 1e6:.1f} MB | {traffic / max(need, 1):.3f} | {read / 1e6:.1f} MB | {written / 1e6:.1f} MB | {launches} |")
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
