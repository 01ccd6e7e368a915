#!/bin/bash
# One round's profiling evidence on the GPU box: rocprofv3 kernel trace of the default bench, two PMC passes (FETCH_SIZE,
# WRITE_SIZE) of the timed scan, one of the batched filter kernel, and the plain bench line.  The raw traces are large (the
# multi-client legs launch > 100 000 kernels), so they are summarised here (tools/rocprof_summary.py) and removed; what
# remains under gpurun_out/ is small enough to travel back.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O && mkdir -p $O
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 $R/bench.py --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "step1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err; echo "step2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err; echo "step3 rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k3b_fetch -- python3 $R/tools/filter_batch_probe.py > $O/k3b_probe_under_pmc.txt 2> $O/pmc_k3b.err; echo "step4 rc=$?"
cd $R
python3 tools/rocprof_summary.py r02_all $(ls $O/all/*/*kernel_trace.csv) && cp profiles/r02_all_kernel_summary.md $O/ && cp $(ls $O/all/*/*kernel_stats.csv) $O/all_kernel_stats.csv
ROWS=$(python3 -c "import json; print(json.load(open('$O/bench_under_rocprof.json'))['roofline']['plane_rows'])")
python3 tools/rocprof_summary.py r02_pmc $(ls $O/fetch/*/*kernel_trace.csv) $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) 10000000 $ROWS && cp profiles/r02_pmc_kernel_summary.md $O/
python3 - <<PY
import csv, glob, json
rows = [r for r in csv.DictReader(open(glob.glob("$O/k3b_fetch/*/*counter_collection.csv")[0])) if r["Counter_Name"] == "FETCH_SIZE" and "k_filter_eval_batch" in r["Kernel_Name"]]
values = [float(r["Counter_Value"]) for r in rows]
k3b = {"kernel": "k_filter_eval_batch", "launches": len(values), "fetch_size_kib_raw_avg": sum(values) / max(1, len(values)),
       "hbm_read_bytes_avg": 2 * 1024 * sum(values) / max(1, len(values))}
json.dump(k3b, open("$O/pmc_k3b.json", "w"), indent=1)
# the batched filter kernel joins the scan's entries in the file bench.py reads
doc = json.load(open("profiles/pmc_traffic.json"))
doc["kernels"]["k_filter_eval_batch@64x306"] = {
    "fetch_size_kib_raw": k3b["fetch_size_kib_raw_avg"], "hbm_read_bytes": k3b["hbm_read_bytes_avg"], "hbm_bytes": k3b["hbm_read_bytes_avg"],
    "launches": k3b["launches"], "programs": 64, "sequences": 10000000,
    "note": "tools/filter_batch_probe.py under rocprofv3 --pmc FETCH_SIZE: 64 programs x 32 leaf columns = 2.56 GB algorithmic; count-only launches write nothing"}
doc["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1, and --pmc FETCH_SIZE of tools/filter_batch_probe.py (tools/profile_round.sh)"
json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf $O/all $O/fetch $O/write $O/k3b_fetch
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "step5 rc=$?"
tail -2 $O/bench_under_rocprof.err; ls -la $O; du -sh $O
