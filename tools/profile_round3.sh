#!/bin/bash
# Round 3's profiling evidence on the GPU box, in two calls (each within gpurun's limit):
#   tools/profile_round3.sh trace   rocprofv3 --kernel-trace --stats of the default bench (every leg), summarised and removed
#   tools/profile_round3.sh pmc     two PMC passes (FETCH_SIZE, WRITE_SIZE) of the timed scan, then the plain bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
if [ "$1" = "trace" ]; then
   rm -rf $O/all
   timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 $R/bench.py --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "trace rc=$?"
   cd $R
   python3 tools/rocprof_summary.py r03_all $(ls $O/all/*/*kernel_trace.csv | head -1) && cp profiles/r03_all_kernel_summary.md $O/ && cp $(ls $O/all/*/*kernel_stats.csv | head -1) $O/all_kernel_stats.csv
   rm -rf $O/all
   tail -2 $O/bench_under_rocprof.err; ls -la $O
else
   rm -rf $O/fetch $O/write
   timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err; echo "fetch rc=$?"
   timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_pmc_write.json 2> $O/pmc_write.err; echo "write rc=$?"
   cd $R
   python3 tools/pmc_entries.py r03 $O/bench_pmc_fetch.json $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) && cp profiles/pmc_traffic.json profiles/r03_pmc_traffic.md $O/
   python3 tools/rocprof_summary.py r03_pmc $(ls $O/fetch/*/*kernel_trace.csv | head -1) && cp profiles/r03_pmc_kernel_summary.md $O/
   rm -rf $O/fetch $O/write
   timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
   tail -2 $O/bench.err; ls -la $O
fi
