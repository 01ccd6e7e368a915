#!/bin/bash
# Kernel trace of a short bench run (no client-thread legs, no CPU baseline), summarised per kernel and grid size into
# gpurun_out/<tag>/.  usage: tools/quick_profile.sh <tag> [bench args...]
TAG=${1:-quick}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/all -- python3 $R/bench.py --no-client-threads --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/bench.json 2> $O/bench.err; echo "rc=$?"
cd $R
python3 tools/rocprof_summary.py $TAG $(ls $O/all/*/*kernel_trace.csv) && mv profiles/${TAG}_kernel_summary.md $O/
cp $(ls $O/all/*/*kernel_stats.csv) $O/kernel_stats.csv
rm -rf $O/all
head -40 $O/${TAG}_kernel_summary.md
