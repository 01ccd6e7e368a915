#!/bin/bash
# Kernel timeline of the last dispatches of a command under rocprofv3 --kernel-trace.
# usage: tools/timeline.sh <tag> <n_dispatches> <python script and args...>
TAG=$1; N=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/"$@" > $O/run.log 2>&1; echo "rc=$?"
cd $R
python3 tools/trace_timeline.py $(ls $O/t/*/*kernel_trace.csv | head -1) $N > $O/timeline.txt
rm -rf $O/t
tail -12 $O/run.log
cat $O/timeline.txt
