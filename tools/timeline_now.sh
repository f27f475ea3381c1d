#!/bin/bash
# launch-by-launch timeline of one overlapped training step (GPU box): bash tools/timeline_now.sh [bench args]   -> gpurun_out/timeline.txt
set -e
R=$GRAFT_REPO_ROOT
export FRHIP_BENCH_INSTEP=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r50 -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/timeline.log 2>&1
cd $R
python tools/trace_timeline.py $(ls /tmp/tl/*kernel_trace.csv | head -1) > gpurun_out/timeline.txt
python tools/trace_summary.py $(ls /tmp/tl/*kernel_trace.csv | head -1) > gpurun_out/timeline_summary.txt
head -3 gpurun_out/timeline.txt
