#!/bin/bash
# End-of-round evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command, serial step anatomy,
# PMC traffic passes.  Everything lands under gpurun_out/final/ ; copy the summaries into profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.log
tail -1 $OUT/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r50 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/serial -o r50 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/serial.log 2>&1
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/swin -o swin -- python3 $R/bench.py --network Swin34 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/swin.log 2>&1
cd $R
python tools/trace_summary.py $(ls $OUT/serial/*kernel_trace.csv | head -1) > $OUT/step_anatomy.txt
python tools/trace_summary.py $(ls $OUT/swin/*kernel_trace.csv | head -1) > $OUT/swin_step_anatomy.txt
rm -f $OUT/serial/*kernel_trace.csv $OUT/swin/*kernel_trace.csv $OUT/stats/*kernel_trace.csv
bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1
ls $OUT $OUT/stats | head -30
