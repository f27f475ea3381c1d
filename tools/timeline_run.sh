set -eu
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl2
rm -rf $OUT; mkdir -p $OUT
export FRHIP_BENCH_INSTEP=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o r50 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/run.log 2>&1
cd $R
python tools/trace_timeline.py $(ls $OUT/tr/*kernel_trace.csv | head -1) > $OUT/timeline.txt
rm -rf $OUT/tr
tail -1 $OUT/run.log | cut -c1-300
