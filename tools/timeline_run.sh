# one overlapped and one serial kernel trace of bench.py; timeline + per-layer averages + anatomy under gpurun_out/tl2/
set -eu
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl2
rm -rf $OUT; mkdir -p $OUT
export FRHIP_BENCH_INSTEP=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o r50 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/run.log 2>&1
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/ser -o r50 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/run_serial.log 2>&1
cd $R
python tools/trace_timeline.py $(ls $OUT/tr/*kernel_trace.csv | head -1) > $OUT/timeline.txt
python tools/trace_summary.py $(ls $OUT/tr/*kernel_trace.csv | head -1) > $OUT/anatomy_overlapped.txt
python tools/trace_summary.py $(ls $OUT/ser/*kernel_trace.csv | head -1) > $OUT/anatomy_serial.txt
python tools/halo_layers.py $(ls $OUT/ser/*kernel_trace.csv | head -1) > $OUT/halo_layers.txt
rm -rf $OUT/tr $OUT/ser
cat $OUT/halo_layers.txt
