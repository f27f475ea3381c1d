set -e
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_train_step_gpu.py -q -m gpu -k "fresh" > $R/gpurun_out/r04_t07.log 2>&1 || true
tail -2 $R/gpurun_out/r04_t07.log
export FRHIP_BENCH_INSTEP=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r50 -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/r04_tl.log 2>&1
cd $R
python tools/trace_timeline.py $(ls /tmp/tl/*kernel_trace.csv | head -1) > gpurun_out/r04_timeline.txt
head -3 gpurun_out/r04_timeline.txt
