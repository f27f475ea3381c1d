#!/bin/bash
# End-of-round evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command, serial step anatomy (weight
# gradients on the main stream so durations add up), the overlapped anatomy (what the step really runs), Swin34 / AlterNet50
# anatomies, PMC traffic passes and the MFMA-utilisation counters of the dominant kernels.
# Everything lands under gpurun_out/final/ ; tools/collect_profiles.py <tag> copies the summaries into profiles/.
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
# QUICK=1: bench line, kernel stats and the step anatomies only -- for a refresh after changes that did not touch the conv kernels (the
# committed PMC traffic report still matches their source digest; counters, clock stamps and the ablation are theirs).
# QUICK=2: the same plus the PMC traffic passes (a change in igemm_nt* / igemm_halo* / common.h moves the digest bench.py checks)
QUICK=${QUICK:-0}
# PMC traffic passes first: bench.py reports roofline.traffic only from a pass whose kernel-source digest matches this tree
if [ "$QUICK" != 1 ]; then    # QUICK=0, 2, 3
rm -rf $R/gpurun_out/pmc
bash tools/pmc_traffic.sh > $OUT/pmc.log 2>&1
python tools/pmc_traffic_report.py gpurun_out/pmc ${TAG:-r03} > $OUT/pmc_report.log 2>&1
fi
python bench.py > $OUT/bench.json 2> $OUT/bench.log
tail -1 $OUT/bench.json | cut -c1-200
export FRHIP_BENCH_INSTEP=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r50 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
cp $(ls $OUT/stats/*kernel_trace.csv | head -1) $OUT/overlap_trace.csv
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/serial -o r50 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/serial.log 2>&1
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/swin -o swin -- python3 $R/bench.py --network Swin34 --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/swin.log 2>&1
FRHIP_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/alt -o alt -- python3 $R/bench.py --network AlterNet50 --fp8 --steps 6 --warmup 3 --no-cpu-baseline --no-extra > $OUT/alt.log 2>&1
cd $R
python tools/trace_summary.py $(ls $OUT/serial/*kernel_trace.csv | head -1) > $OUT/step_anatomy.txt
python tools/trace_summary.py $OUT/overlap_trace.csv > $OUT/step_anatomy_overlapped.txt
python tools/trace_summary.py $(ls $OUT/swin/*kernel_trace.csv | head -1) > $OUT/swin_step_anatomy.txt
python tools/trace_summary.py $(ls $OUT/alt/*kernel_trace.csv | head -1) > $OUT/alt_step_anatomy.txt
rm -f $OUT/serial/*kernel_trace.csv $OUT/swin/*kernel_trace.csv $OUT/alt/*kernel_trace.csv $OUT/stats/*kernel_trace.csv $OUT/overlap_trace.csv
python tools/bench_eval.py ResNet50 512 > $OUT/eval.txt 2>&1 || true
if [ "$QUICK" = 1 ] || [ "$QUICK" = 2 ]; then ls $OUT | head -40; exit 0; fi
# QUICK=3: QUICK=2 plus the MFMA-utilisation counters below, without the diagnostic builds (in-kernel clock, ablation)
# MFMA utilisation / LDS counters of the dominant kernels (256-channel 14x14 layer, B = 512), one rocprofv3 pass per counter group
for what in fwd dgrad wgrad; do
  bash tools/pmc_run.sh $OUT/pmc_$what $what 14 256 256 > $OUT/pmc_$what.log 2>&1 || true
  python tools/pmc_show.py $OUT/pmc_$what > $OUT/pmc_$what.txt 2>/dev/null || true
done
FRHIP_T9_NARROW=0 FRHIP_T9_LDS_PAD=0 bash tools/pmc_run.sh $OUT/pmc_wgrad8 wgrad 14 256 256 > $OUT/pmc_wgrad8.log 2>&1 || true
python tools/pmc_show.py $OUT/pmc_wgrad8 > $OUT/pmc_wgrad8.txt 2>/dev/null || true
rm -rf $OUT/pmc_*/g*      # raw counter CSVs: summarised in pmc_*.txt
if [ "$QUICK" = 3 ]; then ls $OUT | head -40; exit 0; fi
# in-kernel clock of the halo kernels (diagnostic build: `python tools/clock_probe.py build` before the snapshot) and the timing-only
# ablation of the 4-wave halo kernel (`ABL_ONLY=... python tools/ablate.py build`)
CLOCK_SECS=1.5 python tools/clock_probe.py run > $OUT/clock.txt 2>&1 || true
ABL_ONLY=full,nobar,nomfma,nodma,noepi,nolds,mfma_only,lds_only python tools/ablate.py run > $OUT/ablate.txt 2>&1 || true
ls $OUT | head -40
