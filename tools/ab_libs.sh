# step-level A/B of variant libraries on ONE box, alternating: bash tools/ab_libs.sh <passes> <lib>...   ("base" = the in-tree build;
# other names = face-recognition-pytorch_amd/frhip/build/var/libfrhip_<name>.so from FRHIP_VARIANT=<name> FRHIP_CXXFLAGS=... python -m frhip.build)
# BENCH_ARGS (default: headline) e.g. BENCH_ARGS="--network Swin34 --steps 30"
passes=$1; shift
args=${BENCH_ARGS:-}
for pass in $(seq $passes); do
for lib in "$@"; do
  if [ "$lib" = base ]; then echo "== base"; python bench.py --no-cpu-baseline --no-extra $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  else echo "== $lib"; FRHIP_LIB_PATH=face-recognition-pytorch_amd/frhip/build/var/libfrhip_$lib.so python bench.py --no-cpu-baseline --no-extra $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; fi
done
done
