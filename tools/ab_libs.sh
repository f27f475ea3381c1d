# step-level A/B of variant libraries (tools/variants.py): bash tools/ab_libs.sh <passes> <lib>...   ("base" = the in-tree build)
passes=$1; shift
for pass in $(seq $passes); do
for lib in "$@"; do
  if [ "$lib" = base ]; then echo "== base"; python bench.py --no-cpu-baseline --no-extra 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  else echo "== $lib"; FRHIP_LIB_PATH=face-recognition-pytorch_amd/frhip/build/var/libfrhip_$lib.so python bench.py --no-cpu-baseline --no-extra 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; fi
done
done
