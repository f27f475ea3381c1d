# step-level A/B of the element-wise non-temporal policy (bn.hip g_ew_nt; FRHIP_EW_NT bits: 1 loads in residual / backward
# passes, 4 loads in the plain apply too, 2 stores; FRHIP_EW_NT_MB = size threshold), two passes on one box
for pass in 1 2; do
for cfg in "0 0" "1 0" "5 0" "7 0" "5 40"; do set -- $cfg; echo "== FRHIP_EW_NT=$1 MB=$2"; FRHIP_EW_NT=$1 FRHIP_EW_NT_MB=$2 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
done
