python tools/rows14_check.py 2>&1 | grep "n=512"
FRHIP_LIB_PATH=face-recognition-pytorch_amd/frhip/build/var/libfrhip_pin0.so python tools/rows14_check.py 2>&1 | grep "n=512"
python tools/rows14_check.py 2>&1 | grep -v "n=512" | tail -4
bash tools/ab_libs.sh 2 pin0 base
