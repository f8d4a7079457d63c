mkdir -p gpurun_out
SH="624,624,14,5,1 72,72,56,3,1"
for a in 0 1 3 7 8 15 31; do echo "ABL=$a"; FROST_DWM_ABL=$a FROST_DW_MFMA=1 python tests/devtools/pw_micro.py $SH 2>&1 | grep npix | sed 's/bwd_reduce.*//'; done > gpurun_out/abl.log
