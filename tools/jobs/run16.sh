mkdir -p gpurun_out
SH="240,1440,7 288,1728,7 104,624,14 144,864,14 320,1280,7"
for cap in 0 32 64 128; do echo "CAP=$cap"; FROST_PW_STATS_CAP=$cap python tests/devtools/pw_micro.py $SH 2>&1 | grep npix | sed 's/bwd_reduce.*//'; done > gpurun_out/cap.log
