mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_prod.py tests/test_gpu_model.py -m gpu -x -q -k "g3 or prod_layer or g4 or site_by_site" 2>&1 | tail -15 > gpurun_out/t1.log
SH="624,624,14,5,1 360,360,14,3,1 1440,1440,7,5,1 32,32,112,3,1 72,72,56,3,1 168,168,28,3,1"
FROST_DW_MFMA=0 python tests/devtools/pw_micro.py $SH > gpurun_out/micro_old.log 2>&1
FROST_DW_MFMA=1 python tests/devtools/pw_micro.py $SH > gpurun_out/micro_new.log 2>&1
