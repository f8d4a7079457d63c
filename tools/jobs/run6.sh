mkdir -p gpurun_out
python tests/devtools/dbg_resume.py 64 4 > gpurun_out/dbg_resume.log 2>&1
python tests/devtools/dbg_resume.py 128 8 >> gpurun_out/dbg_resume.log 2>&1
