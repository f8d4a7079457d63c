mkdir -p gpurun_out
for n in layer4.3.squeeze_conv layer3.1.reduce_conv layer4.1.squeeze_conv; do python tests/devtools/dbg_solid_flips.py $n 2>&1 | grep -v Warn; done > gpurun_out/dbg_solid.log
