mkdir -p gpurun_out/r05b
for s in 0 1 2 3 4; do IW_STAGE=$s timeout 300 python tools/hwtests/inflate_wave.py 0.25 4096 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05b/stages.txt 2>&1
for g in 1024 2048 8192 16384; do IW_STAGE=4 timeout 300 python tools/hwtests/inflate_wave.py 0.25 $g 2>&1 | grep launches; done >> gpurun_out/r05b/stages.txt 2>&1
cat gpurun_out/r05b/stages.txt
