#!/bin/bash
# Everything the round's numbers come from, in one call on the GPU box: the default bench line, its rocprofv3 kernel stats,
# the PMC passes (tools/prof_pmc.sh) and the inflate kernels' stats.  Usage: bash tools/round_profile.sh <tag>
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${tag}_stats -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 3 --only-step > $OLDPWD/gpurun_out/${tag}_bench_rocprof.json 2> $OLDPWD/gpurun_out/${tag}_bench_rocprof.err); echo "stats rc=$?"
timeout 1800 bash tools/prof_pmc.sh $tag
python tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/${tag}_pmc_summary.txt 2>&1
LANES=8 timeout 600 bash tools/exp_inflate_lanes.sh > gpurun_out/${tag}_inflate.txt 2>&1
timeout 900 bash tools/prof_pmc_inflate.sh > gpurun_out/${tag}_inflate_pmc.txt 2>&1
tail -c 600 gpurun_out/${tag}_bench.json
