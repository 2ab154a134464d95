tag=${1:-r05e}
mkdir -p gpurun_out/$tag
{ timeout 600 python tools/hwtests/inflate_product.py 0.25 3; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/inflate_product.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag/prof -o iw -- python /root/repo/tools/hwtests/inflate_product.py 0.25 3 > /dev/null 2>&1
f=$(find /root/repo/gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
cp "$f" /root/repo/gpurun_out/$tag/kernel_stats.csv; rm -rf /root/repo/gpurun_out/$tag/prof
