tag=${1:-r05m}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
one() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o iw -- python /root/repo/tools/hwtests/inflate_product.py 0.25 2 > /tmp/log_$name.txt 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep -E 'k_inflate_symbols|k_inflate_copy' $f | awk -F, '{gsub(/"/,""); printf "%s calls %s avg %.3f ms total %.3f ms; ", $1, $2, $4/1e6, $3/1e6}')"
}
{
for w in 4 8 12 15; do one full_w$w CHECK_CRC=0 GCI_INFLATE_WAVES=$w GCI_INFLATE_BATCH=16384; done
for w in 4 8 15; do one a6_w$w GCI_IW_CUT_A=6 CHECK_CRC=0 GCI_INFLATE_WAVES=$w GCI_INFLATE_BATCH=16384; done
one a1_w15 GCI_IW_CUT_A=1 CHECK_CRC=0 GCI_INFLATE_BATCH=16384
one a3_w15 GCI_IW_CUT_A=3 CHECK_CRC=0 GCI_INFLATE_BATCH=16384
} 2>&1 | tee /root/repo/gpurun_out/$tag/occ.txt
