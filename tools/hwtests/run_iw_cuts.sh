tag=${1:-r05f}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
one() {  # name, env...
  name=$1; shift
  env "$@" GCI_INFLATE_FALLBACK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o iw -- python /root/repo/tools/hwtests/inflate_product.py 0.25 2 > /tmp/log_$name.txt 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep -E 'k_inflate_symbols|k_inflate_copy' $f | awk -F, '{gsub(/"/,""); printf "%s calls %s avg %.3f ms total %.3f ms; ", $1, $2, $4/1e6, $3/1e6}')"
}
{
one full CHECK_CRC=0
one a1 GCI_IW_CUT_A=1 CHECK_CRC=0
one a4 GCI_IW_CUT_A=4 CHECK_CRC=0
one a6 GCI_IW_CUT_A=6 CHECK_CRC=0
one a2 GCI_IW_CUT_A=2 CHECK_CRC=0
one a3 GCI_IW_CUT_A=3 CHECK_CRC=0
one b1 GCI_IW_CUT_B=1 CHECK_CRC=0
one b2 GCI_IW_CUT_B=2 CHECK_CRC=0
} 2>&1 | tee /root/repo/gpurun_out/$tag/cuts.txt
