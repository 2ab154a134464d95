# k_inflate_wave.hip built with other piece / grain sizes (libgci_hip.so relinked with that object), timed through rocprofv3
tag=${1:-r05q}
mkdir -p /root/repo/gpurun_out/$tag
cd /root/repo/gci_amd/csrc
cp libgci_hip.so /tmp/libgci_hip.so.keep
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v k_inflate_wave)
for v in ${IW_VARIANTS:-9,6,1024 9,6,512}; do
  set -- $(echo $v | tr "," " ")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -DIW_PIECE_LOG2=$1 -DIW_GRAIN_LOG2=$2 -DIW_CP_THREADS=${3:-1024} -c k_inflate_wave.hip -o /tmp/kiw_$1_$2_$3.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgci_hip.so $objs /tmp/kiw_$1_$2_$3.o -lz -lpthread
  cd /tmp; export TMPDIR=/tmp
  CHECK_CRC=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2_$3 -o iw -- python /root/repo/tools/hwtests/inflate_product.py 0.5 3 > /tmp/log_$1_$2_$3.txt 2>&1
  f=$(find /tmp/prof_$1_$2_$3 -name "*kernel_stats.csv" | head -1)
  echo "== piece 2^$1 grain 2^$2 copy threads ${3:-1024}: $(grep -E 'k_inflate_symbols|k_inflate_copy' $f | awk -F, '{gsub(/"/,""); printf "%s avg %.3f ms; ", $1, $4/1e6}') $(grep -o 'equal to the stream: [A-Za-z]*; .*' /tmp/log_$1_$2_$3.txt | tail -1)"
  cd /root/repo/gci_amd/csrc
done 2>&1 | tee /root/repo/gpurun_out/$tag/variants.txt
cp /tmp/libgci_hip.so.keep libgci_hip.so
