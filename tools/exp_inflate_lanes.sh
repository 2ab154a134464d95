# k_bgzf_inflate per members-per-wave setting: kernel time from rocprofv3 (tools/exp_inflate_gpu.py at chr19 scale)
cd /tmp && export TMPDIR=/tmp
for L in ${LANES:-4 8 16 32}; do
  GCI_INFLATE_LANES=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/inflanes/$L -o inf -- python /root/repo/tools/exp_inflate_gpu.py ${SCALE:-1.0} 1 2>&1 | grep -E "crc True|rror"
  python - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/inflanes/$L/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:2]: print("lanes $L", r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e6, "ms")
PY
done
