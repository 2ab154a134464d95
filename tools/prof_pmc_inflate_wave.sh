# PMC passes over the wave inflate (k_inflate_symbols, k_inflate_copy) as the library runs it: tools/hwtests/inflate_product.py at a
# quarter of chr19, one counter group per pass (--kernel-trace + --pmc only); prints the means per dispatch and kernel.
tag=${1:-r05y}
mkdir -p /root/repo/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; CHECK_CRC=0 GCI_INFLATE_STREAMS=1 GCI_INFLATE_BATCH=16384 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/iwpmc/$name -o pmc -- python /root/repo/tools/hwtests/inflate_product.py 0.25 1 > /tmp/iwpmc_$name.log 2>&1; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run c SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
run e FETCH_SIZE
run f WRITE_SIZE
cd /root/repo
python - <<'PY' | tee /root/repo/gpurun_out/$tag/inflate_wave_pmc.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in "abcdef":
    for f in glob.glob("/tmp/iwpmc/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:20]
            if "k_inflate" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 --pmc over tools/hwtests/inflate_product.py 0.25 (14 065 members, 385 MB -> 918 MB), one batch, means per dispatch")
for k,v in sorted(acc.items()):
    print(k)
    for c,vals in sorted(v.items()):
        print("    %-24s %18.0f   (n=%d)" % (c, sum(vals)/len(vals), len(vals)))
PY
