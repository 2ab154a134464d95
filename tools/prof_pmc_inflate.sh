cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /root/repo/gpurun_out/infpmc/$name -o pmc -- python /root/repo/tools/exp_inflate_gpu.py 0.25 1 > /root/repo/gpurun_out/infpmc_$name.log 2>&1; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run d SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU
cd /root/repo
python - <<'PY'
import csv,glob,collections
for d in "abcd":
    f=glob.glob("gpurun_out/infpmc/%s/*counter_collection.csv"%d)
    if not f: print(d,"none"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:14]
        if not k.startswith("k_bgzf"): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items(): print(d,k,dict(v))
PY
