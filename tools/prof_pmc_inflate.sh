# PMC passes over k_bgzf_inflate (tools/exp_inflate_gpu.py at a quarter of chr19); prints the sums per kernel
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /root/repo/gpurun_out/infpmc/$name -o pmc -- python /root/repo/tools/exp_inflate_gpu.py 0.25 1 > /root/repo/gpurun_out/infpmc_$name.log 2>&1; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH
run c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run d SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run e SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH_LEVEL
cd /root/repo
python - <<'PY'
import csv,glob,collections
for d in "abcde":
    f=glob.glob("gpurun_out/infpmc/%s/*counter_collection.csv"%d)
    if not f: print(d,"none"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:24]
        if "k_bgzf_inflate" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items(): print(d,k,{a:int(b) for a,b in v.items()})
PY
