# how long a fresh process takes to start the HIP runtime, with /dev/shm empty and with tens of GB freshly written into it
t() { python -c "import time; t=time.time(); import torch; torch.cuda.init(); torch.zeros(1, device='cuda'); print('init+first op %.2f s' % (time.time()-t))" 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo; }
grep -E "MemTotal|MemAvailable|Shmem:|HugePages_Total|AnonHugePages" /proc/meminfo | tr '\n' ';'; echo
echo "shm empty:"; t; t
for g in 40 120; do
  mkdir -p /dev/shm/fill; for i in $(seq 1 $((g/10))); do head -c 10G /dev/urandom > /dev/shm/fill/f$g_$i & done; wait
  echo "after writing $g GB more into /dev/shm:"; grep -E "MemAvailable|Shmem:" /proc/meminfo | tr '\n' ';'; echo; t; t
done
rm -rf /dev/shm/fill
echo "after removing them:"; t
