# the wave inflate's host-side knobs on one box: streams, copy grid, batch size (tools/hwtests/inflate_product.py at half of chr19)
one() { name=$1; shift; echo "== $name: $(env "$@" CHECK_CRC=0 timeout 200 python /root/repo/tools/hwtests/inflate_product.py 0.5 4 2>&1 | grep -o "calls.*GB/s out; equal to the stream: [A-Za-z]*; .*")"; }
one "2 streams, persistent copy, batch 8192 (default)"
one "1 stream" GCI_INFLATE_STREAMS=1
one "copy grid = members" GCI_INFLATE_COPY_GRID=members
one "1 stream, copy grid = members (round-5 state before)" GCI_INFLATE_STREAMS=1 GCI_INFLATE_COPY_GRID=members GCI_INFLATE_BATCH=16384
one "batch 4096" GCI_INFLATE_BATCH=4096
one "batch 16384" GCI_INFLATE_BATCH=16384
one "default again"
