#!/usr/bin/env python3
"""Host-side timing of gci_bam_heads against the whole-stream inflate + offset chase, chr19 / 40x (no GPU work)."""
import os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gci_amd import synth, hostio
from gci_amd.formats import bam as bamfmt
print("cpu_count", os.cpu_count(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-",
      "affinity", len(os.sched_getaffinity(0)))
tmp = tempfile.mkdtemp(prefix="gci_heads_")
rs = synth.simulate_reads(synth.CHR19, 40, "hifi", seed=synth.seed_for(2, 0))
stream, offs = synth.to_bam_stream(rs)
p = os.path.join(tmp, "a.bam")
bamfmt.write_bam_stream(p, stream, level=1, threads=os.cpu_count())
del stream
raw = np.fromfile(p, dtype=np.uint8)
for th in (8, 16, 32, 64, 128):
    row = []
    for gb in (4 << 20, 16 << 20, 64 << 20):
        ts = []
        for _ in range(3):
            t = time.perf_counter(); H = hostio.bam_heads(raw, threads=th, group_bytes=gb); ts.append(time.perf_counter() - t)
            t = time.perf_counter(); H.close(); tc = time.perf_counter() - t
        row.append("%d MiB %.3f s (close %.3f)" % (gb >> 20, min(ts), tc))
    t = time.perf_counter(); s = hostio.bgzf_inflate(raw, threads=th); o = hostio.bam_record_offsets(s); tl = time.perf_counter() - t
    t = time.perf_counter(); del s; td = time.perf_counter() - t
    print("threads %3d:" % th, " | ".join(row), "| whole stream %.3f s + release %.3f" % (tl, td), flush=True)
shutil.rmtree(tmp)
