#!/usr/bin/env python3
"""The stack dumps of a command line run under GCI_STUCK_TRACE=<seconds> (faulthandler, every thread, most recent call first), one line
per thread and dump: the innermost frames inside this repository.  Usage: stuck_summary.py stderr.txt [first dump] [last dump]"""
import re, sys
txt = open(sys.argv[1]).read()
lo, hi = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
dumps = txt.split("Timeout (")[1:]
for n, d in enumerate(dumps):
    if not lo <= n <= hi:
        continue
    print("== dump %d" % n)
    for th in re.split(r"\n(?=Thread 0x|Current thread)", d):
        frames = re.findall(r'File "([^"]+)", line (\d+) in (\S+)', th)
        mine = [(f.split("/")[-1], l, fn) for f, l, fn in frames if "/gci_amd/" in f or f.endswith("GCI.py")]
        if not frames:
            continue
        top = frames[0]
        print("   %-22s top %s:%s %s | %s" % (th.split("\n")[0][:22], top[0].split("/")[-1], top[1], top[2], " < ".join("%s:%s %s" % m for m in mine[:4])))
