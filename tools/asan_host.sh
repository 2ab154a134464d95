#!/bin/bash
# The host-side entry points of the C ABI (gci_amd/csrc/host_io.cpp: threaded BGZF inflate, the heads pipeline, gzip framing, the
# host PAF filter, FASTA titles -- code the command line executes) through g++ with AddressSanitizer + UBSan, and the host-logic
# tests over that build.  No GPU needed.  usage: bash tools/asan_host.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
so=gci_amd/csrc/libgci_host_asan.so
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    gci_amd/csrc/host_io.cpp tools/asan_host_extra.cpp -o $so -lz -lpthread
asan=$(g++ -print-file-name=libasan.so)
ubsan=$(g++ -print-file-name=libubsan.so)
LD_PRELOAD="$asan $ubsan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    GCI_LIB_PATH=$PWD/$so GCI_HOST_ONLY=1 python -m pytest tests/test_host_logic.py -q -p no:cacheprovider "$@"
