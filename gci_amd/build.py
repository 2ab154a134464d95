"""Build libgci_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libgci_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("api_ctx.hip", "k_filter.hip", "k_join.hip", "k_depth.hip", "k_track.hip", "k_deflate.hip", "k_paf.hip", "k_inflate.hip", "k_inflate_wave.hip", "k_pages.hip", "k_shard.hip", "k_hbm.hip", "host_io.cpp", "staging.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "gci_common.h"), os.path.join(CSRC, "gci_ctx.hpp"),
                  os.path.join(_HERE, "..", "include", "gci_hip.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wall"]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """One hipcc per source file, side by side (the filter kernel alone takes a minute), then one link."""
    if force or needs_build():
        from concurrent.futures import ThreadPoolExecutor
        obj_dir = os.path.join(CSRC, "build")
        os.makedirs(obj_dir, exist_ok=True)
        cc = hipcc()

        def compile_one(src: str) -> str:
            obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
            cmd = [cc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            return obj

        with ThreadPoolExecutor(max(1, min(len(SOURCES), os.cpu_count() or 1))) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
