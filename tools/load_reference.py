"""Load the unmodified reference module /root/reference/GCI.py in THIS container only
(SURVEY.md F2): pysam and Bio are absent, so stand-ins from tools/ref_shim are placed in
sys.modules first.  Used by tools/make_golden.py and by tests that are skipped when
/root/reference does not exist (it never does on the GPU box)."""
import importlib.util
import os
import sys

REF = "/root/reference/GCI.py"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available() -> bool:
    return os.path.exists(REF)


def load():
    if "gci_reference" in sys.modules:
        return sys.modules["gci_reference"]
    shim = os.path.join(_HERE, "ref_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("gci_reference", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gci_reference"] = mod
    spec.loader.exec_module(mod)
    return mod
