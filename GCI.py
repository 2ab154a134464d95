#!/usr/bin/env python3
"""Drop-in for the reference's `python GCI.py ...` command line (same flags, same outputs);
the work is done by the gfx950 HIP path in gci_amd/ -- see gci_amd/cli.py."""
import sys

from gci_amd.cli import main

if __name__ == "__main__":
    main(sys.argv)
