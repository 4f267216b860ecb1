#!/usr/bin/env python3
"""Static check of the hand-written DPP instructions in the built kernels: the scanner lives in
the package (riemann_amd/_dpp_hazards.py, run by riemann_amd.build on every build); this is its
command-line entry.

    python tools/check_dpp_hazards.py [objects...]     # default: every object of the library
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riemann_amd._dpp_hazards import *          # noqa: E402,F401,F403
from riemann_amd._dpp_hazards import main       # noqa: E402

if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1:])[1] else 0)
