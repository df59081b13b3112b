#!/usr/bin/env python3
"""Command-line front end of edison_amd/nnom_import.py: NNoM weights.h -> .ednn model blob.

Usage:  tools/import_weights_h.py /path/to/weights.h out.ednn
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edison_amd.nnom_import import *  # noqa: F401,F403,E402  (parse_weights_h, build_blob, deinterleave_dense_opt, main)
from edison_amd.nnom_import import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(sys.argv))
