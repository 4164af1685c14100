#!/usr/bin/env python3
"""The variable-N decoder step with the first- and second-version filter kernel (aether_set_option dyn_filter_v1)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib
lib = _lib.load()
for mode in (1, 2):
    lib.aether_set_option(b"dyn_filter_v1", mode)
    print("dyn_filter_v1 =", mode, flush=True)
    sys.argv = ["dyn_decoder_time.py"]
    exec(open(os.path.join(os.path.dirname(__file__), "dyn_decoder_time.py")).read())
