"""Run a script against another build of the library:  python tools/_alt_lib_run.py <lib file name in aether_amd/> <script> [args]"""
import os, sys, runpy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from aether_amd import _lib
_lib.LIB_PATH = os.path.join(REPO, "aether_amd", sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
