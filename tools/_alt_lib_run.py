import os, sys, runpy
sys.path.insert(0, os.getcwd())
from aether_amd import _lib
_lib.LIB_PATH = os.path.join(os.getcwd(), "aether_amd", sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
