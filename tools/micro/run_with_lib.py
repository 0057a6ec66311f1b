"""Any tool of this repository on a lab build of the kernel library: python tools/micro/run_with_lib.py <lib.so> <script.py> [arguments]."""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from anemoi_models_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.abspath(sys.argv[2])] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name="__main__")
