"""dev: run bench.py's main() on another build of the library: bench_with_lib.py <name|base> [bench args...]"""
import os, sys
sys.path.insert(0, os.getcwd())
name = sys.argv[1]
from polmux_amd import _abi
if name != "base":
    _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % name)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
