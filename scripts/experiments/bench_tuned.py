# dev tool: bench.py under a plan-time tuning (plx_ssfm_tuning_override: the A/B switches are no longer environment variables).
#   usage: python scripts/experiments/bench_tuned.py field=value [field=value ...] -- <bench.py arguments>
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
split = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
fields = {k: int(v) for k, v in (t.split("=") for t in sys.argv[1:split])}
from polmux_amd import _abi
b = _abi.get()
if fields:
    b.call("plx_ssfm_tuning_override", C.byref(b.tuning(**fields)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[split + 1:]
import bench
bench.main()
