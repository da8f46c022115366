"""Experiment helper (gpurun): runs bench.py's main() against another build of the library, e.g. one compiled with a
different -D tuning macro:  python tools/bench_with_lib.py build/w5/libswfr.so --steps 300 --warmup 30 --no-cpu-baseline"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from swf_renderer_amd import api
api.library_path = lambda: lib
import bench
bench.main()
