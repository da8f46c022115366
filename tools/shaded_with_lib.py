"""tools/shaded_bench.py against another build of the library:  python tools/shaded_with_lib.py build/x/libswfr.so large|magnified [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, sys.argv[1])
from swf_renderer_amd import api
api.library_path = lambda: lib
sys.argv = [os.path.join(ROOT, "tools", "shaded_bench.py")] + sys.argv[2:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"))
