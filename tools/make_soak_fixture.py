"""Writes one child of a soak scene as a JSON fixture (scenes far down a generator's sequence take too long to regenerate in a test):
  SOAK_BIG=1 python tools/make_soak_fixture.py big 7000 2285 3   ->  tests/golden/soak_big_7000_2285_child3.json
  python tools/make_soak_fixture.py mixed 7100 2196 0,1           ->  tests/golden/soak_mixed_7100_2196_child0_1.json  (several children, in order)
Only for children without bitmap fills (the scene's bitmaps are not written)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import soak_scene

name, seed, index, child = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), [int(c) for c in sys.argv[4].split(",")]
sc = soak_scene(name, seed, index)
out = {"width": sc["width"], "height": sc["height"], "even_odd": bool(sc.get("even_odd")), "stage": {"children": [sc["stage"]["children"][c] for c in child]}}
path = os.path.join(ROOT, "tests", "golden", "soak_%s_%d_%d_child%s.json" % (name, seed, index, "_".join(str(c) for c in child)))
json.dump(out, open(path, "w"), indent=1)
print(path, os.path.getsize(path), "bytes")
