import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios
from helpers import oracle_render, product_render, diff_stats
SC = scenarios.scenarios()
names = sys.argv[1:] or sorted(SC)
for name in names:
    sc = SC[name]
    g, o = product_render(sc), oracle_render(sc)
    d = (g != o).any(-1)
    print(name, "diff", int(d.sum()), flush=True)
    if d.any():
        ys, xs = np.nonzero(d)
        for y in sorted(set(ys.tolist()))[:12]:
            xx = xs[ys == y]
            print("  row", y, "cols", int(xx.min()), "..", int(xx.max()), "n", len(xx), "gpu", g[y, xx[0]].tolist(), "oracle", o[y, xx[0]].tolist())
