import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eggshell_amd import capi, scenes
ctx = capi.Context(0)
r = bench.world_leg(ctx, 30)
print({k: v for k, v in r.items() if k != "note"})
ctx.close()
