"""How often do the decode kernels find their first weight rows waiting in LDS, and what is a token worth with / without them?
(kernels.hip.h "CARRY".)   python tools/carry_probe.py [model] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RWKV_CARRY_COUNT", "1")
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7B"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L, D = mf.SHAPES[model]
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True)
m.loadTensors(L, D, t)
m.decode_greedy(5, 16)
m.carry_hits()
t0 = time.perf_counter()
ids = m.decode_greedy(7, steps)
dt = time.perf_counter() - t0
hit, miss = m.carry_hits()
print(f"{model}: RWKV_CARRY={os.environ.get('RWKV_CARRY', 'default')} RWKV_CARRY_EDGES={os.environ.get('RWKV_CARRY_EDGES', 'default')}: "
      f"{steps / dt:.1f} tok/s; workgroup launches that found their rows in LDS {hit}, that did not {miss} "
      f"({100.0 * hit / max(1, hit + miss):.2f} % of {hit + miss}; per token {(hit + miss) / steps:.0f})")
m.close()
