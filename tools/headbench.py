import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
L, D = 4, 4096
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7): m.forward(tk)
r = [p for p in m.profile_batched(token=9, reps=64) if p["name"] == "head"][0]
print("head us %.2f -> %.0f GB/s (%.1f%% of 8 TB/s)" % (r["us"], mf.VOCAB * D / r["us"] / 1e3, mf.VOCAB * D / r["us"] / 1e3 / 80))
m.close()
