"""print the phase timeline of the middle layer's ffn_rk kernel (tuning aid)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
L, D = mf.SHAPES[model]
L = min(L, 8)
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7):
    m.forward(tk)
names = ["entry", "prologue issued", "pre-steps issued", "tuple arrived", "site reduced", "staged", "loop+sync", "end"]
for rep in range(2):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
    t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
    us = (buf - t0) / 100.0
    print(f"rep {rep}: kernel span {us[:, :, 7][buf[:, :, 7] > 0].max():.2f} us (first entry -> last end)")
    for ph in range(8):
        v = us[:, :, ph][buf[:, :, ph] > 0]
        if v.size:
            print(f"  {names[ph]:14s} min {v.min():7.2f}  mean {v.mean():7.2f}  max {v.max():7.2f}  (n={v.size})")
m.close()
