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
    print("  by wave  staged:", np.round([us[:, w, 5][buf[:, w, 5] > 0].mean() if (buf[:, w, 5] > 0).any() else -1 for w in range(8)], 2),
          " loop end:", np.round([us[:, w, 6][buf[:, w, 6] > 0].mean() for w in range(8)], 2),
          " ph2:", np.round([us[:, w, 2][buf[:, w, 2] > 0].mean() if (buf[:, w, 2] > 0).any() else -1 for w in range(8)], 2))
m.close()
# per-workgroup view of the last repetition: is the spread between workgroups systematic?
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7):
    m.forward(tk)
ends = []
for rep in range(4):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
    t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
    ends.append(((buf[:, :, 6] - t0) / 100.0).max(axis=1))      # loop+sync per workgroup
ends = np.array(ends)                                            # [rep][block]
print("loop-end per XCD (block %% 8), mean over reps:", np.round([ends[:, x::8].mean() for x in range(8)], 2))
print("loop-end by block range of 32:", np.round([ends[:, i:i + 32].mean() for i in range(0, 256, 32)], 2))
c = np.corrcoef(ends)
print("rep-to-rep correlation of per-block end times:", np.round(c[0, 1:], 2))
order = np.argsort(ends.mean(axis=0))
print("fastest blocks", order[:10], np.round(ends.mean(axis=0)[order[:10]], 2))
print("slowest blocks", order[-10:], np.round(ends.mean(axis=0)[order[-10:]], 2))
m.close()
