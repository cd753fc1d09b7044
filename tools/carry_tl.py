"""phase timeline of the middle layer's k_ffn_rk (class 3) per WAVE, to see when the carried rows' verdict is in (waves 4..6: phase 3 = past
the order barrier, 4 = verdict in) against when the vectors are staged (5, stamped behind the waits for `staged` and `verified`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402
L, D = 8, 4096
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7):
    m.forward(tk)
for rep in range(2):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
    t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
    us = (buf - t0) / 100.0
    def col(ph):
        return np.round([us[:, w, ph][buf[:, w, ph] > 0].mean() if (buf[:, w, ph] > 0).any() else -1 for w in range(8)], 2)
    print(f"rep {rep}: span {us[:, :, 7][buf[:, :, 7] > 0].max():.2f} us | ph1 {col(1)} | ph3 {col(3)} | ph4 {col(4)} | staged(5) {col(5)} | loop end {col(6)} | loader done {col(2)}")
m.close()
