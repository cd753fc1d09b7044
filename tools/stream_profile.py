"""stream profile of the tile-form kernels (tuning build -DRWKV_TL_STREAM=1, selected with RWKV_LIB): when the loader wave has requested a
quarter / half / three quarters / all of its units and when the last has landed, and the time every consumer wave spent waiting for units.
usage: RWKV_LIB=.../lib_tlstream.so RWKV_TL_CLASS=1..4 python tools/stream_profile.py [7B]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
L, D = mf.SHAPES[model]
L = min(L, 8)
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7):
    m.forward(tk)
if os.environ.get("TL_MODE") == "2":       # RWKV_TL_STREAM=2 build: arrival at / departure from the order barrier, per wave
    for rep in range(3):
        buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
        t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
        us = (buf - t0) / 100.0
        arr = np.concatenate([us[:, :7, 1], us[:, 7:, 3]], axis=1); dep = np.concatenate([us[:, :7, 2], us[:, 7:, 1]], axis=1)
        if rep: print("class %s %s: entry %s | at the order barrier, by wave (7 = loader): %s | behind it: %s" % (os.environ.get("RWKV_TL_CLASS"), model, np.round(us[:, :, 0].mean(axis=0), 2), np.round(arr.mean(axis=0), 2), np.round(dep.mean(axis=0), 2)))
    m.close(); sys.exit(0)
acc = []
for rep in range(4):
    buf = m.debug_timeline(9).reshape(-1, 8, 8)[:256].astype(np.int64)
    t0 = buf[:, :, 0][buf[:, :, 0] > 0].min()
    ld = (buf[:, 7, :] - t0) / 100.0                       # loader wave: 1 stream starts, 3 / 4 / 5 quarters requested, 6 all requested, 2 all landed
    wait = buf[:, :7, 2] / 100.0                           # consumer waves: time waited for units
    staged = (buf[:, :7, 5] - t0) / 100.0; loop = (buf[:, :7, 6] - t0) / 100.0; end = (buf[:, :, 7] - t0) / 100.0
    acc.append([ld[:, 1].mean(), ld[:, 3].mean(), ld[:, 4].mean(), ld[:, 5].mean(), ld[:, 6].mean(), ld[:, 2].mean(), staged.mean(), loop.mean(), wait.mean(), wait.max(axis=1).mean(), end.max(axis=1).mean(), end.max()])
a = np.array(acc[1:]).mean(axis=0)
print("class %s %s: stream starts %.2f | 1/4 %.2f | 1/2 %.2f | 3/4 %.2f | all requested %.2f | landed %.2f || consumers: staged %.2f loop end %.2f, waited for units mean %.2f (slowest wave of a workgroup %.2f) || workgroup end mean %.2f last %.2f"
      % ((os.environ.get("RWKV_TL_CLASS"), model) + tuple(a)))
m.close()
