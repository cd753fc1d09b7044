"""time the device-side typical sampler and the sampled generation loop (tuning aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
L, D = mf.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "7B"]
if len(sys.argv) > 2: L = int(sys.argv[2])
t = mf.synthetic_tensors_torch(L, D, seed=0)
m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
for tk in (5, 6, 7): m.forward(tk)
m.sample_typical(0.9, 0.8, 0.3)
t0 = time.perf_counter()
for i in range(50): m.sample_typical(0.9, 0.8, (i + 0.5) / 50)
print("sample_typical (launch + sync + 8-byte copy): %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
n = 256
m.decode_greedy(9, 16); m.decode_typical(9, 16)
t0 = time.perf_counter(); m.decode_greedy(9, n); tg = time.perf_counter() - t0
t0 = time.perf_counter(); m.decode_typical(9, n, seed=1); tt = time.perf_counter() - t0
print("decode %d tokens: greedy %.3f ms/token, typical %.3f ms/token (sampler adds %.1f us)" % (n, tg / n * 1e3, tt / n * 1e3, (tt - tg) / n * 1e6))
m.close()
