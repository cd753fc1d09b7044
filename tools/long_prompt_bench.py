"""one rwkv_forward call on a long prompt (passes of RWKV_SEQ_ROWS = 64 or 32 rows as a software pipeline over RWKV_SEQ_STAGES streams) and a 96-stream batched
step: python tools/long_prompt_bench.py [model] [tokens]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch                                                 # noqa: E402
from rwkv_cpp_accelerated_amd import engine, modelfile as mf              # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "7B"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L, D = mf.SHAPES[model]
m = engine.RWKV(resident=True)
m.loadTensors(L, D, mf.synthetic_tensors_torch(L, D, seed=0), maxGPT=max(T, 96))
toks = [int(v) for v in np.random.default_rng(3).integers(2, mf.VOCAB, T)]
m.forward(toks, engine.MODE_GPT)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.forward(toks, engine.MODE_GPT); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
par = (toks * (96 // max(1, len(toks)) + 1))[:96]
m.forward(par, engine.MODE_PARRALEL)
bp = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): m.forward(par, engine.MODE_PARRALEL)
    torch.cuda.synchronize(); bp = min(bp, (time.perf_counter() - t0) / 4)
print(f"{model} RWKV_SEQ_STAGES={os.environ.get('RWKV_SEQ_STAGES', 'default')}: {T}-token prompt {T / best:.0f} tok/s ({best * 1e3:.2f} ms); 96-stream step {96 / bp:.0f} tok/s ({bp * 1e3:.2f} ms)")
m.close()
