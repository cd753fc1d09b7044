"""N independent single-stream greedy decodes on ONE GPU, one context (own weights copy, own stream) and one host thread each:
do two sequences hide each other's per-launch start-up and tail the way the prefill pipeline's stages do?  usage: [model] [N] [steps]"""
import sys, os, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L, D = mf.SHAPES[model]
t = mf.synthetic_tensors_torch(L, D, seed=0)
ms = []
for i in range(N):
    m = engine.RWKV(resident=True); m.loadTensors(L, D, t); ms.append(m)
del t; torch.cuda.empty_cache()
for m in ms:
    m.decode_greedy(5, 16)
t0 = time.perf_counter(); ms[0].decode_greedy(7, steps); dt1 = time.perf_counter() - t0
outs = [None] * N
def run(i):
    outs[i] = ms[i].decode_greedy(7 + i, steps)
th = [threading.Thread(target=run, args=(i,)) for i in range(N)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dtn = time.perf_counter() - t0
print(json.dumps(dict(model=model, sequences=N, steps=steps, single_tok_s=round(steps / dt1, 1), aggregate_tok_s=round(N * steps / dtn, 1),
                      per_sequence_tok_s=round(steps / dtn, 1), speedup=round(N * dt1 / dtn, 3))))
