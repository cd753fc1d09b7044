"""time rwkv_load_file on a model.bin of a given size (written here from synthetic tensors) and check that the file-loaded
context computes what the device-tensor context computes.  usage: loadbench.py [model] [dir]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rwkv_cpp_accelerated_amd import engine, modelfile as mf
model = sys.argv[1] if len(sys.argv) > 1 else "7B"
where = sys.argv[2] if len(sys.argv) > 2 else ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
L, D = mf.SHAPES[model]
t = mf.synthetic_tensors_torch(L, D, seed=0)
a = engine.RWKV(resident=True); a.loadTensors(L, D, t)
ref = [np.array(a.forward(tk)[: mf.VOCAB]) for tk in (5, 17, 300)]
a.close()
path = os.path.join(where, f"rwkv_{model}.bin")
t0 = time.perf_counter()
sz = mf.sizes(L, D)
host = [np.zeros(sz[i], dtype=mf.DTYPES[i]) if x is None else x.cpu().numpy() for i, x in enumerate(t)]
del t; torch.cuda.empty_cache()
mf.write_bin(path, L, D, host)
del host
t_write = time.perf_counter() - t0
size = os.path.getsize(path)
out = dict(model=model, file_bytes=size, where=where, write_s=round(t_write, 2))
for rep in range(2):                      # second pass: the file is in the page cache
    b = engine.RWKV(resident=True)
    t0 = time.perf_counter()
    b.loadFile(path)
    dt = time.perf_counter() - t0
    got = [np.array(b.forward(tk)[: mf.VOCAB]) for tk in (5, 17, 300)]
    b.close()
    same = all(np.array_equal(x, y) for x, y in zip(ref, got))
    out[f"load_s_pass{rep}"] = round(dt, 3)
    out[f"GBps_pass{rep}"] = round(size / dt / 1e9, 2)
    out["logits_identical_to_device_tensor_load"] = bool(same)
os.remove(path)
print(json.dumps(out))
