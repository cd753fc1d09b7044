#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN KERNEL (include/rwkv/cuda/rwkv.cu built
unmodified with hipcc into oracle/_ref/libref.so) on seeded synthetic models.  Needs the MI355X box:

    gpurun -- 'python tools/make_golden.py gpurun_out/golden'   then copy the .npz into tests/golden/

The fixtures pin the CPU oracle (tests/test_golden_cpu.py, runs without a GPU) and the HIP engine
(tests/test_ref_parity_gpu.py) to the reference.  Models are re-generated from (L, D, seed) by
rwkv_cpp_accelerated_amd.modelfile.synthetic_tensors, so only inputs/outputs are stored:
per step a strided sample of the logits (every 29th + the top-16), the argmax, and the final state."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rwkv_cpp_accelerated_amd import modelfile as mf
import oracle_lib
import parity

STRIDE = 29
CASES = [  # name, L, D, seed, mode, chunks of tokens (each chunk = one reference forward() call)
    ("gpt_L2_D64", 2, 64, 101, 1, [[5], [77], [1234], [50276], [9], [10]]),
    ("gpt_L3_D768", 3, 768, 102, 1, [[11], [50276], [1], [4097], [11], [333]]),
    ("gptchunk_L2_D768", 2, 768, 103, 1, [[100, 200, 300, 400], [7, 8, 9]]),
    ("parralel_L2_D256", 2, 256, 104, 0, [[3, 4, 5], [6, 7, 3]]),
    ("greedy_L2_D1024", 2, 1024, 105, 1, None),   # 16 greedy steps from token 42 (ids are part of the fixture)
]


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    ref = oracle_lib.Ref()
    for name, L, D, seed, mode, chunks in CASES:
        t = mf.synthetic_tensors(L, D, seed=seed)
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "model.bin")
            mf.write_bin(p, L, D, t)
            maxT = max(len(c) for c in chunks) if chunks else 1
            rm = ref.load_file(p, maxT)
            samples, tops, topv, ids = [], [], [], []
            if chunks is None:
                chunks, tk = [], 42
                for _ in range(16):
                    chunks.append([tk])
                    lg = rm.forward([tk], mode)
                    tk = parity.argmax_ban0(lg[0]); ids.append(tk)
                    samples.append(lg[:, ::STRIDE].copy()); o = np.argsort(lg[0])[-16:]; tops.append(o[None]); topv.append(lg[0][o][None])
            else:
                for c in chunks:
                    lg = rm.forward(c, mode)
                    samples.append(lg[:, ::STRIDE].copy())
                    o = np.argsort(lg, axis=1)[:, -16:]
                    tops.append(o); topv.append(np.take_along_axis(lg, o, axis=1))
                    ids.extend(parity.argmax_ban0(r) for r in lg)
            state = [rm.state(i).copy() for i in range(5)]
        np.savez_compressed(os.path.join(outdir, f"ref_{name}.npz"), L=L, D=D, seed=seed, mode=mode, stride=STRIDE,
                            tokens=np.array([x for c in chunks for x in c]), chunk_len=np.array([len(c) for c in chunks]),
                            sample=np.concatenate(samples), top_idx=np.concatenate(tops), top_val=np.concatenate(topv),
                            argmax=np.array(ids), **{f"state{i}": s for i, s in enumerate(state)})
        print("wrote", name, "steps", sum(len(c) for c in chunks))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden"))
