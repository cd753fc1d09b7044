"""RWKV-v4 .pth -> model.bin converter without the torch C++ extension.

Re-implements what the reference converter does (converter/convert_model.py:16-179 +
converter/cpp_save_tensor.cpp:9-97) on numpy: stack the per-layer parameters into the 46 slots of
the format, quantise the seven matrix families + head with the reference's
per-input-row asymmetric uint8 scheme (`quantize_matrix`, convert_model.py:108-119), and dump the
file with modelfile.write_bin.  torch is only used to read a .pth; a dict of numpy arrays works too.

    python -m rwkv_cpp_accelerated_amd.converter  RWKV-4-....pth  model.bin
"""
from __future__ import annotations

import sys

import numpy as np

from . import modelfile as mf


def _np(t):
    if isinstance(t, np.ndarray):
        return t
    return t.detach().to("cpu").float().numpy() if hasattr(t, "detach") else np.asarray(t)


def quantize_matrix(x: np.ndarray):
    """convert_model.py:108-119.  x: torch weight [out][in].  Returns (u8 [in][out], r f32[in], o f32[in]).
    Arithmetic is f64 like the reference (`mini`/`ran` are .double())."""
    x = np.asarray(x, dtype=np.float32)
    mini = x.min(axis=0).astype(np.float64)
    out = x.astype(np.float64) - mini
    ran = out.max(axis=0) / 255.0
    out = out / ran
    frac = out - np.trunc(out)
    mini = mini + frac.mean(axis=0) * ran            # truncation-bias compensation
    q = np.ascontiguousarray(np.trunc(out).T.astype(np.uint8))
    return q, ran.astype(np.float32), mini.astype(np.float32)


def convert_state_dict(w: dict):
    """-> (n_layers, n_embed, [46 numpy tensors in file order])"""
    dims = len(_np(w["blocks.0.att.key.weight"]))                                   # convert_model.py:209
    layers = len([k for k in w.keys() if "blocks" in k and "ln1.bias" in k])         # :210-211
    t = mf._buffers(layers, dims)                                                    # scratch/state slots (:19-25,98-105)
    t[mf.EMBED] = _np(w["emb.weight"]).astype(np.float32).reshape(-1)
    sn = ["blocks.0.ln0.weight", "blocks.0.ln0.bias"]
    for i in range(layers):
        sn += [f"blocks.{i}.ln1.weight", f"blocks.{i}.ln1.bias", f"blocks.{i}.ln2.weight", f"blocks.{i}.ln2.bias"]
    sn += ["ln_out.weight", "ln_out.bias"]
    t[mf.LAYERNORMS] = np.stack([_np(w[k]).reshape(-1) for k in sn]).astype(np.float64).reshape(-1)

    def stack(fmt):
        return np.stack([_np(w[fmt.format(i)]).reshape(-1) for i in range(layers)]).astype(np.float64)

    t[mf.MIXK] = stack("blocks.{}.att.time_mix_k").reshape(-1)
    t[mf.MIXV] = stack("blocks.{}.att.time_mix_v").reshape(-1)
    t[mf.MIXR] = stack("blocks.{}.att.time_mix_r").reshape(-1)
    t[mf.FFNMIXK] = stack("blocks.{}.ffn.time_mix_k").reshape(-1)
    t[mf.FFNMIXV] = stack("blocks.{}.ffn.time_mix_r").reshape(-1)                    # slot named "v" holds time_mix_r (:160-161)
    t[mf.DECAY] = -np.exp(stack("blocks.{}.att.time_decay")).reshape(-1)             # :57-58
    t[mf.BONUS] = stack("blocks.{}.att.time_first").reshape(-1)

    def family(key, wslot, rslot, oslot):
        qs, rs, os_ = zip(*[quantize_matrix(_np(w[f"blocks.{i}.{key}"])) for i in range(layers)])
        t[wslot] = np.stack(qs).reshape(-1); t[rslot] = np.stack(rs).reshape(-1); t[oslot] = np.stack(os_).reshape(-1)

    family("att.key.weight", mf.KM, mf.KR, mf.O1)
    family("att.value.weight", mf.VM, mf.VR, mf.O2)
    family("att.receptance.weight", mf.RM, mf.RR, mf.O3)
    family("att.output.weight", mf.ATTOUT, mf.ATTOUTR, mf.ATTOUTO)
    family("ffn.key.weight", mf.FFNK, mf.FFNKR, mf.FFNKO)
    family("ffn.value.weight", mf.FFNV, mf.FFNVR, mf.FFNVO)
    family("ffn.receptance.weight", mf.FFNR, mf.FFNRR, mf.FFNRO)
    q, r, o = quantize_matrix(_np(w["head.weight"]))                                 # :92-93
    t[mf.HEAD], t[mf.HEADR], t[mf.HEADO] = q.reshape(-1), r, o
    return layers, dims, t


def is_valid_weights_file(w: dict) -> bool:                                          # convert_model.py:182-191
    return all(k in w for k in ("emb.weight", "ln_out.weight", "ln_out.bias", "blocks.0.ln0.weight", "blocks.0.ln0.bias"))


def convert_pth(path_in: str, path_out: str):
    import torch
    w = torch.load(path_in, map_location="cpu")
    if not is_valid_weights_file(w):
        raise ValueError("Invalid weights file structure. Please provide a valid .pth file.")
    L, D, t = convert_state_dict(w)
    mf.write_bin(path_out, L, D, t)
    return L, D


def synthetic_state_dict(n_layers: int, n_embed: int, seed: int = 0, vocab: int = mf.VOCAB):
    """a random RWKV-v4 state dict with the checkpoint's key names/shapes (float32), for tests"""
    rng = np.random.default_rng(seed)
    D = n_embed
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    w = {"emb.weight": f(vocab, D), "head.weight": f(vocab, D) * 0.1,
         "ln_out.weight": 1 + 0.1 * f(D), "ln_out.bias": 0.1 * f(D),
         "blocks.0.ln0.weight": 1 + 0.1 * f(D), "blocks.0.ln0.bias": 0.1 * f(D)}
    for i in range(n_layers):
        b = f"blocks.{i}."
        for ln in ("ln1", "ln2"):
            w[b + ln + ".weight"] = 1 + 0.1 * f(D); w[b + ln + ".bias"] = 0.1 * f(D)
        for k in ("time_mix_k", "time_mix_v", "time_mix_r"):
            w[b + "att." + k] = rng.random((1, 1, D)).astype(np.float32)
        w[b + "att.time_decay"] = rng.uniform(-6, 1, D).astype(np.float32)
        w[b + "att.time_first"] = (0.3 * f(D)).astype(np.float32)
        for k in ("key", "value", "receptance", "output"):
            w[b + f"att.{k}.weight"] = f(D, D) / np.float32(np.sqrt(D))
        w[b + "ffn.time_mix_k"] = rng.random((1, 1, D)).astype(np.float32)
        w[b + "ffn.time_mix_r"] = rng.random((1, 1, D)).astype(np.float32)
        w[b + "ffn.key.weight"] = f(4 * D, D) / np.float32(np.sqrt(D))
        w[b + "ffn.value.weight"] = f(D, 4 * D) / np.float32(np.sqrt(4 * D))
        w[b + "ffn.receptance.weight"] = f(D, D) / np.float32(np.sqrt(D))
    return w


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    L, D = convert_pth(sys.argv[1], sys.argv[2])
    print(f"wrote {sys.argv[2]}: n_layers={L} n_embed={D} ({mf.file_bytes(L, D)} bytes)")
