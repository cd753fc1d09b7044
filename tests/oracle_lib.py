"""ctypes wrappers around the checkers under oracle/ (TEST INFRASTRUCTURE):
   Oracle -- oracle/librwkv_oracle.so, the CPU restatement of rwkv.cu:493-593
   Ref    -- oracle/_ref/libref.so, the reference's own rwkv.cu + rwkv.h built with hipcc (GPU only)
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librwkv_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
VOCAB = 50277
MODE_PARRALEL, MODE_GPT = 0, 1

vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Oracle:
    def __init__(self):
        L = C.CDLL(ORACLE_SO)
        L.oracle_open_file.argtypes = [C.c_char_p]; L.oracle_open_file.restype = vp
        L.oracle_from_ptrs.argtypes = [u64, u64, C.POINTER(vp)]; L.oracle_from_ptrs.restype = vp
        L.oracle_close.argtypes = [vp]
        L.oracle_n_layers.argtypes = [vp]; L.oracle_n_layers.restype = u64
        L.oracle_n_embed.argtypes = [vp]; L.oracle_n_embed.restype = u64
        L.oracle_forward.argtypes = [vp, C.POINTER(u64), u64, i32, C.POINTER(vp), vp]; L.oracle_forward.restype = i32
        L.oracle_mm8_one_f64.argtypes = [u64, u64, vp, vp, vp, vp, vp, u64, u64]
        L.oracle_mm8_one_f32.argtypes = [u64, u64, vp, vp, vp, vp, vp, u64, u64]
        L.oracle_meanvar.argtypes = [u64, vp, u64, vp, vp]
        L.oracle_layernorm.argtypes = [u64, vp, vp, u64, vp, vp, vp, u64]
        L.oracle_wkv.argtypes = [u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, u64, i32]
        L.oracle_mm8_three.argtypes = [u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64]
        L.oracle_mixatt.argtypes = [u64, vp, vp, vp, vp, vp, vp, u64, u64, u64, i32]
        L.oracle_mixffn.argtypes = [u64, vp, vp, vp, vp, vp, vp, u64, u64, u64, i32]
        L.oracle_quantize_matrix.argtypes = [vp, u64, u64, vp, vp, vp]
        L.oracle_argmax_ban0.argtypes = [vp]; L.oracle_argmax_ban0.restype = u64
        L.oracle_tensor_elems.argtypes = [u64, u64, u64]; L.oracle_tensor_elems.restype = u64
        L.oracle_tensor_type.argtypes = [u64]; L.oracle_tensor_type.restype = u64
        if hasattr(L, "oracle_num_threads"):
            L.oracle_num_threads.argtypes = []; L.oracle_num_threads.restype = i32
        self.L = L

    def set_threads(self, n: int):
        if hasattr(self.L, "oracle_set_threads"):
            self.L.oracle_set_threads.argtypes = [i32]; self.L.oracle_set_threads.restype = None
            self.L.oracle_set_threads(int(n))

    def num_threads(self) -> int:
        return int(self.L.oracle_num_threads()) if hasattr(self.L, "oracle_num_threads") else 1

    # -- whole model -------------------------------------------------------------------------
    def open_file(self, path):
        h = self.L.oracle_open_file(os.fsencode(path))
        if not h:
            raise IOError(path)
        return OracleModel(self, h, None)

    def from_tensors(self, n_layers, n_embed, tensors):
        keep = [None if t is None else np.ascontiguousarray(t) for t in tensors]
        ptrs = (vp * 46)(*[None if t is None else t.ctypes.data for t in keep])
        h = self.L.oracle_from_ptrs(n_layers, n_embed, ptrs)
        return OracleModel(self, h, keep)

    # -- single kernels ----------------------------------------------------------------------
    def mm8_one(self, x, w, r, o, y0=None):
        """x [T][N] f64 or f32, w [N][M] u8 -> y [T][M] f32 (rwkv.cu:267-295, one layer)"""
        x = np.ascontiguousarray(x); T, N = x.shape; M = w.shape[1]
        y = np.zeros((T, M), np.float32) if y0 is None else np.array(y0, np.float32, copy=True)
        f = self.L.oracle_mm8_one_f64 if x.dtype == np.float64 else self.L.oracle_mm8_one_f32
        f(N, M, _p(x), _p(np.ascontiguousarray(w)), _p(y), _p(np.ascontiguousarray(r, np.float32)),
          _p(np.ascontiguousarray(o, np.float32)), 0, T)
        return y

    def layernorm(self, x, lnrows):
        """x [T][D] f64, lnrows [2][D] (weight, bias) -> [T][D] f64 (rwkv.cu:40-57,412-465)"""
        x = np.ascontiguousarray(x, np.float64); T, D = x.shape
        mean = np.zeros(T, np.float32); var = np.zeros(T, np.float32); out = np.zeros_like(x)
        self.L.oracle_meanvar(D, _p(x), T, _p(mean), _p(var))
        ln = np.ascontiguousarray(lnrows, np.float64)
        self.L.oracle_layernorm(D, _p(x), _p(ln), 0, _p(mean), _p(var), _p(out), T)
        return out

    def wkv(self, w, u, k, v, r, aa, bb, pp):
        """one layer, GPT mode; k,v,r [T][C] f32; aa,bb,pp [C] f64 updated in place -> y [T][C]"""
        T, Cn = k.shape
        y = np.zeros((T, Cn))
        self.L.oracle_wkv(Cn, _p(w), _p(u), _p(k), _p(v), _p(r), _p(y), _p(aa), _p(bb), _p(pp), 0, 1, T, MODE_GPT)
        return y

    # -- the pieces of one layer, on a model's flat FILE-layout tensors (layer = l); T = 1, GPT mode ----------
    def mm8_layer(self, x, w, r, o, N, M, layer, y0=None):
        """kernelc_mm8_one on layer `layer` of a stacked matrix w[L][N][M] (rwkv.cu:267-311): x f64[N] or f32[N] -> f32[M] (+ y0)"""
        x = np.ascontiguousarray(x)
        y = np.zeros(M, np.float32) if y0 is None else np.array(y0, np.float32, copy=True)
        f = self.L.oracle_mm8_one_f64 if x.dtype == np.float64 else self.L.oracle_mm8_one_f32
        f(N, M, _p(x), _p(w), _p(y), _p(r), _p(o), layer, 1)
        return y

    def mm8_three(self, xy, km, vm, rm, kr, vr, rr, o1, o2, o3, D, layer):
        """kernel_mm8_threec (rwkv.cu:58-142): xy f32[3][D] -> k, v, r f32[D]"""
        xy = np.ascontiguousarray(xy, np.float32)
        k, v, r = (np.zeros(D, np.float32) for _ in range(3))
        self.L.oracle_mm8_three(D, _p(xy), _p(km), _p(vm), _p(rm), _p(kr), _p(vr), _p(rr), _p(o1), _p(o2), _p(o3), _p(k), _p(v), _p(r), layer, 1)
        return k, v, r

    def mixatt(self, ln, sxy, mixk, mixv, mixr, D, layer, layers):
        """mixatt (rwkv.cu:351-392): ln f64[D], state xy (updated in place, whole [L][D] array) -> f32[3][D]"""
        out = np.zeros(3 * D, np.float32)
        self.L.oracle_mixatt(D, _p(np.ascontiguousarray(ln, np.float64)), _p(sxy), _p(mixk), _p(mixv), _p(mixr), _p(out), layer, layers, 1, MODE_GPT)
        return out

    def mixffn(self, ln, sdd, mixk, mixr, D, layer, layers):
        """mixffn (rwkv.cu:313-349): ln f64[D], state dd (updated in place) -> ffn_k input, ffn_r input, f64[D] each"""
        ok, orr = np.zeros(D), np.zeros(D)
        self.L.oracle_mixffn(D, _p(np.ascontiguousarray(ln, np.float64)), _p(sdd), _p(mixk), _p(mixr), _p(ok), _p(orr), layer, layers, 1, MODE_GPT)
        return ok, orr

    def wkv_layer(self, w, u, k, v, r, aa, bb, pp, D, layer, layers):
        """kernel_wkvc_forward (rwkv.cu:221-265) on layer `layer`: k, v, r f32[D]; aa, bb, pp whole [L][D] state arrays, updated in place -> y f64[D]"""
        y = np.zeros(D)
        self.L.oracle_wkv(D, _p(w), _p(u), _p(k), _p(v), _p(r), _p(y), _p(aa), _p(bb), _p(pp), layer, layers, 1, MODE_GPT)
        return y

    def quantize_matrix(self, xx):
        xx = np.ascontiguousarray(xx, np.float32); n_out, n_in = xx.shape
        q = np.zeros((n_in, n_out), np.uint8); r = np.zeros(n_in, np.float32); o = np.zeros(n_in, np.float32)
        self.L.oracle_quantize_matrix(_p(xx), n_out, n_in, _p(q), _p(r), _p(o))
        return q, r, o

    def argmax_ban0(self, logits):
        return int(self.L.oracle_argmax_ban0(_p(np.ascontiguousarray(logits, np.float32))))


class OracleModel:
    def __init__(self, lib, h, keep):
        self.lib, self.h, self.keep = lib, h, keep
        self.L_, self.D = int(lib.L.oracle_n_layers(h)), int(lib.L.oracle_n_embed(h))

    def new_state(self, slots=1):
        return [np.zeros(slots * self.L_ * self.D) for _ in range(5)]

    def forward(self, tokens, state, mode=MODE_GPT):
        toks = (u64 * len(tokens))(*[int(t) for t in tokens])
        logits = np.zeros((len(tokens), VOCAB), np.float32)
        sp = (vp * 5)(*[s.ctypes.data for s in state])
        rc = self.lib.L.oracle_forward(self.h, toks, len(tokens), mode, sp, _p(logits))
        if rc:
            raise RuntimeError(f"oracle_forward rc={rc}")
        return logits

    def close(self):
        if self.h:
            self.lib.L.oracle_close(self.h); self.h = None


class Ref:
    """the reference's own kernel (needs a GPU)"""

    def __init__(self):
        L = C.CDLL(REF_SO)
        L.ref_load_file.argtypes = [C.c_char_p, u64]; L.ref_load_file.restype = vp
        L.ref_from_ptrs.argtypes = [u64, u64, C.POINTER(vp), u64]; L.ref_from_ptrs.restype = vp
        L.ref_forward.argtypes = [vp, C.POINTER(u64), u64, i32]; L.ref_forward.restype = C.POINTER(C.c_float)
        L.ref_state.argtypes = [vp, i32]; L.ref_state.restype = C.POINTER(C.c_double)
        L.ref_n_layers.argtypes = [vp]; L.ref_n_layers.restype = u64
        L.ref_n_embed.argtypes = [vp]; L.ref_n_embed.restype = u64
        self.L = L

    def load_file(self, path, maxGPT=1):
        return RefModel(self, self.L.ref_load_file(os.fsencode(path), maxGPT), maxGPT)

    def from_ptrs(self, n_layers, n_embed, ptrs46, maxGPT=1):
        arr = (vp * 46)(*ptrs46)
        h = self.L.ref_from_ptrs(n_layers, n_embed, arr, maxGPT)
        if not h:
            raise RuntimeError("ref_from_ptrs failed")
        return RefModel(self, h, maxGPT)


class RefModel:
    def __init__(self, lib, h, maxGPT):
        self.lib, self.h, self.maxGPT = lib, h, maxGPT
        self.L_, self.D = int(lib.L.ref_n_layers(h)), int(lib.L.ref_n_embed(h))

    def forward(self, tokens, mode=MODE_GPT):
        toks = (u64 * len(tokens))(*[int(t) for t in tokens])
        p = self.lib.L.ref_forward(self.h, toks, len(tokens), mode)
        if not p:
            raise RuntimeError("reference forward threw")
        return np.ctypeslib.as_array(p, shape=(len(tokens), VOCAB)).copy()

    def state(self, which):
        p = self.lib.L.ref_state(self.h, which)
        return np.ctypeslib.as_array(p, shape=(self.maxGPT * self.L_ * self.D,))
