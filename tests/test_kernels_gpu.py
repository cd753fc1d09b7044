"""Per-kernel parity of the PRODUCTION decode kernels: a token is launched one kernel at a time through
rwkv_debug_launch -- k_first, then k_att / k_attout / k_ffn_rk / k_ffnv of every layer, then k_head: exactly the launches of the
captured token graph, in the form (row registers, row ring, 16-row tiles, 4-row tiles) the context runs each class in -- and after every
launch its outputs are read back and compared with the oracle's piece for THAT kernel, evaluated on the kernel's OWN inputs as the
engine holds them (the residual vector and the state it was handed):

   k_att     oracle_layernorm + oracle_mixatt + oracle_mm8_three + oracle_wkv   (rwkv.cu:535-545; kernels :351-392, :58-142, :221-265)
   k_attout  oracle_mm8_one_f64 with the accumulator pre-loaded with f32(x)     (:548-553), state xy
   k_ffn_rk  oracle_layernorm + oracle_mixffn + 2 x oracle_mm8_one_f64, sigmoid, relu^2   (:557-573)
   k_ffnv    oracle_mm8_one_f32 + blockout                                      (:574-577), state dd
   k_head    oracle_layernorm + oracle_mm8_one_f64                              (:585-589)

so the five mm8_one shapes (att_out D->D, ffn_r D->D, ffn_k D->4D, ffn_v 4D->D, head D->V) are checked on the kernels that ship, and a
compensated error inside one launch has a test that names the launch.  k_attout and k_ffnv are ALSO checked against their exact
contract -- the f64 dot product of the vector they were actually handed (YBUF / HBUF and the per-workgroup offset partials) with the
uint8 matrix -- which isolates them from the producer's rounding."""
import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf

pytestmark = pytest.mark.gpu
TOL = 3e-5          # relative to the vector's max |.|: an f32 GEMV of <= 20480 terms is good to ~1e-6, a wrong row or scale is O(1)


def _close(got, ref, what, tol=TOL):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    assert np.isfinite(got).all(), what
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max()) / scale
    assert err <= tol, f"{what}: max |d| / max |ref| = {err:.3e} > {tol:.0e}"
    return err


@pytest.mark.parametrize("D,tile", [(768, None), (2048, None), (2048, "15"), (2560, None), (4096, None), (4096, "0"), (5120, None), (5120, "15")])
def test_each_production_decode_kernel_against_its_oracle_piece(built, oracle, monkeypatch, D, tile):
    import torch
    from rwkv_cpp_accelerated_amd import engine
    if tile is None:
        monkeypatch.delenv("RWKV_TILE", raising=False)
    else:
        if torch.cuda.get_device_properties(0).multi_processor_count != 256:
            pytest.skip("the tile forms are laid out for 256 workgroups")
        monkeypatch.setenv("RWKV_TILE", tile)
    L, V = 2, mf.VOCAB
    t = mf.synthetic_tensors(L, D, seed=4000 + D)
    m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
    G = m.debug_grid()
    # a non-trivial recurrent state (the reference starts from zeros: every mix / WKV term must see real numbers)
    rng = np.random.default_rng(D)
    st = m.state
    st.statexy[:] = rng.standard_normal(L * D); st.statedd[:] = rng.standard_normal(L * D)
    st.stateaa[:] = rng.standard_normal(L * D); st.statebb[:] = 0.5 + 1.5 * rng.random(L * D); st.statepp[:] = rng.standard_normal(L * D)
    m.push_state(1)
    ln = t[mf.LAYERNORMS].reshape(4 * (L + 1), D)
    worst = {}

    def note(k, e):
        worst[k] = max(worst.get(k, 0.0), e)

    token = 4242
    m.debug_launch(0, 0, token, 0)                                          # k_first: embedding row + ln0 (rwkv.cu:513-524)
    x = m.debug_read("x")
    emb = t[mf.EMBED].reshape(V, D)[token].astype(np.float64)
    note("first x", _close(x, oracle.layernorm(emb[None, :], ln[0:2])[0], "k_first: x = ln0(embedding row)"))
    for l in range(L):
        lo = slice(l * D, (l + 1) * D)
        # ---- k_att: ln1, mixatt, K/V/R, WKV, from the engine's own x and state ----
        m.pull_state(1)
        sxy, saa, sbb, spp, sdd = (a[: L * D].copy() for a in st.arrays())
        ln1 = oracle.layernorm(x[None, :], ln[4 * l + 2: 4 * l + 4])[0]
        sxy_o = sxy.copy()
        kvr_in = oracle.mixatt(ln1, sxy_o, t[mf.MIXK], t[mf.MIXV], t[mf.MIXR], D, l, L)          # (writes ln1 into sxy_o[l])
        k, v, r = oracle.mm8_three(kvr_in, t[mf.KM], t[mf.VM], t[mf.RM], t[mf.KR], t[mf.VR], t[mf.RR], t[mf.O1], t[mf.O2], t[mf.O3], D, l)
        aa_o, bb_o, pp_o = saa.copy(), sbb.copy(), spp.copy()
        y = oracle.wkv_layer(t[mf.DECAY], t[mf.BONUS], k, v, r, aa_o, bb_o, pp_o, D, l, L)
        m.debug_launch(1, l)
        ybuf = m.debug_read("ybuf"); part_a = m.debug_read("part_att"); pmax_a = m.debug_read("pmax_att")
        m.pull_state(1)
        yf = y.astype(np.float32)                                                                  # the att_out GEMV reads it as f32 (rwkv.cu:290)
        note("att y", _close(ybuf, yf * t[mf.ATTOUTR][lo], f"k_att layer {l}: gated wkv * att_out scale"))
        note("att aa", _close(st.stateaa[lo], aa_o[lo], f"k_att layer {l}: state aa", 1e-4))
        note("att bb", _close(st.statebb[lo], bb_o[lo], f"k_att layer {l}: state bb", 1e-4))
        assert np.array_equal(st.statepp[: L * D], spp), "pp is carried through (rwkv.cu:257)"
        terms = yf.astype(np.float64) * t[mf.ATTOUTO][lo]
        assert abs(part_a.sum() - terms.sum()) <= 1e-5 * np.abs(terms).sum(), f"k_att layer {l}: offset partials"
        assert abs(float(pmax_a.max()) - float(np.abs(ybuf).max())) <= 1e-12, "per-workgroup maxima of YBUF"
        # ---- k_attout: x = f32(x) + att_out . y, state xy = ln1 ----
        acc0 = x.astype(np.float32)
        x_ref = oracle.mm8_layer(y, t[mf.ATTOUT], t[mf.ATTOUTR], t[mf.ATTOUTO], D, D, l, y0=acc0).astype(np.float64)
        w_att = t[mf.ATTOUT].reshape(L, D, D)[l].astype(np.float64)
        x_contract = (acc0 + (ybuf.astype(np.float64) @ w_att + part_a.sum()).astype(np.float32)).astype(np.float64)
        m.debug_launch(2, l)
        x1 = m.debug_read("x")
        m.pull_state(1)
        note("attout x (oracle)", _close(x1 - x, x_ref - x, f"k_attout layer {l}: residual update vs oracle_mm8_one"))
        note("attout x (contract)", _close(x1 - x, x_contract - x, f"k_attout layer {l}: residual update vs the f64 product of its own input", 1e-5))
        note("attout xy", _close(st.statexy[lo], ln1, f"k_attout layer {l}: state xy = ln1 output"))
        # ---- k_ffn_rk: ln2, mixffn, ffn_r + sigmoid, ffn_k + relu^2, from the engine's own x ----
        ln2 = oracle.layernorm(x1[None, :], ln[4 * l + 4: 4 * l + 6])[0]
        sdd_o = sdd.copy()
        k_in, r_in = oracle.mixffn(ln2, sdd_o, t[mf.FFNMIXK], t[mf.FFNMIXV], D, l, L)
        rr = oracle.mm8_layer(r_in, t[mf.FFNR], t[mf.FFNRR], t[mf.FFNRO], D, D, l)
        sig = (1.0 / (1.0 + np.exp(-rr.astype(np.float64)))).astype(np.float32)                    # rwkv.cu:212
        kk = oracle.mm8_layer(k_in, t[mf.FFNK], t[mf.FFNKR], t[mf.FFNKO], D, 4 * D, l)
        h = kk * (kk > 0).astype(np.float32); h = h * h                                            # rwkv.cu:189-190
        m.debug_launch(3, l)
        rgate = m.debug_read("rgate"); hbuf = m.debug_read("hbuf"); part_f = m.debug_read("part_ffn")
        fvr = t[mf.FFNVR][l * 4 * D: (l + 1) * 4 * D]; fvo = t[mf.FFNVO][l * 4 * D: (l + 1) * 4 * D]
        note("ffn sigmoid(r)", _close(rgate, sig, f"k_ffn_rk layer {l}: sigmoid(ffn_r)"))
        note("ffn relu^2(k)", _close(hbuf, h * fvr, f"k_ffn_rk layer {l}: relu(ffn_k)^2 * ffn_v scale"))
        terms = h.astype(np.float64) * fvo
        assert abs(part_f.sum() - terms.sum()) <= 1e-5 * np.abs(terms).sum(), f"k_ffn_rk layer {l}: offset partials"
        # ---- k_ffnv: x += ffn_v . h * sigmoid(r), state dd = ln2 ----
        vv = oracle.mm8_layer(h, t[mf.FFNV], t[mf.FFNVR], t[mf.FFNVO], 4 * D, D, l)
        x2_ref = x1 + (vv * sig).astype(np.float64)                                                # blockout (rwkv.cu:407): f32 product
        w_fv = t[mf.FFNV].reshape(L, 4 * D, D)[l].astype(np.float64)
        v_contract = (hbuf.astype(np.float64) @ w_fv + part_f.sum()).astype(np.float32)
        x2_contract = x1 + (v_contract * rgate).astype(np.float64)
        m.debug_launch(4, l)
        x2 = m.debug_read("x")
        m.pull_state(1)
        note("ffnv x (oracle)", _close(x2 - x1, x2_ref - x1, f"k_ffnv layer {l}: residual update vs oracle_mm8_one"))
        note("ffnv x (contract)", _close(x2 - x1, x2_contract - x1, f"k_ffnv layer {l}: residual update vs the f64 product of its own input", 1e-5))
        note("ffnv dd", _close(st.statedd[lo], ln2, f"k_ffnv layer {l}: state dd = ln2 output"))
        x = x2
    # ---- k_head: ln_out + head ----
    lno = oracle.layernorm(x[None, :], ln[4 * L + 2: 4 * L + 4])[0]
    logits_ref = oracle.mm8_layer(lno, t[mf.HEAD], t[mf.HEADR], t[mf.HEADO], D, V, 0)
    m.debug_launch(5, 0)
    note("head", _close(m.logits(1)[:V], logits_ref, "k_head: logits"))
    print(f"D={D} RWKV_TILE={tile} decode_form={m.decode_form()} grid={G}: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    m.close()


def test_debug_hooks_refuse_what_they_cannot_run_and_round_trip_a_buffer(built):
    """rwkv_debug_launch / _read / _write (include/rwkv_mi355x.h): bad class, foreign layer, bad slot and a wrong-sized write fail with a
    message instead of launching; a vector written into the context comes back unchanged; a context that owns a layer range runs only its own
    layers and, without the head, neither class 5 nor 6."""
    from rwkv_cpp_accelerated_amd import engine, pipeline
    L, D = 3, 768
    t = mf.synthetic_tensors(L, D, seed=77)
    m = engine.RWKV(resident=True); m.loadTensors(L, D, t)
    for bad in (dict(cls=7), dict(cls=-1), dict(cls=2, layer=L), dict(cls=0, slot=5), dict(cls=0, token=mf.VOCAB)):
        with pytest.raises(engine.RWKVError):
            m.debug_launch(bad.get("cls"), bad.get("layer", 0), bad.get("token", 1), bad.get("slot", 0))
    x = np.random.default_rng(1).standard_normal(D)
    m.debug_write("x", x)
    assert np.array_equal(m.debug_read("x"), x)
    with pytest.raises(engine.RWKVError):
        m.debug_write("x", x[: D // 2])
    assert m.debug_grid() >= 1 and m.debug_read("lnstat").shape == (6,)
    m.close()
    st = pipeline.EngineStage(t, L, D, 0, 2, n_slots=1)          # layers [0, 2): no head
    st.m.debug_launch(0, 0, 9, 0)
    for cls in (1, 2, 3, 4):
        st.m.debug_launch(cls, 0)
    st.m.debug_launch(1, 1)
    for cls, layer in ((1, 2), (5, 0), (6, 0)):
        with pytest.raises(engine.RWKVError):
            st.m.debug_launch(cls, layer)
    st.m.close()
