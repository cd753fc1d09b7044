"""RUN the reference's own caller on the GPU box: reference examples/storygen/storygen.cpp, compiled from where it lies in
the authoring container (oracle/Makefile: storygen = against this repo's include/rwkv.h, storygen_l2 = against the
reference's own rwkv.h + integration/rwkv_backend_mi355x.cpp), on a 169M-shaped model.bin (BASELINE config 1: L=12, D=768).

storygen is interactive and SAMPLES with typical(out, 0.8, 0.7) (as compiled: a draw from softmax(logits), see
include/rwkv_sampler.h) from its own random generator, so its ids cannot be predicted; the test drives it with one line
on stdin, decodes the ids it printed, and REPLAYS them through the Python engine along the same call sequence
(storygen.cpp:29-73), teacher-forced: every printed id must be a token the engine's logits at that point give real
probability to, and -- the test model's logits are peaked -- most of them must be the argmax.  A state or plumbing error
anywhere (tokenizer ids, loadContext chunks, sub-state snapshot, the doubled last prompt token, out[0] = -99) makes the
replayed probabilities collapse.  The tokenizer is the reference's own
GPT2Tokenizer (compiled in); its vocab files are not available on the GPU box, so the test writes a byte-level vocab
(256 byte tokens + one unique 5-letter string per remaining id, no merges) -- enough for encode() and decode()."""
import json
import os
import re
import subprocess
import time

import numpy as np
import pytest

from rwkv_cpp_accelerated_amd import modelfile as mf
import parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INIT_PROMPT = "### Instruction: Write a story/book using the themes and details provided\n\n### Input:"   # storygen.cpp:5-7
USER_LINE = "a knight"


def bytes_to_unicode():
    """the GPT-2 byte <-> printable-unicode table the reference tokenizer hard-codes (tokenizer.h:24-33)"""
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def word_of(i):
    s = ""
    for _ in range(4):
        s = chr(97 + i % 26) + s; i //= 26
    return "Z" + s


def write_vocab(d):
    os.makedirs(d, exist_ok=True)
    b2u = bytes_to_unicode()
    vocab = {b2u[b]: b for b in range(256)}
    for i in range(256, mf.VOCAB):
        vocab[word_of(i)] = i
    assert len(vocab) == mf.VOCAB
    with open(os.path.join(d, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")


def encode(text):
    return list(text.encode("utf-8"))       # no merges: one token per byte, id = byte value


def decode_stream(raw):
    ids, i = [], 0
    while i < len(raw):
        if raw[i:i + 1] == b"Z" and len(raw) >= i + 5 and raw[i + 1:i + 5].isalpha() and raw[i + 1:i + 5].islower():
            v = 0
            for ch in raw[i + 1:i + 5]:
                v = v * 26 + (ch - 97)
            ids.append(v); i += 5
        else:
            ids.append(raw[i]); i += 1
    return ids


@pytest.fixture(scope="module")
def world(built, tmp_path_factory):
    root = tmp_path_factory.mktemp("storygen")
    cwd = root / "examples" / "storygen" / "build"
    cwd.mkdir(parents=True)
    write_vocab(str(root / "include" / "rwkv" / "tokenizer" / "vocab"))
    (root / "converter").mkdir()
    L, D = mf.SHAPES["169M"]
    t = mf.synthetic_tensors(L, D, seed=169, head_scale=120.0)      # peaked logits, yet exp(logit) finite: NumCpp's softmax subtracts no max
    path = str(root / "converter" / "model.bin")
    mf.write_bin(path, L, D, t)
    return dict(cwd=str(cwd), model=path, L=L, D=D)


def replay(world, ids):
    """feed storygen's own ids back through the engine along storygen's call sequence; returns the probability the engine's
    logits gave each id and whether it was the argmax"""
    from rwkv_cpp_accelerated_amd import engine
    m = engine.RWKV(resident=False)          # host-authoritative state, as the reference's RWKV::forward (rwkv.h:353,372)
    m.loadFile(world["model"], 1)
    prompt = encode(INIT_PROMPT)
    for tk in prompt:                         # loadContext(initPrompt), maxContext = 1
        m.forward(tk)
    for tk in encode(USER_LINE + "\n\n### Response:"):
        m.forward(tk)
    tk, probs, top = prompt[-1], [], []       # storygen feeds the prompt's last token again (storygen.cpp:33,57,65)
    for want in ids:
        lg = m.forward(tk)[: mf.VOCAB].astype(np.float64)
        lg[0] = -99.0
        e = np.exp(lg - lg.max())
        probs.append(float(e[want] / e.sum())); top.append(int(np.argmax(lg)) == want)
        tk = want
    m.close()
    return np.array(probs), np.array(top)


@pytest.mark.parametrize("exe", ["storygen_mi355x", "storygen_l2"])
def test_reference_storygen_runs_on_the_engine(world, exe):
    binp = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(binp):
        pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference at build time)")
    env = dict(os.environ, RWKV_SAMPLER_SEED="1")
    p = subprocess.Popen([binp], cwd=world["cwd"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    p.stdin.write((USER_LINE + "\n").encode()); p.stdin.flush()      # stdin stays open: the app blocks at its next prompt
    buf, t0 = b"", time.time()
    os.set_blocking(p.stdout.fileno(), False)
    try:
        while b"continue? (y/n):" not in buf and time.time() - t0 < 240 and p.poll() is None:
            chunk = p.stdout.read()
            if chunk:
                buf += chunk
            else:
                time.sleep(0.05)
    finally:
        p.kill(); p.wait()
    assert b"continue? (y/n):" in buf, (buf[-400:], p.stderr.read()[-400:])
    head, gen = buf.split(b"written:>", 1)
    assert b"Loaded model" in head and f"n_layers: {world['L']}".encode() in head
    mt = re.match(rb"(\d+):token", gen)
    assert mt, gen[:60]
    gen = gen[mt.end(): gen.index(b"continue? (y/n):")]
    got = decode_stream(gen)
    assert len(got) >= 150
    assert all(0 < g < mf.VOCAB for g in got)
    probs, top = replay(world, got)
    assert probs.min() > 1e-9, f"id {got[int(probs.argmin())]} at step {int(probs.argmin())} has probability {probs.min():.2e} under the engine's logits"
    assert top.mean() > 0.7, f"only {top.mean():.2f} of the sampled ids are the argmax of peaked logits"
    assert len(set(got)) > 20


def _run_interactive(binp, cwd, stdin_line, until, timeout=240):
    p = subprocess.Popen([binp], cwd=cwd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RWKV_SAMPLER_SEED="1"))
    p.stdin.write((stdin_line + "\n").encode()); p.stdin.flush()
    buf, t0 = b"", time.time()
    os.set_blocking(p.stdout.fileno(), False)
    try:
        while not until(buf) and time.time() - t0 < timeout and p.poll() is None:
            chunk = p.stdout.read()
            if chunk:
                buf += chunk
            else:
                time.sleep(0.05)
    finally:
        p.kill(); p.wait()
    return buf, p.stderr.read()


@pytest.fixture(scope="module")
def flat_world(built, tmp_path_factory):
    """vectordb / terminalchat look for ./vocab/{vocab.json,merges.txt} and ./model.bin in the working directory"""
    root = tmp_path_factory.mktemp("callers")
    write_vocab(str(root / "vocab"))
    L, D = 4, 768
    t = mf.synthetic_tensors(L, D, seed=170, head_scale=120.0)
    mf.write_bin(str(root / "model.bin"), L, D, t)
    return dict(cwd=str(root), model=str(root / "model.bin"), L=L, D=D)


@pytest.mark.parametrize("exe", ["vectordb_mi355x", "vectordb_l2"])
def test_reference_vectordb_runs_on_the_engine(flat_world, exe):
    """RUN reference examples/vectordb/vectordb.cpp (vectordb.cpp:15-59: loadFile("./model.bin", 5), emptyState(), loadContext per
    fact in 5-token GPT chunks, getSubState / setSubState, then the distances between raw state->statedd arrays) and recompute
    the distances it prints with the Python engine along the same call sequence."""
    binp = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(binp):
        pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference at build time)")
    question = "Who is Alice"
    facts = ["Alice is a person", "Nvidia is a company", "The grand canyon is a place"]     # vectordb.cpp:21-25
    buf, err = _run_interactive(binp, flat_world["cwd"], question, lambda b: b.count(b"Diff: ") >= 3 and b.rstrip().endswith(b"Question:>"))
    assert b"Loaded model" in buf and buf.count(b"Diff: ") >= 3, (buf[-600:], err[-400:])
    got = [(float(a), float(b)) for a, b in re.findall(rb"Diff: '[^']*' is ([-+0-9.eE]+|nan|inf):([-+0-9.eE]+|nan|inf)", buf)][:3]
    from rwkv_cpp_accelerated_amd import engine
    m = engine.RWKV(resident=False)
    m.loadFile(flat_world["model"], 5)
    n = flat_world["L"] * flat_world["D"]

    def load_context(text):                       # RWKV::loadContext, rwkv.h:395-413: chunks of maxContext tokens, GPT mode
        ids = encode(text)
        for i in range(0, len(ids), 5):
            m.forward(ids[i:i + 5], engine.MODE_GPT)
        dd = m.state.statedd[:n].copy()
        for a in m.state.arrays():                # Rwkv.state->setSubState(emptyState)
            a[:n] = 0
        return dd
    fact_dd = [load_context(f) for f in facts]
    q = load_context(question)
    for (gd, ge), dd in zip(got, fact_dd):
        diff = float(np.sum(np.abs(dd - q).astype(np.float32) / np.float32(flat_world["D"]), dtype=np.float32))      # float accumulation, vectordb.cpp:47-55
        eu = float(np.sqrt(np.sum(((dd - q) ** 2).astype(np.float32) / np.float32(flat_world["D"]), dtype=np.float32)))
        assert abs(gd - diff) <= 2e-3 * max(1.0, abs(diff)), (gd, diff)
        assert abs(ge - eu) <= 2e-3 * max(1.0, abs(eu)), (ge, eu)
    assert len({round(g[0], 3) for g in got}) == 3, "the three facts must be at three different distances"
    m.close()


@pytest.mark.parametrize("exe", ["terminalchat_mi355x", "terminalchat_l2"])
def test_reference_terminalchat_starts_on_the_engine(flat_world, exe):
    """reference examples/terminalchat/chat.cpp: loads ./model.bin, ingests its built-in chat record token by token, prompts,
    takes one user line and starts to answer (typical sampling; an endless loop: killed once it has printed something)"""
    binp = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(binp):
        pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference at build time)")
    buf, err = _run_interactive(binp, flat_world["cwd"], "hello", lambda b: b"User:>" in b and len(b.split(b"User:>", 1)[1]) > 96)     # >= 8 tokens of the synthetic vocabulary, whatever the pipe delivers at once
    assert b"Loaded model" in buf and b"User:>" in buf, (buf[-400:], err[-400:])
    tail = buf.split(b"User:>", 1)[1]
    assert len(decode_stream(tail)) >= 8, tail[:80]


@pytest.mark.parametrize("exe", ["two_models_mi355x", "two_models_l2"])
def test_two_models_in_one_process(built, tmp_path, exe):
    """every RWKV owns its tensors[] (rwkv.h:248,288; c_binding.cpp:28-33 hands out one per initRwkv): two models loaded side by
    side through the drop-in header (level 1) and through the reference's own header + backend TU (level 2, the engine handle
    travels in tensors[X] / tensors[STATEXY]); interleaved greedy decode, then one model is destroyed and the other goes on"""
    binp = os.path.join(ROOT, "oracle", "_ref", exe)
    if not os.path.exists(binp):
        pytest.skip(f"oracle/_ref/{exe} not built (needs /root/reference at build time)")
    from rwkv_cpp_accelerated_amd import engine
    shapes = [(2, 768, 41), (3, 1024, 42)]
    paths = []
    for L, D, seed in shapes:
        p = str(tmp_path / f"m{seed}.bin")
        mf.write_bin(p, L, D, mf.synthetic_tensors(L, D, seed=seed))
        paths.append(p)
    n = 10
    out = subprocess.run([binp, paths[0], paths[1], str(n)], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-400:] + out.stderr[-400:]
    lines = [l for l in out.stdout.splitlines() if l.strip() and not l.startswith(("n_layers", "n_embed"))]
    ia, ib, ic = ([int(x) for x in l.split()] for l in lines[-4:-1])
    assert lines[-1].split() == ["layers", "3", "1024"]
    want = []
    for (L, D, seed), p, steps in zip(shapes, paths, (n, 2 * n)):
        m = engine.RWKV(resident=True); m.loadFile(p, 1)
        tk, ids = 11, []
        for _ in range(steps):
            tk = parity.argmax_ban0(m.forward(tk)[: mf.VOCAB]); ids.append(tk)
        want.append(ids)
        m.close()
    assert ia == want[0] and ib + ic == want[1]
