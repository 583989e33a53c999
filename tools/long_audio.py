"""Long requests (many 30 s windows behind one prompt) through the engine: TINY dimensions against the oracle, FULL dimensions for
batch invariance (the long request alone vs batched with short ones).  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dataclasses import replace
from sonicscribe_amd import spec, synth, frontend
from sonicscribe_amd.engine import Engine
from oracle import oracle as orc

def windows_of(pcm, d):
    return [pcm[s:e] for s, e in frontend.split_windows(len(pcm), d)]

def prompt_for(d, n):
    n_audio, _ = frontend.request_audio_tokens(n, d)
    return [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 9]

# --- TINY vs oracle: 305 s = 11 windows
d = replace(spec.TINY, eos_ids=())
e = Engine(d, 0, max_batch=32, max_ctx=8192)
e.load_synthetic(5)
om = orc.Model(d, synth.synth_state_dict(d, 5, bf16=True), bf16=True)
for seconds in (65.0, 305.0, 655.0):
    n = int(seconds * 16000)
    pcm = frontend.normalise_to_int16(synth.synth_pcm(9, n).astype(np.float32) / 32768.0)
    wins = windows_of(pcm, d)
    prompt = prompt_for(d, n)
    t0 = time.time()
    ids, logits = e.transcribe_batch(wins, [prompt], [6], req_win=[0, len(wins)], want_logits=True)
    t1 = time.time()
    fm = [orc.logmel(w) for w in wins]
    feats = np.stack([f for f, _ in fm]); nv = [int(m.sum()) for _, m in fm]
    ref = om.transcribe(feats, nv, prompt, 6)
    dl = np.abs(logits[:, 0] - ref["step_logits"]).max()
    print(f"TINY {seconds:.0f} s: {len(wins)} windows, prompt {len(prompt)} tokens, engine {1e3 * (t1 - t0):.0f} ms, ids {ids[0].tolist()} oracle {ref['new_ids'].tolist()}, max|dlogit| {dl:.4f}", flush=True)
e.close()

# --- FULL dimensions: a 185 s request (7 windows, 2300 prompt tokens) alone and batched with two short ones
d = replace(spec.FULL, eos_ids=())
e = Engine(d, 0, max_batch=16, max_ctx=4096)
e.load_synthetic(20260128)
n_long = 185 * 16000
long_pcm = frontend.normalise_to_int16(synth.synth_pcm(3, n_long).astype(np.float32) / 32768.0)
shorts = [frontend.normalise_to_int16(synth.synth_pcm(40 + i, 16000 * (5 + 15 * i)).astype(np.float32) / 32768.0) for i in range(2)]
wl = windows_of(long_pcm, d)
t0 = time.time()
ids_a, lg_a = e.transcribe_batch(wl, [prompt_for(d, n_long)], [8], req_win=[0, len(wl)], want_logits=True)
t1 = time.time()
segs = windows_of(shorts[0], d) + wl + windows_of(shorts[1], d)
ids_b, lg_b = e.transcribe_batch(segs, [prompt_for(d, len(shorts[0])), prompt_for(d, n_long), prompt_for(d, len(shorts[1]))], [8, 8, 8],
                                 req_win=[0, 1, 1 + len(wl), 2 + len(wl)], want_logits=True)
print(f"FULL 185 s: {len(wl)} windows, prompt {len(prompt_for(d, n_long))} tokens, alone {1e3 * (t1 - t0):.0f} ms; ids alone {ids_a[0].tolist()} batched {ids_b[1].tolist()}; "
      f"max|dlogit| alone vs batched {np.abs(lg_a[:, 0] - lg_b[:, 1]).max():.4f}", flush=True)
e.close()
