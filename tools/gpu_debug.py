import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
g = np.load("tests/golden/tiny_bf16.npz")
eng = Engine(spec.TINY, 0, max_batch=16, max_ctx=512); eng.load_synthetic(20260128)
n_new = int(g["n_new"])
segs, prompts = [], []
for si in range(2):
    p = f"s{si}_"
    segs.append(synth.synth_pcm(int(g[p + "seg_index"]), int(g[p + "n_samples"]))); prompts.append(g[p + "prompt_ids"])
from oracle import oracle
for si in range(2):
    feats, mask = oracle.logmel(segs[si])
    eng.encode(feats[None], [int(mask.sum())], want_layers=True, want_enc_out=True)
ids, logits = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
ok = True
for si in range(2):
    p = f"s{si}_"
    ref = g[p + "step_logits"]
    d = np.abs(logits[:len(ref), si] - ref).max(axis=1)
    if d.max() > 0.0625:
        ok = False
        dshift = np.abs(logits[1:len(ref), si] - ref[:-1]).max(axis=1)
        dshift2 = np.abs(logits[:len(ref)-1, si] - ref[1:]).max(axis=1)
        print("seg", si, "BAD steps", np.where(d > 0.0625)[0].tolist(), "ids eq", np.array_equal(ids[si], g[p+"new_ids"]))
        print("   vs shifted(+1):", np.round(dshift, 3).tolist())
        print("   vs shifted(-1):", np.round(dshift2, 3).tolist())
        print("   row0 of dump sum", float(np.abs(logits[0, si]).sum()), "last row sum", float(np.abs(logits[len(ref)-1, si]).sum()))
if not ok:
    np.savez(f"gpurun_out/flake_{os.getpid()}.npz", logits=logits, ids0=ids[0], ids1=ids[1])
    ids2, logits2 = eng.transcribe_batch(segs, prompts, [n_new, n_new], want_logits=True)
    print("  rerun in same process: identical to failing run?", np.array_equal(logits, logits2), "max diff", np.abs(logits - logits2).max())
print("OK" if ok else "FAILED")
