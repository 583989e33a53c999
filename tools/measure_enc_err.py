"""Error distribution of encoder activations (HIP vs bf16 oracle) in units of the bf16 ulp of the reference value: picks the bounds of
tests/test_gpu_parity.py::test_encoder_vs_oracle_and_golden and test_fullwidth_layer_vs_oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
from oracle import oracle as orc

def ulps(got, ref):
    u = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -8))) - 7)      # bf16 ulp of the reference (floor at |x| = 2^-8)
    return np.abs(got - ref) / u

for name, d, seed, layers in (("TINY", spec.TINY, 20260128, 2), ("FULLWIDTH-1", replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992)), 7, 1)):
    e = Engine(d, 0, max_batch=2, max_ctx=512); e.load_synthetic(seed)
    st = {n: orc.synth_fill(seed, n, int(np.prod(s)), *synth.kind_params(k, s), True).reshape(s) for n, s, k in spec.tensor_inventory(d)}
    om = orc.Model(d, st, bf16=True)
    pcm = synth.synth_pcm(11, 320000)
    feats, mask = orc.logmel(pcm)
    emb, n_a, lay, enc_out = e.encode(feats[None], [int(mask.sum())], want_layers=True, want_enc_out=True)
    n_audio = spec.audio_token_count(int(mask.sum()))
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    r = om.transcribe(feats, int(mask.sum()), prompt, 1, want=("enc_layers", "enc_out"))
    for tag, g, rf in [(f"layer{l}", lay[0, l], r["enc_layers"][l]) for l in range(layers)] + [("enc_out", enc_out[0], r["enc_out"]), ("embeds", emb[0, :n_audio], r["audio_embeds"][:n_audio])]:
        u = ulps(g, rf)
        print(f"{name:12s} {tag:8s} |ref| max {np.abs(rf).max():7.2f}  exact {np.mean(u == 0):.4f}  <=1ulp {np.mean(u <= 1):.5f}  <=2 {np.mean(u <= 2):.6f}  <=4 {np.mean(u <= 4):.7f}  max ulp {u.max():.1f}  max abs {np.abs(g - rf).max():.4f} mean abs {np.abs(g - rf).mean():.5f}")
    e.close()
