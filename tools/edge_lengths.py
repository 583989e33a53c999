import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from sonicscribe_amd import spec, synth, frontend
from sonicscribe_amd.asr import ASRModel
from oracle import oracle as orc
d = spec.TINY
m = ASRModel.from_synthetic(d, max_batch=8, max_ctx=512)
om = orc.Model(d, synth.synth_state_dict(d, 20260128, bf16=True), bf16=True)
for n in (0, 1, 159, 160, 161, 399, 400, 401, 640, 1279, 1280, 480000, 480001):
    wav = synth.synth_pcm(7, max(n, 1))[:n].astype(np.float32) / 32768.0
    try:
        t = m.transcribe(wav[None], 16000, max_new_tokens=5)
        pcm = frontend.normalise_to_int16(wav)
        wins = frontend.split_windows(len(pcm), d)
        n_audio, per = frontend.request_audio_tokens(len(pcm), d)
        prompt = m.prompt.build(frontend.build_instruction(None), n_audio)
        if len(wins) == 1:
            feats, mask = orc.logmel(pcm)
            r = om.transcribe(feats, int(mask.sum()), prompt, 5)
            ref = m.prompt.decode(r["new_ids"]).strip()
        else:
            ref = "(multi-window)"
        print(n, "n_audio", n_audio, "->", repr(t), "oracle", repr(ref), "OK" if (ref == t or ref.startswith("(")) else "DIFF")
    except Exception as ex:
        print(n, "EXC", type(ex).__name__, str(ex)[:200])
m.close()
