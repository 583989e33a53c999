#!/usr/bin/env python3
"""Headline benchmark: 20 s-segments/sec (+ RTF) of the SonicScribe hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (log-mel -> encoder -> projector -> prefill -> 150 greedy tokens) over one batch of 32
synthetic 20 s segments per GPU (BASELINE.json configs[1]; GLM-ASR-Nano dimensions, bf16, portable-PRNG weights -- no
checkpoint exists offline).  PCM is HBM-resident before the timed region.  Segments are sharded across ranks (one engine
replica per GPU, no data-path collective): weak scaling, value = all ranks' segments / max-over-ranks time.

Two timed legs per run, both K steps bracketed by barrier + synchronize:
  single_batch  one batch of 32 at a time (rounds 1-3's headline definition, kept so rounds stay comparable); stages_ms_per_step and
                roofline are measured here, where the decode kernels have the GPU to themselves
  value         --slots S (default 2) batches of 32 in flight on ONE engine = ONE weight copy (sonic_slot_create: per-slot stream, buffers,
                KV cache, graphs); the K batches go round-robin to the slots through sonic_run_staged_async / sonic_wait, every batch a
                full 32 x 20 s x 150-token run; ms_per_step = wall / K.  --slots 1: value = single_batch.

The JSON line also carries
  roofline      the time-dominant part of a step, the greedy decode loop (HBM-bound: decoder weights + tied lm_head + KV cache once per
                token step): algorithmic bytes per token step / its average duration, measured with HIP events on the engine's stream
                around the decode loop of every timed step; traffic_from_profile = the committed PMC passes (profiles/, with their commit)
  encoder_gemms / encoder_fc1_gemm / mel_frontend   the MFMA- and HBM-side figures SURVEY.md 8d names
  pcie_inclusive          the same batch through the one-call boundary (host PCM in, host ids out); never the headline
  single_5s / single_20s  BASELINE config 1's call shape: wall latency of ONE B=1 ASRModel.transcribe() call (host float tensor in, string out),
                p50 of 20 calls; cpu_baseline carries the same 5 s segment through the reference's CPU arithmetic
  int8_b64 / bf16_b64     BASELINE config 4 (INT8 weight path, batch 64) and the bf16 figure at the same batch, same process, same box
  streaming               BASELINE config 5's call pattern at its per-GPU share (16 sessions), real-time schedule: partial / final latency
  cpu_baseline  the reference's DEVICE=cpu arithmetic (transformers GlmAsrForConditionalGeneration.generate, bf16, B=1, full depth,
                150 tokens) timed on the host cores, rank 0 at N=1 only, at threads = min(asr.py:96-101's rule, the job's cgroup CPU quota):
                `cores` = those threads, `quota_cpus` / `visible_cpus` say what the host gave; `threads64` = the same pass at 64 threads
                (rounds 1-4's figure, oversubscribed inside the quota).  The worker process is spawned before this process
                initialises the GPU and stays idle until the GPU legs are done (2 warm-ups + up to 3 full passes then)
(--no-extras skips int8_b64 / bf16_b64 / streaming / pcie_inclusive; a default run takes about five minutes, most of it the CPU passes.)

  python bench.py --mode int8 --batch 64     BASELINE config 4 alone
  python bench.py --streaming                BASELINE config 5's call pattern (sessions x partial / final decodes through the coalescer)
  torchrun ... bench.py --gpus N [--dist-backend gloo --share-gpu]    N ranks (gloo + shared device: the N-rank path on fewer GPUs)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from dataclasses import replace

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything starts the HIP runtime (sonicscribe_amd/__init__.py says why)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEG_SECONDS = 20
BATCH = 32
MAX_NEW = 150           # min(50 + 5*20, 200), transcription_manager.py:37
SINGLE5_NEW = 75        # the same rule for a 5 s final (BASELINE config 1's segment)
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
DECODE_CHUNK_STEPS = 2      # token steps per captured decode graph (engine default, option decode_chunk)


def segment_flops(d, prompt_len: int) -> dict:
    """Algorithmic FLOPs of one segment: conv stem + encoder layers + projector (`encoder`, with its GEMM / attention split) and the prompt
    forward of the decoder (`prefill`: every linear over prompt_len tokens, causal attention, lm_head on the last position)."""
    T2, T = d.n_frames, d.enc_T
    conv = 2.0 * T2 * d.enc_d * 3 * d.n_mels + 2.0 * T * d.enc_d * 3 * d.enc_d
    qkvo = 4 * 2.0 * T * d.enc_d * d.enc_d
    att = 2 * 2.0 * T * T * d.enc_d
    mlp = 2 * 2.0 * T * d.enc_d * d.enc_ff
    ta = T // d.merge
    proj = ta * 2.0 * (d.proj_in * d.proj_mid + d.proj_mid * d.dec_d)      # linear_1: merge * enc_d -> 2 * dec_d, linear_2 -> dec_d
    enc_gemm = conv + d.enc_layers * (qkvo + mlp) + proj
    enc_att = d.enc_layers * att
    qd, kvd = d.dec_heads * d.dec_head_dim, d.dec_kv_heads * d.dec_head_dim
    lin = d.dec_layers * (d.dec_d * (qd + 2 * kvd) + qd * d.dec_d + 3 * d.dec_d * d.dec_ff)
    pre = 2.0 * prompt_len * lin + d.dec_layers * 2 * 2.0 * (prompt_len * (prompt_len + 1) / 2) * qd + 2.0 * d.vocab * d.dec_d
    return {"encoder": enc_gemm + enc_att, "encoder_gemm": enc_gemm, "encoder_attention": enc_att, "prefill": pre}


def host_cpu_quota():
    """CPUs this job may actually use: the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) and the scheduler affinity,
    next to the count the OS shows.  The GPU boxes show 256 CPUs and give a job 16 of quota: threads beyond the quota are throttled, not run."""
    import multiprocessing
    visible = multiprocessing.cpu_count()
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = min(affinity, visible)
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    return {"visible_cpus": visible, "affinity_cpus": affinity, "quota_cpus": quota, "usable_cpus": usable}


def _cpu_reference_worker(threads: int, n_timed: int, budget_s: float, wait_go: bool = False, threads2: int = 0):
    """Runs in a subprocess (so a slow host cannot stall the bench): times the reference's DEVICE=cpu arithmetic -- third-party torch +
    transformers, exactly what backend/asr.py drives (processor features -> model.generate(do_sample=False), asr.py:393-422) -- on one
    synthetic 20 s segment at FULL depth (32 encoder + 28 decoder layers, vocabulary 59264), B=1, bf16, greedy, 150 new tokens.
    Two warm-up passes (8 tokens each: oneDNN primitive creation, page faults), then up to `n_timed` full passes while the budget lasts;
    the reported figure is their median.  Nothing is extrapolated.  With wait_go the worker does NOTHING (torch is not even imported)
    until the parent writes a line to stdin: it is spawned before the parent touches the GPU, and the host must be quiet while the GPU
    legs are timed (a worker that built its 2.1 B-parameter model meanwhile cost the headline 30 %: the decode loop is one host-launched
    graph replay per token)."""
    if wait_go:                                                # idle (not even torch imported) until the parent's GPU legs are done
        print("READY", flush=True)
        if sys.stdin.readline().strip() != "go":               # EOF: the parent died before collect() - do not burn 64 threads on an orphan
            return
    import torch
    from sonicscribe_amd import spec, synth
    torch.set_num_threads(threads)
    try:
        torch.set_num_interop_threads(1)                       # asr.py:104
    except RuntimeError:
        pass
    t_start = time.perf_counter()
    from transformers import GlmAsrConfig, GlmAsrForConditionalGeneration, WhisperFeatureExtractor
    try:
        from transformers.initialization import no_init_weights      # transformers >= 5
    except ImportError:
        from transformers.modeling_utils import no_init_weights
    fe = WhisperFeatureExtractor(feature_size=128)
    pcm = synth.synth_pcm(0, SEG_SECONDS * 16000)
    wav = pcm.astype(np.float32) / 32768.0
    n_audio = spec.audio_token_count(spec.valid_frames(len(pcm)))
    cfg = GlmAsrConfig()
    cfg.text_config.eos_token_id = None
    with no_init_weights():                                     # skip HF's slow fp32 random init of 2.1 B parameters ...
        model = GlmAsrForConditionalGeneration(cfg).to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():                                       # ... and fill them with small finite values instead (timing is value-independent,
        for name, prm in model.named_parameters():              # but uninitialised memory can hold NaNs / denormals, which is not)
            prm.uniform_(-0.02, 0.02, generator=g)
            if "norm" in name and name.endswith("weight"):
                prm.add_(1.0)
    model.generation_config.eos_token_id = None
    ids = torch.tensor([[1, 17, 23, 5] + [cfg.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]])

    def run(n_new, wav=wav, ids=ids):
        t0 = time.perf_counter()
        f = fe([wav], sampling_rate=16000, return_attention_mask=True, padding="max_length", return_tensors="pt")
        with torch.no_grad():
            out = model.generate(input_ids=ids, input_features=f["input_features"].to(torch.bfloat16), input_features_mask=f["attention_mask"],
                                 attention_mask=torch.ones_like(ids), max_new_tokens=n_new, min_new_tokens=n_new, do_sample=False)
        assert out.shape[1] == ids.shape[1] + n_new
        return time.perf_counter() - t0

    # BASELINE config 1's shape: one 5 s segment, B=1 (transcription_manager.py:37 gives a 5 s final min(50 + 5*5, 200) = 75 tokens)
    pcm5 = synth.synth_pcm(0, 5 * 16000)
    wav5 = pcm5.astype(np.float32) / 32768.0
    ids5 = torch.tensor([[1, 17, 23, 5] + [cfg.audio_token_id] * spec.audio_token_count(spec.valid_frames(len(pcm5))) + [7, 301, 302, 303, 9, 11]])

    build_s = time.perf_counter() - t_start
    t_go = time.perf_counter()
    warm = [run(8), run(8)]
    runs = []
    for _ in range(n_timed):
        if runs and (time.perf_counter() - t_go) + 1.15 * max(runs) > budget_s:
            break
        runs.append(run(MAX_NEW))
    runs5 = []
    for _ in range(2):
        if runs5 and (time.perf_counter() - t_go) + 1.15 * max(runs5) > budget_s + 45.0:
            break
        runs5.append(run(SINGLE5_NEW, wav5, ids5))
    # second figure: the same pass at `threads2` threads (rounds 1-4 quoted 64 threads whatever the quota), one warm-up + one pass, only if there is time
    runs2 = []
    if threads2 and threads2 != threads and (time.perf_counter() - t_go) + 2.5 * max(runs) < budget_s + 60.0:
        torch.set_num_threads(threads2)
        run(8)
        runs2.append(run(MAX_NEW))
    print(json.dumps({"threads": threads, "runs_s": runs, "median_s": float(np.median(runs)), "warmup_8tok_s": warm, "build_s": build_s,
                      "single_5s_runs_s": runs5, "threads2": threads2, "runs2_s": runs2}), flush=True)


class CpuBaseline:
    """cpu_baseline of the bench line: the reference CPU path at full depth.  start() spawns the worker (a child process: model build
    only) BEFORE the parent initialises the GPU; collect() lets it time 2 warm-ups + up to 5 full passes and reads the result.
    Threads: min(asr.py:96-101's rule, the CPUs the job can really use).  The rule takes all visible cores minus two; the GPU boxes show 256
    CPUs and give the job a cgroup quota of 16, so threads beyond the quota only take turns (rounds 1-4 ran 64 threads inside that quota:
    4x oversubscribed and throttled, 31-37 s per segment).  `cores` of the line = the threads of the timed passes = CPUs actually running;
    the 64-thread figure is kept beside it (`threads64`) when the budget allows."""

    def __init__(self, budget_s: float = 250.0, n_timed: int = 3):
        self.host = host_cpu_quota()
        self.cores = self.host["visible_cpus"]
        self.rule = max(1, self.cores - 2) if self.cores > 4 else self.cores
        self.threads = max(1, min(self.rule, self.host["usable_cpus"]))
        self.threads2 = 64 if (self.cores >= 64 and self.threads != 64) else 0
        self.budget_s, self.n_timed, self.proc = budget_s, n_timed, None

    def start(self):
        import atexit
        import subprocess
        atexit.register(self.kill)
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-threads", str(self.threads),
                                      "--cpu-timed", str(self.n_timed), "--cpu-budget", str(self.budget_s - 70.0), "--cpu-wait-go", "--cpu-threads2", str(self.threads2)],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def kill(self):
        if self.proc is not None and self.proc.poll() is None:
            self.proc.kill()

    def collect(self):
        import subprocess
        res = {}
        try:
            self.proc.stdin.write("go\n"); self.proc.stdin.flush()
            out, err = self.proc.communicate(timeout=self.budget_s + 120.0)
            line = next((l for l in reversed(out.strip().splitlines()) if l.startswith("{")), None)
            res = json.loads(line) if line else {"error": err[-300:]}
        except subprocess.TimeoutExpired:
            self.proc.kill()
            res = {"error": f"did not finish within {self.budget_s + 120:.0f} s"}
        except Exception as ex:
            res = {"error": repr(ex)}
        ok = "median_s" in res
        desc = (f"1 synthetic 20 s segment, B=1, bf16, FULL depth (32+28 layers, vocab 59264), {MAX_NEW} greedy tokens through torch-CPU + transformers "
                f"generate() = the reference's DEVICE=cpu arithmetic (asr.py:393-422); worker spawned before the GPU was initialised, timed after the GPU legs; "
                f"2 warm-ups (8 tokens) then the median of the timed full passes; {self.cores} host CPUs visible, cgroup quota {self.host['quota_cpus']}, affinity "
                f"{self.host['affinity_cpus']}; {self.threads} threads = min(asr.py:96-101's rule = {self.rule}, usable CPUs = {self.host['usable_cpus']}): ")
        desc += (f"runs {[round(x, 2) for x in res['runs_s']]} s, median {res['median_s']:.2f} s/segment (RTF {res['median_s'] / SEG_SECONDS:.2f}); " if ok
                 else f"no result ({res.get('error', '?')}); ")
        r5 = res.get("single_5s_runs_s") or []
        r2 = res.get("runs2_s") or []
        return {"value": (1.0 / res["median_s"]) if ok else None, "unit": "20s-segments/sec", "cores": self.threads, "kind": "reference",
                "threads": self.threads, "quota_cpus": self.host["quota_cpus"], "visible_cpus": self.cores, "affinity_cpus": self.host["affinity_cpus"],
                "reference_rule_threads": self.rule,
                "threads64": ({"threads": res.get("threads2"), "seconds_per_segment": r2[0], "value": 1.0 / r2[0],
                               "note": "the same pass with 64 threads inside the same quota (what rounds 1-4 reported): oversubscribed"} if r2 else None),
                "passes": len(res.get("runs_s", [])), "sample": desc + "random weights",
                "single_5s": {"latency_s": (float(np.median(r5)) if r5 else None), "runs_s": r5, "max_new_tokens": SINGLE5_NEW,
                              "note": "BASELINE config 1: one 5 s segment, B=1, the same CPU arithmetic and threads (the GPU figure is the line's single_5s)"}}


def extra_batch_run(dims, device_index: int, mode: str, B: int, max_new: int, steps: int = 2, slots: int = 1):
    """An extra object of the N=1 line: one engine of `mode` at batch B, PCM staged before the clock.  `value`: `slots` batches in flight on the one
    weight copy (round-robin through sonic_run_staged_async / sonic_wait, 3 * steps batches), `single_batch`: one at a time (1 warm-up + `steps`
    timed steps; stages and roofline are its device times)."""
    from sonicscribe_amd import spec, synth
    from sonicscribe_amd.engine import MODE_INT8, MODE_NATIVE, Engine
    e = Engine(dims, device_index, MODE_INT8 if mode == "int8" else MODE_NATIVE, max_batch=B, max_ctx=512)
    try:
        e.load_synthetic(20260128)
        n_samples = SEG_SECONDS * 16000
        prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]
        segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
        engines = [e] + [e.slot() for _ in range(max(1, slots) - 1)]
        for en in engines:
            en.stage_pcm(segs)
            en.run_staged([prompt] * B, [max_new] * B)
        stage = {"mel_ms": 0.0, "encoder_ms": 0.0, "prefill_ms": 0.0, "decode_ms": 0.0}
        t0 = time.perf_counter()
        for _ in range(steps):
            e.rerun_staged()
            t = e.timings()
            for k in stage:
                stage[k] += t[k]
        dt = time.perf_counter() - t0
        value_single = B * steps / dt
        value, ms = value_single, dt / steps * 1e3
        n_multi = 0
        if len(engines) > 1:
            S, n_multi = len(engines), 3 * steps

            def pipeline(nb):
                started = 0
                for k in range(min(S, nb)):
                    engines[k].run_staged_async(); started += 1
                for j in range(nb):
                    engines[j % S].wait()
                    if started < nb:
                        engines[started % S].run_staged_async(); started += 1
            pipeline(S)
            t1 = time.perf_counter()
            pipeline(n_multi)
            d1 = time.perf_counter() - t1
            value, ms = B * n_multi / d1, d1 / n_multi * 1e3
        d_ = dims
        qd, kvd = d_.dec_heads * d_.dec_head_dim, d_.dec_kv_heads * d_.dec_head_dim
        wb = 1 if mode == "int8" else 2
        w_bytes = wb * d_.dec_layers * (d_.dec_d * (qd + 2 * kvd) + qd * d_.dec_d + 3 * d_.dec_d * d_.dec_ff) + 2 * d_.vocab * d_.dec_d
        n_dec = max(1, max_new - 1)
        kv_bytes = B * d_.dec_layers * 2 * kvd * 2 * (len(prompt) + (n_dec + 1) / 2.0)
        dec_ms = stage["decode_ms"] / steps / n_dec
        gbs = (w_bytes + kv_bytes) / (dec_ms * 1e-3) / 1e9
        return {"value": value, "unit": "20s-segments/sec", "ms_per_step": ms, "batch": B, "mode": mode, "batches_in_flight": len(engines), "steps": n_multi or steps,
                "single_batch": {"value": value_single, "ms_per_step": dt / steps * 1e3, "steps": steps},
                "dtype": "int8" if mode == "int8" else "bf16", "stages_ms_per_step": {k: v / steps for k, v in stage.items()},
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "avg_launch_ms": dec_ms,
                             "bytes_per_launch": w_bytes + kv_bytes, "kernel": "decode token step", "measured_in": "single_batch"},
                "weights_mb": e.weight_bytes() / 2 ** 20,
                "parity": ("int8 = LLM.int8() restated from the published algorithm; bitsandbytes is absent offline: parity unpinned (DESIGN.md 2)" if mode == "int8" else None)}
    finally:
        e.close()


def facade_bulk_measure(dims, device_index: int, B: int, max_new: int, native_dispatch=None, batches: int = 20):
    """The facade's bulk shape: ASRModel.submit() x (batches x B) host float tensors of 20 s - host-side normalisation, H2D, three decode loops + a prefill
    slot behind the row-level dispatcher (native threads in the library by default; native_dispatch=False: the Python class), detokenised strings back."""
    from sonicscribe_amd import synth as _synth
    from sonicscribe_amd.asr import ASRModel
    fm = ASRModel.from_synthetic(dims, seed=20260128, device=f"cuda:{device_index}", mode="native", max_batch=64, max_ctx=512, slots=4, continuous=True, decoders=3,
                                 native_dispatch=native_dispatch)
    wavs = [(_synth.synth_pcm(i, SEG_SECONDS * 16000).astype(np.float32) / np.float32(32768.0))[None] for i in range(B)]
    [f.result() for f in [fm.submit(w, 16000, max_new) for w in wavs]]                  # warm-up
    t1 = time.perf_counter()
    futs = [fm.submit(wavs[i % B], 16000, max_new) for i in range(batches * B)]
    [f.result() for f in futs]
    d1 = time.perf_counter() - t1
    kind = type(fm._dispatcher.replicas[0]).__name__
    fm.close()
    return {"value": batches * B / d1, "unit": "20s-segments/sec", "segments": batches * B, "wall_s": d1, "dispatcher": kind,
            "note": f"ASRModel(max_batch=64, slots=4, continuous=True, decoders=3).submit() x {batches * B} host float tensors of 20 s, {max_new} tokens each: "
                    "host-side normalisation, H2D, the three decode loops + prefill slot behind the row-level dispatcher (rows join and leave one by one; "
                    "csrc/dispatch.cpp native threads unless dispatcher says _ContinuousReplica), detokenised strings back; seven batches are in flight, so the "
                    "figure still contains the fill and drain of a 4 s run; not the headline"}


def run_streaming(a):
    print(json.dumps(streaming_measure(a)), flush=True)


def streaming_measure(a):
    """BASELINE config 5's call pattern against one process: S concurrent sessions, each speaking for 20 s (64 ms chunks), a partial
    decode of the last 20 chunks (1.28 s, 15 tokens: audio_manager.py:106-114, transcription_manager.py:19-28) every second while
    speaking (connection_manager.py:89-92, config.py:40) and a final decode of the whole 20 s segment (150 tokens,
    transcription_manager.py:30-41) 1.28 s after the speech ends (two silent VAD windows).  Sessions start staggered over one second.
    Events are issued in REAL TIME through ASRModel.submit() - the non-blocking entry the asyncio callers await - and every
    request's latency is submit -> result.  One utterance cycle per session (about 22.5 s of wall clock)."""
    from sonicscribe_amd import spec, synth
    from sonicscribe_amd.asr import ASRModel
    base = spec.FULL if a.dims == "full" else spec.TINY
    dims = replace(base, eos_ids=())
    S = a.sessions
    dev = "cuda:*" if a.gpus > 1 else "cuda:" + ",".join(["0"] * max(1, a.replicas_per_gpu))     # several replicas on one GPU fill each other's decode bubbles
    model = ASRModel.from_synthetic(dims, seed=20260128, device=dev, mode=a.mode, max_batch=a.batch, max_ctx=512, slots=getattr(a, "slots", 2),
                                    continuous=getattr(a, "continuous", False), decoders=getattr(a, "decoders", 1), native_dispatch=(False if getattr(a, "python_dispatch", False) else None), _options=dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in getattr(a, "opt", []) or []))
    n_rep = len(model.models)
    speech = SEG_SECONDS * 16000
    wire = [synth.synth_pcm(i, speech) for i in range(S)]                          # int16, as the WebSocket delivers it
    pcm = [w.astype(np.float32) / np.float32(32768.0) for w in wire]
    ring = a.ingest == "ring"
    CH = 1024                                                                      # samples per 64 ms chunk (config.py:23-24)
    n_chunks = speech // CH
    streams = [model.open_stream(f"client-{i}") for i in range(S)] if ring else []
    # warm-up: graphs of the batch sizes the run will see, both step classes
    for n in (1, 2, min(S, a.batch)):
        [f.result() for f in [model.submit(pcm[i % S][None, :20480], 16000, 15) for i in range(n)]]
        [f.result() for f in [model.submit(pcm[i % S][None], 16000, MAX_NEW) for i in range(n)]]
    single = {}
    if getattr(a, "single", False):
        # BASELINE config 1's call shape (the reference is B=1 per call: transcription_manager.py:58-62, main.py:616-624): ONE
        # ASRModel.transcribe() at a time, host float tensor in, transcript string out, nothing else on the GPU; p50 of 20 calls
        for key, secs, mn in (("single_5s", 5, SINGLE5_NEW), ("single_20s", SEG_SECONDS, MAX_NEW)):
            x = (synth.synth_pcm(500, secs * 16000).astype(np.float32) / np.float32(32768.0))[None]
            model.transcribe(x, 16000, mn)
            ts = []
            for _ in range(20):
                t1 = time.perf_counter(); txt = model.transcribe(x, 16000, mn); ts.append(time.perf_counter() - t1)
                assert isinstance(txt, str)
            single[key] = {"latency_ms": {"p50": float(np.percentile(ts, 50)) * 1e3, "p99": float(np.percentile(ts, 99)) * 1e3, "min": min(ts) * 1e3}, "calls": 20,
                           "audio_s": secs, "max_new_tokens": mn, "rtf": float(np.percentile(ts, 50)) / secs,
                           "note": "one B=1 ASRModel.transcribe(tensor[1, N], 16000, max_new_tokens) call at a time: peak-normalise + PCM_16 on the host, H2D, "
                                   "log-mel, encoder, prefill, greedy decode (EOS disabled: the full budget), D2H, detokenise; wall clock around the call"}
    events = []
    for s_ in range(S):
        off = s_ / S
        if ring:                                                                    # every 64 ms chunk arrives at its own time
            for j in range(n_chunks):
                events.append((off + (j + 1) * CH / 16000.0, s_, "chunk", j))
        for k in range(1, SEG_SECONDS + 1):
            events.append((off + k + 1e-4, s_, "partial", k))
        events.append((off + SEG_SECONDS + 1.28, s_, "final", 0))
    events.sort()
    lat = {"partial": [], "final": []}
    pending = []
    append_s, n_app, late = 0.0, 0, 0.0
    t0 = time.perf_counter()
    for t_ev, s_, kind, k in events:
        now = time.perf_counter() - t0
        if t_ev > now:
            time.sleep(t_ev - now)
        else:
            late = max(late, now - t_ev)
        if kind == "chunk":
            ta = time.perf_counter()
            streams[s_].add_audio_chunk(wire[s_][k * CH:(k + 1) * CH].tobytes())
            append_s += time.perf_counter() - ta; n_app += 1
            continue
        if ring:                                                                    # decode a chunk range that is already on the device
            last = streams[s_].next_chunk_id - 1
            ts = time.perf_counter()
            fut = streams[s_].submit_chunks(max(0, last - 19), last, 15) if kind == "partial" else streams[s_].submit_chunks(0, last, MAX_NEW)
            fut.add_done_callback(lambda f, ts=ts, kind=kind: lat[kind].append(time.perf_counter() - ts))
            pending.append(fut)
            continue
        if kind == "partial":
            end = min(speech, k * 16000)
            audio = pcm[s_][None, max(0, end - 20480):end]
            mn = 15
        else:
            audio, mn = pcm[s_][None], MAX_NEW
        ts = time.perf_counter()
        fut = model.submit(audio, 16000, mn, session=f"client-{s_}")
        fut.add_done_callback(lambda f, ts=ts, kind=kind: lat[kind].append(time.perf_counter() - ts))
        pending.append(fut)
    for f in pending:
        f.result()
    wall = time.perf_counter() - t0
    batches = [r.batches for r in model._dispatcher.replicas]
    for st in streams:
        st.close()
    model.close()

    def pct(v, q):
        return float(np.percentile(np.asarray(v) * 1e3, q)) if v else None
    out = {
        "metric": f"streaming: {S} concurrent sessions on {n_rep} MI355X (64 ms chunks, 1 s partials of 1.28 s / 15 tokens, 20 s finals / {MAX_NEW} tokens), {a.mode}",
        "value": len(lat["final"]) / wall, "unit": "finals/sec (real-time schedule)", "n_gpus": n_rep, "higher_is_better": True, "data": "synthetic",
        "dtype": "int8" if a.mode == "int8" else "bf16", "sessions": S, "wall_s": wall,
        "partial_latency_ms": {"p50": pct(lat["partial"], 50), "p99": pct(lat["partial"], 99), "max": pct(lat["partial"], 100), "n": len(lat["partial"])},
        "final_latency_ms": {"p50": pct(lat["final"], 50), "p99": pct(lat["final"], 99), "max": pct(lat["final"], 100), "n": len(lat["final"])},
        "device_batches_per_replica": batches, "slots": model.slots, "continuous": model.continuous, **single,
        "ingest": {"kind": a.ingest, "appends": n_app, "mean_append_us": (append_s / n_app * 1e6) if n_app else None, "max_event_lateness_ms": late * 1e3},
        "config": {"workload": "BASELINE config 5 call pattern, one process, requests through ASRModel.submit() (dispatch.Dispatcher: no linger, "
                               "step-class buckets, session -> replica)" + ("; every 64 ms wire chunk appended to the session's device ring as it "
                               "arrives (AudioStream.add_audio_chunk), decodes name chunk ranges" if ring else ""), "sessions": S, "replicas": n_rep},
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed batches per leg (the bulk pipeline is timed from empty to drained: fewer than ~10 batches mostly measure its fill and drain)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--max-new", type=int, default=MAX_NEW)
    ap.add_argument("--dims", default="full", choices=["full", "tiny"])
    ap.add_argument("--mode", default="native", choices=["native", "int8"], help="native = bf16 (BASELINE config 2); int8 = the repo's quantised option (config 4, use --batch 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-chains", action="store_true", help=argparse.SUPPRESS)     # (round 2-3 flag, accepted and ignored: slots replaced the second engine)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="engine tuning knob for experiments (sonic_set_option)")
    ap.add_argument("--streaming", action="store_true", help="BASELINE config 5: real-time session simulation (partial / final latency), one process")
    ap.add_argument("--sessions", type=int, default=16, help="concurrent sessions of --streaming (128 sessions / 8 GPUs = 16 per GPU)")
    ap.add_argument("--replicas-per-gpu", type=int, default=1, help="--streaming on one GPU: engine replicas sharing it (DESIGN.md 4: concurrent decode chains)")
    ap.add_argument("--python-dispatch", action="store_true", help="row-level scheduling by the Python class (dispatch._ContinuousReplica) instead of the library's native threads (csrc/dispatch.cpp): A/B")
    ap.add_argument("--facade-only", action="store_true", help="only the facade_bulk leg (ASRModel.submit x 640 segments): A/B of the dispatchers")
    ap.add_argument("--continuous", action="store_true", help="--streaming: row-level scheduling (the engine decodes forever over its rows, slots prefill; dispatch._ContinuousReplica)")
    ap.add_argument("--decoders", type=int, default=1, help="--streaming --continuous: decoding handles per replica (each loops over --batch rows); the other slots prefill")
    ap.add_argument("--single", action="store_true", help="--streaming: also time B=1 transcribe() calls of 5 s / 20 s first (BASELINE config 1's call shape)")
    ap.add_argument("--ingest", default="host", choices=["host", "ring"], help="--streaming: decodes hand over host tensors (the reference's call) or name chunk "
                    "ranges of per-session device rings fed chunk by chunk (SURVEY 8 f2)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads2", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-timed", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=150.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-wait-go", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the multi-rank run (nccl == RCCL; gloo for boxes "
                    "with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="ranks use device LOCAL_RANK mod device count (exercise the N-rank path on fewer GPUs; not a scaling measurement)")
    ap.add_argument("--slots", type=int, default=3, help="batches of --batch segments in flight per GPU on one engine / one weight copy (sonic_slot_create); 1 = rounds 1-3's definition")
    ap.add_argument("--pipeline", default="3x64+1", help="the headline leg: bulk pipeline 'DxR+P' = D decoding handles looping continuously over R rows each + P "
                    "prefill slots, all on the engine's one weight copy (sonicscribe_amd/pipeline.py); 'off' = the headline is the --slots leg.  "
                    "(3x64+1 since round 5: 166-169 segments/s against 163-166 for 2x64+1 in alternating runs on one box, profiles/round5_pipeline_shapes.txt)")
    ap.add_argument("--pipeline-host", default="native", choices=["native", "python"], help="who drives the bulk pipeline's hand-overs: threads inside libsonic_hip.so "
                    "(sonic_pipeline_*, round 5) or round 4's Python threads (sonicscribe_amd/pipeline.py ContinuousPipeline; A/B)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects int8_b64 / streaming / pcie_inclusive of the N=1 line")
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        _cpu_reference_worker(a.cpu_threads, a.cpu_timed, a.cpu_budget, a.cpu_wait_go, a.cpu_threads2)
        return
    if a.streaming:
        run_streaming(a)
        return
    if a.facade_only:
        import contextlib
        from sonicscribe_amd import spec as _spec
        with contextlib.redirect_stdout(sys.stderr):
            r = facade_bulk_measure(_spec.FULL if a.dims == "full" else _spec.TINY, 0, BATCH, a.max_new, native_dispatch=(False if a.python_dispatch else None))
        print(json.dumps(r), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = world if world > 1 else 1
    cpu = None
    if n_gpus == 1 and rank == 0 and not a.no_cpu_baseline:
        cpu = CpuBaseline()
        cpu.start()                                      # a child process, spawned before this process initialises the GPU; it idles until collect()
    import torch
    dist = None
    device_index = local_rank % max(1, torch.cuda.device_count()) if a.share_gpu else local_rank
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):   # launched by torch.distributed.run (also at world size 1)
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")               # ranks sharing a GPU: RCCL cannot put two ranks on one device

    from sonicscribe_amd import spec, synth
    from sonicscribe_amd.engine import Engine
    from sonicscribe_amd.sharder import shard_range

    base = spec.FULL if a.dims == "full" else spec.TINY
    dims = replace(base, eos_ids=())       # random weights: never stop early, every row does the full 150 steps
    B = a.batch
    from sonicscribe_amd.engine import MODE_INT8, MODE_NATIVE
    pipe_cfg = None
    if a.pipeline != "off":
        dxr, n_pre = a.pipeline.split("+")
        pipe_cfg = (int(dxr.split("x")[0]), int(dxr.split("x")[1]), int(n_pre))             # decoders, rows per decoder, prefill slots
        if pipe_cfg[1] % B or pipe_cfg[1] > 64:
            raise SystemExit("--pipeline rows must be a multiple of --batch and at most 64")
    eng = Engine(dims, device_index, MODE_INT8 if a.mode == "int8" else MODE_NATIVE, max_batch=max(B, pipe_cfg[1]) if pipe_cfg else B, max_ctx=512)
    eng.load_synthetic(20260128)
    global DECODE_CHUNK_STEPS
    for kv in a.opt:
        if kv.startswith("decode_chunk="):
            DECODE_CHUNK_STEPS = max(1, min(64, int(kv.split("=")[1])))
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    lo, hi = shard_range(n_gpus * B, rank, n_gpus)           # shard g gets segments g*B .. g*B+B-1
    print(f"[bench] rank {rank}/{n_gpus} device {device_index} backend {a.dist_backend if dist is not None else 'none'} segments [{lo}, {hi})", file=sys.stderr, flush=True)
    n_samples = SEG_SECONDS * 16000
    segs = [synth.synth_pcm(i, n_samples) for i in range(lo, hi)]
    n_audio = spec.audio_token_count(spec.valid_frames(n_samples))
    prompt = [1, 17, 23, 5] + [dims.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    eng.stage_pcm(segs)                                        # PCM resident in HBM before any timed region
    eng.run_staged([prompt] * len(segs), [a.max_new] * len(segs))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for en in engines:
            en.lib.sonic_synchronize(en.h)

    def cpu_s():                                               # user + system CPU seconds of this process (all threads, the engine's native ones included)
        import resource
        r = resource.getrusage(resource.RUSAGE_SELF)
        return r.ru_utime + r.ru_stime

    def max_over_ranks(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], device="cuda" if a.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # further batches in flight on the SAME weights: slot engines (their own stream / buffers / KV cache / graphs), each with the batch staged
    n_slots = max(1, a.slots)
    n_handles = max(n_slots, pipe_cfg[0] + pipe_cfg[2]) if pipe_cfg else n_slots
    engines = [eng]
    for k in range(1, n_handles):
        sl = eng.slot()
        sl.stage_pcm(segs)
        sl.run_staged([prompt] * len(segs), [a.max_new] * len(segs))
        engines.append(sl)
    weight_bytes = eng.weight_bytes()
    assert all(sl.weight_bytes() == 0 for sl in engines[1:])
    from sonicscribe_amd.engine import runtime_info
    hwq = runtime_info(device_index)

    # ---- leg A: one batch at a time (the headline definition of rounds 1-3; stages and roofline are measured here)
    for _ in range(max(0, a.warmup - 1)):
        eng.rerun_staged()
    barrier()
    stage = {"mel_ms": 0.0, "encoder_ms": 0.0, "prefill_ms": 0.0, "decode_ms": 0.0}
    c0 = cpu_s()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.rerun_staged()                                     # synchronous: returns after the stream drained
        t = eng.timings()
        for k in stage:
            stage[k] += t[k]
    barrier()
    host_cpu = {"single_batch": (cpu_s() - c0) / (time.perf_counter() - t0)}
    dt_single = max_over_ranks(time.perf_counter() - t0)
    ids = eng.fetch_tokens(len(segs), a.max_new)
    assert all(len(x) == a.max_new for x in ids)

    # ---- leg B: the K batches round-robin over the slots, sonic_run_staged_async / sonic_wait from this one thread
    dt = dt_single
    slots_identical = None
    if n_slots > 1:
        def pipeline(n_batches):
            started = 0
            for k in range(min(n_slots, n_batches)):
                engines[k].run_staged_async(); started += 1
            for j in range(n_batches):
                engines[j % n_slots].wait()
                if started < n_batches:
                    engines[started % n_slots].run_staged_async(); started += 1
        pipeline(max(n_slots, min(a.warmup, 2 * n_slots)))   # warm-up in the same pattern
        barrier()
        c0 = cpu_s()
        t0 = time.perf_counter()
        pipeline(a.steps)
        barrier()
        host_cpu["batches_in_flight_slots"] = (cpu_s() - c0) / (time.perf_counter() - t0)
        dt = max_over_ranks(time.perf_counter() - t0)
        slots_identical = all(all(np.array_equal(x, y) for x, y in zip(sl.fetch_tokens(len(segs), a.max_new), ids)) for sl in engines[1:n_slots])
        assert slots_identical, "a slot's tokens differ from the single-batch run of the same segments"
    dt_slots = dt

    # ---- leg C (headline): the bulk pipeline - decoding handles loop continuously over their rows, prefill slots splice whole batches in
    pipe_info = None
    if pipe_cfg:
        from sonicscribe_amd.pipeline import ContinuousPipeline, NativePipeline
        nd, prow, npre = pipe_cfg
        native = a.pipeline_host == "native"
        want_ids = ids
        prompts_b, budgets_b = [prompt] * len(segs), [a.max_new] * len(segs)
        # Verification pass (untimed): the timed batches are identical, which cannot show a batch that was prefilled with its
        # neighbour's plan.  Here every batch is DIFFERENT - segments rotated and restaged, a different text tail per row, a different budget per
        # row - and every row must give the tokens of the same request in a plain run_staged batch.
        n_var = 3
        var = []
        for v in range(n_var):
            segs_v = segs[v + 1:] + segs[:v + 1]
            prompts_v = [prompt + [40 + v, 50 + (i * 5 + v) % 23] * (1 + (i + v) % 3) for i in range(len(segs))]
            budgets_v = [max(1, a.max_new - (i * 7 + v * 13) % 41) for i in range(len(segs))]
            eng_v = engines[nd]                                  # a prefill slot computes the expected tokens as a plain batch
            eng_v.stage_pcm(segs_v); eng_v.run_staged(prompts_v, budgets_v)
            var.append((segs_v, prompts_v, budgets_v, eng_v.fetch_tokens(len(segs), a.max_new)))
        n_ver = 2 * n_var + 1
        if native:
            pipe = NativePipeline(engines[:nd], engines[nd:nd + npre], block=B)
            tick = [pipe.submit(var[j % n_var][1], var[j % n_var][2], segments=var[j % n_var][0]) for j in range(n_ver)]
            wrong = 0
            for j, t in enumerate(tick):
                got = pipe.wait(t)
                wrong += sum(0 if (len(g) == var[j % n_var][2][i] and np.array_equal(g, var[j % n_var][3][i])) else 1 for i, g in enumerate(got))
            res_v = {"batches": n_ver, "wrong_rows": wrong}
            for p_ in engines[nd:nd + npre]:
                p_.stage_pcm(segs)                              # back to the timed workload's staged PCM
            run_pipe = lambda n: pipe.run(n, prompts_b, budgets_b, lambda i, got: np.array_equal(got, want_ids[i]))
        else:
            pipe = ContinuousPipeline(engines[:nd], engines[nd:nd + npre], block=B)
            run_pipe = lambda n: pipe.run(n, lambda p: p.prefill(prompts_b, budgets_b, wait=False), lambda i, got: np.array_equal(got, want_ids[i]))
            run_pipe(pipe.batches_in_flight)                   # graphs of the row count, every handle touched
            import itertools
            ctr, ctr_lock = itertools.count(), threading.Lock()

            def prefill_var(p):
                with ctr_lock:
                    v = next(ctr) % n_var
                p.stage_pcm(var[v][0])
                p.prefill(var[v][1], var[v][2], wait=False)
                return v
            res_v = pipe.run(n_ver, prefill_var, lambda i, got, v: len(got) == var[v][2][i] and np.array_equal(got, var[v][3][i]))
            for p_ in engines[nd:nd + npre]:
                p_.stage_pcm(segs)
        assert res_v["batches"] == n_ver and res_v["wrong_rows"] == 0, f"pipeline verification pass (varied batches): {res_v}"
        run_pipe(pipe.batches_in_flight)                       # warm-up in the timed pattern
        barrier()
        c0 = cpu_s()
        t0 = time.perf_counter()
        res = run_pipe(a.steps)
        barrier()
        host_cpu["pipeline"] = (cpu_s() - c0) / (time.perf_counter() - t0)
        dt = max_over_ranks(time.perf_counter() - t0)
        assert res["batches"] == a.steps and res["wrong_rows"] == 0, f"pipeline: {res}"
        pipe.close()
        pipe_info = {"decoders": nd, "rows_per_decoder": prow, "prefill_slots": npre, "batches_in_flight": pipe.batches_in_flight, "rows_bit_identical_to_single_batch": True,
                     "decode_chunks_queued": res["decode_chunks"],
                     "host": ("native threads inside libsonic_hip.so (sonic_pipeline_*: condition variables + blocking events, no interpreter in the loop)" if native
                              else "Python threads over the C ABI (round 4's driver)"),
                     "verification_pass": {"batches": res_v["batches"], "wrong_rows": res_v["wrong_rows"],
                                           "what": f"{n_var} different batches (segments rotated and restaged, per-row text tails and budgets) cycled through the pipeline "
                                                   "before the clock: every row equals the same request in a plain batch run"}}

    # one extra, untimed step with HIP events around every encoder-layer GEMM launch (the 256 event records stay out of the timed region)
    eng.set_option("gemm_timing", 1)
    eng.rerun_staged()
    tg = eng.timings()
    eng.set_option("gemm_timing", 0)

    if rank == 0:
        total_segments = n_gpus * B * a.steps
        value = total_segments / dt
        value_single = total_segments / dt_single
        value_slots = total_segments / dt_slots
        in_flight_n = pipe_info["batches_in_flight"] if pipe_info else n_slots
        PEAK_HBM_GBS = 8000.0
        d_ = dims
        qd, kvd = d_.dec_heads * d_.dec_head_dim, d_.dec_kv_heads * d_.dec_head_dim
        wbytes_per_el = 1 if a.mode == "int8" else 2                  # int8 mode: quantised linears stream 1 byte per weight, lm_head stays 16-bit
        w_bytes = wbytes_per_el * d_.dec_layers * (d_.dec_d * (qd + 2 * kvd) + qd * d_.dec_d + 3 * d_.dec_d * d_.dec_ff) + 2 * d_.vocab * d_.dec_d
        n_dec = max(1, a.max_new - 1)                                  # token 1 comes out of prefill
        avg_ctx = len(prompt) + (n_dec + 1) / 2.0
        kv_bytes = B * d_.dec_layers * 2 * kvd * 2 * avg_ctx
        dec_ms = stage["decode_ms"] / a.steps / n_dec
        dec_gbs = (w_bytes + kv_bytes) / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        # Headline leg, honest accounting: a decode loop streams the weights ONCE per token step whatever number of its rows are
        # occupied (two batches of 32 in one 64-row loop share them), every batch reads its own KV.  The steps the loops really ran are counted by
        # the pipeline (decode chunks queued x steps per chunk); without the pipeline every batch runs its own chain.
        if pipe_info:
            loop_steps = pipe_info["decode_chunks_queued"] * max(1, DECODE_CHUNK_STEPS)
            dec_bytes_leg = loop_steps * w_bytes + a.steps * n_dec * kv_bytes
            if_note = (f"{loop_steps} decode-loop token steps queued by the pipeline x weight bytes (once per loop step, shared by the batches riding the loop) + "
                       f"{a.steps} batches x {n_dec} steps x KV bytes, over the wall time of the headline leg (encoder and prefill of other batches run concurrently)")
        else:
            loop_steps = a.steps * n_dec
            dec_bytes_leg = loop_steps * (w_bytes + kv_bytes)
            if_note = "every batch runs its own chain: algorithmic decode bytes of all timed batches / wall time of the headline leg"
        in_flight_obj = {"batches_in_flight": in_flight_n, "achieved": dec_bytes_leg / dt / 1e9, "unit": "GB/s", "frac": dec_bytes_leg / dt / 1e9 / PEAK_HBM_GBS,
                         "decode_loop_steps": loop_steps, "bytes": dec_bytes_leg, "note": if_note}
        # whole step against the two rooflines it is made of: encoder + projector + prefill on the matrix pipe, the decode loop + mel on HBM;
        # the serial bound adds them (nothing overlaps), the overlapped bound takes the larger one
        seg_flops = segment_flops(d_, len(prompt))
        mfma_ms = B * (seg_flops["encoder"] + seg_flops["prefill"]) / (PEAK_BF16_TFLOPS * (2.0 if a.mode == "int8" else 1.0) * 1e12) * 1e3
        hbm_bytes_batch = dec_bytes_leg / a.steps + B * (n_samples * 2 + 128 * 3000 * 2)
        hbm_ms = hbm_bytes_batch / (PEAK_HBM_GBS * 1e9) * 1e3
        whole_step = {"mfma_ms_at_peak": mfma_ms, "hbm_ms_at_peak": hbm_ms, "measured_ms_per_step": dt / a.steps * 1e3,
                      "serial_bound_ms": mfma_ms + hbm_ms, "overlapped_bound_ms": max(mfma_ms, hbm_ms),
                      "frac_of_serial_bound": (mfma_ms + hbm_ms) / (dt / a.steps * 1e3), "frac_of_overlapped_bound": max(mfma_ms, hbm_ms) / (dt / a.steps * 1e3),
                      "flops_per_batch": B * (seg_flops["encoder"] + seg_flops["prefill"]), "hbm_bytes_per_batch": hbm_bytes_batch,
                      "mfma_frac_in_leg": B * (seg_flops["encoder"] + seg_flops["prefill"]) / (dt / a.steps) / 1e12 / (PEAK_BF16_TFLOPS * (2.0 if a.mode == "int8" else 1.0)),
                      "segment_flops": seg_flops,
                      "note": "per batch of the headline leg: FLOPs of encoder + projector + prompt forward at the dense MFMA peak, decode-loop + log-mel bytes at the HBM peak "
                              "(weights charged once per decode-loop step), against the measured wall time per batch"}
        out = {
            "metric": "20s-segments/sec/node + RTF, GLM-ASR-Nano bf16, batch=32, 1/2/4/8 MI355X" if a.mode == "native" and B == BATCH else
                      f"20s-segments/sec/node + RTF, GLM-ASR-Nano {a.mode}, batch={B} (BASELINE config {4 if a.mode == 'int8' else 2} variant)",
            "value": value, "unit": "20s-segments/sec", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8" if a.mode == "int8" else "bf16", "data": "synthetic",
            "rtf": 1.0 / (SEG_SECONDS * value),
            "config": {"workload": ((f"bulk pipeline on one engine / one weight copy ({weight_bytes / 2 ** 20:.0f} MiB): {pipe_info['decoders']} decoding handles each run ONE continuous greedy loop over "
                                     f"{pipe_info['rows_per_decoder']} rows ({pipe_info['rows_per_decoder'] // B} batches of {B} per loop), {pipe_info['prefill_slots']} prefill slot(s) run log-mel + encoder + prompt forward + first "
                                     f"token of whole batches and splice their rows into a free block (sonic_service_* / sonic_prefill / sonic_splice_rows / sonic_fetch_row), hand-overs by {pipe_info['host']}: "
                                     f"up to {pipe_info['batches_in_flight']} batches of {B} in flight; {a.steps} batches timed from an empty pipeline to the last fetched row, every row's tokens equal the single-batch run's; ") if pipe_info else
                                    (f"{n_slots} batches of {B} in flight, one engine, one weight copy ({weight_bytes / 2 ** 20:.0f} MiB; per-slot stream, activation "
                                     f"buffers, KV cache, decode graphs): {a.steps} batches go round-robin to the slots through sonic_run_staged_async / sonic_wait, ") if n_slots > 1
                                    else "one batch in flight: ") +
                                   f"every batch = {B} synthetic {SEG_SECONDS} s 16 kHz segments per GPU, GLM-ASR-Nano dims ({a.dims}), {a.mode}, "
                                   f"log-mel + encoder + prefill + {a.max_new} greedy tokens, portable-PRNG weights; int16 PCM HBM-resident before the "
                                   f"timed region, token ids fetched to the host " + ("row by row as rows finish" if pipe_info else "after the clock stops") + " (see pcie_inclusive for the host-to-host rate); "
                                   f"ms_per_step = wall / batches; batches_in_flight_slots = the same K batches as {n_slots} whole batches in flight (sonic_run_staged_async; round 4's first form); "
                                   f"single_batch = the same K batches one at a time (the headline definition of rounds 1-3)",
                       "pipeline": pipe_info,
                       "hw_queues": hwq,                # hardware queues the HIP runtime of this process really has (measured, sonic_runtime_info), GPU_MAX_HW_QUEUES, wanted
                       "host_cpus_busy": host_cpu,      # CPU seconds per wall second of this process in each timed leg (rank 0): how much host the legs need
                       "batches_in_flight": in_flight_n, "weight_copies": 1, "weight_bytes": weight_bytes, "weights_mb": round(weight_bytes / 2 ** 20, 1), "slots_bit_identical_to_single_batch": slots_identical,
                       "segments_per_gpu": B, "max_new_tokens": a.max_new, "parallelism": f"replica x{n_gpus} (segments sharded, no collective)",
                       "shard_of_rank0": [lo, hi], "dist_backend": (a.dist_backend if dist is not None else None), "share_gpu": bool(a.share_gpu)},
            # one batch at a time: the K steps of leg A (same barriers, same max over ranks); stages and roofline below are ITS device times
            "single_batch": {"value": value_single, "unit": "20s-segments/sec", "ms_per_step": dt_single / a.steps * 1e3, "steps": a.steps,
                             "rtf": 1.0 / (SEG_SECONDS * value_single)},
            # whole batches in flight: the K steps of leg B (each slot runs complete batches through sonic_run_staged_async / sonic_wait; host-robust:
            # a batch is queued whole by a native thread)
            "batches_in_flight_slots": {"value": value_slots, "unit": "20s-segments/sec", "ms_per_step": dt_slots / a.steps * 1e3, "steps": a.steps, "slots": n_slots,
                                        "bit_identical_to_single_batch": slots_identical},
            "stages_ms_per_step": {k: stage[k] / a.steps for k in ("mel_ms", "encoder_ms", "prefill_ms", "decode_ms")},
            "whole_step": whole_step,
            # The time-dominant part of a step is the greedy decode loop (~2/3 of it): every token step streams the decoder's weights,
            # the tied lm_head and each sequence's KV cache exactly once -- HBM-bound.  One "launch" here is one token step (one hipGraph
            # replay of the captured kernel chain); its duration is measured live: HIP events on the engine stream around the decode
            # loop of every timed step, divided by the token steps.  `traffic` (HBM bytes from PMC counters) cannot be read inside
            # the run: the rocprofv3 passes are committed under profiles/ and quoted as traffic_from_profile.
            "roofline": {"bound": "hbm", "achieved": dec_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": dec_gbs / PEAK_HBM_GBS, "traffic": None,
                         "kernel": "decode token step (hipGraph chunk replay: per layer qkv / attention / o_proj / gate-up / down kernels + lm_head + greedy)",
                         "measured_in": "single_batch leg (the chain alone on the GPU); in_flight below is the whole-GPU rate of the headline leg",
                         # headline leg: every batch streams the same algorithmic decode bytes; divided by the WHOLE wall time (encoder and prefill
                         # of the other slot included), i.e. a lower bound of the HBM rate while several chains overlap
                         "in_flight": in_flight_obj,
                         "bytes_per_launch": w_bytes + kv_bytes, "avg_launch_ms": dec_ms, "launches_timed": n_dec * a.steps,
                         "algorithmic_bytes": {"weights": w_bytes, "kv_cache_avg": kv_bytes}},
        }
        # `traffic`: HBM bytes per launch (= per token step) from the PMC counters.  Counters cannot be read inside this run (rocprofv3 --pmc serialises the
        # kernels); the value is the newest committed pass of THIS command line's decode chain (tools/round_profiles.sh: FETCH_SIZE and WRITE_SIZE in separate
        # runs, FETCH_SIZE doubled as the guide prescribes for gfx950), taken at the same --max-new so the contexts match; a pass of another batch size, mode
        # or token budget is quoted under traffic_from_profile only and `traffic` stays null
        for prof in ("round6_pmc_decode.json", "round5_pmc_decode.json", "round4_pmc_decode.json", "round3_pmc_decode.json", "round2_pmc_decode.json"):
            try:
                with open(os.path.join(ROOT, "profiles", prof)) as f:
                    pm = json.load(f)
                out["roofline"]["traffic_from_profile"] = {k: pm[k] for k in ("hbm_bytes_per_token_step", "fetch_bytes_x2", "write_bytes", "note", "source", "commit", "max_new", "batch", "mode") if k in pm}
                out["roofline"]["traffic_from_profile"]["file"] = "profiles/" + prof
                if pm.get("max_new") == a.max_new and pm.get("batch", BATCH) == B and pm.get("mode", "native") == a.mode and a.dims == "full":
                    out["roofline"]["traffic"] = pm["hbm_bytes_per_token_step"]
                    out["roofline"]["traffic_over_algorithmic"] = pm["hbm_bytes_per_token_step"] / (w_bytes + kv_bytes)
                break
            except Exception:
                pass
        # the other rooflines SURVEY.md 8d names:
        #   mel front-end vs HBM: 1.408 MB of algorithmic bytes per 20 s segment (int16 PCM in, bf16 features out), timed region
        #   encoder GEMMs vs MFMA: all encoder-layer GEMM launches (QKV, o, fc1, fc2) with HIP events around each, from the extra step
        mel_bytes = B * (n_samples * 2 + 128 * 3000 * 2)
        mel_gbs = mel_bytes / (stage["mel_ms"] / a.steps * 1e-3) / 1e9 if stage["mel_ms"] > 0 else 0.0
        # ... and what the kernel is actually bound by in its DFT-as-GEMM form: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s peak): four 100-long
        # sums x 128 bin columns per live frame (frames in whole 32-frame tiles)
        live_frames = ((min(n_samples, 480000) + 200 + 159) // 160 + 31) // 32 * 32
        mel_flops = B * live_frames * 4 * 100 * 128 * 2.0
        mel_tf = mel_flops / (stage["mel_ms"] / a.steps * 1e-3) / 1e12 if stage["mel_ms"] > 0 else 0.0
        out["mel_frontend"] = {"bound": "hbm", "achieved": mel_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": mel_gbs / PEAK_HBM_GBS,
                               "bytes_per_step": mel_bytes, "ms_per_step": stage["mel_ms"] / a.steps,
                               "compute": {"bound": "mfma fp32", "achieved": mel_tf, "peak": 157.3, "unit": "TFLOP/s", "frac": mel_tf / 157.3, "flops_per_step": mel_flops,
                                           "note": "the folded 400-point DFT runs as an exact-fp32 MFMA GEMM: by bytes the kernel is HBM-bound, in this form it is not"}}
        peak_mm = PEAK_BF16_TFLOPS * (2.0 if a.mode == "int8" else 1.0)
        if tg["enc_gemm_ms"] > 0:
            eg = tg["enc_gemm_flops"] / (tg["enc_gemm_ms"] * 1e-3) / 1e12
            out["encoder_gemms"] = {"bound": "mfma", "achieved": eg, "peak": peak_mm, "unit": "TFLOP/s", "frac": eg / peak_mm,
                                    "ms_per_step": tg["enc_gemm_ms"], "flops_per_step": tg["enc_gemm_flops"],
                                    "note": "HIP events around each of the 128 encoder-layer GEMM launches of one extra untimed step"}
        if tg["gemm_launches"] > 0:
            gm = tg["gemm_ms"] / tg["gemm_launches"]; fl = tg["gemm_flops"] / tg["gemm_launches"]
            out["encoder_fc1_gemm"] = {"bound": "mfma", "achieved": fl / (gm * 1e-3) / 1e12, "peak": peak_mm, "unit": "TFLOP/s",
                                       "frac": fl / (gm * 1e-3) / 1e12 / peak_mm, "avg_launch_ms": gm, "flops_per_launch": fl, "launches_timed": tg["gemm_launches"]}
        extras = n_gpus == 1 and a.dims == "full" and a.mode == "native" and B == BATCH and not a.no_extras
        if extras:
            # host-to-host rate of the same batch through the one-call boundary (sonic_transcribe_batch: H2D of the int16 PCM, the device
            # work, D2H of the ids) - never the headline `value`
            t1 = time.perf_counter()
            for _ in range(a.steps):
                eng.transcribe_batch(segs, [prompt] * len(segs), [a.max_new] * len(segs))
            d1 = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": B * a.steps / d1, "unit": "20s-segments/sec", "ms_per_step": d1 / a.steps * 1e3,
                                     "note": "sonic_transcribe_batch host buffers in, host ids out (32 x 640 kB of PCM over PCIe per step); not the headline"}
        if extras:
            import contextlib
            eng.close(); eng = None; engines = []
            _quiet = contextlib.redirect_stdout(sys.stderr)       # the facade prints a banner; stdout carries the one JSON line only
            _quiet.__enter__()
            # BASELINE config 4 (the repo's INT8 option, batch 64) and the bf16 batch-64 figure it has to beat, same process, same box
            for key, mode_ in (("int8_b64", "int8"), ("bf16_b64", "native")):     # (separate try blocks: one failure must not erase the other's figure)
                try:
                    out[key] = extra_batch_run(dims, device_index, mode_, 64, a.max_new, steps=2, slots=a.slots)
                except Exception as ex:
                    out[key] = {"value": None, "note": f"not measured: {ex!r}"}
            # the facade's bulk shape: the same pipeline behind ASRModel.submit() - host float tensors in (peak-normalise + PCM_16 on the host, H2D),
            # transcripts out - 10 batches' worth of segments submitted at once
            try:
                out["facade_bulk"] = facade_bulk_measure(dims, device_index, B, a.max_new)
            except Exception as ex:
                out["facade_bulk"] = {"value": None, "note": f"not measured: {ex!r}"}
            # BASELINE config 5's call pattern at its per-GPU share (128 sessions / 8 GPUs = 16), real-time schedule, device-resident ingest
            try:
                # continuous scheduling (dispatch._ContinuousReplica: the engine decodes forever over its rows, one slot prefills), decode chunks of 2 steps
                sa = argparse.Namespace(dims="full", sessions=16, gpus=1, replicas_per_gpu=1, mode="native", batch=BATCH, ingest="ring", slots=2, single=True,
                                        continuous=True, opt=["decode_chunk=2"])
                st = streaming_measure(sa)
                out["streaming"] = {k: st[k] for k in ("sessions", "partial_latency_ms", "final_latency_ms", "value", "unit", "wall_s", "ingest", "device_batches_per_replica", "slots", "continuous")}
                for k in ("single_5s", "single_20s"):            # BASELINE config 1's call shape, measured on the same model object
                    if k in st:
                        out[k] = st[k]
                out["streaming"]["note"] = ("BASELINE config 5 call pattern: 16 sessions (128 / 8 GPUs) x (64 ms chunks into device rings, 1 s partials of 1.28 s / 15 "
                                            "tokens, one 20 s final / 150 tokens), real-time schedule through ASRModel.submit(); latency = submit -> transcript; row-level "
                                            "scheduling (ASRModel(continuous=True)): requests join and leave a running greedy loop row by row")
            except Exception as ex:
                out["streaming"] = {"value": None, "note": f"not measured: {ex!r}"}
            _quiet.__exit__(None, None, None)
        if cpu is not None:
            try:
                out["cpu_baseline"] = cpu.collect()
            except Exception as ex:   # transformers missing on the box: report that rather than a wrong number
                out["cpu_baseline"] = {"value": None, "unit": "20s-segments/sec", "cores": 0, "kind": "reference", "sample": f"unavailable: {ex!r}"}
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
