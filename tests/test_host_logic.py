"""CPU tests: host-side restatement of the reference glue, C-ABI surface, weight plumbing, sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sonicscribe_amd import frontend, spec, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_audio_token_counts_match_survey_shapes():
    # 5 s -> 500 frames -> 62 tokens; 20 s -> 2000 frames -> 250 tokens; 30 s -> 375
    assert spec.valid_frames(80000) == 500 and spec.audio_token_count(500) == 62
    assert spec.valid_frames(320000) == 2000 and spec.audio_token_count(2000) == 250
    assert spec.audio_token_count(3000) == 375
    assert spec.audio_token_count(spec.valid_frames(20480)) == 16      # 1.28 s partial
    assert spec.audio_token_count(1) == 0 and spec.audio_token_count(0) == 0


def test_param_count_matches_survey():
    assert abs(spec.param_count(spec.FULL) - 2.136e9) / 2.136e9 < 0.01     # 2.136 B incl. tied embedding once


def test_hotwords_prompt():
    assert frontend.format_hotwords_prompt([]) == ""
    assert frontend.format_hotwords_prompt(["  ", None, 3]) == ""
    s = frontend.format_hotwords_prompt(["Brand", "brand ", "Model X", "Brand"])      # set() drops exact repeats of the RAW string only (asr.py:318-322)
    assert s == '. Pay special attention to these important terms: "brand", "brand", "model x"'
    assert frontend.format_hotwords_prompt(["Alpha", "alpha "]) == '. Pay special attention to these important terms: "alpha", "alpha"'
    many = [f"w{i}" for i in range(20)]
    assert frontend.format_hotwords_prompt(many).count('"') == 20      # capped at 10 hotwords
    assert frontend.build_instruction(None) == "Please transcribe this audio into text"


def test_pcm_bytes_and_max_new_tokens():
    b = np.array([0, 16384, -32768, 32767], np.int16).tobytes()
    x = frontend.pcm_bytes_to_float(b)
    assert x.shape == (1, 4) and x.dtype == np.float32 and x[0, 1] == 0.5 and x[0, 2] == -1.0
    assert frontend.max_new_tokens_committed(20.0) == 150 and frontend.max_new_tokens_committed(40.0) == 200
    assert frontend.max_new_tokens_committed(0.5) == 52


def test_normalise_properties():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1000) * 0.01).astype(np.float32)
    q = frontend.normalise_to_int16(x)
    assert np.abs(q).max() == 32767                                      # peak normalised
    assert np.array_equal(q, frontend.normalise_to_int16(x * 7.5))       # scale invariant (asr.py:265-267)
    assert np.array_equal(frontend.normalise_to_int16(x[None, :]), q)   # [1, N] accepted
    tiny = np.full(10, 5e-7, np.float32)
    assert not frontend.normalise_to_int16(tiny).any()                  # max <= 1e-6: passed unnormalised -> rounds to 0


def test_split_windows():
    assert frontend.split_windows(320000) == [(0, 320000)]
    assert frontend.split_windows(0) == [(0, 0)]
    w = frontend.split_windows(30 * 16000 * 25)
    assert len(w) == 21 and w[-1][1] == 21 * 480000                      # truncated to 655 s / 21 windows


def test_header_declares_what_library_exports():
    hdr = open(os.path.join(ROOT, "include", "sonic_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sonic_[a-z_0-9]+)\s*\(", hdr)))
    from sonicscribe_amd import engine
    assert sorted(engine.EXPORTS) == declared
    if not os.path.exists(engine.LIB_PATH):
        pytest.skip("libsonic_hip.so not built")
    lib = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sonic_abi_version() == engine.ABI_VERSION == int(re.search(r"#define SONIC_ABI_VERSION (\d+)", hdr).group(1))
    # the dynamic symbol table is the C ABI and nothing else (-fvisibility=hidden + csrc/exports.map): no mangled launchers, no kernel stubs
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], capture_output=True, text=True).stdout
        exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
        assert exported == declared, sorted(set(exported) ^ set(declared))[:10]


def test_library_refuses_without_gpu_or_bad_args():
    from sonicscribe_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        pytest.skip("libsonic_hip.so not built")
    lib = engine.load_library()
    h = C.c_void_p()
    cd = engine.make_dims(spec.TINY)
    rc = lib.sonic_create(C.byref(cd), 0, 7, 4, 512, C.byref(h))       # bad mode
    assert rc != 0 and b"mode must be either" in lib.sonic_last_error(None)
    if lib.sonic_device_count() == 0:
        rc = lib.sonic_create(C.byref(cd), 0, engine.MODE_INT8, 4, 512, C.byref(h))   # int8 mode is built: it fails for the missing GPU only
        assert rc == 2 and b"no HIP device" in lib.sonic_last_error(None)
        rc = lib.sonic_create(C.byref(cd), 0, 0, 4, 512, C.byref(h))
        assert rc != 0 and not h.value                                    # fails loudly, no CPU fallback
        with pytest.raises(RuntimeError):
            engine.Engine(spec.TINY)


def test_asrmodel_argument_errors():
    from sonicscribe_amd.asr import ASRModel
    with pytest.raises(ValueError):
        ASRModel("x", mode="fp8")
    with pytest.raises(RuntimeError):
        ASRModel("x", device="cpu")
    with pytest.raises(RuntimeError):
        ASRModel("x", device="cpu", mode="int8")        # int8 mode exists (asr.py:148-210), but only on a GPU


def test_checkpoint_name_mapping_and_config(tmp_path):
    from sonicscribe_amd import weights
    assert weights.canonical_name("audio_tower.conv1.weight") == "model.audio_tower.conv1.weight"
    assert weights.canonical_name("language_model.model.layers.0.mlp.up_proj.weight") == "model.language_model.layers.0.mlp.up_proj.weight"
    assert weights.canonical_name("language_model.lm_head.weight") == "lm_head.weight"
    weights.save_synthetic_checkpoint(str(tmp_path), spec.TINY, 3)
    d = weights.load_dims(str(tmp_path))
    assert d == spec.TINY
    names = {}
    for name, arr, bits in weights.iter_safetensors(str(tmp_path)):
        names[name] = (arr, bits)
    inv = {n: s for n, s, _ in spec.tensor_inventory(spec.TINY)}
    assert set(names) == set(inv)
    ref = synth.synth_state_dict(spec.TINY, 3, bf16=True)
    k = "model.audio_tower.layers.1.mlp.fc1.weight"
    assert names[k][1] and np.array_equal(names[k][0], synth.to_bf16_bits(ref[k]))


def test_shard_segments():
    from sonicscribe_amd.sharder import shard_range
    got = [shard_range(256, r, 8) for r in range(8)]
    assert got[0] == (0, 32) and got[7] == (224, 256)
    cover = []
    for r in range(3):
        a, b = shard_range(10, r, 3)
        cover += list(range(a, b))
    assert cover == list(range(10))


def test_header_is_plain_c_and_links(tmp_path):
    """include/sonic_hip.h is the drop-in boundary: it must compile as C99 (no C++-isms outside the extern "C" guard) and a C program
    that names every entry point must link against the built library (no compute: there is no GPU here)."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "sonic_hip.h")).read()
    names = sorted(set(re.findall(r"\b(sonic_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    src = tmp_path / "abi.c"
    body = "\n".join(f"    p[{i}] = (fn){n};" for i, n in enumerate(names))
    src.write_text('#include "sonic_hip.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n    fn p[%d];\n%s\n'
                   '    printf("%%d entry points, %%d non-null\\n", %d, (int)(p[0] != 0));\n    return 0;\n}\n' % (len(names), body, len(names)))
    lib_dir = os.path.join(root, "sonicscribe_amd", "csrc")
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L", lib_dir, "-lsonic_hip", "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
