"""The three scenarios of the reference's only test script (test_all.sh:61-70: default text, 2-item en+ko batch with two
voices, ~600-character long-form text) through the native CLI `supertonic_amd/example_native`, which takes the flags of
cpp/example_onnx.cpp:35-50.  Pass criterion of the reference = exit code + WAV files; here the WAVs are also parsed."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "supertonic_amd", "example_native")
LONG = ("The engine synthesizes long passages by splitting them into chunks. Each chunk is synthesized on its own. "
        "The chunks are then joined with a short silence between them. This keeps the memory footprint small! "
        "Does it also keep the prosody natural? Mostly, yes. " * 3).strip()


def _wav(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
    fmt, ch, sr, _, _, bits = struct.unpack("<hhiihh", b[20:36])
    n = struct.unpack("<i", b[40:44])[0]
    assert (fmt, ch, bits) == (1, 1, 16) and n == len(b) - 44
    return sr, np.frombuffer(b[44:], dtype="<i2")


def _run(args, cwd):
    p = subprocess.run([CLI, "--synthetic"] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_default_scenario(tmp_path):
    out = _run(["--onnx-dir", "no_assets_here", "--n-test", "2", "--save-dir", "res", "--seed", "7"], tmp_path)
    assert "synthetic weights" in out and out.count("Saved: ") == 2 and "completed in" in out
    files = sorted(os.listdir(tmp_path / "res"))
    assert files == ["This_morning__I_took_1.wav", "This_morning__I_took_2.wav"]  # sanitizeFilename(text, 20) + "_" + n
    sr, pcm = _wav(tmp_path / "res" / files[0])
    assert sr == 44100 and len(pcm) > 4410 and np.abs(pcm).max() > 0


def test_batch_en_ko_two_voices(tmp_path):
    out = _run(["--batch", "--voice-style", "M1,F1", "--text", "Hello there, how are you today?|안녕하세요 반갑습니다", "--lang", "en,ko",
                "--n-test", "1", "--save-dir", "res", "--total-step", "3", "--seed", "5"], tmp_path)
    assert out.count("Saved: ") == 2
    files = sorted(os.listdir(tmp_path / "res"))
    assert len(files) == 2 and any(f.startswith("안녕하세요") for f in files)
    for f in files:
        sr, pcm = _wav(tmp_path / "res" / f)
        assert sr == 44100 and len(pcm) > 0


def test_long_form_chunks_and_silence(tmp_path):
    assert len(LONG) > 600
    out = _run(["--text", LONG, "--n-test", "1", "--save-dir", "res", "--seed", "3", "--total-step", "2"], tmp_path)
    assert out.count("Saved: ") == 1
    (f,) = os.listdir(tmp_path / "res")
    sr, pcm = _wav(tmp_path / "res" / f)
    # chunks are joined by 0.3 s of exact zeros (cpp/helper.cpp:706-715): find at least one such run inside the file
    z = (pcm == 0).astype(np.int8)
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], z, [0]]))))[::2]
    assert runs.size and runs.max() >= int(0.3 * sr) - 1


def test_mismatched_counts_exit_code(tmp_path):
    p = subprocess.run([CLI, "--batch", "--voice-style", "M1", "--text", "a|b", "--lang", "en,en"], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 1 and "must match number of texts" in p.stderr  # cpp/example_onnx.cpp:66-70
    p = subprocess.run([CLI, "--synthetic", "--text", "x", "--lang", "de", "--n-test", "1"], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 2 and "Invalid language: de" in p.stderr  # cpp/helper.cpp:193


def test_missing_assets_are_an_error_without_the_opt_in(tmp_path):
    """cpp/helper.cpp:805: unreadable assets throw; synthetic weights are an explicit opt-in (--synthetic)."""
    p = subprocess.run([CLI, "--onnx-dir", "no_assets_here", "--n-test", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "Saved: " not in p.stdout
    assert "Failed to open" in (p.stdout + p.stderr)


def test_voice_style_files_are_loaded(tmp_path):
    """--voice-style with real paths goes through loadVoiceStyle (cpp/helper.cpp:829-897), not the synthetic-by-name styles."""
    import json
    rng = np.random.default_rng(3)
    paths = []
    for name in ("M1", "F1"):
        p = tmp_path / f"{name}.json"
        d = {"style_ttl": {"data": (rng.standard_normal((1, 50, 256)) * 0.1).astype(np.float32).tolist(), "dims": [1, 50, 256], "type": "float32"},
             "style_dp": {"data": (rng.standard_normal((1, 8, 16)) * 0.1).astype(np.float32).tolist(), "dims": [1, 8, 16], "type": "float32"}}
        p.write_text(json.dumps(d))
        paths.append(str(p))
    out = _run(["--batch", "--voice-style", ",".join(paths), "--text", "Hello there.|Good morning to you.", "--lang", "en,en",
                "--n-test", "1", "--save-dir", "res", "--total-step", "2", "--seed", "5"], tmp_path)
    assert "Loaded 2 voice styles" in out and "synthetic styles keyed by name" not in out and out.count("Saved: ") == 2
    # a path that does not exist beside one that does is an error (the reference throws), not a silent synthetic style
    p = subprocess.run([CLI, "--synthetic", "--batch", "--voice-style", paths[0] + "," + str(tmp_path / "missing.json"), "--text", "a|b", "--lang", "en,en",
                        "--n-test", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
