#!/usr/bin/env python3
"""Generate tests/golden/host_fixtures.json from the reference's Python host.

Run ONLY in the build container (needs /root/reference); the output JSON is
committed, this script is committed, the reference source never is.

How the reference is loaded: `py/helper.py` does `import onnxruntime as ort`
at module top (py/helper.py:9) and uses it only in type annotations of the
neural loaders (py/helper.py:145-148,284-285).  onnxruntime is not installed
here, so an EMPTY placeholder module of that name is registered first; it
provides no functionality and none of the neural functions are called.  Only
the host-side functions are exercised:

  UnicodeProcessor._preprocess_text / __call__      py/helper.py:21-136
  length_to_mask / get_latent_mask                  py/helper.py:263-290
  TextToSpeech.sample_noisy_latent (shapes, mask)   py/helper.py:160-175
  chunk_text / sanitize_filename                    py/helper.py:378-429

The four ONNX graphs are absent, so nothing neural can be pinned: the neural
oracle (oracle/stn_ref.c) is "parity unpinned" — see DESIGN.md.
"""
import json
import os
import sys
import types

import numpy as np

REF_PY = "/root/reference/py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "host_fixtures.json")

VOCAB = 512


def synthetic_indexer():
    """unicode_indexer.json is absent: a synthetic table of the same
    shape (flat int64 array indexed by UTF-16 code unit, cpp/helper.cpp:1054-1064)."""
    idx = np.zeros(65536, dtype=np.int64)
    cp = np.arange(65536)
    idx[:] = np.where(cp < 384, cp, 384 + (cp % 128))
    return idx.tolist()


def main():
    ph = types.ModuleType("onnxruntime")
    ph.InferenceSession = type("InferenceSession", (), {})
    ph.SessionOptions = type("SessionOptions", (), {})
    sys.modules["onnxruntime"] = ph
    sys.path.insert(0, REF_PY)
    import helper  # noqa: E402  (the reference's Python host)

    fx = {"_generated_by": "tools/gen_golden.py from /root/reference/py/helper.py",
          "vocab": VOCAB}

    # ---- text normalisation ------------------------------------------------
    up = helper.UnicodeProcessor.__new__(helper.UnicodeProcessor)
    up.indexer = synthetic_indexer()
    pre_cases = [
        ("Hello world", "en"),
        ("The quick brown fox jumps over the lazy sleeping dog.", "en"),
        ("This morning, I took a walk in the park , and it was nice !", "en"),
        ("Wait — what?  Really–yes_no [note] a|b a/b #tag", "en"),
        ("“Quoted” and ‘single’ and `tick` and ´acute´", "en"),
        ("mail me @ home, e.g., now; i.e., today", "en"),
        ("He said \"\"hi\"\" and ''bye''", "en"),
        ("tabs\tand\nnewlines   and   spaces  ", "en"),
        ("Ends with colon:", "en"),
        ("Ends with paren)", "en"),
        ("arrow → left ← end", "en"),
        ("hearts ♥ stars ☆ ♡ copy © back\\slash", "en"),
        ("안녕하세요 반갑습니다", "ko"),
        ("¿Cómo estás? Mañana será mejor", "es"),
        ("Olá, você está bem? Ação e coração", "pt"),
        ("Ça va très bien, merci à vous. Noël", "fr"),
        ("Emoji \U0001F600 gone \U0001F680 too", "en"),
        ("ellipsis at end…", "en"),
        ("guillemet end »", "fr"),
        ("x ,y .z !w ?v ;u :t 's", "en"),
    ]
    fx["preprocess"] = [{"text": t, "lang": l, "out": up._preprocess_text(t, l)} for t, l in pre_cases]
    try:
        up._preprocess_text("x", "de")
        fx["preprocess_bad_lang_raises"] = False
    except ValueError:
        fx["preprocess_bad_lang_raises"] = True

    # ---- text -> ids / mask (synthetic indexer) ------------------------------
    # only inputs whose normalised form stays inside the BMP (py/helper.py:112
    # casts to uint16; cp > 0xFFFF is a documented divergence)
    id_batches = [
        (["Hello world"], ["en"]),
        (["The quick brown fox jumps over the lazy sleeping dog.", "Hi"], ["en", "en"]),
        (["안녕하세요", "Good morning to you"], ["ko", "en"]),
        (["¿Cómo estás?", "Olá você", "Ça va"], ["es", "pt", "fr"]),
    ]
    fx["text_ids"] = []
    for texts, langs in id_batches:
        ids, mask = up(texts, langs)
        fx["text_ids"].append({
            "texts": texts, "langs": langs,
            "text_ids": ids.tolist(),
            "lengths": mask.sum(axis=(1, 2)).astype(int).tolist(),
            "mask_shape": list(mask.shape),
        })

    # ---- masks -------------------------------------------------------------
    fx["length_to_mask"] = []
    for lengths, max_len in [([3, 1, 5], None), ([4], None), ([2, 2], 6), ([7, 0, 3], None)]:
        m = helper.length_to_mask(np.array(lengths, dtype=np.int64), max_len)
        fx["length_to_mask"].append({"lengths": lengths, "max_len": max_len,
                                     "shape": list(m.shape), "mask": m.reshape(m.shape[0], -1).tolist()})
    fx["latent_mask"] = []
    for wl, bcs, ccf in [([148400, 3072, 3073, 1], 512, 6), ([44100, 88200], 512, 6), ([1000, 5000, 9999], 256, 4)]:
        m = helper.get_latent_mask(np.array(wl, dtype=np.int64), bcs, ccf)
        fx["latent_mask"].append({"wav_lengths": wl, "base_chunk_size": bcs, "chunk_compress_factor": ccf,
                                  "shape": list(m.shape), "latent_lengths": m.sum(axis=(1, 2)).astype(int).tolist()})

    # ---- noisy latent geometry (values are unseeded noise: only shapes/masks) -
    cfg = {"ae": {"sample_rate": 44100, "base_chunk_size": 512},
           "ttl": {"chunk_compress_factor": 6, "latent_dim": 24}}
    tts = helper.TextToSpeech(cfg, up, None, None, None, None)
    fx["noisy_latent"] = []
    for durs in [[3.2, 1.1], [3.3650794], [0.05, 0.5, 9.99, 2.0], [0.6965736]]:
        xt, lm = tts.sample_noisy_latent(np.array(durs, dtype=np.float32))
        zero_outside = bool(np.all(xt * (1 - lm) == 0))
        fx["noisy_latent"].append({"duration": durs, "cfg": cfg, "xt_shape": list(xt.shape),
                                   "mask_shape": list(lm.shape),
                                   "latent_lengths": lm.sum(axis=(1, 2)).astype(int).tolist(),
                                   "zero_outside_mask": zero_outside})

    # ---- chunker / filenames -----------------------------------------------
    chunk_cases = [
        ("Hello world.", 300),
        ("First paragraph here.\n\nSecond paragraph there.\n\n\nThird one.", 300),
        ("Dr. Smith went home. He slept! Did he? Yes.", 25),
        ("One sentence that is fairly long indeed. Another sentence that is also long enough. Third.", 45),
        ("A. B. C. D.", 300),
        ("  leading and trailing   ", 300),
        ("", 300),
        ("No terminal punctuation at all", 10),
    ]
    fx["chunk_text"] = [{"text": t, "max_len": n, "py_chunks": helper.chunk_text(t, n)} for t, n in chunk_cases]
    san_cases = [("Hello, world! 123", 20), ("under_score stays", 20), ("a/b\\c:d*e?f", 20),
                 ("This morning, I took a walk in the park", 20), ("안녕 hi", 20),
                 ("Héllo wörld", 8), ("", 5)]
    fx["sanitize_filename"] = [{"text": t, "max_len": n, "out": helper.sanitize_filename(t, n)} for t, n in san_cases]

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(fx, f, ensure_ascii=True, indent=1)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
