"""ORACLE (test infrastructure, never shipped, never on the product path).

Pure-Python / numpy restatement of the reference's C++ HOST behaviour on the
synthesis path — the code that sits around the four `Ort::Session::Run` sites of
`cpp/helper.cpp`.  The C++ host is the contract (BASELINE.json north_star keeps
the `cpp/` surface); where the reference's Python twin (`py/helper.py`) behaves
differently this file follows the C++ and the tests list
the divergent inputs explicitly.

Pinned by: tests/golden/host_fixtures.json (generated from the reference's
`py/helper.py` by tools/gen_golden.py) on every input where C++ == Python.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

AVAILABLE_LANGS = ("en", "ko", "es", "pt", "fr")  # cpp/helper.cpp:15

_C_SPACE = b" \t\n\v\f\r"  # std::isspace in the "C" locale (cpp/helper.cpp:30-42)


def _trim(b: bytes) -> bytes:
    """cpp/helper.cpp:30-42 (byte-wise isspace trim)."""
    s, e = 0, len(b)
    while s < e and b[s] in _C_SPACE:
        s += 1
    while e > s and b[e - 1] in _C_SPACE:
        e -= 1
    return b[s:e]


def _replace_all(b: bytes, frm: bytes, to: bytes) -> bytes:
    """find/replace loop advancing past the replacement (cpp/helper.cpp:89-95)."""
    return b.replace(frm, to)  # bytes.replace is the same left-to-right non-overlapping scan


# cpp/helper.cpp:69-87 — order matters
_REPLACEMENTS = [
    ("–", "-"), ("‑", "-"), ("—", "-"), ("_", " "),
    ("“", '"'), ("”", '"'), ("‘", "'"), ("’", "'"),
    ("´", "'"), ("`", "'"), ("[", " "), ("]", " "), ("|", " "), ("/", " "),
    ("#", " "), ("→", " "), ("←", " "),
]
_SPECIAL = ["♥", "☆", "♡", "©", "\\"]  # cpp/helper.cpp:105
_EXPR = [("@", " at "), ("e.g.,", "for example, "), ("i.e.,", "that is, ")]  # :114-118
_PUNCT_FIX = [b" ,", b" .", b" !", b" ?", b" ;", b" :", b" '"]  # :129-135
_END_ASCII = b".!?;:,'\")]}>"  # :158-163
_END_3BYTE = [s.encode("utf-8") for s in
              ("…", "。", "」", "』", "】", "〉", "》", "›",
               "»", "“", "”", "‘", "’")]  # :167-174 ("»" is 2 bytes: never matches)


def preprocess_text(text: str, lang: str) -> str:
    """UnicodeProcessor::preprocessText, cpp/helper.cpp:52-200 (operates on UTF-8 bytes)."""
    b = text.encode("utf-8")
    for frm, to in _REPLACEMENTS:
        b = _replace_all(b, frm.encode("utf-8"), to.encode("utf-8"))
    # :99-102 — drop every 4-byte sequence F0 9F xx xx
    out = bytearray()
    i = 0
    while i < len(b):
        if (i + 3 < len(b) and b[i] == 0xF0 and b[i + 1] == 0x9F
                and 0x80 <= b[i + 2] <= 0xBF and 0x80 <= b[i + 3] <= 0xBF):
            i += 4
            continue
        out.append(b[i])
        i += 1
    b = bytes(out)
    for sym in _SPECIAL:
        b = b.replace(sym.encode("utf-8"), b"")
    for frm, to in _EXPR:
        b = _replace_all(b, frm.encode("utf-8"), to.encode("utf-8"))
    for pat in _PUNCT_FIX:  # single left-to-right regex_replace each
        b = b.replace(pat, pat[1:])
    while b'""' in b:  # :138-149
        b = b.replace(b'""', b'"', 1)
    while b"''" in b:
        b = b.replace(b"''", b"'", 1)
    while b"``" in b:
        b = b.replace(b"``", b"`", 1)
    # :152 — \s+ -> " "
    out = bytearray()
    in_ws = False
    for c in b:
        if c in _C_SPACE:
            if not in_ws:
                out.append(0x20)
            in_ws = True
        else:
            out.append(c)
            in_ws = False
    b = _trim(bytes(out))
    if b:  # :156-182 (empty text gets no period in the C++ host)
        ends = b[-1] in _END_ASCII
        if not ends and len(b) >= 3 and b[-3:] in _END_3BYTE:
            ends = True
        if not ends:
            b += b"."
    if lang not in AVAILABLE_LANGS:  # :185-194
        raise ValueError("Invalid language: " + lang + ". Available: en, ko, es, pt, fr")
    b = b"<" + lang.encode() + b">" + b + b"</" + lang.encode() + b">"
    return b.decode("utf-8", errors="surrogateescape")


# cpp/helper.cpp:214-269: (base letter, combining mark) by precomposed code point
_LATIN = {}
for _mark, _letters in ((0x0301, "AEIOUaeiou"), (0x0300, "AEIOUaeiou"), (0x0302, "AEIOUaeiou"),
                        (0x0303, "ANOano"), (0x0308, "AEIOUaeiou"), (0x0327, "Cc")):
    import unicodedata as _ud
    for _ch in _letters:
        _LATIN[ord(_ud.normalize("NFC", _ch + chr(_mark)))] = (ord(_ch), _mark)
del _mark, _letters, _ch


def text_to_unicode_values(text: str) -> List[int]:
    """UnicodeProcessor::textToUnicodeValues + decomposeCharacter, cpp/helper.cpp:272-347."""
    b = text.encode("utf-8", errors="surrogateescape")
    n = len(b)
    vals: List[int] = []
    i = 0
    while i < n:
        c = b[i]
        if c & 0x80 == 0:
            cp = c
            i += 1
        elif c & 0xE0 == 0xC0 and i + 1 < n:
            cp = ((c & 0x1F) << 6) | (b[i + 1] & 0x3F)
            i += 2
        elif c & 0xF0 == 0xE0 and i + 2 < n:
            cp = ((c & 0x0F) << 12) | ((b[i + 1] & 0x3F) << 6) | (b[i + 2] & 0x3F)
            i += 3
        elif c & 0xF8 == 0xF0 and i + 3 < n:
            cp = ((c & 0x07) << 18) | ((b[i + 1] & 0x3F) << 12) | ((b[i + 2] & 0x3F) << 6) | (b[i + 3] & 0x3F)
            i += 4
        else:
            i += 1
            continue
        if 0xAC00 <= cp < 0xAC00 + 11172:  # Hangul syllable -> Jamo, :274-287
            s = cp - 0xAC00
            vals.append(0x1100 + s // 588)
            vals.append(0x1161 + (s % 588) // 28)
            if s % 28:
                vals.append(0x11A7 + s % 28)
        elif cp in _LATIN:  # :290-296
            vals.extend(_LATIN[cp])
        else:
            vals.append(cp & 0xFFFF)  # :299
    return vals


def length_to_mask(lengths: Sequence[int], max_len: Optional[int] = None) -> np.ndarray:
    """lengthToMask, cpp/helper.cpp:740-757 -> float32 [B,1,max_len]."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if max_len is None or max_len == -1:
        max_len = int(lengths.max())
    ids = np.arange(max_len)
    return (ids[None, :] < lengths[:, None]).astype(np.float32).reshape(-1, 1, max_len)


def get_latent_mask(wav_lengths: Sequence[int], base_chunk_size: int, chunk_compress_factor: int) -> np.ndarray:
    """getLatentMask, cpp/helper.cpp:759-770."""
    cs = base_chunk_size * chunk_compress_factor
    wl = np.asarray(wav_lengths, dtype=np.int64)
    return length_to_mask((wl + cs - 1) // cs)


def unicode_processor_call(indexer: Sequence[int], text_list: Sequence[str], lang_list: Sequence[str]
                           ) -> Tuple[np.ndarray, np.ndarray]:
    """UnicodeProcessor::call, cpp/helper.cpp:355-390 -> (int64 [B,Lt], float32 [B,1,Lt])."""
    vals = [text_to_unicode_values(preprocess_text(t, l)) for t, l in zip(text_list, lang_list)]
    lens = [len(v) for v in vals]
    lt = max(lens)
    ids = np.zeros((len(vals), lt), dtype=np.int64)
    n_idx = len(indexer)
    for i, v in enumerate(vals):
        for j, u in enumerate(v):
            if u < n_idx:  # :383-385 (out-of-table stays 0)
                ids[i, j] = indexer[u]
    return ids, length_to_mask(lens)


def latent_geometry(duration: Sequence[float], sample_rate: int, base_chunk_size: int,
                    chunk_compress_factor: int, latent_dim: int) -> Tuple[int, int, np.ndarray]:
    """Shape part of TextToSpeech::sampleNoisyLatent, cpp/helper.cpp:424-440,457.

    Returns (D, L, latent_lengths[B]).  Float32 arithmetic as in the C++ host:
    wav_len_max = max(dur)*sr (float), L = int((wav_len_max + cs - 1) / cs),
    wav_lengths[b] = int64(dur_b * sr) (float32 product truncated).
    """
    d = np.asarray(duration, dtype=np.float32)
    sr = np.float32(sample_rate)
    cs = base_chunk_size * chunk_compress_factor
    wav_len_max = np.float32(d.max() * sr)
    L = int(np.float32(np.float32(wav_len_max + np.float32(cs)) - np.float32(1)) / np.float32(cs))
    wav_lengths = (d * sr).astype(np.int64)
    lat = (wav_lengths + cs - 1) // cs
    return latent_dim * chunk_compress_factor, L, lat


def sample_noisy_latent(duration, sample_rate, base_chunk_size, chunk_compress_factor, latent_dim,
                        rng: np.random.Generator) -> Tuple[np.ndarray, np.ndarray]:
    """sampleNoisyLatent with an injectable generator (the reference is unseeded, :442-444)."""
    D, L, lat = latent_geometry(duration, sample_rate, base_chunk_size, chunk_compress_factor, latent_dim)
    mask = length_to_mask(lat, L)
    xt = rng.standard_normal((len(lat), D, L)).astype(np.float32) * mask
    return xt, mask


def _split_paragraphs(b: bytes) -> List[bytes]:
    """regex \\n\\s*\\n+ token split, cpp/helper.cpp:1121-1131."""
    parts: List[bytes] = []
    cur = bytearray()
    i, n = 0, len(b)
    while i < n:
        if b[i] == 0x0A:
            # try to match \n \s* \n+ with backtracking: find the longest \s* run that still ends in \n
            j = i + 1
            while j < n and b[j] in _C_SPACE:
                j += 1
            # within b[i+1:j] (all whitespace) we need at least one more '\n'
            k = j
            while k > i + 1 and b[k - 1] != 0x0A:
                k -= 1
            if k > i + 1:  # b[k-1] is a '\n' at index >= i+1  -> match is b[i:k]
                parts.append(bytes(cur))
                cur = bytearray()
                i = k
                continue
        cur.append(b[i])
        i += 1
    parts.append(bytes(cur))
    return parts


def chunk_text(text: str, max_len: int = 300) -> List[str]:
    """chunkText, cpp/helper.cpp:1117-1186 (byte lengths; sentences keep their delimiter run)."""
    b = text.encode("utf-8", errors="surrogateescape")
    paragraphs = [p for p in (_trim(x) for x in _split_paragraphs(b)) if p]
    chunks: List[bytes] = []
    for para in paragraphs:
        # split on [.!?]\s+ ; each non-empty token gets the delimiter that follows it appended (:1144-1155)
        sentences: List[bytes] = []
        n = len(para)
        start = 0
        i = 0
        while i < n:
            if para[i] in b".!?" and i + 1 < n and para[i + 1] in _C_SPACE:
                j = i + 1
                while j < n and para[j] in _C_SPACE:
                    j += 1
                tok = para[start:i]
                if tok:
                    sentences.append(tok + para[i:j])
                start = j
                i = j
                continue
            i += 1
        tok = para[start:]
        if tok:
            sentences.append(tok)
        cur = b""
        for s in sentences:  # :1161-1173
            if len(cur) + len(s) + 1 <= max_len:
                if cur:
                    cur += b" "
                cur += s
            else:
                if cur:
                    chunks.append(_trim(cur))
                cur = s
        if cur:
            chunks.append(_trim(cur))
    if not chunks:  # :1181-1183
        chunks.append(_trim(b))
    return [c.decode("utf-8", errors="surrogateescape") for c in chunks]


def sanitize_filename(text: str, max_len: int) -> str:
    """sanitizeFilename, cpp/helper.cpp:1070-1111."""
    b = text.encode("utf-8", errors="surrogateescape")
    out = bytearray()
    i, cnt, n = 0, 0, len(b)
    while i < n and cnt < max_len:
        c = b[i]
        if (0x30 <= c <= 0x39) or (0x41 <= c <= 0x5A) or (0x61 <= c <= 0x7A) or c == 0x5F:
            out.append(c); i += 1
        elif c & 0xE0 == 0xC0 and i + 1 < n:
            out += b[i:i + 2]; i += 2
        elif c & 0xF0 == 0xE0 and i + 2 < n:
            out += b[i:i + 3]; i += 3
        elif c & 0xF8 == 0xF0 and i + 3 < n:
            out += b[i:i + 4]; i += 4
        else:
            out.append(0x5F); i += 1
        cnt += 1
    return bytes(out).decode("utf-8", errors="surrogateescape")


def wav_bytes(audio: np.ndarray, sample_rate: int) -> bytes:
    """writeWavFile, cpp/helper.cpp:943-990: RIFF PCM16 mono, int16(clamp(x,-1,1)*32767) truncated."""
    a = np.asarray(audio, dtype=np.float32)
    pcm = (np.clip(a, -1.0, 1.0) * np.float32(32767)).astype(np.int16)  # astype truncates toward zero
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<i", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<ihhiihh", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<i", len(data))
    return hdr + data


def concat_chunks(wavs: Sequence[np.ndarray], durs: Sequence[float], sample_rate: int,
                  silence_duration: float = 0.3) -> Tuple[np.ndarray, float]:
    """Long-form join of TextToSpeech::call, cpp/helper.cpp:703-716 (untrimmed chunk waves)."""
    out = None
    dur = np.float32(0)
    for w, d in zip(wavs, durs):
        if out is None:
            out, dur = np.asarray(w, np.float32), np.float32(d)
        else:
            sil = np.zeros(int(np.float32(silence_duration) * np.float32(sample_rate)), np.float32)
            out = np.concatenate([out, sil, np.asarray(w, np.float32)])
            dur = np.float32(dur + np.float32(np.float32(d) + np.float32(silence_duration)))
    return out, float(dur)
