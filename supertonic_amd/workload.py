"""Synthetic benchmark workloads of BASELINE.json (no datasets offline): seeded pseudo-English utterances.

Word list: 512 pronounceable lowercase pseudo-words (2-10 letters) from a fixed syllable generator, seed 1234.
Durations are forced to n_chars / 15 s (the speech rate the reference's published tables imply,
/root/reference/README.md:190-254) because synthetic weights predict meaningless durations."""
import numpy as np

SPEECH_RATE_CPS = 15.0
C1_SENTENCE = "The quick brown fox jumps over the lazy sleeping dog."

_ONSETS = ["b", "br", "c", "ch", "d", "dr", "f", "fl", "g", "gr", "h", "j", "k", "l", "m", "n", "p", "pl", "pr", "r", "s",
           "sh", "st", "t", "th", "tr", "v", "w", "y", "z", ""]
_VOWELS = ["a", "e", "i", "o", "u", "ai", "ea", "ee", "oo", "ou"]
_CODAS = ["", "", "n", "r", "s", "t", "l", "m", "nd", "st", "ck", "ng"]


def word_list(n=512, seed=1234):
    rng = np.random.default_rng(seed)
    words, seen = [], set()
    while len(words) < n:
        syl = 1 + int(rng.random() < 0.3)
        w = "".join(_ONSETS[rng.integers(len(_ONSETS))] + _VOWELS[rng.integers(len(_VOWELS))] + _CODAS[rng.integers(len(_CODAS))]
                    for _ in range(syl))
        if 2 <= len(w) <= 10 and w not in seen:
            seen.add(w)
            words.append(w)
    return words


def utterances(n, words_per_utt=10, seed=1234, min_words=None, max_words=None):
    """n utterances; fixed word count, or uniform in [min_words, max_words] (the mixed-length C4 workload)."""
    rng = np.random.default_rng(seed)
    wl = word_list()
    out = []
    for _ in range(n):
        k = words_per_utt if min_words is None else int(rng.integers(min_words, max_words + 1))
        ws = [wl[int(i)] for i in rng.integers(0, len(wl), k)]
        s = " ".join(ws)
        out.append(s[0].upper() + s[1:] + ".")
    return out


def forced_durations(texts):
    """seconds BEFORE the /speed division: n_chars / 15."""
    return np.array([len(t) / SPEECH_RATE_CPS for t in texts], np.float32)


def synthetic_styles(arch, utt_ids, seed=1234):
    """Voice styles are asset files we do not have: seeded N(0, 0.1^2) of the model's shapes, keyed by utterance id
    so that a sharded batch sees the same styles as an unsharded one."""
    ttl = np.empty((len(utt_ids), arch.n_style_ttl, arch.d_style_ttl), np.float32)
    dp = np.empty((len(utt_ids), arch.n_style_dp, arch.d_style_dp), np.float32)
    for i, u in enumerate(utt_ids):
        rng = np.random.default_rng([seed, int(u)])
        ttl[i] = rng.standard_normal(ttl.shape[1:]) * 0.1
        dp[i] = rng.standard_normal(dp.shape[1:]) * 0.1
    return ttl, dp
