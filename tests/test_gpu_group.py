"""include/stn_group.h on a GPU box: the group path at N = 1 against stn_batch_fetch_pcm16 byte for byte; the multi-rank path
rehearsed on one GPU (ranks share the device, the gather is a device copy: deal, worker threads, block layout and the reorder into
caller order are the real path's) against single-engine syntheses of exactly the dealt shards; refusal of more devices than the box
has; the C++ host and CLI with --gpus / --devices; the RCCL calls of the exchange with one rank sending to itself.  Two or more RCCL ranks need
as many GPUs and have not run here."""
import os
import subprocess

import numpy as np
import pytest

from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch, tiny_arch
from gpu_util import make_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3_like(n, seed):
    arch = default_arch()
    texts = workload.utterances(n, min_words=3, max_words=12, seed=seed)
    ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * n)
    sttl, sdp = workload.synthetic_styles(arch, list(range(n)))
    return arch, ids, mask, sttl, sdp, workload.forced_durations(texts)


def test_group_of_one_equals_the_engine_byte_for_byte():
    arch, ids, mask, sttl, sdp, durs = _c3_like(12, 3)
    g = binding.Group(1, "bf16")
    assert not g.uses_rccl
    g.load_synthetic(arch, 7)
    pcm, dur = g.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=99)
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(arch, 7)
    # the group sorts its one shard by length: upload the same order, with the caller's indices as the noise keys
    rank_of, row_of = binding.group_deal(mask.sum(axis=(1, 2)).astype(np.int32), 1)
    order = np.argsort(row_of)
    Lt = int(mask[order].sum(axis=(1, 2)).max())
    eng.batch_upload(ids[order][:, :Lt], mask[order][:, :, :Lt], sttl[order], sdp[order], duration_override=durs[order], utt_ids=order.astype(np.int64))
    eng.batch_run(3, 1.05, 99)
    ref, dref = eng.batch_fetch_pcm16()
    assert pcm.shape == ref.shape and np.array_equal(pcm[order], ref) and np.array_equal(dur[order], dref)
    assert np.abs(pcm.astype(np.int32)).max() > 0
    g.close()


@pytest.mark.parametrize("n_ranks,B", [(2, 9), (3, 10), (4, 3)])
def test_rehearsal_of_the_multi_rank_path_on_one_gpu(n_ranks, B):
    arch, ids, mask, sttl, sdp, durs = _c3_like(B, 40 + n_ranks)
    g = binding.Group([0] * n_ranks, "bf16")
    assert g.n == n_ranks and not g.uses_rccl
    g.load_synthetic(arch, 7)
    pcm, dur = g.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=durs, noise_seed=5)
    rows, samples = g.last_shards()
    lengths = mask.sum(axis=(1, 2)).astype(np.int32)
    rank_of, row_of = binding.group_deal(lengths, n_ranks)
    assert np.array_equal(rows, np.bincount(rank_of, minlength=n_ranks)) and pcm.shape == (B, samples.max())
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(arch, 7)
    for r in range(n_ranks):
        mine = np.where(rank_of == r)[0]
        if len(mine) == 0:
            assert rows[r] == 0
            continue
        order = mine[np.argsort(row_of[mine])]
        Lt = int(lengths[order].max())
        eng.batch_upload(ids[order][:, :Lt], mask[order][:, :, :Lt], sttl[order], sdp[order], duration_override=durs[order], utt_ids=order.astype(np.int64))
        eng.batch_run(2, 1.05, 5)
        ref, dref = eng.batch_fetch_pcm16()
        W = ref.shape[1]
        assert W == samples[r]
        assert np.array_equal(pcm[order][:, :W], ref) and np.all(pcm[order][:, W:] == 0) and np.array_equal(dur[order], dref), r
    # a second synthesis of another shape on the same group (buffers regrow, graphs re-key)
    arch2, ids2, mask2, sttl2, sdp2, durs2 = _c3_like(B + 5, 77)
    pcm2, dur2 = g.synthesize(ids2, mask2, sttl2, sdp2, 2, 1.05, duration_override=durs2, noise_seed=5)
    assert pcm2.shape[0] == B + 5 and np.all(dur2 > 0) and np.abs(pcm2.astype(np.int32)).max() > 0
    g.close()


def test_more_devices_than_the_box_has_is_an_error():
    n = binding.device_count()
    with pytest.raises(binding.StnError) as ei:
        binding.Group(n + 1, "bf16")
    assert ei.value.code == -2 and "visible" in str(ei.value) and str(n) in str(ei.value)
    with pytest.raises(binding.StnError):
        binding.Group([0, 0, n], "bf16")


def test_a_rank_failure_names_the_rank_and_leaves_nothing_to_fetch():
    arch, ids, mask, sttl, sdp, durs = _c3_like(5, 8)
    g = binding.Group([0, 0], "bf16")
    g.load_synthetic(arch, 7)
    pcm, _ = g.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=durs, noise_seed=5)
    bad = durs.copy()
    bad[binding.group_deal(mask.sum(axis=(1, 2)).astype(np.int32), 2)[0] == 1] = 0.0  # rank 1's shard: not a duration
    with pytest.raises(binding.StnError) as ei:
        g.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=bad, noise_seed=5)
    assert "rank 1" in str(ei.value) and "device 0" in str(ei.value) and "duration override must be > 0" in str(ei.value)
    out = np.zeros(pcm.size, np.int16)
    rc = g._lib.stn_group_fetch_pcm16(g._g, out.ctypes.data, out.size, None)
    assert rc == -3 and b"no synthesis" in g._lib.stn_group_last_error(g._g)  # STN_ERR_STATE: the earlier result is not handed out as this one's
    pcm2, _ = g.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=durs, noise_seed=5)  # and the group is usable afterwards
    assert np.array_equal(pcm, pcm2)
    g.close()


SELF_RCCL_CHILD = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch
arch = default_arch()
texts = workload.utterances(12, min_words=3, max_words=12, seed=3)
ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * 12)
sttl, sdp = workload.synthetic_styles(arch, list(range(12)))
g = binding.Group(1, "bf16")
g.load_synthetic(arch, 7)
for seed in (99, 100):   # twice: the communicator and the receive block are reused
    pcm, dur = g.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=workload.forced_durations(texts), noise_seed=seed)
    print("RESULT", int(g.uses_rccl), hashlib.sha256(pcm.tobytes()).hexdigest(), int(np.abs(pcm.astype(np.int32)).max()))
g.close()
"""


def test_the_rccl_calls_of_the_exchange_on_one_gpu(tmp_path):
    """A group of one whose own block goes through ncclSend / ncclRecv to itself (measurement switch STN_GROUP_SELF_RCCL=1): librccl is opened, the
    communicator created, a grouped send / receive enqueued on the engine's stream and the received block fetched — the bytes are the plain group's.
    More than one rank needs more than one GPU and has not run here."""
    import sys
    outs = []
    for extra in ({}, {"STN_DEV_SWITCHES": "1", "STN_GROUP_SELF_RCCL": "1"}):
        env = dict(os.environ); env.pop("STN_GROUP_SELF_RCCL", None); env.pop("STN_DEV_SWITCHES", None); env.update(extra)
        r = subprocess.run([sys.executable, "-c", SELF_RCCL_CHILD, ROOT], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-4000:])
        outs.append([l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")])
    plain, via = outs
    assert len(plain) == 2 and len(via) == 2
    assert all(p[1] == "0" for p in plain) and all(v[1] == "1" for v in via)   # uses_rccl
    assert [p[2] for p in plain] == [v[2] for v in via] and all(int(p[3]) > 0 for p in plain)


def test_cli_gpus_flag(tmp_path):
    exe = os.path.join(ROOT, "supertonic_amd", "example_native")
    n = binding.device_count()
    base = [exe, "--onnx-dir", str(tmp_path / "no_assets"), "--synthetic", "--batch", "--n-test", "1", "--total-step", "2", "--seed", "7",
            "--voice-style", "F1,M2,F3", "--text", "Hello there.|A second, longer sentence for the batch.|Third one.", "--lang", "en,en,en"]
    # more GPUs than the box has: the reference-style error exit, both numbers in the message
    p = subprocess.run(base + ["--gpus", str(n + 1), "--save-dir", str(tmp_path / "a")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "visible" in p.stderr, (p.returncode, p.stderr[-500:])
    # the rehearsal: three ranks on device 0, against the single-device run — the same three WAV files, byte for byte
    outs = {}
    for tag, extra in (("one", []), ("three", ["--devices", "0,0,0"])):
        d = tmp_path / tag
        p = subprocess.run(base + extra + ["--save-dir", str(d)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-1500:]
        outs[tag] = {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}
        assert len(outs[tag]) == 3 and all(len(b) > 1000 for b in outs[tag].values())
    assert "by device copies" in p.stdout
    # (one utterance per rank vs three in one batch: the kernels' regimes differ with the row count, so equal to rounding, not bit for bit)
    for f in outs["one"]:
        a = np.frombuffer(outs["one"][f][44:], "<i2").astype(np.int32)
        b = np.frombuffer(outs["three"][f][44:], "<i2").astype(np.int32)
        assert a.shape == b.shape and np.abs(a - b).max() <= 0.12 * max(1.0, np.sqrt(np.mean(a.astype(np.float64) ** 2))), f
