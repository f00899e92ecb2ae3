"""The hipGraph cache of stn_batch_run on a real MI355X: several shapes stay captured at once (the reference's call() chunk
loop and n_test loop alternate shapes, /root/reference/cpp/helper.cpp:697-719, cpp/example_onnx.cpp:88), and nothing a captured
graph points at may be freed under it (weights reloaded on the handle, staging that outgrows its block)."""
import numpy as np
import pytest

from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
from gpu_util import make_inputs

pytestmark = pytest.mark.gpu


def _ins(a, shapes):
    return {k: make_inputs(a, B, Lt, np.array(lens), seed=3) + (np.array(d, np.float32),) for k, (B, Lt, lens, d) in shapes.items()}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_three_alternating_shapes_all_replay(dtype):
    a = tiny_arch()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    shapes = {"A": (4, 13, [13, 4, 9, 2], [0.9, 0.3, 0.6, 0.2]), "B": (3, 9, [9, 5, 2], [0.5, 0.4, 0.2]), "C": (2, 17, [17, 8], [1.2, 0.5])}
    ins = _ins(a, shapes)
    # eager references (graph mode off)
    eng.set_graph_mode(False)
    want = {}
    for k in "CBA":  # largest buffers first: nothing is reallocated afterwards
        ids, mask, sttl, sdp, durs = ins[k]
        want[k], _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
    eng.set_graph_mode(True)
    r0 = eng.graph_replays
    seq = "ABCABCABCABC"
    for k in seq:
        ids, mask, sttl, sdp, durs = ins[k]
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
        np.testing.assert_array_equal(w, want[k], err_msg=k)
    # every shape is captured by its second sighting in graph mode at the latest (the eager passes above may have warmed it, a
    # buffer that grew in between un-warms it) and replayed from then on
    assert eng.graphs_cached == 3
    assert eng.graph_replays - r0 >= len(seq) - 6
    r1 = eng.graph_replays
    for k in "CABBCA":
        ids, mask, sttl, sdp, durs = ins[k]
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
        np.testing.assert_array_equal(w, want[k], err_msg=k)
    assert eng.graph_replays - r1 == 6 and eng.graphs_cached == 3  # all three shapes replay, none was re-captured


def test_cache_evicts_least_recently_used():
    a = tiny_arch()
    eng = binding.Engine(0, "f32")
    eng.load_synthetic(a, 7)
    want = {}
    for Lt in list(range(19, 8, -1)) * 3:  # 11 shapes, three passes: more than the cache holds
        ids, mask, sttl, sdp = make_inputs(a, 2, Lt, np.array([Lt, 3]), seed=Lt)
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=np.array([0.6, 0.2], np.float32), noise_seed=5)
        if Lt in want:
            np.testing.assert_array_equal(w, want[Lt])
        want[Lt] = w
    assert 1 <= eng.graphs_cached <= 8


def test_reloading_weights_drops_the_captured_graphs():
    """A captured graph points into the weight allocations: after stn_load_synthetic on the same handle the same batch shape must
    run on the NEW weights (bit for bit what a fresh engine gives), not replay a graph that reads freed memory."""
    a = tiny_arch()
    ids, mask, sttl, sdp = make_inputs(a, 3, 11, np.array([11, 6, 2]), seed=1)
    durs = np.array([0.8, 0.4, 0.2], np.float32)
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 7)
    for _ in range(3):  # eager, capture, replay
        w7, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
    assert eng.graphs_cached == 1
    eng.load_synthetic(a, 8)
    assert eng.graphs_cached == 0
    outs = [eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)[0] for _ in range(3)]
    fresh = binding.Engine(0, "bf16")
    fresh.load_synthetic(a, 8)
    w8, _ = fresh.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
    for o in outs:
        np.testing.assert_array_equal(o, w8)
    assert not np.array_equal(w7, w8)


def test_batch_size_sequence_4_5_4():
    """B = 4 captured, B = 5 in between, B = 4 again: the second B = 4 run must not replay a graph whose copy nodes read
    staging that the B = 5 upload replaced."""
    a = tiny_arch()
    eng = binding.Engine(0, "f32")
    eng.load_synthetic(a, 7)
    i4 = make_inputs(a, 4, 10, np.array([10, 7, 4, 2]), seed=2) + (np.array([0.7, 0.5, 0.3, 0.2], np.float32),)
    i5 = make_inputs(a, 5, 10, np.array([10, 8, 6, 4, 2]), seed=2) + (np.array([0.7, 0.6, 0.5, 0.3, 0.2], np.float32),)
    ref = binding.Engine(0, "f32")
    ref.load_synthetic(a, 7)
    ref.set_graph_mode(False)
    want4, _ = ref.synthesize(*i4[:4], 2, 1.0, duration_override=i4[4], noise_seed=9)
    want5, _ = ref.synthesize(*i5[:4], 2, 1.0, duration_override=i5[4], noise_seed=9)
    for k in "444" + "555" + "444" + "5454":
        i, want = (i4, want4) if k == "4" else (i5, want5)
        w, _ = eng.synthesize(*i[:4], 2, 1.0, duration_override=i[4], noise_seed=9)
        np.testing.assert_array_equal(w, want, err_msg=k)
    assert eng.graph_replays >= 6


def test_shape_buckets_let_unlike_requests_share_a_graph():
    """stn_set_shape_buckets: two length-aware requests of different lengths that fall into the same buckets replay ONE captured graph
    (replays rise, cached graphs do not), and every utterance's frames and samples are bit-identical to the unbucketed run — the rows
    are merely longer.  (the call() chunk loop /root/reference/cpp/helper.cpp:697-719, the service
    /root/reference/py/service.py:79-136.)"""
    from supertonic_amd import host, workload
    from supertonic_amd.arch import default_arch
    a = default_arch()
    up = host.UnicodeProcessor(host.synthetic_indexer())

    def request(seed, words):
        texts = workload.utterances(6, min_words=words[0], max_words=words[1], seed=seed)
        ids, mask = up(texts, ["en"] * 6)
        sttl, sdp = workload.synthetic_styles(a, list(range(6)))
        return ids, mask, sttl, sdp, workload.forced_durations(texts)

    reqs = [request(s, (7, 9)) for s in (1, 2, 3, 4, 5, 6)]
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 7)
    eng.set_vocoder_mode(True)
    exact = []
    for ids, mask, sttl, sdp, durs in reqs:  # the unbucketed results (each request its own shapes: eager or its own graph)
        wav, dur = eng.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=durs, noise_seed=4)
        exact.append((wav.copy(), dur.copy(), eng.batch_fetch_latent().copy()))
    eng.set_shape_buckets(True)
    keys = set()
    for rnd in range(3):  # first pass: eager (warm), second: capture, third: replay
        c0, r0 = eng.graphs_cached, eng.graph_replays
        for (ids, mask, sttl, sdp, durs), (wav0, dur0, lat0) in zip(reqs, exact):
            wav, dur = eng.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=durs, noise_seed=4)
            B, L, W = eng.batch_dims()
            keys.add((L, eng.ve_rows))
            lat = eng.batch_fetch_latent()
            assert np.array_equal(dur, dur0) and W >= wav0.shape[1] and L >= lat0.shape[2]
            ns = np.floor(dur * a.sample_rate).astype(int)
            for b in range(6):
                assert np.array_equal(wav[b, :ns[b]], wav0[b, :ns[b]]), (rnd, b)          # its own samples: the same bits
            assert np.array_equal(lat[:, :, :lat0.shape[2]], lat0) and np.all(lat[:, :, lat0.shape[2]:] == 0)
        if rnd == 2:
            assert eng.graph_replays - r0 == len(reqs) and eng.graphs_cached == c0  # all six replayed, nothing new captured
    # six requests of six different length profiles: fewer bucket combinations than requests, and at most that many graphs
    n_exact = len({(r[0].shape[1], e[2].shape[2], int(sum(np.ceil(np.floor(e[1] * a.sample_rate) / a.chunk_size)))) for r, e in zip(reqs, exact)})
    assert len(keys) < n_exact and eng.graphs_cached <= len(keys), (keys, n_exact, eng.graphs_cached)
    print(f"shape buckets: {n_exact} exact shapes -> {len(keys)} bucketed shapes, {eng.graphs_cached} graphs cached")
    eng.close()
