#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: audio-seconds per second (RTF^-1) + p50 utterance latency,
66 M-parameter model, batch = 128 utterances per GPU, bf16, 5 Euler steps (config C3).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one full synthesis of this rank's resident batch (DP -> text encoder -> noise -> 5x vector estimator ->
vocoder) with every input already in HBM; for N > 1 the step also converts the waveforms to 16-bit PCM (what the
reference writes to disk) and gathers them to rank 0 over RCCL, the gather overlapping the next step's synthesis.
`value` is that device-resident rate (the headline, as the task contract defines it); `value_host` beside it is the
host-to-host rate of _infer's own contract (/root/reference/cpp/helper.cpp:674-682: host text in, host waveform out): text
frontend -> pinned upload -> synthesis -> 16-bit PCM back in host memory, the copy of batch i overlapping batch i+1.
--scaling weak (default): 128 utterances per GPU; --scaling strong: 128 utterances in all, 128/N per GPU.
Synthetic text / styles / weights (no assets offline).  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="utterances per GPU")
    ap.add_argument("--words", type=int, default=10)
    ap.add_argument("--mixed", action="store_true",
                    help="config C4 instead of C3: mixed-length utterances (4..48 words), length-sorted and dealt round-robin")
    ap.add_argument("--total-step", type=int, default=5)
    ap.add_argument("--speed", type=float, default=1.05)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu-sample", type=int, default=32, help="utterances timed on the CPU oracle (0 = skip): 1 warm-up + 3 timed runs, ~15 s at 32")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event timing of the dominant kernel")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --batch utterances per GPU; strong: --batch utterances in all")
    ap.add_argument("--no-host-loop", action="store_true", help="skip the host-to-host loop (value_host)")
    ap.add_argument("--no-b1", action="store_true", help="skip the single-utterance record (config C2)")
    ap.add_argument("--eager", action="store_true", help="launch every step eagerly instead of replaying the captured hipGraphs (profiler runs: rocprofv3's kernel "
                                                         "trace of many graph launches crashes inside the ROCm 7.2 runtime; kernel durations are the same)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself (0: pick a free one)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process could use, WITHOUT loading a HIP runtime here (the parent of the ranks stays clear of the GPU: a runtime handle
    held by it for the whole run serves nothing): the *_VISIBLE_DEVICES lists when set, else the KFD topology (nodes with SIMDs are
    GPUs).  None when neither can be read — the ranks then check for themselves and fail with exit code 2."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here — one process per GPU through
    torch.distributed.run, as a CHILD process — pass rank 0's JSON line through (the children inherit stdout) and return their exit
    code.  This parent never loads a HIP runtime (visible_gpus()).  Fewer than N visible GPUs is an error, never a silent one-rank run."""
    import socket
    import subprocess
    stub = os.environ.get("STN_BENCH_STUB") == "1"
    if not stub:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            print(f"[bench] error: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to run a smaller job under that label",
                  file=sys.stderr)
            return 2
    port = args.master_port
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def stub_main(args):
    """STN_BENCH_STUB=1: the launch / rendezvous / max-over-ranks / one-JSON-line skeleton with a stand-in step and the gloo backend —
    what the CPU test of the spawn path runs (tests/test_bench_spawn_cpu.py).  Nothing here is a measurement."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from supertonic_amd.dist import bench_shards
    texts_all, shards = bench_shards(args.batch, world, args.scaling, args.mixed, args.words)
    mine = shards[rank]
    audio = float(sum(len(texts_all[i]) for i in mine)) / 15.0 / args.speed

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    fence()
    el = torch.tensor([time.perf_counter() - t0, audio], dtype=torch.float64)
    if world > 1:
        mx = el.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(el, op=dist.ReduceOp.SUM)
        el[0] = mx[0]
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(float(el[1]) * args.steps / float(el[0]), 1), "unit": "audio-sec/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(el[0]) / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "stub", "batch_per_gpu": len(mine), "global_batch": len(texts_all)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if os.environ.get("STN_BENCH_STUB") == "1":
        sys.exit(stub_main(args))
    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner at communicator
    # creation, for one) is diverted to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] error: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    # the multi-GPU code path (process group, shared stream, waveform gather); STN_BENCH_FORCE_DIST=1 exercises it at world 1
    use_dist = world > 1 or os.environ.get("STN_BENCH_FORCE_DIST") == "1"
    # One GPU needs nothing from PyTorch: the run then stays on the HIP runtime libstn.so was built against (the system ROCm) instead of
    # the copy bundled with the torch wheel, which importing torch first would bind the library to — measure on the runtime you ship.
    # Only the N > 1 ranks (torch.distributed over RCCL for the gather) import torch, before the library (binding.load()).
    torch = dist = None
    if use_dist:
        import torch
        import torch.distributed as dist
    else:
        os.environ["STN_NO_TORCH_PRELOAD"] = "1"
    from supertonic_amd import binding, host, workload
    from supertonic_amd.arch import default_arch
    if use_dist:
        from supertonic_amd.dist import GatherPlan
    n_dev = torch.cuda.device_count() if use_dist else binding.device_count()
    if n_dev <= local:
        print(f"[bench] error: rank {rank} wants GPU {local} but only {n_dev} are visible", file=sys.stderr)
        sys.exit(2)
    dev = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    rt_info = binding.runtime_info()

    # ---- workload: 128*N utterances, sorted by length and dealt round-robin ---------------------
    arch = default_arch()
    from supertonic_amd.dist import bench_shards
    up = host.UnicodeProcessor(host.synthetic_indexer())

    def make_shard(scaling):
        texts_all_, shards_ = bench_shards(args.batch, world, scaling, args.mixed, args.words)
        mine_ = shards_[rank]
        texts_ = [texts_all_[i] for i in mine_]
        ids_, mask_ = up(texts_, ["en"] * len(texts_))
        sttl_, sdp_ = workload.synthetic_styles(arch, mine_)
        return dict(n_total=len(texts_all_), mine=mine_, texts=texts_, ids=ids_, mask=mask_, sttl=sttl_, sdp=sdp_, durs=workload.forced_durations(texts_))

    sh = make_shard(args.scaling)
    n_total, mine, texts, ids, mask, sttl, sdp, durs = (sh[k] for k in ("n_total", "mine", "texts", "ids", "mask", "sttl", "sdp", "durs"))

    eng = binding.Engine(local, args.dtype)
    eng.load_synthetic(arch, 7)
    side = None
    if use_dist:
        # one dedicated (non-default) stream carries the engine's kernels AND the RCCL gather, so the collective is ordered
        # after the waveform copy without a host sync.  (The default stream's handle is 0 = "engine's own stream".)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
        eng.set_stream(side.cuda_stream)
    if args.eager:
        eng.set_graph_mode(False)
    eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs, utt_ids=mine)

    gather_buf = {}
    step_no = [0]
    cur = {"durs": durs}

    def step():
        if use_dist and "plan" in gather_buf:
            gather_buf["plan"].wait(step_no[0] & 1)  # the slot's previous gather (two steps ago) must have drained
        eng.batch_run(args.total_step, args.speed, 1234)
        if use_dist:
            B, L, W = eng.batch_dims()
            if gather_buf.get("shape") != (B, W):  # first step of a shape only: buffers + the one-time shape exchange
                gather_buf["dur"] = torch.tensor(cur["durs"] / args.speed, dtype=torch.float32, device=dev)
                gather_buf["plan"] = GatherPlan((B, W), dev, torch.int16, dst=0, slots=2)
                gather_buf["shape"] = (B, W)
            plan, k = gather_buf["plan"], step_no[0] & 1
            # int16 PCM (the reference's final product, writeWavFile) straight into the gather payload, then the gather
            # itself starts behind it and overlaps the NEXT step's synthesis (two payload slots)
            eng.batch_copy_pcm16_device(plan.wav_ptr(k), plan.stride)
            plan.set_durations(gather_buf["dur"], k)
            plan.launch(k)
        step_no[0] += 1

    def fence():
        if use_dist:
            if "plan" in gather_buf:
                gather_buf["plan"].wait(0)
                gather_buf["plan"].wait(1)
            dist.barrier()
        eng.sync()
        if use_dist:
            torch.cuda.synchronize()
        else:
            binding.device_sync(local)  # hipDeviceSynchronize through libstn.so: the same device-wide fence, no PyTorch in the process

    def timed(k_steps):
        """K steps between barrier + synchronize on both sides; the MAX over ranks."""
        fence()
        t0 = time.perf_counter()
        for _ in range(k_steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def audio_total(durs_rank):
        a_ = float((durs_rank / np.float32(args.speed)).sum())
        if use_dist:
            t = torch.tensor([a_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            a_ = float(t.item())
        return a_

    # ---- warm-up (sizes the workspace; the shape is captured as a hipGraph the second time it is seen and replayed from the third
    # step on: at least three untimed steps, so that the timed region below is what a host's n_test loop runs — graph replays,
    # /root/reference/cpp/example_onnx.cpp:88-97)
    n_warm = max(3, args.warmup)
    for _ in range(n_warm):
        step()
    replays0 = eng.graph_replays

    # ---- timed region: exactly K steps (hipGraph replays of the post-duration pipeline) between barrier + synchronize -----------
    elapsed = timed(args.steps)
    replays_timed = eng.graph_replays - replays0

    # ---- p50 per-utterance latency: completion time of the batch that contains the utterance ---------------------
    lat = []
    for _ in range(min(5, max(2, args.steps))):
        fence()
        t1 = time.perf_counter()
        step()
        fence()
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat))
    # ---- the same lone batch on the critical path of a PREDICTED-duration run: the duration predictor, the device->host read of its
    # result and the wait for it precede everything else (the forced-duration runs above skip the read: durations are known on the
    # host).  Shapes stay the forced ones (stn_set_duration_read: synthetic weights predict meaningless lengths).
    lat_pred = []
    eng.set_duration_read(True)
    for _ in range(min(5, max(2, args.steps)) + 1):
        fence()
        t1 = time.perf_counter()
        step()
        fence()
        lat_pred.append((time.perf_counter() - t1) * 1e3)
    eng.set_duration_read(False)
    p50_pred = float(np.median(lat_pred[1:]))

    # ---- the gathered waveforms against the engine's own PCM fetch (N > 1 code path): rank 0's block of the last gather must be the
    # bytes stn_batch_fetch_pcm16 returns for the same resident batch, and every rank's block must have arrived with its shape ---------
    gather_check = None
    if use_dist:
        import zlib
        fence()
        k_last = (step_no[0] - 1) & 1
        wavs, durs_g = gather_buf["plan"].result(k_last)
        if rank == 0:
            pcm_ref, dur_ref = eng.batch_fetch_pcm16()
            got = wavs[0].cpu().numpy()
            gather_check = {"equal_rank0_block": bool(np.array_equal(got, pcm_ref)), "crc32_gathered": zlib.crc32(got.tobytes()),
                            "crc32_fetch_pcm16": zlib.crc32(pcm_ref.tobytes()), "durations_equal": bool(np.allclose(durs_g[0].cpu().numpy(), dur_ref, rtol=0, atol=1e-6)),
                            "blocks": [list(w.shape) for w in wavs], "nonzero_blocks": [bool(int(w.abs().max()) > 0) for w in wavs], "bytes_per_gather": int(sum(w.numel() for w in wavs) * 2)}

    # ---- kernel timing, separate from the timed region: events ride on dispatch packets and cannot live inside a graph, so these
    # passes run eager.  One fully instrumented step finds the dominant family; then the same K steps with events on every 7th
    # launch of that family give its live average duration (roofline.achieved) -------------------------------------------------
    dominant, fam_stats, roof, eager = None, {}, None, None
    sample_every = 1
    if not args.no_profile:
        eng.profile_filter(None)
        eng.profile_enable(True)
        eng.profile_reset()
        step()
        fence()
        fam_stats = eng.profile()
        eng.profile_enable(False)
        dominant = max(fam_stats, key=lambda k: fam_stats[k]["ms"])
        eng.profile_filter(dominant)
        # an event-carrying launch does not overlap its neighbours (~4 us each); 7 is coprime to the launches per step, so the sample
        # walks through every layer position
        sample_every = 7 if fam_stats[dominant]["launches"] >= 28 else 1
        eng.profile_sample(sample_every)
        eng.profile_reset()
        eng.profile_enable(True)
        el_e = timed(args.steps)
        st = eng.profile()[dominant]
        eng.profile_enable(False)
        eager = {"ms_per_step": round(el_e / args.steps * 1e3, 3),
                 "note": "the same K steps launched eagerly with HIP events on every %d-th launch of the dominant family (the pass roofline.avg_launch_us comes from)" % sample_every}
        avg_ms = st["ms"] / max(st["launches"], 1)
        flops_per_launch = st["flops"] / max(st["launches"], 1)
        bytes_per_launch = st["bytes"] / max(st["launches"], 1)
        is_gemm = "gemm" in dominant or "attention" in dominant or "ffn" in dominant or "xattn" in dominant
        if is_gemm:
            peak = 2500.0 if args.dtype in ("bf16", "f16") else 157.3
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4))
        else:
            ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4))
        roof.update(kernel=dominant, avg_launch_us=round(avg_ms * 1e3, 2), launches_timed=st["launches"], sampled_every=sample_every,
                    algorithmic_flops_per_launch=flops_per_launch, algorithmic_bytes_per_launch=bytes_per_launch)
        pmc, stale = _pmc_traffic(dominant)
        roof["traffic"] = pmc.get("hbm_bytes_per_launch") if pmc and not stale else None
        if stale:  # the committed counters were taken on other kernel sources: not this build's traffic
            roof["profile_stale"] = True
            roof["profile_stale_note"] = "profiles/pmc_traffic.json was recorded at another source hash (tools/src_hash.py); re-run tools/profile_round.sh"
            pmc = None
        mu = None if stale else _pmc_mfma(dominant)
        if mu:
            roof["mfma_util_pmc"] = round(mu["mfma_util"], 4)
            roof["mfma_util_source"] = ("profiles/mfma_util.json: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of this kernel in a "
                                        "rocprofv3 --pmc pass of this command (matrix-pipe busy share of the dispatch at the clock the chip held; a lower "
                                        "bound on dispatches this short)")
        if pmc:
            roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
            if "rocprof_avg_us" in pmc:
                roof["rocprof_avg_us"] = round(pmc["rocprof_avg_us"], 2)
                roof["timing_note"] = ("avg_launch_us: HIP events attached to each launch's dispatch packet on the engine's stream "
                                       "(hipExtLaunchKernelGGL start/stop = the kernel's own begin/end), live over K eagerly launched steps, where "
                                       "its un-instrumented neighbours may overlap its first and last microsecond (a fully instrumented "
                                       "step, `roofline_other`, serialises every launch and reads ~8 % lower); "
                                       "rocprof_avg_us: the same kernel in the committed rocprofv3 kernel trace")

    # ---- host-to-host: text frontend -> pinned upload -> synthesis -> 16-bit PCM in host memory, per batch --------------------
    hostrec = None
    if not args.no_host_loop:
        hostrec = host_loop(eng, up, texts, sttl, sdp, durs, mine, args, fence)

    audio_per_step = audio_total(durs)
    value = audio_per_step * args.steps / elapsed

    if hostrec and use_dist:  # every rank takes part in the max over ranks
        t = torch.tensor([hostrec["elapsed"]], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hostrec["elapsed"] = float(t.item())

    # ---- the other scaling mode, in the same run (N > 1): north_star words the job as "a 128-utterance batch at 1, 2, 4 and 8
    # MI355X" (strong: 128 in all), the driver's efficiency curve wants fixed work per GPU (weak: 128 each).  `value` is --scaling's;
    # the other figure is reported beside it ------------------------------------------------------------------------------------
    other_scaling = None
    if world > 1:
        om = "strong" if args.scaling == "weak" else "weak"
        so = make_shard(om)
        fence()
        eng.batch_upload(so["ids"], so["mask"], so["sttl"], so["sdp"], duration_override=so["durs"], utt_ids=so["mine"])
        cur["durs"] = so["durs"]
        gather_buf.clear()
        for _ in range(3):
            step()
        el_o = timed(args.steps)
        a_o = audio_total(so["durs"])
        other_scaling = {"scaling": om, "value": round(a_o * args.steps / el_o, 1), "ms_per_step": round(el_o / args.steps * 1e3, 3),
                         "global_batch": so["n_total"], "batch_per_gpu": len(so["texts"]), "audio_sec_per_step": round(a_o, 2)}
        fence()
        eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs, utt_ids=mine)
        cur["durs"] = durs
        gather_buf.clear()

    if rank == 0:
        B, L, W = eng.batch_dims()
        out = {
            "metric": "audio-sec/sec (RTF^-1), 66M model, batch=128 per GPU" if args.scaling == "weak" else "audio-sec/sec (RTF^-1), 66M model, batch=128 in all (strong scaling)",
            "value": round(value, 1), "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_run": n_warm,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"C4: batch={args.batch} mixed-length (4..48 words) English utterances per GPU, "
                                    f"{args.total_step} Euler steps, {args.dtype} (BASELINE.json configs[3])") if args.mixed else
                                   (f"C3: batch={args.batch} {args.words}-word English utterances {'per GPU' if args.scaling == 'weak' else 'in all (' + str(len(texts)) + ' on rank 0)'}, "
                                    f"{args.total_step} Euler steps, {args.dtype} (BASELINE.json configs[2])"),
                       "batch_per_gpu": len(texts), "global_batch": n_total, "total_step": args.total_step,
                       "durations": "forced to n_chars / 15 s before /speed (synthetic weights predict meaningless lengths); the duration predictor still runs, only its device->host read is skipped",
                       "speed": args.speed, "params": eng.param_count, "text_tokens_max": int(ids.shape[1]),
                       "latent_frames_max": L, "audio_sec_per_step": round(audio_per_step, 2),
                       "weights": "synthetic (descriptor include/stn_arch.h, seed 7)",
                       "in_flight_batches": 1,
                       "timed_region": f"{args.steps} syntheses of the resident batch as hipGraph replays of the post-duration pipeline ({replays_timed} replays counted); "
                                       f"{n_warm} untimed warm-up steps (eager, capture, first replay); every synthesis runs the duration predictor, the text encoder, the noise, "
                                       "all Euler steps and the vocoder on the uploaded inputs — the one thing computed once per (total_step, batch size, weights) instead of per "
                                       "synthesis is the estimator's time conditioning, which depends on nothing the caller uploads",
                       "warmup_steps_run": n_warm,
                       "hip_built": rt_info["hip_built"], "hip_runtime": rt_info["hip_runtime"], "torch_in_process": bool(rt_info["torch_preloaded"]),
                       "parallelism": f"utterance-sharded x{world}, RCCL gather of int16 PCM to rank 0 overlapped with the next step" if world > 1 else "single GPU"},
            "p50_latency_ms": round(p50, 3),
            "lone_batch_predicted_path": {"p50_ms": round(p50_pred, 3),
                                          "note": "one batch alone with the duration predictor, the device->host read of its durations and the wait for it on the critical "
                                                  "path, as when durations are predicted (the other figures run with durations known on the host: no read); shapes forced"},
            "latency_note": "per-utterance latency = completion time of its 128-utterance batch (submit -> waveform in HBM), graph replay",
            "graph_replays_in_timed_region": replays_timed,
            "roofline": roof,
        }
        if eager:
            out["eager_sampled"] = eager
        if other_scaling:
            out["other_scaling"] = other_scaling
        if gather_check:
            out["gather_check"] = gather_check
        if hostrec:
            el_h = hostrec.pop("elapsed")
            out["value_host"] = round(audio_per_step * args.steps / el_h, 1)
            out["ms_per_step_host"] = round(el_h / args.steps * 1e3, 3)
            out["p50_latency_host_ms"] = hostrec.pop("p50")
            el2 = hostrec.pop("elapsed_resident_two_in_flight")
            if world == 1:
                out["two_in_flight"] = {
                    "ms_per_step": round(el2 / args.steps * 1e3, 3), "value": round(audio_per_step * args.steps / el2, 1),
                    "note": "the same K syntheses of the resident batch issued alternately on two engine handles (two hipGraph replays in flight "
                            "on two streams: the estimator phase of one batch fills the CUs the vocoder phase of the other leaves idle); NOT `value`, "
                            "which runs one batch at a time; per-batch latency roughly doubles in this mode"}
            out["host_loop"] = hostrec
        if fam_stats:
            tot = sum(v["ms"] for v in fam_stats.values())
            top = sorted(fam_stats.items(), key=lambda kv: -kv[1]["ms"])[:8]
            out["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in top}
            # secondary rooflines from the one fully-profiled step (per-kernel HIP-event spans, as above):
            # GEMM / attention families against dense MFMA peak, the conv / norm families against HBM peak
            other = {}
            for k, v in top:
                ms = v["ms"] / max(v["launches"], 1)
                if "gemm" in k or "attention" in k or "ffn" in k or "xattn" in k:
                    peak = 2500.0 if args.dtype in ("bf16", "f16") else 157.3
                    a_ = v["flops"] / max(v["launches"], 1) / (ms * 1e-3) / 1e12
                    other[k] = {"bound": "mfma", "achieved": round(a_, 1), "unit": "TFLOP/s", "frac": round(a_ / peak, 4), "avg_us": round(ms * 1e3, 1)}
                else:
                    a_ = v["bytes"] / max(v["launches"], 1) / (ms * 1e-3) / 1e9
                    other[k] = {"bound": "hbm", "achieved": round(a_, 1), "unit": "GB/s", "frac": round(a_ / 8000.0, 4), "avg_us": round(ms * 1e3, 1)}
            out["roofline_other"] = other
            # HBM-roofline fraction of the vocoder STAGE under the fused-ideal byte model (each block reads
            # its input once and writes its output once), with the stage's MFMA utilisation beside it.  Low by design
            # for a compute-dense stage: the fused-ideal vocoder sits ~6x above the ridge point.
            vo_ms = sum(v["ms"] for k, v in fam_stats.items() if k.startswith("vo."))
            if vo_ms > 0:
                es = 2 if args.dtype in ("bf16", "f16") else 4
                frames = B * L * arch.chunk_compress_factor
                bytes_frame = es * arch.latent_dim + arch.vo_blocks * 2 * arch.vo_dim * es + arch.base_chunk_size * 4
                flops_frame = (arch.vo_blocks * (2 * arch.vo_dim * arch.vo_hidden * 2 + 2 * arch.vo_kernel * arch.vo_dim)
                               + 2 * arch.vo_in_kernel * arch.latent_dim * arch.vo_dim + 2 * arch.vo_dim * arch.base_chunk_size)
                peak = 2500.0 if args.dtype in ("bf16", "f16") else 157.3
                gbs = bytes_frame * frames / (vo_ms * 1e-3) / 1e9
                tfs = flops_frame * frames / (vo_ms * 1e-3) / 1e12
                out["vocoder_stage"] = {"frames": frames, "ms": round(vo_ms, 3), "bytes_per_frame_fused_ideal": bytes_frame,
                                        "flops_per_frame": flops_frame,
                                        "hbm": {"achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)},
                                        "mfma": {"achieved": round(tfs, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tfs / peak, 4)},
                                        "note": "event-timed spans of every vo.* launch in one fully profiled step"}
            st_ms = {}
            for k, v in fam_stats.items():
                st_ms[k.split(".")[0]] = st_ms.get(k.split(".")[0], 0.0) + v["ms"]
            out["stage_ms_fully_profiled_step"] = {k: round(v, 3) for k, v in sorted(st_ms.items())}
        if world == 1 and not args.no_b1 and not args.mixed:
            out["single_utterance"] = single_utterance(eng, arch, up, args)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(arch, texts, ids, mask, sttl, sdp, durs, args)
        log_path = os.environ.get("STN_LAUNCH_LOG")
        if world == 1 and log_path:
            # profiler runs (tools/profile_round.sh): the LAST step of the process is one fully tagged step of the bench batch,
            # and its (family, kernel) sequence is written out for tools/pmc_families.py to align rocprofv3's dispatch rows with
            eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs, utt_ids=mine)
            eng.set_graph_mode(False)
            eng.batch_run(args.total_step, args.speed, 1234)
            eng.sync()
            # family tags only: a filter that no family matches keeps every launch free of events (the profiler times the kernels itself;
            # event-carrying dispatch packets — hipExtLaunchKernelGGL — under rocprofv3's interception crash inside the ROCm 7.2 runtime)
            eng.profile_filter("-")
            eng.profile_sample(1)
            eng.profile_enable(True)
            eng.launch_log_enable(True)
            eng.profile_reset()
            eng.batch_run(args.total_step, args.speed, 1234)
            eng.sync()
            with open(log_path, "w") as f:
                json.dump({"entries": eng.launch_log()}, f)
            eng.launch_log_enable(False)
            eng.profile_enable(False)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        eng.sync()
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(kernel):
    """(entry, stale): HBM bytes per launch of the dominant kernel from an offline `rocprofv3 --pmc` pass of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_families.py), and whether that pass was taken on other kernel sources than
    the tree's (the file records tools/src_hash.py's hash); (None, False) until such a pass exists."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from src_hash import source_sha
            return d.get(kernel), d.get("_source_sha") != source_sha(ROOT)
        except Exception:
            return None, False
    return None, False


def _pmc_mfma(kernel):
    """Matrix-pipe utilisation of the dominant kernel from the committed PMC pass (tools/pmc_families.py); None if absent."""
    p = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(kernel)
        except Exception:
            return None
    return None


def arch_device(eng):
    return int(getattr(eng, "device", 0))


def host_loop(eng, up, texts, sttl, sdp, durs, utt_ids, args, fence):
    """K batches host-to-host, as _infer's contract has it (/root/reference/cpp/helper.cpp:469-683: strings and host tensors in,
    a host waveform out): per batch the C++ text frontend (text -> ids, mask), the upload of ids / mask / styles from page-locked
    host memory, the synthesis, and the 16-bit PCM (what writeWavFile stores, cpp/helper.cpp:986-987) back in page-locked host
    memory — the device->host copy of batch i runs on a second stream under the upload and synthesis of batch i+1."""
    from supertonic_amd import binding
    B = len(texts)
    langs = ["en"] * B
    ids0, mask0 = up(texts, langs)
    p_ids = binding.pinned_array(ids0.shape, np.int64)
    p_mask = binding.pinned_array(mask0.shape, np.float32)
    p_ttl = binding.pinned_array(sttl.shape, np.float32)
    p_dp = binding.pinned_array(sdp.shape, np.float32)
    p_ttl[...] = sttl
    p_dp[...] = sdp
    # Two engine handles on the device (weights twice: 2 x 134 MB of 288 GB) take the batches alternately: stn_batch_upload waits
    # for ITS handle's previous batch only, so batch i+1 is uploaded and queued while batch i computes and the GPU goes from one
    # batch straight into the next.  (With one handle the upload waits for the running batch: ~0.7 ms of idle GPU per batch.)
    eng2 = binding.Engine(arch_device(eng), args.dtype)
    eng2.load_synthetic(eng.arch, 7)
    engs = (eng, eng2)
    t_front = [0.0]
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(1)  # the text of batch i+1 is prepared while the GPU runs batch i (the ctypes call drops the GIL)

    def front():
        t0 = time.perf_counter()
        r = up(texts, langs)
        t_front[0] += time.perf_counter() - t0
        return r

    def one(k, fut=None, more=True):
        e = engs[k]
        ids, mask = fut.result() if fut is not None else front()
        p_ids[...] = ids
        p_mask[...] = mask
        e.batch_upload(p_ids, p_mask, p_ttl, p_dp, duration_override=durs, utt_ids=utt_ids)  # (synchronous copies: the staging is free again)
        nxt = pool.submit(front) if (fut is not None and more) else None
        e.batch_run(args.total_step, args.speed, 1234)
        e.fetch_pcm16_begin(0)
        return nxt

    checksum = 0
    for i in range(6):  # warm-up of both handles: eager, capture, replay
        one(i & 1)
        engs[i & 1].fetch_pcm16_end(0, copy=False)
    fence()
    eng2.sync()
    t_front[0] = 0.0
    t0 = time.perf_counter()
    fut = pool.submit(front)
    for i in range(args.steps):
        fut = one(i & 1, fut, i + 1 < args.steps)
        if i:
            pcm, _ = engs[(i - 1) & 1].fetch_pcm16_end(0, copy=False)
            checksum ^= int(pcm[0, 0])  # the host owns the waveform here
    pcm, d = engs[(args.steps - 1) & 1].fetch_pcm16_end(0, copy=False)
    fence()
    eng2.sync()
    elapsed = time.perf_counter() - t0
    pool.shutdown()
    n_pcm = int(pcm.size)
    lat = []
    for _ in range(5):  # one batch alone, nothing to overlap with
        fence()
        t1 = time.perf_counter()
        one(0)
        eng.fetch_pcm16_end(0, copy=False)
        lat.append((time.perf_counter() - t1) * 1e3)
    # the same two handles with RESIDENT inputs (no text frontend, no PCIe): K syntheses, two batches in flight
    for e in engs:
        e.batch_upload(p_ids, p_mask, p_ttl, p_dp, duration_override=durs, utt_ids=utt_ids)
    for i in range(6):
        engs[i & 1].batch_run(args.total_step, args.speed, 1234)
    fence()
    eng2.sync()
    t2 = time.perf_counter()
    for i in range(args.steps):
        engs[i & 1].batch_run(args.total_step, args.speed, 1234)
    fence()
    eng2.sync()
    el2 = time.perf_counter() - t2
    del eng2
    h2d = int(p_ids.nbytes + p_mask.nbytes + p_ttl.nbytes + p_dp.nbytes + durs.nbytes + 8 * B)
    return {"elapsed": elapsed, "p50": round(float(np.median(lat)), 3), "elapsed_resident_two_in_flight": el2,
            "in_flight_batches": 2,
            "text_frontend_ms_per_step": round(t_front[0] / args.steps * 1e3, 3),
            "pcie_bytes_per_step": {"h2d": h2d, "d2h": n_pcm * 2 + 4 * B},
            "note": "per batch: text -> ids (C++ frontend), upload from pinned memory, synthesis, int16 PCM into pinned host memory; the "
                    "device->host copy of batch i and the text frontend of batch i+1 (one worker thread) overlap the synthesis (stn_batch_fetch_pcm16_begin/_end); "
                    "two engine handles alternate so that batch i+1 is uploaded, queued AND STARTED while batch i computes (two batches in flight: this rate "
                    "is bounded by `two_in_flight`, not by `value`, which runs one batch at a time); "
                    "every batch is tokenised anew; p50_latency_host_ms = one batch alone, nothing overlapped"}


def single_utterance(eng_bf16, arch, up, args):
    """BASELINE.json configs[1] (C2) as a secondary record: the fixed 10-word sentence, batch of one, default steps; p50 / p90 of
    200 resident-batch syntheses after 20 warm-ups (graph replay from the third call on), fp32 and the bench dtype."""
    from supertonic_amd import binding, workload
    text = [workload.C1_SENTENCE]
    ids, mask = up(text, ["en"])
    sttl, sdp = workload.synthetic_styles(arch, [0])
    durs = workload.forced_durations(text)
    out = {"workload": f"C2: '{text[0]}' ({len(text[0])} chars, {float(durs[0] / args.speed):.2f} s of audio), batch 1, {args.total_step} Euler steps, resident inputs"}
    for name in ("f32", args.dtype):
        if name in out:
            continue
        eng = eng_bf16 if name == args.dtype else binding.Engine(0, name)
        if eng is not eng_bf16:
            eng.load_synthetic(arch, 7)
        eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs, utt_ids=[0])
        for _ in range(20):
            eng.batch_run(args.total_step, args.speed, 1234)
        eng.sync()
        lat = []
        for _ in range(200):
            t1 = time.perf_counter()
            eng.batch_run(args.total_step, args.speed, 1234)
            eng.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        out[name] = {"p50_ms": round(float(np.percentile(lat, 50)), 3), "p90_ms": round(float(np.percentile(lat, 90)), 3),
                     "audio_sec_per_sec": round(float(durs[0] / args.speed) / (float(np.percentile(lat, 50)) * 1e-3), 1)}
        if eng is not eng_bf16:
            eng.close()
    return out


def cpu_baseline(arch, texts, ids, mask, sttl, sdp, durs, args):
    """The CPU oracle (plain-C fp32 restatement, OpenMP) timed on a bounded sample of the SAME workload: 1 warm-up + 3 timed
    runs, the median reported.  kind = "port": the reference's ORT-CPU path cannot run here (no ONNX
    Runtime, no ONNX graphs)."""
    from oracle import neural_ref
    n = min(args.cpu_sample, len(texts))
    lens = mask[:n].sum(axis=(1, 2)).astype(int)
    lt = int(lens.max())
    ref = neural_ref.RefModel(arch, 7)
    times, d = [], None
    for run in range(4):
        t0 = time.perf_counter()
        _, d = ref.synthesize(ids[:n, :lt], mask[:n, :, :lt], sttl[:n], sdp[:n], args.total_step, args.speed,
                              lambda B, D, L: neural_ref.randn(1234, B, D, L), duration_override=durs[:n])
        if run:
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"value": round(float(d.sum()) / dt, 2), "unit": "audio-sec/sec", "cores": neural_ref.threads(), "kind": "port",
            "sample": f"first {n} utterances of the same batch as one padded batch, fp32, {args.total_step} Euler steps; 1 warm-up + 3 timed "
                      f"runs, median {dt:.1f} s wall ({min(times):.1f}-{max(times):.1f}) on {neural_ref.threads()} OpenMP threads (oracle/stn_ref.c)"}


if __name__ == "__main__":
    main()
