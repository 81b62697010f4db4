#!/usr/bin/env python3
"""Benchmark of the classify+segment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without a launcher (no WORLD_SIZE in the environment) the
script starts its N ranks itself as fresh child processes, before anything touches a GPU, relays
rank 0's JSON line and exits non-zero if a rank fails.

Workload: synthetic 1280x720 u8 frames resident in HBM, classified in batches of 256 frames
(fp32) and segmented.  One STEP = one whole pass over this rank's frame range: every batch
through the fused resize+prod_net forward, then the tail - the per-frame (class, max) streams
are all-gathered (N > 1: the ONE collective of the job), run-length encoded on the GPU, repaired
and merged on the host and rendered to CSV bytes.  Everything is inside the timed region.

  default (weak scaling)    every rank owns a 10 000-frame range of an (N x 10 000)-frame game
  --total-frames T (strong) the game has T frames; rank r owns the r-th contiguous range of
                            ceil(T / N) frames (T = 200 000 on 8 GPUs is the full-game shape)
  --independent-clips       every rank segments its own clip: no gather, no collective

A range that does not fit the per-rank HBM budget (--resident-frames) is classified in chunks; the
regeneration of the next chunk's frames is then excluded from the timed region (it stands in for
frames that would already be resident) and the JSON says so in config.timed_region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))

CLIP_FRAMES = 10_000
BATCH = 256
H0, W0 = 720, 1280
SEED = 1234

# algorithmic work per frame, independent of the source resolution (2 flops per MAC):
#   block 1: 144*256 px * 48 ch * 27 ; block 2: 48*85 * 48 * 432 ; block 3: 16*28 * 48 * 432 ; FC 768*32 + 32*3
FLOPS_CONV1 = 2 * 144 * 256 * 48 * 27
FLOPS_CONV2 = 2 * 48 * 85 * 48 * 432
FLOPS_CONV3 = 2 * 16 * 28 * 48 * 432
FLOPS_FC = 2 * (768 * 32 + 32 * 3)
FLOPS_FRAME = FLOPS_CONV1 + FLOPS_CONV2 + FLOPS_CONV3 + FLOPS_FC          # 283 386 048


def compulsory_bytes(h0, w0, net_h):
    """Source rows the resize actually samples x row bytes + 12 B of logits (552 972 B at 720p: an odd
    integer scale is a pure point sample, anything else touches two source rows per output row)."""
    point = h0 % net_h == 0 and (h0 // net_h) % 2 == 1
    return min(h0, net_h if point else 2 * net_h) * w0 * 3 + 12


PEAK_F32_MFMA_TFLOPS = 157.3                                                # dense fp32 matrix peak, MI355X
PEAK_BF16_MFMA_TFLOPS = 2500.0                                              # dense bf16 matrix peak (--dtype bf16 only)
PEAK_HBM_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is a whole pass over the rank's range (10 000 frames = 40 batches, ~22 ms), so a few
    # untimed passes are more than the ~20 ms of load the chip needs to reach its steady clocks
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clip-frames", type=int, default=CLIP_FRAMES,
                    help="weak scaling: frames every rank owns")
    ap.add_argument("--total-frames", type=int, default=None,
                    help="strong scaling: frames of the whole game, split into contiguous per-rank ranges")
    ap.add_argument("--resident-frames", type=int, default=None,
                    help="most frames a rank keeps in HBM at once (default: 160 GB worth, divided by the "
                         "ranks that share a GPU); a longer range is classified in chunks")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--height", type=int, default=H0, help="source frame height (default: the 720p headline)")
    ap.add_argument("--width", type=int, default=W0, help="source frame width")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="with --stage-timing dominant: bracket every n-th launch of the dominant kernel (an event "
                         "pair costs about 4 us of stream time; the roofline's launch time is the average over "
                         "that sample: 200 of the default run's 800 launches)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16_split"], default="f32",
                    help="f32 is the headline (the reference's arithmetic); bf16 / bf16_split run the library's "
                         "bf16 matrix-core modes - separate, non-headline measurements")
    ap.add_argument("--independent-clips", action="store_true",
                    help="multi-GPU: every rank segments its own clip (no stream gather, no collective at all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="geometries that need a real resize (not 1280x720): one fused call per batch instead of "
                         "the resize of batch i+1 on a second stream beside the classification of batch i")
    ap.add_argument("--streams", type=int, choices=[1, 2], default=1,
                    help="2: batches alternate between two library handles on two HIP streams, so one batch's "
                         "kernel start-ups and tails fill with the other's work (the bracketed block-2 time then "
                         "includes sharing the chip, so the per-kernel roofline figure is not comparable)")
    ap.add_argument("--stage-timing", choices=["dominant", "all", "off"], default="dominant",
                    help="which kernels are bracketed with HIP events inside the timed region: the dominant "
                         "one (block 2, what the roofline object reports), every stage (each event pair costs "
                         "a few microseconds of stream time per launch), or none")
    ap.add_argument("--cpu-sample-batches", type=int, default=40,
                    help="128-frame batches the CPU baseline times (40 = 5120 frames, about 11 s on 16 host threads)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N without a launcher
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Starts n fresh rank processes of this script (the parent never touches a GPU), relays rank
    0's stdout, returns the job's exit code: 0 only if every rank exits 0."""
    import threading
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))

    def relay():
        # only the JSON line belongs on stdout; whatever else a library prints there goes to stderr
        for line in procs[0].stdout:
            text = line.decode("utf-8", "replace")
            out = sys.stdout if text.lstrip().startswith("{") else sys.stderr
            out.write(text)
            out.flush()

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    for p in procs:                      # the exact processes started above, nothing else
        if p.poll() is None:
            p.kill()
        p.wait()
    reader.join(timeout=10)
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        return failed[1] if failed[1] > 0 else 1
    return 0


# ---------------------------------------------------------------------------------------------
def cpu_baseline(frames_u8_cpu, n_warm_batches, n_timed_batches):
    """The CPU oracle (port of the reference's CPU path) on a bounded sample of the same clip:
    numpy resize + /255, the traced torch prod_net at batch 128 (segment_video.py:104 default),
    then segmentation.  Returns the dict reported as `cpu_baseline`."""
    import torch
    from oracle import net as onet, preprocess, segmentation
    bs = 128
    w = onet.load_blob()
    traced = onet.traced_cpu_net(w)
    frames = frames_u8_cpu.numpy()

    def run(a, b):
        x = torch.from_numpy(preprocess.preprocess_batch(frames[a:b]))
        with torch.no_grad():
            return traced(x)

    for i in range(n_warm_batches):
        run(i * bs, (i + 1) * bs)
    t0 = time.perf_counter()
    outs, t_net = [], 0.0
    a0 = n_warm_batches * bs
    for i in range(n_timed_batches):
        a = a0 + i * bs
        x = torch.from_numpy(preprocess.preprocess_batch(frames[a:a + bs]))
        t1 = time.perf_counter()
        with torch.no_grad():
            outs.append(traced(x))
        t_net += time.perf_counter() - t1
    scores = torch.cat(outs).numpy()
    try:
        segmentation.segment_clip(scores)
    except IndexError:
        pass
    dt = time.perf_counter() - t0
    n = n_timed_batches * bs
    return {
        "value": round(n / dt, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
        "net_only_frames_per_s": round(n / t_net, 2), "host_cpus": os.cpu_count(),
        "sample": f"{n} frames (batches of {bs}) of the same {frames.shape[2]}x{frames.shape[1]} clip after "
                  f"{n_warm_batches} warm-up batches: numpy resize + /255, torch.jit-traced fp32 prod_net, segmentation",
    }


def rehearse_plumbing(args, world, rank):
    """CUTDET_BENCH_REHEARSE=plumbing: no GPU work at all.  The ranks rendezvous over gloo, exchange
    the (class, max) streams the run table prescribes with the job's one all-gather, and rank 0
    prints a line without a metric value.  It exists so that the launch / rendezvous / relay / exit
    code logic can be exercised on a machine without a GPU."""
    import torch
    import torch.distributed as dist
    from cutdet import synth as csynth
    from cutdet.pipeline import gather_streams, shard_range
    if os.environ.get("CUTDET_BENCH_FAIL_RANK") == str(rank):     # for the test of the exit-code relay
        sys.exit(7)
    if world > 1:
        dist.init_process_group("gloo")
    total = args.total_frames if args.total_frames else world * args.clip_frames
    lo, hi = shard_range(total, rank, world)
    classes = csynth.classes_from_runs(csynth.run_table(total, SEED))
    cls = torch.tensor(classes[lo:hi], dtype=torch.int64)
    mx = torch.full((hi - lo,), 1.0 + rank, dtype=torch.float32)
    g_cls, g_mx = gather_streams(cls, mx, dist.group.WORLD if world > 1 else None, total)
    ok = g_cls.tolist() == classes and g_mx.shape[0] == total
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal: launch and stream gather only (no GPU work)", "value": None,
                          "unit": "frames/s", "n_gpus": world, "n_ranks_seen": int(seen.item()),
                          "gathered_frames": int(g_cls.shape[0]), "gather_ok": bool(ok)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


class RegionTimer:
    """Wall time of the timed region: one interval, or several when chunk regeneration is excluded.
    Every interval starts and ends with a device synchronisation (and a barrier across ranks)."""

    def __init__(self, sync):
        self.sync, self.total, self.t0, self.intervals = sync, 0.0, None, 0

    def start(self):
        self.sync()
        self.t0 = time.perf_counter()

    def stop(self):
        self.sync()
        self.total += time.perf_counter() - self.t0
        self.t0 = None
        self.intervals += 1


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("CUTDET_BENCH_REHEARSE") == "plumbing":
        return rehearse_plumbing(args, world, rank)

    import torch
    import torch.distributed as dist

    # CUTDET_BENCH_BACKEND=gloo + CUTDET_BENCH_ONE_GPU=1 rehearse the multi-rank path on a 1-GPU box
    backend = os.environ.get("CUTDET_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("CUTDET_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {dev_index} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cutdet import api, synth as csynth
    from cutdet.pipeline import ClipClassifier, shard_range

    eng = api.default_engine(dev, args.dtype)
    peak_tflops = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    bs = args.batch
    strong = args.total_frames is not None
    total = args.total_frames if strong else world * args.clip_frames
    gather = world > 1 and not args.independent_clips
    # this rank's contiguous frame range [lo, hi) of the total-frame game
    lo, hi = shard_range(total, rank, world)
    n_local = hi - lo
    classes_all = csynth.classes_from_runs(csynth.run_table(total, SEED))
    frame_bytes = args.height * args.width * 3
    # Frames a rank keeps in HBM at once.  Every rank cuts its range at the same multiples of
    # `resident` (computed from the longest range), so that all ranks pause the timer together.
    per = -(-total // world)
    budget = args.resident_frames or int(160e9 / frame_bytes) // (world if one_gpu else 1)
    resident = per if per <= budget else max(bs, budget // bs * bs)
    chunks = [(min(n_local, a), min(n_local, a + resident)) for a in range(0, per, resident)]
    frames = torch.empty((max(1, min(resident, n_local)), args.height, args.width, 3), dtype=torch.uint8, device=dev)
    cls_dev = torch.tensor(classes_all[lo:hi], dtype=torch.uint8, device=dev)

    def fill(chunk):
        a, b = chunk
        if b > a:
            eng.synth_frames(cls_dev[a:b], args.height, args.width, SEED, first_index=lo + a, out=frames[: b - a])

    if chunks:
        fill(chunks[0])
    net_h = int(args.height * (256 / args.width))
    eng.reserve(bs, net_h)
    twin = None
    if args.streams == 2:
        from cutdet.engine import Engine
        twin = Engine(api.default_blob(), dev, args.dtype)
        twin.reserve(bs, net_h)
    clip = ClipClassifier(eng, max(n_local, 1), twin=twin)
    group = dist.group.WORLD if gather else None
    torch.cuda.synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    timer = RegionTimer(sync)

    def one_pass(timed):
        """Classify this rank's whole range, then segment (every rank takes part in the gather;
        rank 0's result is the job's output).  With several chunks the timer pauses while the next
        chunk's frames are regenerated."""
        for k, chunk in enumerate(chunks):
            if one_pass.loaded != k:
                if timed:
                    timer.stop()
                fill(chunk)
                one_pass.loaded = k
                if timed:
                    timer.start()
            a0, b0 = chunk
            if b0 > a0:
                clip.classify_range(frames[:b0 - a0], a0, bs, overlap=not args.no_overlap)
        if not gather:
            return clip.segment(n_local, 100, 10) if n_local else None
        return clip.segment(n_local, 100, 10, group=group, total=total)

    one_pass.loaded = 0          # index of the chunk whose frames are in `frames`

    # ---- warm-up (untimed): W whole passes, so RCCL and every kernel are initialised
    result = None
    for _ in range(args.warmup):
        try:
            result = one_pass(False)
        except IndexError:   # a range too short to hold one full-length shot
            pass
    torch.cuda.synchronize()

    # ---- timed region
    for e in [eng] + ([twin] if twin is not None else []):
        if args.stage_timing == "dominant":
            e.timing_enable(True, stages=["conv2"], every=args.timing_every)
        else:
            e.timing_enable(args.stage_timing == "all")
    timer.start()
    for _ in range(args.steps):
        result = one_pass(True)
    timer.stop()
    dt = timer.total
    timing = eng.timing_read()
    eng.timing_enable(False)
    if twin is not None:
        for k, (cnt, ms_) in twin.timing_read().items():
            timing[k] = (timing[k][0] + cnt, timing[k][1] + ms_)
        twin.timing_enable(False)

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    seen = torch.ones((1,), dtype=torch.float64, device=t.device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    dt_max = float(t.item())
    frames_per_step = total
    total_frames = frames_per_step * args.steps

    if rank == 0:
        merged = result["merged"]
        n_launch, ms = timing["conv2"]
        # the bracketed launches: every batch of every timed pass, or every timing_every-th one
        every = args.timing_every if args.stage_timing == "dominant" else 1
        launches = [min(b0, a + bs) - a for _ in range(args.steps) for a0, b0 in chunks for a in range(a0, b0, bs)]
        sampled = launches[::every]
        assert n_launch in (0, len(sampled)) or args.streams == 2, (n_launch, len(sampled))
        frames_per_launch = sum(sampled) / max(1, len(sampled))
        conv2_ms = ms / max(1, n_launch)
        achieved = FLOPS_CONV2 * frames_per_launch / (conv2_ms * 1e-3) / 1e12 if n_launch else 0.0
        stages = {k: {"launches": v[0], "ms_total": round(v[1], 3)} for k, v in timing.items() if v[0]}
        traffic = None      # measured separately under rocprofv3 --pmc; read back if the summary is present
        try:
            with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
                tj = json.load(f)
            if args.dtype == "f32" and (args.height, args.width, bs) == (H0, W0, BATCH):
                traffic = tj["conv2_kernel"]["traffic_bytes"]
            else:
                traffic = tj["other_configs"][f"{args.dtype}_{args.width}x{args.height}_b{bs}"]["conv2_kernel"]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        fps = total_frames / dt_max
        bytes_frame = compulsory_bytes(args.height, args.width, net_h)
        arith = {"f32": "fp32", "bf16": "bf16-matrix-core", "bf16_split": "split-bf16 (hi+lo, 4 products) matrix-core"}[args.dtype]
        out = {
            "metric": f"frames/sec classified+segmented, {args.width}x{args.height} RGB",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_max * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"{total}-frame synthetic {args.width}x{args.height} u8 game split into {world} contiguous "
                                    f"range(s)" if strong else
                                    f"{args.clip_frames} synthetic {args.width}x{args.height} u8 frames per GPU") +
                                   f" resident in HBM, batch={bs}, {arith} prod_net, fused resize+classify, GPU run-length "
                                   f"encode, host glue/merge, CSV; one step = one pass over the range(s)",
                       "frames_per_step": frames_per_step, "frames_per_gpu_timed": n_local * args.steps,
                       "segments_out": int(len(merged["start_frames"])),
                       "n_ranks_seen": int(seen.item()),
                       "timed_region": "one interval" if timer.intervals == 1 else
                                       f"{timer.intervals} intervals: regeneration of non-resident frame chunks excluded",
                       "resident_frames_per_rank": resident, "streams": args.streams,
                       "stitch": "one all-gather of (class,max) streams over " + ("RCCL" if backend == "nccl" else backend)
                                 if group is not None else "none"},
            "roofline": {
                "bound": "mfma", "kernel": "conv2_kernel", "achieved": round(achieved, 2),
                "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(achieved / peak_tflops, 4),
                "traffic": traffic, "traffic_unit": "bytes per full-batch launch (rocprofv3 PMC, profiles/roofline_traffic.json)",
                "launch_ms": round(conv2_ms, 4), "launches_timed": n_launch, "flops_per_frame": FLOPS_CONV2,
                "frames_per_launch": round(frames_per_launch, 2),
                # the kernel does not compute conv column 84 of 85, which the floor-mode pooling discards:
                # `achieved` counts the algorithmic FLOPs, this is the rate of the MFMAs actually issued
                "issued_frac_of_peak": round(achieved * (84.0 / 85.0) / peak_tflops, 4),
                "whole_net_tflops": round(FLOPS_FRAME * fps / world / 1e12, 2),
                "whole_net_frac_of_mfma_peak": round(FLOPS_FRAME * fps / world / 1e12 / peak_tflops, 4),
                "hbm_compulsory_GBps": round(bytes_frame * fps / world / 1e9, 2),
                "hbm_frac_of_peak": round(bytes_frame * fps / world / 1e9 / PEAK_HBM_GBS, 5),
            },
            "stages": stages,
        }
        if world == 1 and not args.no_cpu_baseline:
            nb = args.cpu_sample_batches
            if one_pass.loaded != 0:
                fill(chunks[0])
            nb = max(1, min(nb, min(resident, n_local) // 128 - 2))
            sample = frames[: (2 + nb) * 128].cpu()
            out["cpu_baseline"] = cpu_baseline(sample, 2, nb)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
