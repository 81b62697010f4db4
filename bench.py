#!/usr/bin/env python3
"""Benchmark of the classify+segment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without a launcher (no WORLD_SIZE in the environment) the
script starts its N ranks itself as fresh child processes, before anything touches a GPU, relays
rank 0's JSON line and exits non-zero if a rank fails.

Headline workload: synthetic 1280x720 u8 frames resident in HBM, classified in batches of 256
frames (fp32) and segmented.  One STEP = one whole pass over this rank's frame range: the range goes
through the fused resize+prod_net forward (one cd_forward_u8_range call per resident chunk: batches
alternate between two streams inside the library), then the tail - the per-frame (class, max)
streams are all-gathered (N > 1: the ONE collective of the job), run-length encoded on the GPU,
repaired and merged on the host and rendered to CSV bytes.  Everything is inside the timed region.

  default (weak scaling)    every rank owns a 10 000-frame range of an (N x 10 000)-frame game
  --total-frames T (strong) the game has T frames; rank r owns the r-th contiguous range of
                            ceil(T / N) frames (T = 200 000 on 8 GPUs is the full-game shape)
  --independent-clips       every rank segments its own clip: no gather, no collective, no process group

Besides the headline the one JSON line carries, measured in the same run and outside the headline's
timed interval:
  roofline            the dominant kernel (block 2) bracketed with HIP events in a separate
                      single-stream pass (one fused call per batch), so that the launch time says
                      something about the kernel and not about what shared the chip with it
  config4   (N = 1)   1920x1080 frames, split-bf16 matrix-core arithmetic, batch 512, with its HBM roofline
  strong_200k (N > 1) the 200 000-frame game split into N contiguous ranges (one all-gather)
  independent_clips (N > 1)  one 10 000-frame clip per GPU, run after the process group is gone
  cpu_baseline (N = 1) the CPU oracle on a bounded sample of the headline clip

A range that does not fit the per-rank HBM budget (--resident-frames) is classified in chunks; the
regeneration of the next chunk's frames is then excluded from the timed region (it stands in for
frames that would already be resident) and the JSON says so in timed_region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))

from cutdet import launch  # noqa: E402  (standard library only: the parent of a self-launch never loads torch)

CLIP_FRAMES = 10_000
BATCH = 256
H0, W0 = 720, 1280
SEED = 1234
GAME_FRAMES = 200_000                      # the full-game shape of the strong-scaling leg
C4 = dict(height=1080, width=1920, batch=512, frames=16384, dtype="bf16_split")     # the 1080p bf16 configuration

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
PEAK_BF16_MFMA_TFLOPS = 2500.0                                              # dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0
# measured on an MI355X of this pool (profiles/r01_microbench.txt, profiles/r03_microbench.txt): back-to-back
# v_mfma_f32_16x16x4_f32 on every SIMD, and a 4 GiB read-only stream (16 bytes per lane)
MEASURED_F32_MFMA_TFLOPS = 156.0
MEASURED_HBM_READ_GBS = 6390.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is a whole pass over the rank's range (10 000 frames = 40 batches, ~21 ms), so a few
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
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16_split"], default="f32",
                    help="f32 is the headline (the reference's arithmetic); bf16 / bf16_split run the library's "
                         "bf16 matrix-core modes - separate, non-headline measurements")
    ap.add_argument("--independent-clips", action="store_true",
                    help="multi-GPU: the headline itself is one clip per rank (no gather, no process group at all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f32-block1-fp32-mfma", action="store_true",
                    help="fp32 only: block 1 on the fp32 matrix instructions too (default: the frame's bytes times three "
                         "exact bfloat16 weight terms on the bf16 matrix pipe; the default run reports this variant as "
                         "the fp32_mfma_block1 object)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the headline (and its roofline pass): no config4 / strong_200k / independent_clips legs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one fused cd_forward_u8 call per batch on one stream instead of one cd_forward_u8_range "
                         "call per chunk (two streams inside the library, resize pass ahead where one is needed)")
    ap.add_argument("--roofline-timing", choices=["dominant", "all"], default="dominant",
                    help="the separate single-stream pass brackets block 2 only, or every stage")
    ap.add_argument("--leg-steps", type=int, default=3, help="timed passes of each extra leg")
    ap.add_argument("--leg-warmup", type=int, default=1, help="untimed passes of each extra leg")
    ap.add_argument("--cpu-sample-batches", type=int, default=40,
                    help="128-frame batches the CPU baseline times (40 = 5120 frames, about 11 s on 16 host threads)")
    return ap.parse_args(argv)


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
    run_dir = [tempfile.mkdtemp(prefix="cutdet_bench_") if rank == 0 else None]
    if world > 1:
        dist.all_reduce(seen)
        dist.broadcast_object_list(run_dir, src=0)
        dist.barrier()
        dist.destroy_process_group()
    # the no-process-group rendezvous of the independent-clips leg: a file barrier and a result exchange
    fb = launch.FileBarrier(rank, world, run_dir[0], timeout=120)
    fb.wait()
    texts = fb.exchange("plumbing", str(hi - lo))
    if rank == 0:
        print(json.dumps({"metric": "rehearsal: launch and stream gather only (no GPU work)", "value": None,
                          "unit": "frames/s", "n_gpus": world, "n_ranks_seen": int(seen.item()),
                          "gathered_frames": int(g_cls.shape[0]), "gather_ok": bool(ok),
                          "file_barrier_ranks": len(texts), "file_barrier_frames": sum(int(t) for t in texts)}))
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


class Job:
    """What every leg shares: the rank's place in the job, its GPU, and how ranks line up."""

    def __init__(self, world, rank, dev, one_gpu, backend):
        self.world, self.rank, self.dev, self.one_gpu, self.backend = world, rank, dev, one_gpu, backend
        self.group = None          # the process group while one exists
        self.files = None          # launch.FileBarrier once the process group is gone
        self._frames = None

    def frames_buffer(self, n, height, width):
        """One u8 buffer reused by the legs (reallocated only when a leg needs a bigger one)."""
        import torch
        need = max(1, n) * height * width * 3
        if self._frames is None or self._frames.numel() < need:
            self._frames = None
            torch.cuda.empty_cache()
            self._frames = torch.empty((need,), dtype=torch.uint8, device=self.dev)
        return self._frames[:need].view(max(1, n), height, width, 3)

    def sync(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            if self.group is not None:
                dist.barrier()
                torch.cuda.synchronize()
            else:
                self.files.wait()

    def max_over_ranks(self, dt):
        """(max of dt over the ranks, number of ranks that took part)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return dt, 1
        if self.group is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
            seen = torch.ones((1,), dtype=torch.float64, device=t.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(seen)
            return float(t.item()), int(seen.item())
        self._tag = getattr(self, "_tag", 0) + 1
        texts = self.files.exchange(f"dt{self._tag}", repr(dt))
        return max(float(t) for t in texts), len(texts)


def run_leg(job, eng, *, total, gather, steps, warmup, height, width, batch, resident_frames=None, overlap=True,
            roofline_pass=None):
    """One measurement: `steps` timed passes over this rank's range of a `total`-frame game after
    `warmup` untimed ones.  gather: the ranks' streams are all-gathered and the whole game segmented
    (needs the process group); otherwise every rank segments its own range as a clip of its own.
    roofline_pass: None, or "dominant" / "all": afterwards, outside the timed interval, one single-stream
    pass over the resident chunk with those stages bracketed by HIP events.
    Returns a dict (the same on every rank where it matters: dt_max, ranks_seen)."""
    import torch
    from cutdet import synth as csynth
    from cutdet.pipeline import ClipClassifier, shard_range
    world, rank, dev = job.world, job.rank, job.dev
    lo, hi = shard_range(total, rank, world)
    n_local = hi - lo
    classes_all = csynth.classes_from_runs(csynth.run_table(total, SEED))
    frame_bytes = height * width * 3
    # Frames a rank keeps in HBM at once.  Every rank cuts its range at the same multiples of
    # `resident` (computed from the longest range), so that all ranks pause the timer together.
    per = -(-total // world)
    budget = resident_frames or int(160e9 / frame_bytes) // (world if job.one_gpu else 1)
    resident = per if per <= budget else max(batch, budget // batch * batch)
    chunks = [(min(n_local, a), min(n_local, a + resident)) for a in range(0, per, resident)]
    frames = job.frames_buffer(min(resident, n_local), height, width)
    cls_dev = torch.tensor(classes_all[lo:hi], dtype=torch.uint8, device=dev)

    def fill(chunk):
        a, b = chunk
        if b > a:
            eng.synth_frames(cls_dev[a:b], height, width, SEED, first_index=lo + a, out=frames[: b - a])

    if chunks:
        fill(chunks[0])
    net_h = int(height * (256 / width))
    eng.reserve(batch, net_h)
    clip = ClipClassifier(eng, max(n_local, 1))
    group = job.group if gather and world > 1 else None
    torch.cuda.synchronize()
    timer = RegionTimer(job.sync)
    state = {"loaded": 0}          # index of the chunk whose frames are in `frames`

    def one_pass(timed, overlapped=overlap):
        """Classify this rank's whole range, then segment (every rank takes part in the gather;
        rank 0's result is the job's output).  With several chunks the timer pauses while the next
        chunk's frames are regenerated."""
        for k, chunk in enumerate(chunks):
            if state["loaded"] != k:
                if timed:
                    timer.stop()
                fill(chunk)
                state["loaded"] = k
                if timed:
                    timer.start()
            a0, b0 = chunk
            if b0 > a0:
                clip.classify_range(frames[:b0 - a0], a0, batch, overlap=overlapped)
        if group is None:
            return clip.segment(n_local, 100, 10) if n_local else None
        return clip.segment(n_local, 100, 10, group=group, total=total)

    result = None
    for _ in range(warmup):
        try:
            result = one_pass(False)
        except IndexError:   # a range too short to hold one full-length shot
            pass
    torch.cuda.synchronize()
    timer.start()
    for _ in range(steps):
        try:
            result = one_pass(True)
        except IndexError:
            if gather or world == 1:
                raise
    timer.stop()
    dt_max, seen = job.max_over_ranks(timer.total)
    out = {"dt_max": dt_max, "ranks_seen": seen, "total": total, "n_local": n_local, "resident": resident,
           "intervals": timer.intervals, "result": result, "net_h": net_h, "chunks": chunks,
           "fps": total * steps / dt_max, "ms_per_step": dt_max * 1e3 / steps}

    if roofline_pass:
        # separate single-stream pass: one fused call per batch, the chosen stages bracketed by events
        if state["loaded"] != 0 and chunks:
            fill(chunks[0])
            state["loaded"] = 0
        a0, b0 = chunks[0] if chunks else (0, 0)
        eng.timing_read()
        if roofline_pass == "dominant":
            eng.timing_enable(True, stages=["conv2"], every=1)
        else:
            eng.timing_enable(True)
        torch.cuda.synchronize()
        if b0 > a0:
            for _ in range(2):
                clip.classify_range(frames[:b0 - a0], a0, batch, overlap=False)
        torch.cuda.synchronize()
        out["timing"] = eng.timing_read()
        eng.timing_enable(False)
        out["timing_launch_frames"] = [min(b0, a + batch) - a for _ in range(2) for a in range(a0, b0, batch)]
    return out


def traffic_from_profiles(key):
    """HBM bytes per full-batch launch of block 2 / per batch of the whole path, measured separately under
    rocprofv3 --pmc and committed as profiles/roofline_traffic.json (None when the file has no such entry)."""
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
            tj = json.load(f)
        node = tj
        for k in key:
            node = node[k]
        return node
    except (OSError, KeyError, ValueError, TypeError):
        return None


def block2_roofline(leg, dtype, height, width, batch, fps_per_gpu):
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    n_launch, ms = leg["timing"]["conv2"]
    sizes = leg["timing_launch_frames"]
    full = [k for k in sizes if k == batch] or sizes or [batch]
    # every launch of the pass is bracketed; partial last batches are in the average and in its frame count
    frames_per_launch = sum(sizes) / max(1, len(sizes)) if n_launch == len(sizes) else sum(full) / len(full)
    conv2_ms = ms / max(1, n_launch)
    achieved = FLOPS_CONV2 * frames_per_launch / (conv2_ms * 1e-3) / 1e12 if n_launch else 0.0
    if dtype == "f32" and (height, width, batch) == (H0, W0, BATCH):
        traffic = traffic_from_profiles(("conv2_kernel", "traffic_bytes"))
    else:
        traffic = traffic_from_profiles(("other_configs", f"{dtype}_{width}x{height}_b{batch}", "conv2_kernel", "traffic_bytes"))
    bytes_frame = compulsory_bytes(height, width, leg["net_h"])
    return {
        "bound": "mfma", "kernel": "conv2_kernel", "achieved": round(achieved, 2),
        "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "peak_measured": MEASURED_F32_MFMA_TFLOPS if dtype == "f32" else None,
        "frac_of_measured_peak": round(achieved / MEASURED_F32_MFMA_TFLOPS, 4) if dtype == "f32" else None,
        "traffic": traffic, "traffic_unit": "bytes per full-batch launch (rocprofv3 PMC, profiles/roofline_traffic.json)",
        "launch_ms": round(conv2_ms, 4), "launches_timed": n_launch, "flops_per_frame": FLOPS_CONV2,
        "frames_per_launch": round(frames_per_launch, 2),
        "measured_in": "a separate single-stream pass of the same run (one fused call per batch, every block-2 launch "
                       "bracketed by HIP events on its stream), outside the interval `value` is timed over",
        # the kernel does not compute conv column 84 of 85, which the floor-mode pooling discards:
        # `achieved` counts the algorithmic FLOPs, this is the rate of the MFMAs actually issued
        "issued_frac_of_peak": round(achieved * (84.0 / 85.0) / peak, 4),
        "whole_net_tflops": round(FLOPS_FRAME * fps_per_gpu / 1e12, 2),
        "whole_net_frac_of_mfma_peak": round(FLOPS_FRAME * fps_per_gpu / 1e12 / peak, 4),
        "whole_net_note": ("algorithmic FLOPs of all three blocks over the fp32 matrix peak; block 1 (95.6 of 283.4 MFLOP per "
                           "frame) runs on the bf16 matrix pipe with exact three-term weights unless --f32-block1-fp32-mfma, so "
                           "this ratio is not bounded by 1; fp32_mfma_block1 has it with every block on the fp32 pipe")
                          if dtype == "f32" else None,
        "hbm_compulsory_GBps": round(bytes_frame * fps_per_gpu / 1e9, 2),
        "hbm_frac_of_peak": round(bytes_frame * fps_per_gpu / 1e9 / PEAK_HBM_GBS, 5),
    }


def stage_table(timing):
    return {k: {"launches": v[0], "ms_total": round(v[1], 3)} for k, v in timing.items() if v[0]}


def main():
    args = parse()
    if not launch.in_rank_process() and args.gpus > 1:
        return launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                                  stdout_filter=lambda text: text.lstrip().startswith("{"))
    rank, local_rank, world = launch.rank_info()
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
    job = Job(world, rank, dev, one_gpu, backend)
    # The run directory of the file barrier: the launcher's (self-launch), else made by rank 0 and passed
    # round through the process group while one exists (torch.distributed.run sets no such directory).
    run_dir = os.environ.get("CUTDET_RUN_DIR")
    pg_wanted = world > 1 and not args.independent_clips
    if world > 1 and (pg_wanted or run_dir is None):
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        job.group = dist.group.WORLD
        if run_dir is None:
            box = [tempfile.mkdtemp(prefix="cutdet_bench_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            run_dir = box[0]
        if not pg_wanted:          # only needed to agree on the directory
            dist.barrier()
            dist.destroy_process_group()
            job.group = None
    if world > 1:
        job.files = launch.FileBarrier(rank, world, run_dir)

    from cutdet import api

    eng = api.default_engine(dev, args.dtype)
    if args.f32_block1_fp32_mfma:
        eng.set_option("f32_block1_fp32_mfma", 1)
    strong = args.total_frames is not None
    total = args.total_frames if strong else world * args.clip_frames
    gather = world > 1 and not args.independent_clips
    head = run_leg(job, eng, total=total, gather=gather, steps=args.steps, warmup=args.warmup, height=args.height,
                   width=args.width, batch=args.batch, resident_frames=args.resident_frames,
                   overlap=not args.no_overlap, roofline_pass=args.roofline_timing)

    extra = {}
    cpu_sample = None
    if world == 1 and not args.no_cpu_baseline:
        nb = max(1, min(args.cpu_sample_batches, min(head["resident"], head["n_local"]) // 128 - 2))
        a0, b0 = head["chunks"][0]
        cpu_sample = job.frames_buffer(b0 - a0, args.height, args.width)[: (2 + nb) * 128].cpu()   # chunk 0 is loaded
    default_run = (args.dtype, args.height, args.width, args.batch) == ("f32", H0, W0, BATCH) and not strong \
        and not args.independent_clips and not args.no_extra_legs
    if default_run and world == 1 and not args.f32_block1_fp32_mfma:
        # ---- the headline configuration once more with block 1 on the fp32 matrix instructions as well
        from cutdet.engine import Engine
        eng_s = Engine(api.default_blob(), dev, "f32")
        eng_s.set_option("f32_block1_fp32_mfma", 1)
        leg = run_leg(job, eng_s, total=total, gather=False, steps=args.leg_steps + 2, warmup=args.leg_warmup + 1,
                      height=args.height, width=args.width, batch=args.batch, resident_frames=args.resident_frames)
        extra["fp32_mfma_block1"] = {
            "value": round(leg["fps"], 1), "unit": "frames/s", "ms_per_step": round(leg["ms_per_step"], 4),
            "steps": args.leg_steps + 2, "warmup": args.leg_warmup + 1,
            "what": "the headline workload with block 1 on v_mfma_f32_16x16x4_f32 like blocks 2 and 3 "
                    "(CD_OPT_F32_BLOCK1_FP32_MFMA); logits differ from the default's in the last bits, both within "
                    "1e-5 of the reference's",
            "whole_net_tflops": round(FLOPS_FRAME * leg["fps"] / 1e12, 2),
            "whole_net_frac_of_mfma_peak": round(FLOPS_FRAME * leg["fps"] / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        }
        eng_s.close()
    if default_run and world == 1:
        # ---- configuration 4: 1920x1080, split-bf16 matrix cores, batch 512
        eng4 = api.default_engine(dev, C4["dtype"])
        leg = run_leg(job, eng4, total=C4["frames"], gather=False, steps=args.leg_steps + 2, warmup=args.leg_warmup + 1,
                      height=C4["height"], width=C4["width"], batch=C4["batch"], roofline_pass="all")
        bytes_frame = compulsory_bytes(C4["height"], C4["width"], leg["net_h"])
        gbs = bytes_frame * leg["fps"] / 1e9
        key = f"{C4['dtype']}_{C4['width']}x{C4['height']}_b{C4['batch']}"
        extra["config4"] = {
            "value": round(leg["fps"], 1), "unit": "frames/s", "ms_per_step": round(leg["ms_per_step"], 4),
            "steps": args.leg_steps + 2, "warmup": args.leg_warmup + 1, "dtype": "bf16 (split: hi+lo bfloat16 operands, "
            "fp32 accumulate; logits within 1e-3 of fp32)",
            "workload": f"{C4['frames']} synthetic {C4['width']}x{C4['height']} u8 frames resident in HBM, batch={C4['batch']}, "
                        "resize pass on a third stream one batch ahead + classify on two alternating streams, GPU "
                        "run-length encode, host glue/merge, CSV; one step = one pass over the clip",
            "roofline": {"bound": "hbm", "kernel": "whole path (resize + blocks 1-3 per 512-frame batch)",
                         "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "peak_measured": MEASURED_HBM_READ_GBS, "frac_of_measured_peak": round(gbs / MEASURED_HBM_READ_GBS, 4),
                         "algorithmic_bytes_per_frame": bytes_frame,
                         "traffic": traffic_from_profiles(("other_configs", key, "path", "traffic_bytes")),
                         "traffic_unit": "HBM bytes per 512-frame batch over all kernels of the path (rocprofv3 PMC, "
                                         "profiles/roofline_traffic.json)",
                         "whole_net_frac_of_bf16_mfma_peak": round(FLOPS_FRAME * leg["fps"] / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)},
            "stages_single_stream_pass": stage_table(leg["timing"]),
            "segments_out": int(len(leg["result"]["merged"]["start_frames"])),
        }
    if default_run and world > 1:
        # ---- configuration 3: the full game, strong-scaled over the ranks, one all-gather
        leg = run_leg(job, eng, total=GAME_FRAMES, gather=True, steps=args.leg_steps, warmup=args.leg_warmup,
                      height=H0, width=W0, batch=BATCH, resident_frames=args.resident_frames)
        extra["strong_200k"] = {
            "value": round(leg["fps"], 1), "unit": "frames/s", "ms_per_step": round(leg["ms_per_step"], 4),
            "n_ranks_seen": leg["ranks_seen"], "scaling": "strong", "steps": args.leg_steps, "warmup": args.leg_warmup,
            "frames_per_step": GAME_FRAMES, "frames_per_rank": leg["n_local"], "resident_frames_per_rank": leg["resident"],
            "timed_region": "one interval" if leg["intervals"] == 1 else
                            f"{leg['intervals']} intervals: regeneration of non-resident frame chunks excluded",
            "stitch": "one all-gather of (class,max) streams over " + ("RCCL" if backend == "nccl" else backend),
            "segments_out": int(len(leg["result"]["merged"]["start_frames"])) if leg["result"] else None,
        }
    if job.group is not None:
        dist.barrier()
        dist.destroy_process_group()
        job.group = None
    if default_run and world > 1:
        # ---- configuration 5: one clip per GPU; the process group is gone, ranks line up through files
        assert not dist.is_initialized()
        leg = run_leg(job, eng, total=world * CLIP_FRAMES, gather=False, steps=args.leg_steps, warmup=args.leg_warmup,
                      height=H0, width=W0, batch=BATCH, resident_frames=args.resident_frames)
        extra["independent_clips"] = {
            "value": round(leg["fps"], 1), "unit": "frames/s", "ms_per_step": round(leg["ms_per_step"], 4),
            "n_ranks_seen": leg["ranks_seen"], "scaling": "weak", "steps": args.leg_steps, "warmup": args.leg_warmup,
            "frames_per_step": world * CLIP_FRAMES, "clips": world, "process_group": "none",
            "rendezvous": "file barrier (no collective, no network)",
            "segments_out_rank0": int(len(leg["result"]["merged"]["start_frames"])) if leg["result"] else None,
        }

    if rank == 0:
        fps = head["fps"]
        merged = head["result"]["merged"]
        net_h = head["net_h"]
        arith = {"f32": "fp32 (blocks 2-3 on v_mfma_f32_16x16x4_f32; block 1 " +
                        ("on it too" if args.f32_block1_fp32_mfma else
                         "multiplies the frame's bytes with the fp32 weights as three exact bfloat16 terms on "
                         "v_mfma_f32_16x16x32_bf16: exact products, fp32 accumulation") + ")",
                 "bf16": "bf16-matrix-core", "bf16_split": "split-bf16 (hi+lo, 4 products) matrix-core"}[args.dtype]
        out = {
            "metric": f"frames/sec classified+segmented, {args.width}x{args.height} RGB",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"{total}-frame synthetic {args.width}x{args.height} u8 game split into {world} contiguous "
                                    f"range(s)" if strong else
                                    f"{args.clip_frames} synthetic {args.width}x{args.height} u8 frames per GPU") +
                                   f" resident in HBM, batch={args.batch}, {arith} prod_net, fused resize+classify, GPU run-length "
                                   f"encode, host glue/merge, CSV; one step = one pass over the range(s)",
                       "frames_per_step": total, "frames_per_gpu_timed": head["n_local"] * args.steps,
                       "segments_out": int(len(merged["start_frames"])),
                       "n_ranks_seen": head["ranks_seen"],
                       "timed_region": "one interval" if head["intervals"] == 1 else
                                       f"{head['intervals']} intervals: regeneration of non-resident frame chunks excluded",
                       "resident_frames_per_rank": head["resident"],
                       "streams": "one fused call per batch on one stream" if args.no_overlap else
                                  "one cd_forward_u8_range call per chunk: batches alternate between two streams inside the library",
                       "stitch": "one all-gather of (class,max) streams over " + ("RCCL" if backend == "nccl" else backend)
                                 if gather else "none",
                       "process_group": ("RCCL" if backend == "nccl" else backend) if pg_wanted else "none"},
            "roofline": block2_roofline(head, args.dtype, args.height, args.width, args.batch, fps / world),
            "stages": stage_table(head["timing"]),
        }
        out.update(extra)
        if cpu_sample is not None:
            nb = cpu_sample.shape[0] // 128 - 2
            out["cpu_baseline"] = cpu_baseline(cpu_sample, 2, nb)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
