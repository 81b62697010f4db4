#!/usr/bin/env python3
"""Benchmark of the classify+segment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (per GPU): a 10 000-frame synthetic 1280x720 clip resident in HBM as u8, classified in
batches of 256 frames (fp32) and segmented.  One STEP = one batch through the fused
resize+prod_net forward; whenever a step finishes the clip (every 40th step) and once more at
the end of the timed region, the per-frame streams are (all-gathered and) run-length encoded on
the GPU, repaired and merged on the host and rendered to CSV bytes -- all inside the timed
region.  With N > 1 every rank owns its own 10 000-frame range (weak scaling); the only
exchange is the all-gather of the (class, max) streams at clip end.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cut-detection_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIP_FRAMES = 10_000
BATCH = 256
H0, W0 = 720, 1280
NET_H = 144
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: one untimed pass over the clip (40 batches: the chip needs ~20 ms of load to reach its
    # steady clocks; with 4 warm-up steps the timed region starts ~6 % slow), then five timed passes
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--clip-frames", type=int, default=CLIP_FRAMES)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--height", type=int, default=H0, help="source frame height (default: the 720p headline)")
    ap.add_argument("--width", type=int, default=W0, help="source frame width")
    ap.add_argument("--timing-every", type=int, default=1,
                    help="with --stage-timing dominant: bracket every n-th launch of the dominant kernel "
                         "(the roofline's launch time is the average over that sample)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 is the headline (the reference's arithmetic); bf16 runs the library's bf16 mode, "
                         "whose logits differ by a few 1e-2 - a separate, non-headline measurement")
    ap.add_argument("--independent-clips", action="store_true",
                    help="multi-GPU: every rank segments its own clip (no stream gather, no collective at all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-timing", choices=["dominant", "all", "off"], default="dominant",
                    help="which kernels are bracketed with HIP events inside the timed region: the dominant "
                         "one (block 2, what the roofline object reports), every stage (each event pair costs "
                         "a few microseconds of stream time per launch), or none")
    ap.add_argument("--cpu-sample-batches", type=int, default=40,
                    help="128-frame batches the CPU baseline times (40 = 5120 frames, about 11 s on 16 host threads)")
    return ap.parse_args()


def cpu_baseline(frames_u8_cpu, n_warm_batches, n_timed_batches):
    """The CPU oracle (port of the reference's CPU path) on a bounded sample of the same clip:
    numpy resize + /255, the traced torch prod_net at batch 128 (segment_video.py:104 default),
    then segmentation.  Returns the dict reported as `cpu_baseline`."""
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
        "sample": f"{n} frames (batches of {bs}) of the same 1280x720 clip after {n_warm_batches} warm-up "
                  f"batches: numpy point-sample resize + /255, torch.jit-traced fp32 prod_net, segmentation",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    # CUTDET_BENCH_BACKEND=gloo + CUTDET_BENCH_ONE_GPU=1 rehearse the multi-rank path on a 1-GPU box
    backend = os.environ.get("CUTDET_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("CUTDET_BENCH_ONE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cutdet import api, synth as csynth
    from cutdet.pipeline import ClipClassifier

    eng = api.default_engine(dev, args.dtype)
    peak_tflops = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    n_clip, bs = args.clip_frames, args.batch
    # this rank's contiguous frame range of the (world * n_clip)-frame game
    runs = csynth.run_table(world * n_clip, SEED)
    classes_all = csynth.classes_from_runs(runs)
    first = rank * n_clip
    classes = torch.tensor(classes_all[first:first + n_clip], dtype=torch.uint8, device=dev)
    frames = eng.synth_frames(classes, args.height, args.width, SEED, first_index=first)   # u8 clip in HBM
    eng.reserve(bs, int(args.height * (256 / args.width)))
    clip = ClipClassifier(eng, n_clip)
    batches = [(a, min(n_clip, a + bs)) for a in range(0, n_clip, bs)]
    group = dist.group.WORLD if world > 1 and not args.independent_clips else None
    torch.cuda.synchronize()

    def step(i):
        a, b = batches[i % len(batches)]
        clip.classify(frames[a:b], a)
        return b - a

    def clip_tail(n_local):
        # every rank takes part in the gather; rank 0's result is the job's output
        return clip.segment(n_local, 100, 10, group=group)

    # ---- warm-up (untimed): W steps + one segmentation so RCCL and every kernel are initialised
    for i in range(args.warmup):
        step(i)
    if args.warmup > 0:
        try:
            clip_tail(batches[min(args.warmup, len(batches)) - 1][1])
        except IndexError:   # a warm-up prefix too short to hold one full-length shot
            pass
    torch.cuda.synchronize()

    # ---- timed region
    if args.stage_timing == "dominant":
        eng.timing_enable(True, stages=["conv2"], every=args.timing_every)
    else:
        eng.timing_enable(args.stage_timing == "all")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames_done, result, pending = 0, None, 0
    for i in range(args.steps):
        frames_done += step(i)
        pending = batches[i % len(batches)][1]
        if (i + 1) % len(batches) == 0:
            result = clip_tail(n_clip)
            pending = 0
    if pending:
        result = clip_tail(pending)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = eng.timing_read()
    eng.timing_enable(False)

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    total_frames = frames_done * world

    if rank == 0:
        # sanity: the segmentation of the synthetic clip must recover its long shots
        merged = result["merged"]
        n_launch, ms = timing["conv2"]
        # the bracketed launches: every step, or every timing_every-th one counted from the first timed step
        every = args.timing_every if args.stage_timing == "dominant" else 1
        sampled = [batches[i % len(batches)] for i in range(args.steps) if i % every == 0]
        assert n_launch in (0, len(sampled)), (n_launch, len(sampled))
        frames_per_launch = sum(b - a for a, b in sampled) / max(1, len(sampled))
        conv2_ms = ms / max(1, n_launch)
        achieved = FLOPS_CONV2 * frames_per_launch / (conv2_ms * 1e-3) / 1e12 if n_launch else 0.0
        stages = {k: {"launches": v[0], "ms_total": round(v[1], 3)} for k, v in timing.items() if v[0]}
        traffic = None      # measured separately under rocprofv3 --pmc; read back if the summary is present
        try:
            with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
                traffic = json.load(f)["conv2_kernel"]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        fps = total_frames / dt_max
        net_h = int(args.height * (256 / args.width))
        bytes_frame = compulsory_bytes(args.height, args.width, net_h)
        if args.dtype != "f32" or (args.height, args.width) != (H0, W0):
            traffic = None          # the committed PMC figure is of the fp32 720p kernel
        out = {
            "metric": f"frames/sec classified+segmented, {args.width}x{args.height} RGB",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_max * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{n_clip} synthetic {args.width}x{args.height} u8 frames per GPU resident in HBM, batch={bs}, "
                                   f"{'fp32' if args.dtype == 'f32' else 'bf16-matrix-core'} prod_net, fused resize+classify, GPU run-length encode, host glue/merge, CSV",
                       "frames_per_gpu_timed": frames_done, "segments_out": int(len(merged["start_frames"])),
                       "stitch": "all-gather of (class,max) streams over RCCL" if group is not None else "none"},
            "roofline": {
                "bound": "mfma", "kernel": "conv2_kernel", "achieved": round(achieved, 2),
                "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(achieved / peak_tflops, 4),
                "traffic": traffic, "traffic_unit": "bytes per 256-frame launch (rocprofv3 PMC, profiles/)",
                "launch_ms": round(conv2_ms, 4), "launches_timed": n_launch, "flops_per_frame": FLOPS_CONV2,
                "frames_per_launch": round(frames_per_launch, 2),
                "whole_net_tflops": round(FLOPS_FRAME * fps / world / 1e12, 2),
                "whole_net_frac_of_mfma_peak": round(FLOPS_FRAME * fps / world / 1e12 / peak_tflops, 4),
                "hbm_compulsory_GBps": round(bytes_frame * fps / world / 1e9, 2),
                "hbm_frac_of_peak": round(bytes_frame * fps / world / 1e9 / PEAK_HBM_GBS, 5),
            },
            "stages": stages,
        }
        if world == 1 and not args.no_cpu_baseline:
            nb = args.cpu_sample_batches
            sample = frames[: (2 + nb) * 128].cpu()
            out["cpu_baseline"] = cpu_baseline(sample, 2, nb)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
