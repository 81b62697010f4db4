"""Segment a video into All-22 / Endzone / blank shots on an MI355X.

Command-line compatible with the reference driver (segment_video.py:81-120): same positional
argument, same options and defaults, `@file` argument expansion, the same log lines and the same
CSV (`start_frame,type` rows).  The per-batch work -- resize, normalise, classify, argmax -- is
one call into the native library on u8 frames; the float frames of the reference's loop
(segment_video.py:42-45) are never materialised.

The `--cpu` switch of the reference selects a code path this build does not have; it is
accepted by the parser and rejected with an error.

`--ranks N` runs the job on N GPUs of the node, one process per GPU (the reference is one process on
one device, segment_video.py:25-29).  With one input the clip's frames are split into N contiguous
ranges: every rank opens the video, seeks to its range, classifies it, the per-frame (class, max)
streams are exchanged in ONE all-gather (RCCL; gloo when ranks share a GPU) and rank 0 segments the
whole clip and writes the CSV - the same bytes a single process writes.  With several inputs
(`--more-inputs`) the clips are dealt round-robin to the ranks, there is no process group at all, and
every rank appends its finished clips to the one CSV under a file lock.
"""
import argparse
import json
import logging
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import torch  # noqa: E402

from frameID.data import VideoDataset  # noqa: E402
from frameID.net import load_default_net  # noqa: E402
from frameID.segmentation import Segmentation  # noqa: E402
from cutdet import launch  # noqa: E402
from cutdet.pipeline import COLUMNS, SegmentCsvStream, check_columns  # noqa: E402

logging.basicConfig(
    level="INFO",
    format="[%(asctime)s] %(levelname)s [%(name)s.%(funcName)s:%(lineno)d] %(message)s",
)

OPTIONS = (
    # flag, type, default, help
    ("--output_path", str, None, "Path to output csv"),
    ("--base-threshold", int, 100,
     "Number of frames below which an A22 or EZ segment will be considered an orphan."),
    ("--blank-threshold", int, 10,
     "Number of frames below which a blank segment will be considered an orphan."),
    ("--batch-size", int, 128, "Batch size for loading frames."),
    ("--print-every", int, 50, "Log message every n batches. 0 to disable."),
    ("--frame-limit", int, None, "Limit how many frames are processed. Mainly for testing."),
)


def build_parser():
    p = argparse.ArgumentParser("Segment a video into scenes.", fromfile_prefix_chars="@")
    p.add_argument("input_path", type=str, help="Path to video to segment.")
    for flag, typ, default, text in OPTIONS:
        p.add_argument(flag, type=typ, default=default, help=text)
    p.add_argument("--cpu", action="store_true", help="Not available in this build (GPU only).")
    p.add_argument("--device", type=str, default="cuda:0", help="GPU to run on.")
    p.add_argument("--dtype", choices=["f32", "bf16", "bf16_split"], default="f32",
                   help="Arithmetic of the classifier: f32 (the reference's); bf16_split (bf16 matrix cores on "
                        "hi+lo split values, logits within 1e-3 of f32); bf16 (plain bf16 operands, logits move "
                        "by a few 1e-2).")
    p.add_argument("--ranks", type=int, default=1,
                   help="Processes, one per GPU: one input is split into contiguous frame ranges (one all-gather, "
                        "rank 0 writes the CSV); several inputs are dealt one clip at a time to the ranks.")
    p.add_argument("--rank-devices", type=str, default=None,
                   help="GPU index of every rank, comma separated (default 0,1,..,N-1).  Ranks that share a GPU "
                        "exchange their streams over gloo instead of RCCL.")
    p.add_argument("--columns", type=str, default="start,type",
                   help=f"CSV columns, comma separated, from {','.join(COLUMNS)} (default: the reference's start,type).")
    p.add_argument("--keep-lone-orphan", action="store_true",
                   help="A clip that collapses to one short segment is written as that segment instead of failing.")
    p.add_argument("--more-inputs", type=str, nargs="*", default=[],
                   help="More videos: all clips go to ONE csv (output_path, default <first input>_segments.csv) "
                        "with a leading clip column, each clip's rows appended as soon as it is done.")
    return p


sv_parser = build_parser()


def classify_video(ds, net, args, log=True):
    """The reference's hot loop (segment_video.py:42-58) over the frames `ds` yields: returns the
    per-frame (argmax, max) streams as two GPU tensors."""
    # size the activation workspace once, before the first batch, instead of inside it
    net.engine.reserve(args.batch_size, ds.batch_net_height or ds.new_height)
    classes, best = [], []
    for i, frames in enumerate(ds.frame_batches(args.batch_size)):
        _, mx, am = net.classify_frames(frames, chan_order=ds.chan_order, net_h=ds.batch_net_height)
        best.append(mx)
        classes.append(am)
        if log and args.print_every > 0 and i % args.print_every == args.print_every - 1:
            logging.info(f"Scored batch {i+1} ({(i+1) * args.batch_size} frames).")
        # same stopping rule as the reference: the batch that crosses the limit is still scored
        if args.frame_limit is not None and (i + 1) * args.batch_size > args.frame_limit:
            break
    dev = net.engine.device
    if not classes:
        return torch.empty((0,), dtype=torch.int64, device=dev), torch.empty((0,), dtype=torch.float32, device=dev)
    return torch.cat(classes), torch.cat(best)


def finish_segments(classes, best, args):
    """Streams of the whole clip -> repaired Segmentation (reference segment_video.py:62-66)."""
    if classes.shape[0] == 0:
        raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
    seg = Segmentation.from_stream(classes, best)
    logging.info(f"Found {len(seg)} initial segments")
    seg.glue_orphans(args.base_threshold, args.blank_threshold,
                     keep_lone_orphan=getattr(args, "keep_lone_orphan", False))
    logging.info(f"Revised to {len(seg)} segments through orphan combination.")
    seg.combine_adjacent_segments()
    logging.info(f"Revised to {len(seg)} segments through matching adjacent combination.")
    return seg


def segment_one(path, net, args, device):
    """Classifies and segments one video; returns the Segmentation (reference segment_video.py:28-66)."""
    ds = VideoDataset(path, resize=256, device=device)
    return finish_segments(*classify_video(ds, net, args), args)


def frames_to_process(length, args):
    """Frames the single-process loop scores: all of them, or, with --frame-limit, whole batches up to
    and including the one that crosses the limit (reference segment_video.py:54-58)."""
    if args.frame_limit is None:
        return length
    return min(length, (args.frame_limit // args.batch_size + 1) * args.batch_size)


def rank_devices(args, world):
    spec = getattr(args, "rank_devices", None)
    devs = [int(v) for v in spec.split(",")] if spec else list(range(world))
    if len(devs) != world:
        raise ValueError(f"--rank-devices names {len(devs)} GPUs for {world} ranks")
    return devs


def load_net(args, device):
    net, params = load_default_net(getattr(args, "dtype", "f32"))
    net.eval()
    net.to(device)
    logging.info("Loaded default classifier.")
    return net


def output_path(args):
    if args.output_path is None:
        return os.path.splitext(args.input_path)[0] + "_segments.csv"
    return args.output_path


def run_rank(args):
    """Body of one rank process of a --ranks job (started by main() in the parent, or by
    torch.distributed.run)."""
    import torch.distributed as dist
    from cutdet.pipeline import gather_streams, shard_range
    rank, _, world = launch.rank_info()
    inputs = [args.input_path] + list(getattr(args, "more_inputs", None) or [])
    devs = rank_devices(args, world)
    if devs[rank] >= torch.cuda.device_count():
        raise RuntimeError(f"rank {rank}: no GPU {devs[rank]} on this node ({torch.cuda.device_count()} visible)")
    device = f"cuda:{devs[rank]}"
    torch.cuda.set_device(devs[rank])
    columns = check_columns(getattr(args, "columns", "start,type").split(","))
    out_path = output_path(args)
    logging.info(f"Rank {rank} of {world} using {device}")
    net = load_net(args, device)

    if len(inputs) > 1:
        # one clip per GPU at a time: no exchange between ranks, no process group
        with SegmentCsvStream(out_path, columns, lock=True) as stream:
            for path in inputs[rank::world]:
                seg = segment_one(path, net, args, device)
                stream.write_clip(os.path.basename(path), {k: v.numpy() for k, v in seg.te.items()})
                logging.info(f"Rank {rank} appended {len(seg)} segments of {path} to {out_path}")
        return out_path

    shared_gpu = len(set(devs)) < world
    if shared_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device(device))      # RCCL
    try:
        ds = VideoDataset(args.input_path, resize=256, device=device)
        total = frames_to_process(ds.video_info["length"], args)
        lo, hi = shard_range(total, rank, world)
        ds.seek(lo, hi - lo)
        local = argparse.Namespace(**dict(vars(args), frame_limit=None))     # the range already encodes the limit
        classes, best = classify_video(ds, net, local, log=rank == 0)
        classes, best = gather_streams(classes, best, dist.group.WORLD, total)
        if rank == 0:
            seg = finish_segments(classes, best, args)
            logging.info(f"Writing {len(seg)} segments to {out_path}")
            seg.write_csv(out_path, columns)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return out_path


def main(args):
    inputs = [args.input_path] + list(getattr(args, "more_inputs", None) or [])
    for path in inputs:
        if not os.path.isfile(path):
            raise ValueError(f"{path} does not exist.")
    if args.cpu:
        raise RuntimeError("--cpu: this build has no CPU classifier; run on an MI355X")
    ranks = int(getattr(args, "ranks", 1) or 1)
    if launch.in_rank_process() and launch.rank_info()[2] > 1:
        return run_rank(args)
    if ranks > 1:
        # the parent starts the ranks before anything here touches a GPU, and only waits for them
        rank_devices(args, ranks)
        check_columns(getattr(args, "columns", "start,type").split(","))
        out_path = output_path(args)
        if len(inputs) > 1:
            open(out_path, "wb").close()
        rc = launch.spawn_ranks(ranks, [os.path.abspath(__file__)],
                                extra_env={"CUTDET_SV_ARGS": json.dumps(vars(args))})
        if rc != 0:
            raise RuntimeError(f"a rank of the {ranks}-rank job failed (exit code {rc})")
        return out_path
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this build runs the classifier on an MI355X only")
    columns = check_columns(getattr(args, "columns", "start,type").split(","))
    device = getattr(args, "device", "cuda:0")
    logging.info(f"Using {device}")
    net = load_net(args, device)
    out_path = output_path(args)

    if len(inputs) == 1:
        seg = segment_one(args.input_path, net, args, device)
        logging.info(f"Writing {len(seg)} segments to {out_path}")
        seg.write_csv(out_path, columns)
        return out_path

    # several clips: one CSV, each clip appended (and flushed) when it is done
    open(out_path, "wb").close()
    with SegmentCsvStream(out_path, columns) as stream:
        for path in inputs:
            seg = segment_one(path, net, args, device)
            stream.write_clip(os.path.basename(path), {k: v.numpy() for k, v in seg.te.items()})
            logging.info(f"Appended {len(seg)} segments of {path} to {out_path}")
    return out_path


if __name__ == "__main__":
    if "CUTDET_SV_ARGS" in os.environ and launch.in_rank_process():      # a rank started by main() above
        main(argparse.Namespace(**json.loads(os.environ["CUTDET_SV_ARGS"])))
    else:
        main(sv_parser.parse_args())
