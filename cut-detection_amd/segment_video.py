"""Segment a video into All-22 / Endzone / blank shots on an MI355X.

Command-line compatible with the reference driver (segment_video.py:81-120): same positional
argument, same options and defaults, `@file` argument expansion, the same log lines and the same
CSV (`start_frame,type` rows).  The per-batch work -- resize, normalise, classify, argmax -- is
one call into the native library on u8 frames; the float frames of the reference's loop
(segment_video.py:42-45) are never materialised.

The `--cpu` switch of the reference selects a code path this build does not have; it is
accepted by the parser and rejected with an error.
"""
import argparse
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
    p.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                   help="Arithmetic of the classifier: f32 (the reference's) or bf16 matrix cores.")
    p.add_argument("--columns", type=str, default="start,type",
                   help=f"CSV columns, comma separated, from {','.join(COLUMNS)} (default: the reference's start,type).")
    p.add_argument("--keep-lone-orphan", action="store_true",
                   help="A clip that collapses to one short segment is written as that segment instead of failing.")
    p.add_argument("--more-inputs", type=str, nargs="*", default=[],
                   help="More videos: all clips go to ONE csv (output_path, default <first input>_segments.csv) "
                        "with a leading clip column, each clip's rows appended as soon as it is done.")
    return p


sv_parser = build_parser()


def segment_one(path, net, args, device):
    """Classifies and segments one video; returns the Segmentation (reference segment_video.py:28-66)."""
    ds = VideoDataset(path, resize=256, device=device)
    # size the activation workspace once, before the first batch, instead of inside it
    net.engine.reserve(args.batch_size, ds.batch_net_height or ds.new_height)
    classes, best = [], []
    for i, frames in enumerate(ds.frame_batches(args.batch_size)):
        _, mx, am = net.classify_frames(frames, chan_order=ds.chan_order, net_h=ds.batch_net_height)
        best.append(mx)
        classes.append(am)
        if args.print_every > 0 and i % args.print_every == args.print_every - 1:
            logging.info(f"Scored batch {i+1} ({(i+1) * args.batch_size} frames).")
        # same stopping rule as the reference: the batch that crosses the limit is still scored
        if args.frame_limit is not None and (i + 1) * args.batch_size > args.frame_limit:
            break
    if not classes:
        raise IndexError("index 0 is out of bounds for dimension 0 with size 0")

    seg = Segmentation.from_stream(torch.cat(classes), torch.cat(best))
    logging.info(f"Found {len(seg)} initial segments")
    seg.glue_orphans(args.base_threshold, args.blank_threshold,
                     keep_lone_orphan=getattr(args, "keep_lone_orphan", False))
    logging.info(f"Revised to {len(seg)} segments through orphan combination.")
    seg.combine_adjacent_segments()
    logging.info(f"Revised to {len(seg)} segments through matching adjacent combination.")
    return seg


def main(args):
    inputs = [args.input_path] + list(getattr(args, "more_inputs", None) or [])
    for path in inputs:
        if not os.path.isfile(path):
            raise ValueError(f"{path} does not exist.")
    if args.cpu:
        raise RuntimeError("--cpu: this build has no CPU classifier; run on an MI355X")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this build runs the classifier on an MI355X only")
    columns = check_columns(getattr(args, "columns", "start,type").split(","))
    device = getattr(args, "device", "cuda:0")
    logging.info(f"Using {device}")

    net, params = load_default_net(getattr(args, "dtype", "f32"))
    net.eval()
    net.to(device)
    logging.info("Loaded default classifier.")

    if args.output_path is None:
        out_path = os.path.splitext(args.input_path)[0] + "_segments.csv"
    else:
        out_path = args.output_path

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
    main(sv_parser.parse_args())
