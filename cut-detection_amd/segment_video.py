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
    return p


sv_parser = build_parser()


def main(args):
    if not os.path.isfile(args.input_path):
        raise ValueError(f"{args.input_path} does not exist.")
    if args.cpu:
        raise RuntimeError("--cpu: this build has no CPU classifier; run on an MI355X")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this build runs the classifier on an MI355X only")
    device = getattr(args, "device", "cuda:0")
    logging.info(f"Using {device}")

    ds = VideoDataset(args.input_path, resize=256, device=device)
    net, params = load_default_net(getattr(args, "dtype", "f32"))
    net.eval()
    net.to(device)
    logging.info("Loaded default classifier.")
    # size the activation workspace once, before the first batch, instead of inside it
    net.engine.reserve(args.batch_size, ds.new_height)

    scores, classes, best = [], [], []
    for i, frames in enumerate(ds.frame_batches(args.batch_size)):
        logits, mx, am = net.classify_frames(frames, chan_order=ds.chan_order, net_h=ds.batch_net_height)
        scores.append(logits)
        best.append(mx)
        classes.append(am)
        if args.print_every > 0 and i % args.print_every == args.print_every - 1:
            logging.info(f"Scored batch {i+1} ({(i+1) * args.batch_size} frames).")
        # same stopping rule as the reference: the batch that crosses the limit is still scored
        if args.frame_limit is not None and (i + 1) * args.batch_size > args.frame_limit:
            break
    if not scores:
        raise IndexError("index 0 is out of bounds for dimension 0 with size 0")

    seg = Segmentation.from_stream(torch.cat(classes), torch.cat(best))
    logging.info(f"Found {len(seg)} initial segments")
    seg.glue_orphans(args.base_threshold, args.blank_threshold)
    logging.info(f"Revised to {len(seg)} segments through orphan combination.")
    seg.combine_adjacent_segments()
    logging.info(f"Revised to {len(seg)} segments through matching adjacent combination.")

    if args.output_path is None:
        out_path = os.path.splitext(args.input_path)[0] + "_segments.csv"
    else:
        out_path = args.output_path
    logging.info(f"Writing {len(seg)} segments to {out_path}")
    seg.write_csv(out_path)
    return out_path


if __name__ == "__main__":
    main(sv_parser.parse_args())
