"""Oracle for the segmentation stage (test infrastructure).  Pinned by tests/golden/seg_golden.json.

Restates reference frameID/segmentation.py on plain Python lists:
  segments_from_scores  :35-60   argmax/max, run-length encoding, per-run score mean
  is_orphan             :12-17
  absorb                :69-89   neighbour swallows a segment (including the score-mean
                                 expression exactly as the reference evaluates it)
  glue_orphans          :91-166
  combine_adjacent      :168-183
  csv_bytes             :185-196 with the class names of frameID/data.py:116
A segment table is a dict of five equally long lists:
  end_frames, frame_types, run_lengths, start_frames (Python ints), score_means (np.float32).
Like the reference, an empty score array and a table that collapses to one orphan segment
raise IndexError.
"""
import numpy as np
import torch

CLASS_NAMES = {0: "a22", 1: "ez", 2: "b"}
KEYS = ("end_frames", "frame_types", "run_lengths", "start_frames", "score_means")


def sum_f32_torch_order(x):
    """fp32 sum of a contiguous 1-D float32 array in the order torch's CPU kernel adds it
    (ATen SumKernel: 8-float vectors, four interleaved vector accumulators, the cascade levels of
    multi_row_sum, then the scalar tail and the eight lanes one after the other; fewer than 8
    elements: four interleaved scalar accumulators).  `highest_scores[a:b].mean()` in the reference
    (segmentation.py:54-59) is this sum divided by the length, in fp32, for slices below the 32768
    elements from which torch splits a reduction over threads.  Checked bit for bit against
    torch.mean in tests/test_oracle_golden.py."""
    x = np.asarray(x, dtype=np.float32)
    n, V = int(x.shape[0]), 8
    f = np.float32

    def row_sum(load, size, width):
        size_ilp = size // 4
        ceil_log2 = 1 if size_ilp <= 2 else (size_ilp - 1).bit_length()
        power = max(4, ceil_log2 // 4)
        step, mask = 1 << power, (1 << power) - 1
        zero = lambda: np.zeros(width, dtype=np.float32)  # noqa: E731
        acc = [[zero() for _ in range(4)] for _ in range(4)]
        i = 0
        while i + step <= size_ilp:
            for _ in range(step):
                for k in range(4):
                    acc[0][k] = acc[0][k] + load(4 * i + k)
                i += 1
            for j in range(1, 4):
                for k in range(4):
                    acc[j][k] = acc[j][k] + acc[j - 1][k]
                    acc[j - 1][k] = zero()
                if i & (mask << (j * power)):
                    break
        while i < size_ilp:
            for k in range(4):
                acc[0][k] = acc[0][k] + load(4 * i + k)
            i += 1
        for j in range(1, 4):
            for k in range(4):
                acc[0][k] = acc[0][k] + acc[j][k]
        ps = acc[0]
        for i in range(size_ilp * 4, size):
            ps[0] = ps[0] + load(i)
        for k in range(1, 4):
            ps[0] = ps[0] + ps[k]
        return ps[0]

    with np.errstate(all="ignore"):
        if n < V:
            return f(row_sum(lambda i: x[i:i + 1], n, 1)[0])
        vec_size = n // V
        lanes = row_sum(lambda i: x[i * V:(i + 1) * V], vec_size, V)
        total = f(0)
        for k in range(vec_size * V, n):
            total = f(total + x[k])
        for k in range(V):
            total = f(total + lanes[k])
        return total


def mean_f32_torch_order(x):
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(all="ignore"):
        return np.float32(sum_f32_torch_order(x) / np.float32(x.shape[0]))


def segments_from_scores(scores):
    """scores float32 [N,3] -> segment table."""
    s = torch.as_tensor(np.asarray(scores, dtype=np.float32))
    if s.shape[0] == 0:
        raise IndexError("no frames")
    best, cls = torch.max(s, dim=1)
    cls_l = cls.tolist()
    best_np = best.numpy()
    n = len(cls_l)
    ends = [i for i in range(n - 1) if cls_l[i + 1] != cls_l[i]] + [n - 1]
    starts = [0] + [e + 1 for e in ends[:-1]]
    return {
        "end_frames": ends,
        "frame_types": [cls_l[e] for e in ends],
        "run_lengths": [e - b + 1 for b, e in zip(starts, ends)],
        "start_frames": starts,
        # fp32 torch.mean over the slice, as the reference does (its summation order restated, so
        # that the result does not depend on the torch build of the machine the tests run on)
        "score_means": [mean_f32_torch_order(best_np[b:e + 1]) for b, e in zip(starts, ends)],
    }


def is_orphan(t, length, real_threshold, blank_threshold):
    return length < (blank_threshold if t == 2 else real_threshold)


def absorb(te, gone, into):
    """Segment `into` swallows its neighbour `gone` (row `gone` is NOT removed here)."""
    if gone < into:
        te["start_frames"][into] = te["start_frames"][gone]
    else:
        te["end_frames"][into] = te["end_frames"][gone]
    f = np.float32
    with np.errstate(all="ignore"):
        ln, lg = f(te["run_lengths"][into]), f(te["run_lengths"][gone])
        # (a*ln + b*lg) / ln + lg : the division binds tighter than the last addition
        te["score_means"][into] = f(f(f(te["score_means"][into] * ln) + f(te["score_means"][gone] * lg)) / ln) + lg
        te["score_means"][into] = f(te["score_means"][into])
    te["run_lengths"][into] = te["end_frames"][into] - te["start_frames"][into] + 1


def _drop(te, i):
    for k in KEYS:
        del te[k][i]


def glue_orphans(te, real_threshold=100, blank_threshold=10):
    while True:
        orphans = [i for i in range(len(te["end_frames"]))
                   if is_orphan(te["frame_types"][i], te["run_lengths"][i], real_threshold, blank_threshold)]
        if not orphans:
            return te
        # least confident first; ties resolve to the lowest index (stable order)
        target = min(orphans, key=lambda i: (te["score_means"][i], i))
        last = len(te["end_frames"]) - 1
        if target == 0:
            if last == 0:
                raise IndexError("index 1 is out of bounds for dimension 0 with size 1")
            absorb(te, 0, 1)
        elif target == last:
            absorb(te, target, target - 1)
        elif te["run_lengths"][target - 1] > te["run_lengths"][target + 1]:
            absorb(te, target, target - 1)
        else:
            absorb(te, target, target + 1)
        _drop(te, target)


def combine_adjacent(te):
    i = 0
    while i < len(te["end_frames"]) - 1:
        if te["frame_types"][i] == te["frame_types"][i + 1]:
            absorb(te, i, i + 1)
            _drop(te, i)
        else:
            i += 1
    return te


def csv_bytes(te):
    return b"".join(f"{s},{CLASS_NAMES[t]}\r\n".encode() for s, t in zip(te["start_frames"], te["frame_types"]))


def segment_clip(scores, real_threshold=100, blank_threshold=10):
    """scores -> (table after init, after glue, after merge, csv bytes); tables are deep copies."""
    def snap(te):
        return {k: list(v) for k, v in te.items()}
    te = segments_from_scores(scores)
    a = snap(te)
    glue_orphans(te, real_threshold, blank_threshold)
    b = snap(te)
    combine_adjacent(te)
    return a, b, snap(te), csv_bytes(te)
