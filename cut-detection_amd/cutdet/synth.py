"""Synthetic clip description used by the benchmark and the tests (product side).

Only the clip STRUCTURE lives here (which frame has which class); the pixels are produced on
the GPU by Engine.synth_frames.  Classes: 0 = a22 (noise), 1 = ez (striped green), 2 = b
(near-black), the ids of frameID/data.py:116 in the reference.
"""


def classes_from_runs(runs):
    """[(class, length), ...] -> per-frame class list."""
    out = []
    for c, n in runs:
        out.extend([int(c)] * int(n))
    return out


def run_table(n_frames, seed=1234):
    """Deterministic run table for an n_frames clip: long a22/ez shots (300..2000 frames)
    separated by short blank runs (15..60), with a sub-threshold orphan every few shots (a
    5-frame blank or a 30-frame shot) so that orphan gluing and merging have work to do."""
    state = seed & 0xFFFFFFFF

    def rnd(lo, hi):
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return lo + (state >> 8) % (hi - lo + 1)

    runs, total, k = [], 0, 0
    while total < n_frames:
        cls, length = k % 2, rnd(300, 2000)
        if k % 3 == 1:   # an orphan in the middle of the shot: 5 blank frames or 30 frames of the other class
            half = length // 2
            orphan = (2, 5) if k % 2 else (1 - cls, 30)
            group = [(cls, half), orphan, (cls, length - half)]
        else:
            group = [(cls, length)]
        group.append((2, rnd(15, 60)))
        for c, n in group:
            n = min(n, n_frames - total)
            if n > 0:
                runs.append((c, n))
                total += n
        k += 1
    return runs
