"""Timeline of a rocprofv3 kernel trace: which kernels ran when, and how much they overlapped.

usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [first_ms last_ms]
Prints, for the window (default: the last 40 % of the trace), every dispatch as
  start_us  dur_us  queue  kernel(grid)
and per kernel: busy time (union of its intervals), and the share of it during which a kernel of
another name was running too.
"""
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("cd::", "")
    return n[:48]


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def overlap(a, b):
    """total length of the intersection of two interval unions"""
    i = j = 0
    t = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            t += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


def main(root, w0=None, w1=None):
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    lo = t0 + (float(w0) * 1e6 if w0 is not None else 0.6 * (t1 - t0))
    hi = t0 + (float(w1) * 1e6 if w1 is not None else (t1 - t0))
    sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"trace {1e-6 * (t1 - t0):.1f} ms, {len(rows)} dispatches; window {1e-6 * (lo - t0):.1f}..{1e-6 * (hi - t0):.1f} ms, {len(sel)} dispatches")
    for s, e, k, q, g in sel[:160]:
        print(f"{1e-3 * (s - lo):10.1f} {1e-3 * (e - s):8.1f}  q{q:>3}  {k}({g})")
    names = sorted({r[2] for r in sel})
    un = {n: union([(r[0], r[1]) for r in sel if r[2] == n]) for n in names}
    allu = union([(r[0], r[1]) for r in sel])
    print(f"\nwindow {1e-3 * (hi - lo):.0f} us; some kernel running {1e-3 * sum(b - a for a, b in allu):.0f} us")
    for n in names:
        busy = sum(b - a for a, b in un[n])
        others = union([(r[0], r[1]) for r in sel if r[2] != n])
        print(f"{n:50s} launches {sum(1 for r in sel if r[2] == n):4d}  busy {1e-3 * busy:9.1f} us  "
              f"with another kernel {100.0 * overlap(un[n], others) / max(1, busy):5.1f} %")


if __name__ == "__main__":
    main(*sys.argv[1:])
