"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) into one per-kernel table.

usage: python tools/summarize_prof.py gpurun_out/prof > profiles/rNN_summary.md
Per kernel: launches, average duration from the kernel trace, and per-launch averages of every
counter found in the pass directories.  FETCH_SIZE / WRITE_SIZE are reported in bytes
(counter unit is KiB); FETCH_SIZE is also shown doubled, the gfx950 correction for wide
coalesced reads (the raw counter tallies 128-byte requests as 64 bytes).
Only full-batch launches (the modal grid size of each kernel) enter the averages.
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("cd::", "")


def main(root):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    grids = collections.defaultdict(collections.Counter)
    for path in glob.glob(os.path.join(root, "*", "*", "*_kernel_trace.csv")):
        if "/trace/" not in path:
            continue
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
            grids[k][g] += 1
            rows[k]["dur_ns"].append((g, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for path in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            rows[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            rows[k]["_vgpr"] = [(0, int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))]
            rows[k]["_lds"] = [(0, int(r["LDS_Block_Size"]))]
    print("| kernel | launches (modal grid) | avg us | counters per launch |")
    print("|---|---|---|---|")
    for k in sorted(rows, key=lambda k: -sum(d for _, d in rows[k].get("dur_ns", [(0, 0)]))):
        d = rows[k]
        if "dur_ns" not in d:
            continue
        modal = grids[k].most_common(1)[0][0]
        durs = [v for g, v in d["dur_ns"] if g == modal]
        cells = []
        for c in sorted(d):
            if c in ("dur_ns",) or c.startswith("_"):
                continue
            gm = collections.Counter(g for g, _ in d[c]).most_common(1)[0][0]
            vals = [v for g, v in d[c] if g == gm]
            avg = sum(vals) / len(vals)
            if c == "FETCH_SIZE":
                cells.append(f"FETCH {avg * 1024 / 1e6:.2f} MB raw, {2 * avg * 1024 / 1e6:.2f} MB corrected x2")
            elif c == "WRITE_SIZE":
                cells.append(f"WRITE {avg * 1024 / 1e6:.2f} MB")
            else:
                cells.append(f"{c} {avg:.4g}")
        if "_vgpr" in d:
            cells.append(f"regs {d['_vgpr'][0][1]}, LDS {d['_lds'][0][1]} B")
        print(f"| {k} | {len(durs)} ({modal}) | {sum(durs) / len(durs) / 1e3:.1f} | {'; '.join(cells)} |")


if __name__ == "__main__":
    main(sys.argv[1])
