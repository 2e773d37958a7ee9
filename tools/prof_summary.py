#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the small summaries kept under profiles/.

  stats   <dir> <out.md> <title>       *_kernel_stats.csv  -> markdown table (+ copy of the csv next to it)
  traffic <fetch_dir> <write_dir> <out.json>
          *_counter_collection.csv of a `--pmc FETCH_SIZE` pass and of a `--pmc WRITE_SIZE` pass
          -> per-kernel HBM bytes per launch.  Units and corrections follow MI355X_MICROARCH.md (HBM section):
          both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled;
          WRITE_SIZE is exact for 16-B-per-lane stores.  The two counters need separate passes (TCC slots).
"""
import csv
import glob
import json
import os
import shutil
import sys


def _one(dirname, pattern):
    hits = sorted(glob.glob(os.path.join(dirname, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {dirname}")
    return hits[0]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:96]


def stats(dirname, out_md, title):
    src = _one(dirname, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(src)))
    with open(out_md, "w") as f:
        f.write(f"# {title}\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
    shutil.copy(src, os.path.splitext(out_md)[0] + ".csv")


def _per_kernel(dirname, counter):
    acc = {}
    for r in csv.DictReader(open(_one(dirname, "*counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        d = acc.setdefault(short(r["Kernel_Name"]), [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
    return acc


def traffic(fetch_dir, write_dir, out_json):
    fe, wr = _per_kernel(fetch_dir, "FETCH_SIZE"), _per_kernel(write_dir, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, [0.0, 1]); w, nw = wr.get(k, [0.0, 1])
        rd, wb = 2.0 * f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": rd + wb}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    json.dump({"note": "FETCH_SIZE x2 x1024 + WRITE_SIZE x1024 (gfx950 corrections of MI355X_MICROARCH.md), "
                       "separate --pmc passes, averaged per launch", "kernels": out}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    {"stats": stats, "traffic": traffic}[sys.argv[1]](*sys.argv[2:])
