#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC counter passes) into a small JSON/markdown
summary for profiles/.  Usage:
  python tools/rocprof_summary.py --trace DIR [--fetch DIR] [--write DIR] --out profiles/NAME
HBM traffic follows MI355X_MICROARCH.md 'HBM': FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads, so reads are doubled.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

FAMILIES = [("conv_igemm_f32<9>", "conv3x3"), ("conv_igemm_f32<1>", "conv1x1"), ("conv_igemm<9", "conv3x3"),
            ("conv_igemm<1", "conv1x1"), ("conv_x3_glds<9", "conv3x3"), ("conv_x3_glds<1", "conv1x1"), ("conv_x3_patch", "conv3x3"), ("attn_fwd", "attention"),
            ("pixnorm_k", "pixnorm"), ("qkv_split_k", "qkv_split"), ("embed_k", "embed"), ("linear_k", "embed"),
            ("assemble_k", "assemble"), ("precond_out_k", "assemble"), ("sampler_step_k", "sampler"),
            ("prep_weight_k", "prep"), ("warp_features_k", "warp"), ("split_k", "split"), ("layout_k", "assemble"), ("axpy_k", "sampler")]


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\s*\(", n)
    return m.group(1) if m else n[:60]


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--stats-csv", default=None, help="also write the per-kernel statistics as CSV (when the input is a rocpd database)")
    a = ap.parse_args()
    rows = []
    stats_csv = find(a.trace, "*kernel_stats.csv")
    if stats_csv:
        for r in csv.DictReader(open(stats_csv)):
            rows.append(dict(kernel=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                             avg_ms=float(r["AverageNs"]) / 1e6, pct=float(r["Percentage"]),
                             min_ms=float(r["MinNs"]) / 1e6, max_ms=float(r["MaxNs"]) / 1e6))
    else:
        # rocprofv3's default output on this image is a rocpd SQLite database (views `kernels`, `counters_collection`)
        import sqlite3
        db = sqlite3.connect(find(a.trace, "*_results.db"))
        per = collections.OrderedDict()
        for name, dur in db.execute("select name, duration from kernels"):
            e = per.setdefault(name, [0, 0.0, 1e30, 0.0])
            e[0] += 1; e[1] += dur; e[2] = min(e[2], dur); e[3] = max(e[3], dur)
        tot = sum(e[1] for e in per.values())
        for name, (n, t, mn, mx) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            rows.append(dict(kernel=short(name), calls=n, total_ms=t / 1e6, avg_ms=t / n / 1e6, pct=100.0 * t / tot, min_ms=mn / 1e6, max_ms=mx / 1e6))
        if a.stats_csv:
            with open(a.stats_csv, "w") as f:
                f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
                for name, (n, t, mn, mx) in sorted(per.items(), key=lambda kv: -kv[1][1]):
                    f.write(f'"{name}",{n},{t:.0f},{t / n:.1f},{100.0 * t / tot:.4f},{mn:.0f},{mx:.0f}\n')
    pmc = {}
    for cname, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not d:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        ccsv = find(d, "*counter_collection.csv")
        if ccsv:
            recs = ((r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(ccsv)) if r["Counter_Name"] == cname)
        else:
            import sqlite3
            recs = sqlite3.connect(find(d, "*_results.db")).execute("select kernel_name, value from counters_collection where counter_name = ?", (cname,))
        for kname, val in recs:
            k = short(kname)
            agg[k][0] += 1
            agg[k][1] += float(val)
        for k, (n, v) in agg.items():
            pmc.setdefault(k, {})[cname] = dict(calls=n, kib_per_call=v / n)
    for r in rows:
        p = pmc.get(r["kernel"], {})
        if "FETCH_SIZE" in p:
            r["hbm_read_mb_per_call"] = 2 * p["FETCH_SIZE"]["kib_per_call"] * 1024 / 1e6      # gfx950 x2 correction
        if "WRITE_SIZE" in p:
            r["hbm_write_mb_per_call"] = p["WRITE_SIZE"]["kib_per_call"] * 1024 / 1e6
    fam = {}
    for r in rows:
        f = next((v for k, v in FAMILIES if k in r["kernel"]), None)
        if f is None or "hbm_read_mb_per_call" not in r:
            continue
        e = fam.setdefault(f, dict(calls=0, bytes=0.0, ms=0.0))
        e["calls"] += r["calls"]
        e["bytes"] += r["calls"] * (r["hbm_read_mb_per_call"] + r.get("hbm_write_mb_per_call", 0.0)) * 1e6
        e["ms"] += r["total_ms"]
    traffic = {f: dict(hbm_bytes_per_launch=e["bytes"] / e["calls"], avg_launch_ms=e["ms"] / e["calls"], launches=e["calls"])
               for f, e in fam.items()}
    out = dict(note=a.note, kernels=[r for r in rows if r["pct"] >= 0.001], families=traffic)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".md", "w") as f:
        f.write(f"# rocprofv3 summary — {a.note}\n\n")
        f.write("| kernel | calls | total ms | avg ms | % | HBM read MB/call (FETCH_SIZE x2) | HBM write MB/call |\n|---|---|---|---|---|---|---|\n")
        for r in out["kernels"]:
            f.write(f"| `{r['kernel']}` | {r['calls']} | {r['total_ms']:.2f} | {r['avg_ms']:.4f} | {r['pct']:.2f} | "
                    f"{r.get('hbm_read_mb_per_call', float('nan')):.1f} | {r.get('hbm_write_mb_per_call', float('nan')):.1f} |\n")
    print(open(a.out + ".md").read())


if __name__ == "__main__":
    main()
