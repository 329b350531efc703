#!/bin/bash
# SQ counters of one op under rocprofv3 (own passes, no tracing):  bash tools/pmc_sq.sh attn 4 4 16384 49152
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VARIANTS="base=" ROUNDS=2 N=3
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/p1 -o p -- python3 $R/tools/ab.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -o p -- python3 $R/tools/ab.py "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$OUT/" + p + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn_fwd" in k or "conv_x3_glds" in k:
                a = acc[(k[:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:60s} {c:28s} {v / n:16.0f}  (mean of {n} dispatches)")
PY
