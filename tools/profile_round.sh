#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun):  bash tools/profile_round.sh r02_c2_bf16x3_v11
# 1. kernel trace + stats of the bench command, 2./3. FETCH_SIZE / WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: TCC slots),
# 4. per-layer tables of the four BASELINE workloads.  Raw output under gpurun_out/prof_<tag>/, summaries under profiles/.
set -e
TAG=${1:-r02_c2_bf16x3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.err
CMD2="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD2 > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD2 > $OUT/bench_write.json 2> $OUT/write.err
cd $R
python3 tools/rocprof_summary.py --trace $OUT/trace --fetch $OUT/fetch --write $OUT/write --out profiles/$TAG --note "$TAG: bench.py --steps 4 --warmup 1 (C2, bf16x3); PMC passes --steps 2" > $OUT/summary.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats.csv
cp $OUT/bench_traced.json profiles/${TAG}_bench_under_rocprof.json
python3 tools/op_profile.py --res 256 --batch 16 > profiles/${TAG%%_c2*}_layers_c2_net.txt 2>/dev/null
python3 tools/op_profile.py --res 256 --batch 16 --uncond > profiles/${TAG%%_c2*}_layers_c2_gnet.txt 2>/dev/null
python3 tools/op_profile.py --res 1024 --batch 4 --sr > profiles/${TAG%%_c2*}_layers_c4_sr1024_b4.txt 2>/dev/null
python3 tools/op_profile.py --res 256 --batch 16 --warp > profiles/${TAG%%_c2*}_layers_c5_warp.txt 2>/dev/null
python3 - <<PY
import json
d = json.load(open("profiles/$TAG.json"))
json.dump({"source": "profiles/$TAG.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of \`bench.py --steps 2 --no-extras\`, library built from the commit the file was committed with)",
           "families": d["families"]}, open("profiles/traffic_c2_bf16x3.json", "w"))
PY
rm -rf gpurun_out/profiles_new && mkdir -p gpurun_out/profiles_new && cp profiles/${TAG}* profiles/${TAG%%_c2*}_layers_* profiles/traffic_c2_bf16x3.json gpurun_out/profiles_new/
tail -30 $OUT/summary.txt
