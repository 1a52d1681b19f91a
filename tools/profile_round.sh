#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*, all from ONE build: for the default bench (headline) and for the other BASELINE configs
# (tools/bench_configs.py: config 2, 3, 4 on the fc32 engine, its 32-level sibling, config 4's ConvectiveAdjustmentNDE half on tile16/RKC2, config 5,
# the implicit steps, ADAM) — kernel-trace stats, FETCH_SIZE and WRITE_SIZE (separate --pmc passes: TCC has 4 slots, FETCH_SIZE takes 3), and the SQ split.
# Counters are never combined with API traces.  Usage (GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...   (copy the summaries you keep into profiles/)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES"
B="python3 $R/bench.py --no-cpu-baseline --no-configs"
C="python3 $R/tools/bench_configs.py 4 4n32 4ca"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- $B --steps 3 --warmup 1 > $OUT/ks.log 2>&1
echo "headline: kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -o pf -- $B --steps 1 --warmup 1 > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -o pw -- $B --steps 1 --warmup 1 > $OUT/pw.log 2>&1
echo "headline: FETCH_SIZE / WRITE_SIZE done"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/sq -o sq -- $B --steps 1 --warmup 1 > $OUT/sq.log 2>&1
echo "headline: SQ counters done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/oc -o oc -- python3 $R/tools/bench_configs.py 2 3 4 4n32 4ca 5 ca mpp adam > $OUT/oc.log 2>&1
echo "other configs: kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cf -o cf -- $C > $OUT/cf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cw -o cw -- $C > $OUT/cw.log 2>&1
echo "config 4: FETCH_SIZE / WRITE_SIZE done"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/cs -o cs -- $C > $OUT/cs.log 2>&1
echo "config 4: SQ counters done"
SQCSV=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python3 $R/tools/traffic_from_pmc.py $(find $OUT/pf -name "*counter_collection.csv" | head -1) $(find $OUT/pw -name "*counter_collection.csv" | head -1) $TAG "$SQCSV" > $OUT/traffic.log 2>&1 || true
cp $R/profiles/traffic.json $OUT/traffic.json || true
for d in sq cs pf pw cf cw; do
  f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/sq_table.py $f > $OUT/${d}_table.csv
done
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/ks_kernel_stats.csv
cp $(find $OUT/oc -name "*kernel_stats.csv" | head -1) $OUT/oc_kernel_stats.csv
# the raw per-dispatch dumps are large: keep the tables only
rm -rf $OUT/ks $OUT/oc $OUT/pf $OUT/pw $OUT/sq $OUT/cf $OUT/cw $OUT/cs
ls -la $OUT
