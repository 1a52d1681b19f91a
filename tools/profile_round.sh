#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats of the default bench, the two PMC passes (FETCH_SIZE, WRITE_SIZE — separate
# runs, counters never combined with API traces), and kernel-trace stats of the other BASELINE configs.  Usage (GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...   (copy the summaries you keep into profiles/)
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $OUT/ks.log 2>&1
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -o pf -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $OUT/pf.log 2>&1
echo "FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -o pw -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $OUT/pw.log 2>&1
echo "WRITE_SIZE done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/oc -o oc -- python3 $R/tools/bench_configs.py 2 3 4 4n32 5 ca adam > $OUT/oc.log 2>&1
echo "other configs done"
find $OUT -name "*.csv" | head -20
