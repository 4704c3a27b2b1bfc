#!/bin/bash
# tools/kstats.sh <bench.py args ...> -- run on the GPU box (gpurun): rocprofv3 --kernel-trace --stats of a short bench run,
# the per-kernel table (calls, average / min / max duration in us) printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/kst; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --only-main > /dev/null 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        print('%-70s %6s  avg %9.2f  min %9.2f  max %9.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
