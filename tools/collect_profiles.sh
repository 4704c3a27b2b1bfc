#!/bin/bash
# tools/collect_profiles.sh -- copy what tools/profile_gpu.sh left under gpurun_out/prof_$ROUND/ into profiles/$ROUND_* (ROUND: r05)
# (run in the build container after the gpurun call; profiles/ is what the judge reads, gpurun_out/ is scratch).
# The raw counter rows are trimmed to the first 12 dispatches of the dominant kernel (the means over all ~200 are in
# <round>_traffic.json): once per round, a few dozen KB.
set -e
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r05}
S=gpurun_out/prof_$ROUND
# gpurun MERGES into gpurun_out/: keep only the newest run's file(s) in every pass directory before summarising
for d in $S/stats_* $S/pmc_*; do
  newest=$(ls -t $d/runc/ | head -1 | sed 's/_.*//')
  for f in $d/runc/*; do case $(basename $f) in ${newest}_*) ;; *) rm -f $f ;; esac; done
done
python3 tools/pmc_summarise.py $S $S/traffic.json > /dev/null
for m in mesh grid; do
  cp $(ls -t $S/stats_$m/*/*_kernel_stats.csv | head -1) profiles/${ROUND}_${m}_1M_x512_kernel_stats.csv
  for c in fetch write sq; do
    f=$(ls -t $S/pmc_${m}_$c/*/*_counter_collection.csv.mbes | head -1)
    python3 - "$f" profiles/${ROUND}_${m}_pmc_$c.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r['Kernel_Name'].startswith('void k_mbes_sweep') and ', false, false>' in r['Kernel_Name']]
ids = sorted({int(r['Dispatch_Id']) for r in keep})[:12]
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep:
        if int(r['Dispatch_Id']) in ids:
            w.writerow(r)
PY
  done
  grep '^{"metric"' $S/bench_$m.log | tail -1 > profiles/${ROUND}_${m}_bench_only_main.json
done
for m in config5 soup; do
  f=$(ls -t $S/stats_$m/*/*_kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f profiles/${ROUND}_${m}_kernel_stats.csv; fi
done
cp $S/traffic.json profiles/${ROUND}_traffic.json
bash tools/isa.sh /tmp/isa > /dev/null 2>&1 && grep -v 'rocprim::' /tmp/isa/resources.tsv > profiles/${ROUND}_isa_resources.tsv   # (the library's own kernels)
ls -la profiles/${ROUND}_*
