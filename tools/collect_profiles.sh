#!/bin/bash
# tools/collect_profiles.sh -- copy what tools/profile_gpu.sh left under gpurun_out/prof_r03/ into profiles/r03_*
# (run in the build container after the gpurun call; profiles/ is what the judge reads, gpurun_out/ is scratch).
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r03
for m in mesh grid; do
  cp $(ls -t $S/stats_$m/*/*_kernel_stats.csv | head -1) profiles/r03_${m}_1M_x512_kernel_stats.csv
  for c in fetch write sq; do
    cp $(ls -t $S/pmc_${m}_$c/*/*_counter_collection.csv.mbes | head -1) profiles/r03_${m}_pmc_$c.csv
  done
  grep '^{"metric"' $S/bench_$m.log | tail -1 > profiles/r03_${m}_bench_only_main.json
done
cp $S/traffic.json profiles/r03_traffic.json
ls -la profiles/r03_*
