#!/bin/bash
# tools/collect_profiles.sh -- copy what tools/profile_gpu.sh left under gpurun_out/prof_$ROUND/ into profiles/$ROUND_* (ROUND: r06)
# (run in the build container after the gpurun call; profiles/ is what the judge reads, gpurun_out/ is scratch).
# The raw counter rows were trimmed on the box to the first 12 dispatches of every kernel (the means over all dispatches
# are in <round>_traffic.json); here only the rows of the dominant sweep kernels are kept.
set -e
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r06}
S=gpurun_out/prof_$ROUND
name() { echo "$1" | sed 's/mesh-tin/tin/'; }
for m in mesh grid mesh-tin mesh_tempered mesh-tin_tempered; do
  f=$(ls -t $S/stats_$m/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f profiles/${ROUND}_$(name $m)_1M_x512_kernel_stats.csv
  [ -f $S/bench_$m.log ] && grep '^{"metric"' $S/bench_$m.log | tail -1 > profiles/${ROUND}_$(name $m)_bench_only_main.json
done
for m in mesh grid mesh-tin; do
  for c in fetch write sq cache; do
    for f in $(ls $S/pmc_${m}_$c/*/*_counter_collection.csv.trim 2>/dev/null); do
      python3 - "$f" profiles/${ROUND}_$(name $m)_pmc_$c.csv <<'PY'
import csv, os, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r['Kernel_Name'].startswith('void k_mbes_sweep') and ', false, false>' in r['Kernel_Name']]
new = not os.path.exists(sys.argv[2])
with open(sys.argv[2], 'a', newline='') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    if new:
        w.writeheader()
    w.writerows(keep)
PY
    done
  done
done
for m in mesh mesh-tin; do
  for c in sq cache; do
    for f in $(ls $S/pmc_${m}_tempered_$c/*/*_counter_collection.csv.trim 2>/dev/null); do
      python3 - "$f" profiles/${ROUND}_$(name $m)_tempered_pmc_$c.csv <<'PY'
import csv, os, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if r['Kernel_Name'].startswith('void k_mbes_sweep') and ', false, false>' in r['Kernel_Name']]
new = not os.path.exists(sys.argv[2])
with open(sys.argv[2], 'a', newline='') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    if new:
        w.writeheader()
    w.writerows(keep)
PY
    done
  done
done
# the streaming kernels' rows of the two SQ passes on the mesh leg
rm -f profiles/${ROUND}_mesh_pmc_streaming.csv
for f in $(ls $S/pmc_mesh_sq/*/*_counter_collection.csv.trim $S/pmc_mesh_sq2/*/*_counter_collection.csv.trim 2>/dev/null); do
  python3 - "$f" profiles/${ROUND}_mesh_pmc_streaming.csv <<'PY'
import csv, os, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_predict_pose', 'k_quantise_tiles', 'k_cdf_expand', 'k_resample_gather', 'k_visit_scan'))]
new = not os.path.exists(sys.argv[2])
with open(sys.argv[2], 'a', newline='') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    if new:
        w.writeheader()
    w.writerows(keep)
PY
done
for m in config5 soup config4_rccl tin_hole tin_ragged; do
  f=$(ls -t $S/stats_$m/*/*_kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f profiles/${ROUND}_${m}_kernel_stats.csv; fi
done
cp $S/traffic.json profiles/${ROUND}_traffic.json
bash tools/isa.sh /tmp/isa > /dev/null 2>&1 && grep -v 'rocprim::' /tmp/isa/resources.tsv > profiles/${ROUND}_isa_resources.tsv   # (the library's own kernels)
ls -la profiles/${ROUND}_*
