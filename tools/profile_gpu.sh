#!/bin/bash
# tools/profile_gpu.sh -- run on the GPU box (gpurun): kernel-trace stats and the PMC passes of the round; results under
# gpurun_out/prof_${ROUND:-r06}/.  Counters are collected in their own runs (never together with a trace), the program
# itself follows `--` (no wrapper that would re-exec).  Round 6 adds the irregular TIN (the adjacency walk: the path of
# any real survey mesh) with L2 / vector-L1 counters, the healthy-cloud filter on both surfaces, and a second SQ pass on
# the mesh leg for the streaming kernels' table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${ROUND:-r06}
rm -rf $O && mkdir -p $O
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
SQ2="SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"
pass() {  # pass <dir> <counters or --stats> <bench args...>
  local d=$1 c=$2; shift 2
  if [ "$c" = "--stats" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --only-main > $O/bench_${d#stats_}.log 2>&1
  else
    rocprofv3 --pmc $c --output-format csv -d $O/$d -- python3 $R/bench.py "$@" --steps 4 --warmup 1 --only-main > /dev/null 2>&1
  fi
}
for m in mesh grid mesh-tin; do
  pass stats_$m --stats --map $m
  pass pmc_${m}_fetch FETCH_SIZE --map $m
  pass pmc_${m}_write WRITE_SIZE --map $m
  pass pmc_${m}_sq "$SQ1" --map $m
done
pass pmc_mesh_sq2 "$SQ2" --map mesh
# cache counters of the two sweeps (L2: TCC_HIT / TCC_MISS; vector L1: accesses and what went on to L2)
for m in mesh mesh-tin; do
  pass pmc_${m}_cache "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --map $m
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_${m}_cache -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --only-main > /dev/null 2>&1
done
# the filter that keeps a healthy spread (likelihood tempered by 1 / beams), lattice and TIN: stats + SQ + L2
for m in mesh mesh-tin; do
  pass stats_${m}_tempered --stats --map $m --temper
  rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_${m}_tempered_sq -- python3 $R/bench.py --map $m --temper --steps 60 --warmup 30 --only-main > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_${m}_tempered_cache -- python3 $R/bench.py --map $m --temper --steps 60 --warmup 30 --only-main > /dev/null 2>&1
done
# BASELINE config 5's shard (524 288 particles, fused landmark step) and the irregular soup: kernel tables only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -- python3 $R/tools/experiments/leg.py config5 30 > $O/bench_config5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_soup -- python3 $R/bench.py --map mesh-soup --steps 10 --warmup 2 --only-main > $O/bench_soup.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config4_rccl -- python3 $R/tools/experiments/leg.py config4_rccl 30 > $O/bench_config4_rccl.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tin_hole -- python3 $R/tools/experiments/leg.py tin_hole 50 > $O/bench_tin_hole.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tin_ragged -- python3 $R/tools/experiments/leg.py tin_ragged 50 > $O/bench_tin_ragged.log 2>&1
cd $R && ROUND=${ROUND:-r06} python3 tools/pmc_summarise.py $O $O/traffic.json > $O/summarise.log 2>&1
# keep only what is small enough to travel back: the stats CSVs, the kernels' counter rows (first dispatches), the summary
for f in $(find $O -name "*counter_collection.csv"); do python3 - $f <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
seen, keep = {}, []
for r in rows:   # the first 12 dispatches of every kernel of the library
    k = r['Kernel_Name']
    if 'rocprim' in k or k.startswith('__amd'):
        continue
    ids = seen.setdefault(k, [])
    if r['Dispatch_Id'] not in ids:
        if len(ids) >= 12:
            continue
        ids.append(r['Dispatch_Id'])
    keep.append(r)
if rows:
    with open(p + '.trim', 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
PY
rm $f; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
head -c 3000 $O/traffic.json
