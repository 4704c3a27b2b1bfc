#!/bin/bash
# tools/profile_gpu.sh -- run on the GPU box (gpurun): kernel-trace stats and the three PMC passes for the
# mesh and grid workloads; results under gpurun_out/prof_${ROUND:-r05}/.  Counters are collected in their own runs
# (never together with a trace), the program itself follows `--` (no wrapper that would re-exec).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${ROUND:-r05}
rm -rf $O && mkdir -p $O
for m in mesh grid; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/bench.py --map $m --steps 50 --warmup 5 --only-main > $O/bench_$m.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${m}_fetch -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --only-main > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${m}_write -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --only-main > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $O/pmc_${m}_sq -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --only-main > /dev/null 2>&1
done
# BASELINE config 5's shard (524 288 particles, fused landmark step) and the irregular soup: kernel tables only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -- python3 $R/tools/experiments/leg.py config5 30 > $O/bench_config5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_soup -- python3 $R/bench.py --map mesh-soup --steps 10 --warmup 2 --only-main > $O/bench_soup.log 2>&1
cd $R && python3 tools/pmc_summarise.py $O $O/traffic.json > $O/summarise.log 2>&1
# keep only what is small enough to travel back: the stats CSVs, the dominant kernel's counter rows, the summary
find $O -name "*kernel_stats.csv" | head
for f in $(find $O -name "*counter_collection.csv"); do grep -E "Correlation_Id|k_mbes_fast|k_mbes_sweep|k_mbes_slice" $f > $f.mbes; rm $f; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
head -c 1500 $O/traffic.json
