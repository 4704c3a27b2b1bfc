#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for m in mesh grid; do
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc6_${m}_fetch -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc6_${m}_write -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc6_${m}_sq -- python3 $R/bench.py --map $m --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
echo done
