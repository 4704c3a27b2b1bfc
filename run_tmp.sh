#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for m in mesh grid; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof7_${m} -- python3 $R/bench.py --map $m --steps 50 --warmup 5 > $R/gpurun_out/bench7_${m}_prof.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof7_meshgen -- python3 $R/bench.py --map mesh --mesh-general --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench7_meshgen.log 2>&1
cd $R
python bench.py > gpurun_out/bench7_mesh.json 2> gpurun_out/bench7_mesh.err
python bench.py --map grid > gpurun_out/bench7_grid.json 2> gpurun_out/bench7_grid.err
MCL_FORCE_COMM=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench7_mesh_rccl1.json 2> gpurun_out/bench7_rccl1.err
wc -l gpurun_out/bench7_mesh.json gpurun_out/bench7_mesh_rccl1.json
cut -c1-160 gpurun_out/bench7_mesh.json
