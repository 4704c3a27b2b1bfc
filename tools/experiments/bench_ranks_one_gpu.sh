#!/bin/bash
# tools/experiments/bench_ranks_one_gpu.sh [N ...] -- on a 1-GPU box: bench.py's multi-rank worker in the driver's launch
# form (python -m torch.distributed.run --nproc-per-node N) with N REAL processes on the one GPU, librccl answered by the
# cross-process test double (tests/fake_nccl; what tests/test_gpu_bench_multirank.py does for N = 2).  Not a measurement:
# it shows that the N-rank job starts, steps, reports ONE line and ends.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
L=$R/build/fake_nccl/libfake_nccl.so
mkdir -p $(dirname $L)
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -Wall -o $L $R/tests/fake_nccl/fake_nccl.cpp || exit 1
for n in ${@:-4 8}; do
  port=$(python3 -c 'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])')
  LD_PRELOAD=$L FAKE_NCCL_SHM=1 FAKE_NCCL_TIMEOUT_S=60 MCL_BENCH_ONE_DEVICE=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $port $R/bench.py --gpus $n --steps 5 --warmup 2 --only-main 2> /tmp/ranks_$n.err | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('n_gpus', 'rccl_ranks', 'value', 'ms_per_step', 'particles_total', 'scaling', 'rccl_library', 'pose_rmse_m')}, d.get('exchange', {}).get('lost_slots_per_step_by_rank'))"
  echo "rc=$? ranks=$n"; tail -2 /tmp/ranks_$n.err
done
