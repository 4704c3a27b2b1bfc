"""bench.py's own multi-rank launcher (`python bench.py --gpus N` without WORLD_SIZE): spawns N ranks before
anything touches the GPU, relays rank 0's JSON line, refuses a rank count that differs from --gpus and never
hangs (a failing or stuck rank ends the whole job with a non-zero exit).  CPU only: --dry-run stops after the
gloo rendezvous, barrier and rank count -- the code path up to there is the one the GPU run takes."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, universal_newlines=True)
    return p.returncode, p.stdout, p.stderr, time.time() - t0


def test_launcher_spawns_two_gloo_ranks_and_prints_one_json_line():
    rc, out, err, _ = _run(['--gpus', '2', '--dry-run'])
    assert rc == 0, err
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['gloo_ranks'] == 2 and d['spawned_by_bench'] is True


def test_launcher_propagates_a_failing_rank():
    rc, out, err, el = _run(['--gpus', '2', '--dry-run', '--dry-run-fail-rank', '1'])
    assert rc == 7
    assert 'rank 1 exited with code 7' in err
    assert not [l for l in out.splitlines() if l.startswith('{')]
    assert el < 120


def test_launcher_never_hangs_on_a_stuck_rank():
    rc, out, err, el = _run(['--gpus', '2', '--dry-run', '--dry-run-hang-rank', '1', '--launch-timeout', '20'])
    assert rc == 124
    assert el < 120


def test_rank_count_mismatch_is_refused():
    rc, out, err, _ = _run(['--gpus', '4', '--dry-run'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert rc == 2 and 'refusing' in err


def test_external_launcher_contract_still_works():
    # python -m torch.distributed.run ... bench.py --gpus 2 (the driver's form): ranks come from the environment
    port = subprocess.check_output([sys.executable, '-c',
                                    'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])'],
                                   universal_newlines=True).strip()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', port, BENCH, '--gpus', '2', '--dry-run'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, universal_newlines=True)
    assert p.returncode == 0, p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['gloo_ranks'] == 2 and d['spawned_by_bench'] is False
