"""bench.py's own launcher: `python bench.py --gpus N` with no launcher around it must start N rank processes (and a
mismatch between --gpus and WORLD_SIZE must fail instead of reporting the wrong GPU count).  The ranks run bench.py's
--worker-stub body (process group on gloo, barrier, MAX over ranks, one JSON line from rank 0): no GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def test_gpus_flag_starts_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--worker-stub'],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                     # one line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['gpus_flag'] == 2
    assert out['steps'] == 3 and out['warmup'] == 1      # the flags reach the ranks
    assert out['max_over_ranks'] == 2.0                  # rank 1 took part in the MAX reduction


def test_single_gpu_run_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--worker-stub'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['ranks_seen'] == 1


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--worker-stub'], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert '--gpus 4 but WORLD_SIZE=1' in r.stderr
