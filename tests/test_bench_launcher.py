"""bench.py --gpus N outside a launcher starts N ranks itself (the reference's tools/dist_train.sh:5:
`python -m torch.distributed.launch --nproc_per_node=$GPUS`), and never runs a smaller job under N's name."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env_without_launcher():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_gpus_2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this box has two GPUs')
    out = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1'], env=_env_without_launcher(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert '--gpus 2' in out.stderr and 'GPU(s)' in out.stderr
    assert not any(l.startswith('{') for l in out.stdout.splitlines()), 'no JSON line may be printed for a job that did not run'


def test_gpus_flag_must_agree_with_the_launcher():
    env = _env_without_launcher()
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, BENCH, '--gpus', '1'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'disagrees with WORLD_SIZE=2' in out.stderr


def test_spawn_path_sets_rank_environment(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    probe = tmp_path / 'probe.py'
    probe.write_text(
        "import json, os, sys\n"
        "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "open(os.path.join(sys.argv[1], 'rank%s.json' % os.environ['RANK']), 'w').write(json.dumps({k: os.environ.get(k) for k in keys}))\n"
        "sys.exit(3 if os.environ['RANK'] == '1' and len(sys.argv) > 2 else 0)\n")
    assert bench.launch_ranks(2, [str(tmp_path)], device_count=2, script=str(probe)) == 0
    seen = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(2)]
    for r, e in enumerate(seen):
        assert e['RANK'] == str(r) and e['LOCAL_RANK'] == str(r) and e['WORLD_SIZE'] == '2'
        assert e['MASTER_ADDR'] == '127.0.0.1' and e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert seen[0]['MASTER_PORT'] == seen[1]['MASTER_PORT'] and int(seen[0]['MASTER_PORT']) > 0
    # the worst exit code of the children is the parent's
    assert bench.launch_ranks(2, [str(tmp_path), 'fail'], device_count=2, script=str(probe)) == 3
    # fewer devices than ranks: refused before anything is started
    assert bench.launch_ranks(4, [str(tmp_path)], device_count=2, script=str(probe)) == 2
