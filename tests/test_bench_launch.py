"""`python bench.py --gpus N` starts its N ranks itself (stands in for Lightning's DDP launcher,
regression/main/entry.py:110-113).  CPU test of the launch logic with N = 2 up to the point of device selection
(HOTK_BENCH_DRY_LAUNCH=1: rendezvous over gloo, the timing MAX all-reduce, rank 0's line), of the exit code when a
rank fails, and of the n_gpus == --gpus contract."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_forwards_one_line():
    for mode in ('graph-minibatch', 'edge-shard'):
        r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--mode', mode], HOTK_BENCH_DRY_LAUNCH='1')
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout                      # stdout carries rank 0's JSON line and nothing else
        line = json.loads(lines[0])
        assert line['n_gpus'] == 2 and line['gpus_arg'] == 2 and line['mode'] == mode
        assert line['slowest_rank_plus_1'] == 2.0             # the MAX over ranks saw both ranks


def test_rank_failure_gives_nonzero_exit():
    # --gpus 2 under a launcher that started ONE rank: the contract check fails in the child -> parent exits non-zero
    r = _run(['--gpus', '2'], HOTK_BENCH_DRY_LAUNCH='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    assert r.returncode != 0 and 'n_gpus' in (r.stderr + r.stdout)
    # a child that dies during start-up (unknown flag) must surface through the parent's exit code
    r = _run(['--gpus', '2', '--no-such-flag'], HOTK_BENCH_DRY_LAUNCH='1')
    assert r.returncode != 0


def test_parent_makes_no_gpu_call():
    """The launching parent must not initialise the GPU (no torch.cuda call, no library load) before it spawns."""
    src = open(BENCH).read()
    body = src[src.index('def launch_ranks'):src.index('def dry_launch')]
    assert 'torch.cuda' not in body and '_lib' not in body
    main = src[src.index('def main():'):]
    assert main.index('launch_ranks(') < main.index('torch.cuda')
