"""`python bench.py --gpus N` starts its own N rank processes (the driver launches it without torch.distributed.run):
rendezvous on 127.0.0.1, barrier, max-over-ranks reduction, rank 0's single JSON line, non-zero exit when a rank dies.
Rehearsed on CPU over gloo with --rehearse-launch (no GPU, no model, never a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, n=2, extra_args=()):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--rehearse-launch'] + list(extra_args),
                          env=env, capture_output=True, text=True, timeout=600)


def test_bench_launches_its_own_ranks():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line['rehearsal'] is True and line['n_gpus'] == 2 and line['ranks'] == 2 and line['backend'] == 'gloo'
    assert line['max_dt'] == 2.0                          # MAX over ranks of (1 + rank)


def test_a_failed_rank_fails_the_run():
    r = _run({'TAI_BENCH_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert 'rank 1 exited with code 3' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_under_an_external_launcher_the_rank_does_not_relaunch():
    # RANK present (torch.distributed.run's environment): the process IS a rank and must not spawn children
    r = _run({'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'}, n=1)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])['n_gpus'] == 1


def test_the_multi_rank_training_leg_runs_over_gloo_and_keeps_replicas_identical():
    """configs[2] under `--gpus N`: every rank runs the G-then-D update through the bucketed all-reduces
    (/root/reference/src/environments/environments.py:348-355's order) and rank 0 reports ms per update, all-reduce bytes and the
    replica-identity check.  Rehearsed here on CPU over gloo with a reduced model: the plumbing, not a measurement."""
    r = _run(extra_args=['--rehearse-train'])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    rec = line['train_step_dp']
    assert 'error' not in rec, rec
    assert rec['ranks'] == 2 and rec['backend'] == 'gloo' and rec['updates_timed'] == 2
    assert rec['replicas_identical'] is True and rec['replica_checksum_max_abs_delta'] == 0.0     # ranks started from DIFFERENT weights and clips
    assert rec['allreduce_bytes_per_update']['generator'] > 0 and rec['allreduce_bytes_per_update']['discriminator'] > 0
    assert rec['ms_per_update'] > 0 and rec['losses_finite'] is True
    assert 'REHEARSAL' in rec['workload']
