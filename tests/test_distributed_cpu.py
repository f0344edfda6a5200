"""The N > 1 path (tf_raft_amd/parallel.py) on CPU: two processes, gloo backend, 127.0.0.1.
The per-shard 'model' is the CPU oracle on tiny inputs (tests may use the oracle); what is under
test is the sharding + all-gather logic that bench.py and multi-GPU users run over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from tf_raft_amd.parallel import all_gather_batch, all_gather_batch_async, predict_sharded, shard_range


def test_shard_range_partitions_every_total():
    for total in range(0, 20):
        for world in range(1, 9):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(64, 3, 8) == (24, 32)                       # BASELINE config 3: 8 pairs per GPU
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oracle
        from tf_raft_amd import weights as wm
        rng = np.random.default_rng(0)
        i1 = rng.uniform(0, 255, (total, 64, 64, 3)).astype(np.float32)
        i2 = rng.uniform(0, 255, (total, 64, 64, 3)).astype(np.float32)
        model = oracle.SmallRAFT(wm.init_weights('small', 0), iters_pred=2)

        def predict(inputs):
            return [torch.as_tensor(p) for p in model(inputs)]

        last = predict_sharded(predict, i1, i2)
        every = predict_sharded(predict, i1, i2, gather_all_iterations=True)
        # ragged all-gather on its own
        lo, hi = shard_range(total, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
        g = all_gather_batch(local, total)
        # the same gathers left in flight (bench.py --gpus N: step i's predictions travel while step i + 1 computes):
        # two outstanding at once, waited in order, identical to the blocking result
        p1 = all_gather_batch_async(local, total)
        p2 = all_gather_batch_async(local * 2, total)
        assert torch.equal(p1.wait(), g) and torch.equal(p2.wait(), g * 2) and p1.wait() is p1.wait()
        if rank == 0:
            np.save(os.path.join(tmp, 'last.npy'), last.numpy())
            np.save(os.path.join(tmp, 'every1.npy'), every[1].numpy())
            np.save(os.path.join(tmp, 'g.npy'), g.numpy())
            np.save(os.path.join(tmp, 'n_every.npy'), np.array(len(every)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('total', [4, 3, 1])
def test_two_rank_sharded_prediction_equals_single_process(tmp_path, total):
    """total=3 is a ragged split (2 + 1); total=1 leaves rank 1 with an empty shard."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    import oracle
    from tf_raft_amd import weights as wm
    rng = np.random.default_rng(0)
    i1 = rng.uniform(0, 255, (total, 64, 64, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (total, 64, 64, 3)).astype(np.float32)
    want = oracle.SmallRAFT(wm.init_weights('small', 0), iters_pred=2)([i1, i2])
    last = np.load(tmp_path / 'last.npy')
    assert last.shape == (total, 64, 64, 2)
    np.testing.assert_allclose(last, want[-1], atol=1e-4)
    np.testing.assert_allclose(np.load(tmp_path / 'every1.npy'), want[1], atol=1e-4)
    assert int(np.load(tmp_path / 'n_every.npy')) == 2
    np.testing.assert_array_equal(np.load(tmp_path / 'g.npy')[:, 0], np.arange(total))


def test_single_process_passthrough():
    t = torch.arange(6.0).reshape(3, 2)
    assert all_gather_batch(t, 3) is t
    assert all_gather_batch_async(t, 3).wait() is t
    out = predict_sharded(lambda inp: [inp[0] * 2, inp[0] * 3], t, t)
    np.testing.assert_array_equal(out.numpy(), (t * 3).numpy())


def test_empty_global_batch_is_rejected():
    """An empty GLOBAL batch has no prediction to return (individual ranks may still get an empty shard)."""
    import torch
    from tf_raft_amd.parallel import predict_sharded
    with pytest.raises(ValueError):
        predict_sharded(lambda xs: [xs[0]], torch.zeros((0, 8, 8, 3)), torch.zeros((0, 8, 8, 3)))


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tf_raft_amd.parallel import all_reduce_gradients
    rng = np.random.default_rng(100 + rank)
    grads = {f'layer{i}/kernel': torch.as_tensor(rng.normal(size=(3, 3, 4, 5 + i)).astype(np.float32)) for i in range(6)}
    grads['layer0/bias'] = torch.as_tensor(rng.normal(size=(5,)).astype(np.float32))
    all_reduce_gradients(grads, bucket_bytes=1024)           # small buckets: several collectives, ragged last one
    np.savez(os.path.join(out_dir, f'g{rank}.npz'), **{k.replace('/', '|'): v.numpy() for k, v in grads.items()})
    dist.destroy_process_group()


def test_gradient_all_reduce_averages_over_ranks(tmp_path):
    """The N > 1 path of train_step (BASELINE config 5): bucketed all-reduce of the gradient dict, world size 2, gloo."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(tmp_path / f'g{r}.npz') for r in range(2)]
    for key in got[0].files:
        name = key.replace('|', '/')
        parts = []
        for r in range(2):
            rng = np.random.default_rng(100 + r)
            ref = {f'layer{i}/kernel': rng.normal(size=(3, 3, 4, 5 + i)).astype(np.float32) for i in range(6)}
            ref['layer0/bias'] = rng.normal(size=(5,)).astype(np.float32)
            parts.append(ref[name])
        want = (parts[0] + parts[1]) / 2
        np.testing.assert_allclose(got[0][key], want, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(got[0][key], got[1][key])          # every rank holds the same average


def test_learning_rate_schedule_and_scalers():
    """reference training.py:10-23 + tfa CyclicalLearningRate (train_sintel.py:83-89): triangular first cycle, then flat."""
    from tf_raft_amd import training
    assert training.first_cycle_scaler(1) == 1.0 and training.first_cycle_scaler(2) == 0.0
    assert training.inverse_scaler(4) == 0.25
    sched = training.CyclicalLearningRate(1e-4, 2e-4, step_size=1000, scale_fn=training.first_cycle_scaler, scale_mode='cycle')
    assert sched(0) == pytest.approx(1e-4) and sched(500) == pytest.approx(1.5e-4) and sched(1000) == pytest.approx(2e-4)
    assert sched(1500) == pytest.approx(1.5e-4) and sched(2000) == pytest.approx(1e-4) and sched(2600) == pytest.approx(1e-4)
    inv = training.CyclicalLearningRate(1e-4, 2e-4, step_size=10, scale_fn=training.inverse_scaler)
    assert inv(30) == pytest.approx(1e-4 + 1e-4 * 0.5)                   # peak of the second cycle, scaled by 1/2
    with pytest.raises(ValueError):
        training.CyclicalLearningRate(1e-4, 2e-4, 10, scale_mode='epoch')


def test_bench_gpus2_self_launches_without_torchrun():
    """`python bench.py --gpus 2` typed WITHOUT a launcher (what a driver may do) must become two ranks on its own and
    print rank 0's single JSON line.  RAFT_BENCH_DRY_RUN=cpu swaps the model for a zero tensor so that the launcher,
    rendezvous on 127.0.0.1, in-flight gathers, closing barrier and max-over-ranks timing run in this CPU container."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['RAFT_BENCH_DRY_RUN'] = 'cpu'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['config']['global_batch'] == 2 * line['config']['pairs_per_gpu'] and 'dry_run' in line


def test_bench_gpus8_dry_run_announces_every_rank_before_timing():
    """The driver's 8-GPU launch shape, control flow only (RAFT_BENCH_DRY_RUN=cpu, gloo): eight ranks rendezvous on 127.0.0.1,
    the preflight table (one entry per rank, printed to stderr BEFORE the timed region) names all of them, and rank 0's line
    reports ranks_seen == 8 with BASELINE's full metric string."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['RAFT_BENCH_DRY_RUN'] = 'cpu'
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8
    assert line['metric'].startswith('image-pairs/sec at 448') and 'final-iter EPE' in line['metric']
    pre = [l for l in out.stderr.splitlines() if l.startswith('[bench preflight] ')]
    assert len(pre) == 1, out.stderr[-2000:]
    table = json.loads(pre[0][len('[bench preflight] '):])
    assert table['summary']['ranks_seen'] == 8 and sorted(t['rank'] for t in table['ranks']) == list(range(8))


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0', RAFT_BENCH_DRY_RUN='cpu')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and 'WORLD_SIZE=3' in out.stderr


def _bn_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tf_raft_amd.parallel import all_reduce_mean_
        a = torch.full((3,), float(rank + 1))
        b = torch.arange(4, dtype=torch.float32).reshape(2, 2) * (rank + 1)
        all_reduce_mean_([a, b])
        if rank == 0:
            np.save(os.path.join(tmp, 'a.npy'), a.numpy())
            np.save(os.path.join(tmp, 'b.npy'), b.numpy())
    finally:
        dist.destroy_process_group()


def test_batch_statistics_all_reduce_mean(tmp_path):
    """train_step averages the batch-norm batch statistics over the ranks before the moving-average update (one flattened
    all-reduce), so that every rank keeps the same moving statistics."""
    from tf_raft_amd.parallel import all_reduce_mean_
    mp.spawn(_bn_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    np.testing.assert_allclose(np.load(tmp_path / 'a.npy'), np.full(3, 1.5))
    np.testing.assert_allclose(np.load(tmp_path / 'b.npy'), np.arange(4).reshape(2, 2) * 1.5)
    t = [torch.ones(2)]
    assert all_reduce_mean_(t) is t and torch.equal(t[0], torch.ones(2))      # one process: untouched
