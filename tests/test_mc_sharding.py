"""N>1 path on CPU: two gloo ranks shard the realisations (r -> rank r mod G), exchange ONE all-reduce of
integer error counts per round and replay ber_estimate's sequential recursion -- results equal the one-rank
sequential loop bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _errors_of(indices):
    """deterministic per-realisation error counts (stands in for the device pipeline)"""
    out = []
    for r in indices:
        g = np.random.default_rng([7, int(r)])
        out.append(int(g.binomial(4096, 0.004)))
    return np.array(out, dtype=np.int64)


def _sequential(x, M, limit):
    from polmux_amd import mc
    st = mc._State()
    res, n = None, 0
    for r in range(limit):
        res = mc.ber_estimate_counts(int(_errors_of([r])[0]), M, x, _state=st)
        n += 1
        if not res[0][0]:
            break
    return res, n


def _worker(rank, world, port, per_round, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polmux_amd import mc
    calls = []

    def simulate(idx):
        calls.extend(idx)
        return _errors_of(idx)

    x = dict(stop=(0.05, 95), nmin=10)
    runner = mc.ShardedBer(simulate, 4096, x, per_rank_per_round=per_round)
    res = runner.run(max_realisations=5000)
    q.put((rank, [np.asarray(v, dtype=float).tolist() for v in res], len(runner.counts), runner.rounds, calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,per_round", [(2, 4), (2, 1), (3, 5)])
def test_sharded_ber_equals_sequential(world, per_round):
    sys.path.insert(0, ROOT)
    x = dict(stop=(0.05, 95), nmin=10)
    ref, nseq = _sequential(x, 4096.0, 5000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * world + per_round) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_round, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, naccepted, rounds, calls in outs:
        assert naccepted == nseq                                    # realisations beyond the stop are discarded
        for got, want in zip(res, ref):
            np.testing.assert_array_equal(np.asarray(got), np.asarray(want, dtype=float))   # bit for bit
        assert all(c % world == rank for c in calls)               # r -> rank r mod G
        assert rounds == -(-nseq // (per_round * world))
    assert sorted(c for o in outs for c in o[4])[:nseq] == list(range(nseq))


def test_shard_indices():
    from polmux_amd.mc import shard_indices
    assert shard_indices(0, 8, 1, 4) == [1, 5]
    assert shard_indices(8, 8, 0, 4) == [8, 12]
    assert sum(len(shard_indices(0, 1024, r, 8)) for r in range(8)) == 1024
