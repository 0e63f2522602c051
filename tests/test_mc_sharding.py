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
    # BASELINE config[3] as stated (bench.py --mc-total 1024): one round of 1024 realisations over 8 ranks = 128 each,
    # realisation r on rank r mod 8, every index taken exactly once
    shards = [shard_indices(0, 1024, r, 8) for r in range(8)]
    assert all(len(s_) == 128 for s_ in shards) and all(i % 8 == r for r, s_ in enumerate(shards) for i in s_)
    assert sorted(i for s_ in shards for i in s_) == list(range(1024))


class _FakeCampaign:
    """the launch()/collect() face of pipeline.McCampaign (the next round is enqueued before the current one is read)"""

    def __init__(self):
        self.launched = []

    def launch(self, idx):
        self.launched.append(list(idx))
        return list(idx)

    def collect(self, handle, with_samples=False):
        e = _errors_of(handle)
        if with_samples:       # a continuous sample per realisation beside the count (the device campaign's EVM)
            return e, np.array([0.05 + 1e-3 * np.random.default_rng([9, int(r)]).standard_normal() for r in handle])
        return e

    def simulate(self, idx):
        return self.collect(self.launch(idx))


def test_split_pool_and_rank_share_keep_one_exchange_and_the_statistics(monkeypatch):
    """bench.py's mc.strong_scaling_rank_share: a rank's fixed share of a campaign dealt over several campaign instances in
    flight (McCampaignPool(split=True)) seen through McRankShare(rank, world).  Host logic only (the campaign is a fake): local
    index i is realisation rank + world i, the parts are contiguous and collected in order, the round ends in ONE exchange, and
    the statistics equal the undivided round's."""
    sys.path.insert(0, ROOT)
    from polmux_amd import mc, pipeline

    class Fake(_FakeCampaign):
        def __init__(self, cfg, frames_per_call, noise_sigma=0.0, noise_provider=None):
            super().__init__()
            self.F = frames_per_call

        def launch(self, idx, keep=None):
            assert len(idx) <= self.F
            return super().launch(idx)

        bits_per_realisation = 4096

        def close(self):
            pass
    monkeypatch.setattr(pipeline, "McCampaign", Fake)
    x = dict(stop=(1e-9, 95), nmin=10)
    out = {}
    for parts in (1, 2, 4):
        pool = pipeline.McCampaignPool(None, frames_per_call=16 // parts, n=parts, split=True)
        view = pipeline.McRankShare(pool, 3, 8)
        sb = mc.ShardedBer(view.simulate, 4096, x, per_rank_per_round=16)
        res = sb.run(max_realisations=16, depth=1)
        assert sb.exchanges == 1 and sb.rounds == 1 and len(sb.counts) == 16
        got = [c.launched for c in pool.camps]
        assert [r for c in got for l in c for r in l] == [3 + 8 * i for i in range(16)]          # contiguous parts, in order
        assert all(len(c) == 1 and len(c[0]) == 16 // parts for c in got)
        out[parts] = (list(sb.counts), [np.asarray(v, dtype=float).tolist() for v in res])
    assert out[1] == out[2] == out[4]
    assert out[1][0] == list(_errors_of([3 + 8 * i for i in range(16)]))


def test_pipelined_rounds_give_the_sequential_statistics():
    """With a simulator that offers launch()/collect(), ShardedBer enqueues round k+1 before it reduces round k; the round
    computed past the stop is discarded: avgber / nruns / stdber still equal the one-realisation-at-a-time loop bit for bit."""
    sys.path.insert(0, ROOT)
    from polmux_amd import mc
    x = dict(stop=(0.05, 95), nmin=10)
    ref, nseq = _sequential(x, 4096.0, 5000)
    camp = _FakeCampaign()
    runner = mc.ShardedBer(camp.simulate, 4096, x, per_rank_per_round=4)
    res = runner.run(max_realisations=5000)
    assert len(runner.counts) == nseq
    for got, want in zip(res, ref):
        np.testing.assert_array_equal(np.asarray(got, dtype=float), np.asarray(want, dtype=float))
    assert len(camp.launched) == runner.rounds + 1                 # exactly one speculative round beyond the stop
    # several rounds in flight (the campaign pool's receivers run beside each other): the same statistics, `depth` rounds dropped
    camp4 = _FakeCampaign()
    r4 = mc.ShardedBer(camp4.simulate, 4096, x, per_rank_per_round=4)
    res4 = r4.run(max_realisations=5000, depth=3)
    assert r4.counts == runner.counts and r4.rounds == runner.rounds
    for got, want in zip(res4, ref):
        np.testing.assert_array_equal(np.asarray(got, dtype=float), np.asarray(want, dtype=float))
    assert len(camp4.launched) == r4.rounds + 3
    assert camp4.launched == [list(range(4 * k, 4 * k + 4)) for k in range(r4.rounds + 3)]   # in order, fixed size
    # the continuous samples of the same rounds go through mc_estimate, block by block (mc_estimate.m:133-212)
    camp3 = _FakeCampaign()
    r3 = mc.ShardedBer(camp3.simulate, 4096, x, per_rank_per_round=4, x_samples=dict(stop=(1e-4, 95), nmin=10))
    r3.run(max_realisations=5000)
    st = mc._State()
    want = None
    for k in range(r3.rounds):
        blk = camp3.collect(list(range(4 * k, 4 * k + 4)), with_samples=True)[1]
        want = mc.mc_estimate(blk, dict(stop=(1e-4, 95), nmin=10), _state=st)
    np.testing.assert_array_equal(r3.samples_result[1]["mean"], want[1]["mean"])
    np.testing.assert_array_equal(r3.samples_result[1]["varlim"], want[1]["varlim"])
    assert abs(r3.samples_result[1]["mean"][0] - 0.05) < 1e-3
    assert r3.exchanges == r3.rounds                                # counts and samples travel in ONE exchange per round
    # a campaign that ends on max_realisations has no speculative round left over
    camp2 = _FakeCampaign()
    r2 = mc.ShardedBer(camp2.simulate, 4096, dict(stop=(1e-9, 95), nmin=10), per_rank_per_round=4)
    r2.run(max_realisations=12)
    assert len(r2.counts) == 12 and len(camp2.launched) == 3


def _sample_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polmux_amd import mc
    camp = _FakeCampaign()
    runner = mc.ShardedBer(camp.simulate, 4096, dict(stop=(0.05, 95), nmin=10), per_rank_per_round=4,
                           x_samples=dict(stop=(1e-4, 95), nmin=10))
    res = runner.run(max_realisations=5000, depth=2)
    so = runner.samples_result[1]
    q.put((rank, [np.asarray(v, dtype=float).tolist() for v in res], list(runner.counts), runner.rounds, runner.exchanges,
           [float(so["mean"][0]), float(so["var"][0]), float(so["nruns"][0])]))
    dist.destroy_process_group()


def test_counts_and_samples_share_one_exchange_per_round_on_two_ranks():
    """World 2 over gloo: the round's int64 counts and the bit patterns of its float64 samples travel in ONE all-reduce
    (disjoint slots, SUM exact -- negative samples included); counts, BER statistics and the samples' mc_estimate equal
    the one-rank campaign's bit for bit."""
    sys.path.insert(0, ROOT)
    from polmux_amd import mc
    camp = _FakeCampaign()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 101) % 2000
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, nseq = _sequential(dict(stop=(0.05, 95), nmin=10), 4096.0, 5000)
    rounds = outs[0][3]
    st = mc._State()
    want = None
    for k in range(rounds):                       # rounds of 8 realisations, fed to mc_estimate in blocks of 4
        for b in range(2):
            idx = list(range(8 * k + 4 * b, 8 * k + 4 * b + 4))
            want = mc.mc_estimate(camp.collect(idx, with_samples=True)[1], dict(stop=(1e-4, 95), nmin=10), _state=st)
    for rank, res, counts, rnds, exch, smp in outs:
        assert len(counts) == nseq and rnds == rounds and exch == rounds       # one exchange per round
        for got, w in zip(res, ref):
            np.testing.assert_array_equal(np.asarray(got), np.asarray(w, dtype=float))
        assert smp == [float(want[1]["mean"][0]), float(want[1]["var"][0]), float(want[1]["nruns"][0])]


def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PLX_SSFM_NO_FUSE"] = "1"       # the ranks of this rehearsal share ONE GPU: barrier-free sweeps
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polmux_amd import mc, pipeline
    torch.cuda.set_device(0)
    cfg = pipeline.HotPathConfig(nsymb=256, nt=16, flag="gps-", nplates=10, dgd=0.2, length=4e4, pavg_mw=1.0, cma_mu=1 / 600,
                                 freqavg=50, dphimax=2e-2)
    camp = pipeline.McCampaign(cfg, frames_per_call=4, noise_sigma=0.28)
    runner = mc.ShardedBer(camp.simulate, camp.bits_per_realisation, dict(stop=(0.01, 99), nmin=50), per_rank_per_round=4,
                           x_samples=dict(stop=(1e-3, 95), nmin=10))
    res = runner.run(max_realisations=48)
    so = runner.samples_result[1]
    q.put((rank, [np.asarray(v, dtype=float).tolist() for v in res], list(runner.counts),
           [float(so["mean"][0]), float(so["var"][0]), float(so["stdmean"][0]), float(so["nruns"][0])] + [float(v) for v in so["varlim"][:, 0]]))
    camp.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_real_campaign_on_two_ranks_equals_one_rank():
    """The real McCampaign (random PMD + noise keyed by the realisation index) sharded over two ranks -- both on cuda:0,
    collectives over gloo: the rehearsal form of the N-GPU run -- gives the one-rank campaign's counts and statistics
    bit for bit."""
    ctx = mp.get_context("spawn")
    outs = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 29500 + (os.getpid() + 13 * world) % 2000
        procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        outs[world] = got
    ref = outs[1][0]
    for rank, res, counts, evm in outs[2]:
        assert counts == ref[2]
        for a, b in zip(res, ref[1]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
        assert evm == ref[3]                                            # mc_estimate of the device EVM samples: sharding-invariant
    assert 0 < ref[3][0] < 1 and ref[3][3] == len(ref[2]) and ref[3][4] <= ref[3][1] <= ref[3][5]   # mean, nruns, variance limits
    assert len(ref[2]) >= 8 and sum(ref[2]) > 0


@pytest.mark.gpu
def test_campaign_pool_with_rounds_in_flight_equals_the_single_campaign():
    """pipeline.McCampaignPool + ShardedBer.run(depth=3): four rounds in flight, each on its own campaign (plans, receiver
    buffers, receiver stream) -- the counts, the BER statistics and the EVM samples' mc_estimate equal the single campaign's
    with one round in flight, bit for bit (realisations are keyed by their index, rounds are replayed in order)."""
    sys.path.insert(0, ROOT)
    from polmux_amd import mc, pipeline
    cfg = pipeline.HotPathConfig(nsymb=256, nt=16, flag="gps-", nplates=10, dgd=0.2, length=4e4, pavg_mw=1.0, cma_mu=1 / 600,
                                 freqavg=50, dphimax=2e-2)
    x = dict(stop=(0.01, 99), nmin=50)
    outs = []
    for pool_n, depth in ((0, 1), (4, 3)):
        camp = pipeline.McCampaignPool(cfg, 4, n=pool_n, noise_sigma=0.28) if pool_n else pipeline.McCampaign(cfg, 4, noise_sigma=0.28)
        runner = mc.ShardedBer(camp.simulate, camp.bits_per_realisation, x, per_rank_per_round=4, x_samples=dict(stop=(1e-3, 95), nmin=10))
        res = runner.run(max_realisations=48, depth=depth)
        so = runner.samples_result[1]
        outs.append((list(runner.counts), [np.asarray(v, dtype=float).tolist() for v in res], float(so["mean"][0]), float(so["var"][0])))
        camp.close()
    assert outs[0] == outs[1]
    assert len(outs[0][0]) >= 8 and sum(outs[0][0]) > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself (fresh child processes, before any GPU call);
    on this one-GPU box as a rehearsal (both on cuda:0, gloo).  One JSON line, n_gpus 2, frames of both ranks counted, the
    Monte-Carlo leg through ShardedBer with its all-reduce."""
    import json
    import subprocess
    env = dict(os.environ, PLX_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "8",
                          "--nsymb", "256", "--nt", "32", "--variants", "2", "--mc-rounds", "2", "--mc-frames", "4", "--mc-total", "12",
                          "--no-cpu-baseline", "--no-single-frame"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rehearsal_all_ranks_on_one_gpu"] is True
    assert d["config"]["bits"] == 2 * 8 * 4 * 256
    assert d["mc"]["realisations"] == 2 * 2 * 4 and d["mc"]["rounds"] == 2 and "gloo" in d["mc"]["exchange"]
    # the strong-scaling leg (BASELINE config[3] as stated: a fixed total): 12 realisations in total = 6 per rank, ONE round, one exchange
    ss = d["mc"]["strong_scaling"]
    assert ss["realisations_total"] == 12 and ss["per_gpu"] == 6 and ss["realisations"] == 12 and ss["rounds"] == 1 and ss["exchanges"] == 1
    # without a rehearsal switch and without that many devices it refuses instead of silently running on one GPU
    env.pop("PLX_BENCH_REHEARSAL")
    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
        assert bad.returncode != 0 and "device(s) visible" in bad.stderr


@pytest.mark.gpu
def test_default_bench_line_carries_configs_2_and_4_as_stated():
    """The driver runs `python bench.py` with no workload flags: after the headline (BASELINE config[1]) the line must carry bounded
    legs of config[2] (16 channels x 10 spans x 32 frames) and config[4] (2^20 samples x 40 spans, rank 0's share of the 64-point
    ladder over 8 GPUs) AS STATED, each with value, roofline (HBM and FP64 fractions, offline traffic) and its own CPU baseline."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PLX_BENCH_REHEARSAL"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--mc-rounds", "0", "--no-gateway",
                          "--no-single-frame", "--no-cohmix-line", "--cpu-frames", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "BASELINE config[1]" in d["config"]["workload"] and set(d["configs"]) == {"c2", "c4"}
    c2, c4 = d["configs"]["c2"], d["configs"]["c4"]
    assert "BASELINE config[2]: 16 channels, 10x80 km" in c2["config"]["workload"] and c2["config"]["channels_per_frame"] == 16
    assert c2["config"]["frames_per_gpu_per_step"] == 32 and c2["steps"] == 2
    assert "BASELINE config[4]: 2^20-sample frame x 40 spans x 64-point ladder)" in c4["config"]["workload"]
    assert c4["config"]["frames_per_gpu_per_step"] == 8 and len(c4["config"]["ladder_dbm"]) == 8
    np.testing.assert_allclose(c4["config"]["ladder_dbm"], -4.0 + 12.0 * 8 * np.arange(8) / 63.0, atol=1e-9)      # every 8th point of the ladder
    for leg in (c2, c4):
        assert leg["value"] > 0 and leg["unit"] == "Gsample/s" and leg["dtype"] == "f64" and leg["ms_per_step"] > 0
        r = leg["roofline"]
        assert r["kernel"] == "k_colx16" and 0.05 < r["frac"] < 1.0 and r["bound"] in ("hbm", "fp64-valu")
        assert r["fp64"] is not None and 0.0 < r["fp64"]["frac"] < 1.0 and r["fp64"]["peak"] == pytest.approx(78.6)
        assert r["traffic"] is not None and 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.3
        cb = leg["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and 0 < cb["value"] < leg["value"]
    assert d["roofline"]["fp64"]["frac"] > 0.05 and d["roofline"]["traffic"] is not None


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    """One small `bench.py` run at N = 1: ONE JSON line with the driver's fields, the roofline object of the dominant
    kernel (real bytes per launch over a live HIP-event duration) and the CPU baseline timed beside it."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PLX_BENCH_REHEARSAL"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "32", "--nsymb", "256",
                          "--nt", "64", "--variants", "2", "--mc-rounds", "1", "--mc-frames", "8", "--mc-total", "16", "--cpu-frames", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "Gsample/s" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] in r["kernels"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    k = r["kernels"][r["kernel"]]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (k["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["bytes_per_sample_per_launch"] == 64.0 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Gsample/s" and c["value"] > 0 and "frame" in c["sample"]
    assert d["mc"]["realisations"] == 8 and d["mc"]["evm_mc_estimate"]["nruns"] == 8
    assert d["mc"]["strong_scaling"]["per_gpu"] == 16 and d["mc"]["strong_scaling"]["realisations"] == 16
    g = d["gateway"]["plx_cmapolardemux"]
    assert g["passes"] == g["per_pass_mex_loop_passes"] == g["oracle_passes"] and g["allocations_in_repeats"] == 0
    assert "fibre_step" in d["config"] and d["config"]["channels_per_frame"] == 1


def _nccl_worker(q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29500 + (os.getpid() + 211) % 2000)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # RCCL, as bench.py does for N > 1
    from polmux_amd import mc, pipeline
    cfg = pipeline.HotPathConfig(nsymb=256, nt=16, flag="gps-", nplates=10, dgd=0.2, length=4e4, pavg_mw=1.0, cma_mu=1 / 600,
                                 freqavg=50, dphimax=2e-2)
    camp = pipeline.McCampaign(cfg, frames_per_call=4, noise_sigma=0.28)
    out = {}
    for dev in ("cuda", "cpu"):               # the collective's buffer on the device (RCCL) / the host reference, no process group use
        runner = mc.ShardedBer(camp.simulate, camp.bits_per_realisation, dict(stop=(1e-9, 99), nmin=50), per_rank_per_round=4,
                               device=dev if dev == "cuda" else None, x_samples=dict(stop=(1e-9, 95), nmin=10))
        if dev == "cpu":
            runner._dist = lambda: (None, 0, 1)
        res = runner.run(max_realisations=8)  # two rounds
        out[dev] = (list(runner.counts), runner.rounds, runner.exchanges, float(runner.samples_result[1]["mean"][0]),
                    [np.asarray(v, dtype=float).tolist() for v in res])
    t = torch.ones(4, dtype=torch.int64, device="cuda")
    dist.all_reduce(t)
    out["allreduce"] = t.cpu().tolist()
    q.put(out)
    camp.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_backend_world_one_runs_the_rccl_exchange():
    """The N-GPU code path of bench.py -- init_process_group('nccl', device_id=...), ShardedBer with its exchange buffer in
    device memory -- executes on a one-GPU box: world size 1, two rounds, the all-reduce issued through RCCL; results equal
    the same campaign with no process group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(q,))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["cuda"] == out["cpu"]
    assert out["cuda"][1] == 2 and out["cuda"][2] == 2 and len(out["cuda"][0]) == 8
    assert out["allreduce"] == [1, 1, 1, 1]
