"""Monte-Carlo estimators and the multi-GPU realisation sharding.

  ber_estimate(pat_hat, pat, x[, nind])   ber_estimate.m:1, recursion :97-143
  mc_estimate(s, x[, nind])               mc_estimate.m:1, recursion :133-212

Both keep MATLAB's ``persistent`` state (module level here), cleared when every
entry of ``cond`` has turned false (ber_estimate.m:132,137).

Sharding: realisation r runs on rank r mod G; a round of W*G
realisations ends with ONE all-reduce (RCCL over xGMI when the backend is nccl)
of the per-realisation integer error counts, after which every rank replays the
reference's sequential recursion in realisation order and discards whatever lies
beyond the first index at which ``cond`` turned false -- so avgber/stdber/nruns
equal the one-GPU sequential loop bit for bit, for any G.
"""
import math

import numpy as np


def erfcinv(y):
    """MATLAB erfcinv (builtin, not in the reference) by Newton-Halley on math.erfc."""
    if y <= 0:
        return math.inf
    if y >= 2:
        return -math.inf
    if y == 1:
        return 0.0
    yy = y if y < 1 else 2 - y
    t = math.sqrt(-2 * math.log(yy / 2))
    x = -0.70711 * ((2.30753 + t * 0.27061) / (1 + t * (0.99229 + t * 0.04481)) - t)
    for _ in range(60):
        err = math.erfc(x) - yy
        dx = err / (-2 / math.sqrt(math.pi) * math.exp(-x * x))
        step = dx / (1 + x * dx)
        x -= step
        if abs(step) <= 1e-17 * abs(x):
            break
    return x if y < 1 else -x


def _get(x, name, default=None):
    if isinstance(x, dict):
        return x.get(name, default)
    return getattr(x, name, default)


def _has(x, name):
    return (name in x) if isinstance(x, dict) else hasattr(x, name)


class _State:
    """the persistent variables of mc_run / complete_mc"""

    def __init__(self):
        self.first = False

    def init(self, dim, stop, two):
        self.n = np.ones(dim)
        self.avg = np.zeros(dim)
        self.var = np.zeros(dim)
        self.varlim = np.zeros((2, dim))
        self.cond = np.ones(dim, dtype=bool)
        self.first = True
        self.epsilon = [0.0, 0.0]
        if stop is not None:
            self.epsilon[0] = math.sqrt(2) * erfcinv(1 - stop[1] / 100)       # ber_estimate.m:113
            if two:
                self.epsilon[1] = math.sqrt(2) * erfcinv(1 + stop[1] / 100)   # mc_estimate.m:155


_ber_state = _State()
_mc_state = _State()


def reset_persistent():
    """`clear functions` in MATLAB terms."""
    _ber_state.first = False
    _mc_state.first = False


def _ber_update(st, err, M, x, nind):
    """mc_run with the error count already formed (ber_estimate.m:105-142)."""
    nmin = _get(x, "nmin", 1) if _has(x, "nmin") else 1                       # :75
    has_dim = _has(x, "dim")
    if not has_dim:
        if nind is not None:
            raise ValueError("missing variable nind in ber_estimate")         # :76-83 (sic)
        dim, nind = 1, 1
    else:
        dim = int(_get(x, "dim"))
    stop = _get(x, "stop") if _has(x, "stop") else None
    if stop is not None and (stop[1] > 100 or stop[1] < 0):
        raise ValueError("the Gaussian confidence must be < 100 and > 0")     # :85-87
    k = nind - 1
    if not st.first:
        st.init(dim, stop, False)
    nnew = st.n[k] * M                                                        # :117
    N = (st.n - 1) * M                                                        # :121
    varerr = (err - err ** 2 / M) / (M - 1)                                   # :122
    avgerr = err / M                                                          # :123
    st.var[k] = ((N[k] - 1) * st.var[k] + (M - 1) * varerr +
                 N[k] * M / (N[k] + M) * (st.avg[k] - avgerr) ** 2) / (N[k] + M - 1)   # :125-126
    st.avg[k] = ((st.n[k] - 1) * st.avg[k] + avgerr) / st.n[k]                # :127
    stdber = np.sqrt(st.var / (N + M))                                        # :128
    clear = False
    if stop is not None:
        if (st.epsilon[0] * stdber[k] < stop[0] * st.avg[k]) and (st.avg[k] * nnew >= nmin):   # :131
            st.cond[k] = False
            clear = not st.cond.any()
    else:
        if st.avg[k] * nnew > nmin:                                           # :137
            st.cond[k] = False
            clear = not st.cond.any()
    st.n[k] = st.n[k] + 1
    out = (st.cond.copy(), st.avg.copy(), (st.n - 1) * M, stdber)
    if clear:
        st.first = False
    return out


def ber_estimate(pat_hat, pat, x, nind=None, _state=None):
    """[cond,avgber,nruns,stdber] = ber_estimate(pat_hat,pat,x[,nind])."""
    pat = np.asarray(pat)
    err = float(np.sum(pat != np.asarray(pat_hat)))                           # :119
    return _ber_update(_state or _ber_state, err, float(pat.size), x, nind)


def ber_estimate_counts(err, M, x, nind=None, _state=None):
    """Same recursion fed with an error COUNT (what the device returns, plx_decide_count_dev)."""
    return _ber_update(_state or _ber_state, float(err), float(M), x, nind)


def mc_estimate(s, x, nind=None, _state=None):
    """[cond,out] = mc_estimate(s,x[,nind])  (mc_estimate.m:104-212)."""
    st = _state or _mc_state
    nmin = _get(x, "nmin", 50) if _has(x, "nmin") else 50                     # DEFMIN :105
    method = _get(x, "method", "mean") if _has(x, "method") else "mean"
    if method not in ("mean", "var"):
        raise ValueError("field method must be 'mean' or 'var'")
    if not _has(x, "dim"):
        if nind is not None:
            raise ValueError("missing variable nind mc_estimate")
        dim, nind = 1, 1
    else:
        dim = int(_get(x, "dim"))
    stop = _get(x, "stop") if _has(x, "stop") else None
    if stop is not None and (stop[1] > 100 or stop[1] < 0):
        raise ValueError("the Gaussian confidence must be < 100 and > 0")
    s = np.asarray(s, dtype=float)
    if min(s.shape) == 1 or s.ndim == 1:
        s = s.reshape(-1, 1)
        nind2 = 0
    else:
        nind2 = nind - 1
    M = s.shape[0]
    k = nind - 1
    if not st.first:
        st.init(dim, stop, True)
    runs = st.n * M                                                           # :160
    N = (st.n - 1) * M
    if M == 1 and st.n[k] == 1:
        st.var[k] = 0.0
        st.avg[k] = s[0, 0]
    else:
        varblk = np.var(s, axis=0, ddof=1)
        avgblk = np.mean(s, axis=0)
        st.var[k] = ((N[k] - 1) * st.var[k] + (M - 1) * varblk[nind2] +
                     N[k] * M / (N[k] + M) * (st.avg[k] - avgblk[nind2]) ** 2) / (N[k] + M - 1)
        st.avg[k] = ((st.n[k] - 1) * st.avg[k] + avgblk[nind2]) / st.n[k]
    stdmean = np.sqrt(st.var / (N + M))                                       # :174
    x21mdh = 0.5 * (st.epsilon[0] + math.sqrt(2 * (N[k] + M) - 3)) ** 2       # :175
    x2dh = 0.5 * (st.epsilon[1] + math.sqrt(2 * (N[k] + M) - 3)) ** 2
    st.varlim[0, k] = (N[k] + M - 1) * st.var[k] / x21mdh
    st.varlim[1, k] = (N[k] + M - 1) * st.var[k] / x2dh
    clear = False
    if stop is not None:
        if method == "mean":
            if (st.epsilon[0] * stdmean[k] < stop[0] * abs(st.avg[k])) and (runs[k] >= nmin):
                st.cond[k] = False
                clear = not st.cond.any()
        else:
            if (st.varlim[1, k] - st.varlim[0, k]) / st.var[k] < stop[0] and (runs[k] >= nmin):
                st.cond[k] = False
                clear = not st.cond.any()
    else:
        if runs[k] > nmin:
            st.cond[k] = False
            clear = not st.cond.any()
    st.n[k] = st.n[k] + 1
    out = dict(nruns=(st.n - 1) * M, mean=st.avg.copy(), var=st.var.copy(), varlim=st.varlim.copy(), stdmean=stdmean)
    cond = st.cond.copy()
    if clear:
        st.first = False
    return cond, out


# ------------------------------------------------------------------ sharding ---
def shard_indices(start, count, rank, world):
    """realisation r -> rank r mod world."""
    return [r for r in range(start, start + count) if r % world == rank]


class ShardedBer:
    """Drives a Monte-Carlo BER campaign over torch.distributed ranks.

    simulate(indices) -> int64 array of error counts for those realisation indices (the
    device pipeline); bits_per_realisation = numel(pat).  One all_reduce(SUM) per round."""

    def __init__(self, simulate, bits_per_realisation, x, per_rank_per_round=8, group=None, device=None, x_samples=None):
        """x_samples: optional mc_estimate options (mc_estimate.m:104-131).  When given and the simulator's collect()
        can return a continuous sample per realisation beside the error count (McCampaign: the EVM), every round's
        sample vector travels in the SAME all-reduce as the counts (bit patterns in the int64 buffer) and is fed to mc_estimate in blocks of
        per_rank_per_round samples: `samples_result` = (cond, out) of the last call."""
        self.x_samples = x_samples
        self.samples_state = _State()
        self.samples_result = None
        self.simulate = simulate
        self.M = float(bits_per_realisation)
        self.x = x
        self.w = int(per_rank_per_round)
        self.group = group
        self.device = device
        self.state = _State()
        self.counts = []          # accepted per-realisation error counts, in index order
        self.rounds = 0
        self.exchanges = 0        # all-reduces issued (one per round)

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(self.group), dist.get_world_size(self.group)
        return None, 0, 1

    def run(self, max_realisations=1 << 30, depth=1):
        """Rounds of per_rank_per_round realisations per rank until the reference's stop rule fires.  When the simulator
        offers launch()/collect() (pipeline.McCampaign, McCampaignPool) up to `depth` FURTHER rounds are already enqueued
        while a round's counts are reduced and replayed: rounds computed past the stop are simply discarded, the
        statistics are unchanged (the rounds are fixed in size and replayed in order whatever the depth)."""
        import collections
        import torch
        dist, rank, world = self._dist()
        sim = self.simulate
        owner = getattr(sim, "__self__", None)
        launch = getattr(owner, "launch", None) if owner is not None and getattr(sim, "__name__", "") == "simulate" else None
        collect = getattr(owner, "collect", None) if launch is not None else None
        depth = max(1, int(depth))

        def plan(start):
            n_round = min(self.w * world, max_realisations - start)
            return n_round, shard_indices(start, n_round, rank, world)

        start = 0
        result = None
        cond = True
        inflight = collections.deque()   # (start, n_round, mine, handle) of the rounds enqueued and not yet collected
        nxt = 0

        def top_up(limit):
            nonlocal nxt
            while len(inflight) < limit and nxt < max_realisations:
                n_round, mine = plan(nxt)
                inflight.append((nxt, n_round, mine, launch(mine) if mine else None))
                nxt += n_round

        if launch is not None:
            top_up(1)
        while cond and start < max_realisations:
            if launch is not None:
                start, n_round, mine, handle = inflight.popleft()
                top_up(depth)               # speculative: enqueue the next round(s) before looking at this one
                if self.x_samples is not None and mine:
                    local, local_s = collect(handle, with_samples=True)
                    local = np.asarray(local, dtype=np.int64)
                else:
                    local = np.asarray(collect(handle), dtype=np.int64) if mine else np.zeros(0, np.int64)
                    local_s = None
            else:
                n_round, mine = plan(start)
                local = np.asarray(sim(mine), dtype=np.int64) if mine else np.zeros(0, np.int64)
                local_s = None
            # ONE exchange step per round: every rank contributes its slots of the round's vector -- the int64
            # error counts and, when a continuous sample travels with them, the BIT PATTERNS of the float64 samples in the
            # second half of the same int64 buffer.  Slots are disjoint (every other rank holds 0 there), so SUM is exact
            # for both halves.
            with_s = self.x_samples is not None and launch is not None
            vec = torch.zeros(n_round * (2 if with_s else 1), dtype=torch.int64, device=self.device or "cpu")
            if mine:
                slots = torch.as_tensor([r - start for r in mine], device=vec.device)
                vec[slots] = torch.as_tensor(local, device=vec.device)
                if with_s and local_s is not None:
                    bits = np.ascontiguousarray(np.asarray(local_s, dtype=np.float64)).view(np.int64)
                    vec[slots + n_round] = torch.as_tensor(bits, device=vec.device)
            if dist is not None:            # (also with one rank: the collective is the same code path on every world size)
                dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
            self.exchanges += 1
            host = vec.cpu().numpy()
            counts = host[:n_round]
            self.rounds += 1
            if with_s:
                # blocks of per_rank_per_round samples in realisation order, whatever the number of ranks: the recursion
                # then sees the same sequence of blocks for any sharding (bit-equal statistics)
                svn = np.ascontiguousarray(host[n_round:]).view(np.float64)
                for b0 in range(0, n_round, self.w):
                    self.samples_result = mc_estimate(svn[b0:b0 + self.w], self.x_samples, _state=self.samples_state)
            # replay the reference's sequential recursion in realisation order (ber_estimate.m:116-141)
            for c in counts:
                result = ber_estimate_counts(int(c), self.M, self.x, _state=self.state)
                self.counts.append(int(c))
                if not result[0][0]:
                    cond = False
                    break
            start += n_round
        while inflight:                      # speculative rounds past the stop: wait for them, drop them
            h = inflight.popleft()[3]
            if h is not None:
                collect(h)
        return result
