#!/usr/bin/env python
"""bench.py -- dual-pol Gsample/s through SSFM + Rx-DSP on MI355X, and Monte-Carlo realisations/s.

One "step" = one pass of the hot path (fiber 'g-s-' split-step Fourier propagation
-> 2-sps pick -> CDE_OFDE overlap-save -> DspPdmCohQpsk with CMA + carrier recovery
-> decisions / error count) over ONE BATCH of synthetic PDM-QPSK frames that is
already resident in HBM when the timed region starts.  Workload at N=1: BASELINE
config[1] (Run_my_PDM_QPSK-style: 28 Gbaud PDM-QPSK, 2^16-sample frame, one 80 km
SSMF span + CMA demux), batched over --frames independent frames per GPU; the frames
of a batch carry --variants different de Bruijn sequences (and, with --power-ladder,
BASELINE config[4]'s -4...+8 dBm launch powers, i.e. data-dependent step counts).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  `--gpus N` with no
WORLD_SIZE in the environment starts the N ranks itself, as fresh child processes, BEFORE
anything touches a GPU; under torchrun (RANK/LOCAL_RANK/WORLD_SIZE set) it is one rank.
Frames are independent units, so ranks shard them with no data-path collective (weak
scaling); the exchanges are the all-reduce of the error counters and, in the Monte-Carlo
leg, one all-reduce of the round's per-realisation error counts.

After the timed region (outside it) every run also does a short Monte-Carlo leg -- BASELINE
config[3]: fresh random birefringence + amplifier ASE per realisation, through
mc.ShardedBer (realisation r on rank r mod G, one all-reduce per round, host replay of
ber_estimate.m:116-141) -- and reports realisations/s including the reduce in "mc".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured float4 copy (79 % of spec)
INPLACE_RW_GBS = 5100.0        # scripts/experiments/micro/seg_copy.hip: in-place read-modify-write of this access pattern
SWEEP_BYTES = 64.0             # one sweep over a dual-pol complex128 field: 32 B read + 32 B written per sample
SURVEY_BYTES_PER_SAMPLE_STEP = 272.0   # FOUR sweeps + 16 B of tables (an upper-bound budget, not what runs)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--nsymb", type=int, default=1024)
    ap.add_argument("--nt", type=int, default=64)
    ap.add_argument("--pavg", type=float, default=2.0)
    ap.add_argument("--flag", default=None, help="fiber() flag (default 'g-s-'; 'gps-' with --nch > 1: BASELINE config[2])")
    ap.add_argument("--nch", type=int, default=1,
                    help="channels per frame: N > 1 makes every frame a 'sepfields' WDM field of N columns (ex10_wdm.m:9-11) that share "
                         "the step length (fiber.m:694-698); BASELINE config[2] is --nch 16 --spans 10 --nf 5 --frames 32")
    ap.add_argument("--chspacing", type=float, default=0.4, help="channel spacing [nm] of the WDM comb (ex10_wdm.m: 0.4)")
    ap.add_argument("--variants", type=int, default=16, help="distinct Tx sequences cycled over the frames of a batch")
    ap.add_argument("--power-ladder", action="store_true",
                    help="frame f is launched at point f %% 64 of BASELINE config[4]'s ladder (-4...+8 dBm in equal dB steps): "
                         "frames of one batch then need 5...90+ SSFM steps each (fiber.m:518,534)")
    ap.add_argument("--noise", type=float, default=0.05, help="receiver noise sigma per quadrature (full scale 1)")
    ap.add_argument("--frontend", default="pick", choices=["pick", "cohmix"],
                    help="pick: 2-sps sampling supplied by the harness (SURVEY 8d C1); cohmix: receiver_cohmix + ADC + decimate on the device")
    ap.add_argument("--mc", action="store_true",
                    help="timed region in BASELINE config[3] style: fiber('gps-') with a fresh random-birefringence draw per "
                         "frame and per step; the Monte-Carlo leg proper (ShardedBer) runs after the timed region in every mode")
    ap.add_argument("--mc-rounds", type=int, default=2, help="rounds of the Monte-Carlo leg (0: skip it; 2 x 512 = config[3]'s 1024 realisations per GPU)")
    ap.add_argument("--mc-depth", type=int, default=1, help="Monte-Carlo rounds enqueued ahead of the one being reduced (their receivers run beside each other)")
    ap.add_argument("--mc-frames", type=int, default=512,
                    help="realisations per GPU per round.  The CMA of a noise-loaded realisation runs its 299 passes whatever the batch (~57 ms "
                         "of a serial recurrence) and its waves keep column workgroups off their CUs meanwhile, so few LARGE rounds beat many "
                         "small ones: 128 x 8 rounds, 8 in flight: 3800/s; 512 x 2, 2 in flight: 5000/s (profiles/r03_mc.txt); the statistics "
                         "are identical (rounds are replayed in realisation order)")
    ap.add_argument("--mc-total", type=int, default=1024,
                    help="second Monte-Carlo leg, STRONG scaling: this many realisations IN TOTAL (BASELINE config[3]: 1024 over 8 GPUs = "
                         "128 per GPU), one round of mc_total / N per GPU, one all-reduce (0: skip)")
    ap.add_argument("--mc-nf", type=float, default=31.0, help="noise figure [dB] of the amplifier in the Monte-Carlo leg")
    ap.add_argument("--spans", type=int, default=1, help="spans per step, with in-line amplifiers between them (config[4]: 40)")
    ap.add_argument("--nf", type=float, default=None, help="noise figure [dB] of the in-line amplifiers (default: noiseless)")
    ap.add_argument("--no-overlap", action="store_true", help="run the receiver on the fibre stream (no stream overlap)")
    ap.add_argument("--share-device", default="auto", choices=["auto", "yes", "no"],
                    help="fibre plan on the barrier-free three-sweep step (plx_ssfm_create_ex, PLX_SSFM_SHARE_DEVICE) so that a frame "
                         "that is the whole grid of the fused sweep (2^20 samples, 16 channels) can propagate BESIDE the previous "
                         "batch's receiver; auto: where such a plan's receiver takes more than 0.18 of its fibre's time (the three-sweep "
                         "step costs the fibre ~15 %%: profiles/r04_three_sweep_rows_ab.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rx-thread", action="store_true", help="enqueue the receiver from the fibre's host thread (A/B)")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame latency measurement after the timed region")
    ap.add_argument("--no-cohmix-line", action="store_true", help="skip the short side measurement with the reference's own front end")
    ap.add_argument("--no-gateway", action="store_true", help="skip the per-call timing of the MEX-shaped gateway tier")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the batch the one-core CPU baseline processes (~0.5 s each)")
    ap.add_argument("--ladder-world", type=int, default=0,
                    help="with --power-ladder: take the ladder points of rank 0 of THIS many GPUs (frame f at point (rank + W f) %% 64) "
                         "whatever the real world size; 0 = the real world size.  8 gives a one-GPU run the share one GPU of BASELINE "
                         "config[4]'s eight has: every 8th point of the ladder, -4 ... +6.7 dBm")
    ap.add_argument("--configs", default="auto", choices=["auto", "yes", "no"],
                    help="after the headline (BASELINE config[1]) also run bounded legs of config[2] (--nch 16 --spans 10 --nf 5 "
                         "--frames 32) and config[4] (--nsymb 16384 --frames 8 --spans 40 --power-ladder --ladder-world 8) AS STATED, each "
                         "with its own roofline and CPU baseline, into \"configs\" of the same JSON line; auto: when the headline is the "
                         "default workload on one GPU")
    return ap.parse_args()


# ------------------------------------------------------------------------------------- launcher ---
def launch_ranks(a):
    """`bench.py --gpus N` outside a launcher: start N ranks as fresh child processes of THIS interpreter, which has not
    touched (and never touches) a GPU.  Only rank 0's JSON line reaches stdout: the ranks' own stdout (the gloo backend of
    the rehearsal mode chats there) goes to a temporary file / stderr."""
    import torch       # device_count() does not initialise the GPU on this image
    rehearsal = os.environ.get("PLX_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if a.gpus > ndev and not rehearsal:
        sys.stderr.write("bench.py: --gpus %d but %d device(s) visible (PLX_BENCH_REHEARSAL=1 rehearses the N-rank path on "
                         "one GPU with gloo)\n" % (a.gpus, ndev))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:
                    rc = r
            time.sleep(0.05)
    finally:
        for p in procs:              # a rank failed (or we were interrupted): stop exactly the children we started
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        out0.seek(0)
        for line in out0:
            if line.startswith("{"):
                sys.stdout.write(line)
            elif line.strip():
                sys.stderr.write(line)
        sys.stdout.flush()
        out0.close()
    return rc


class HipEvents:
    """HIP events on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.live = []

    def create(self):
        e = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(e)) == 0
        self.live.append(e)
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value

    def destroy_all(self):
        for e in self.live:
            self.hip.hipEventDestroy(e)
        self.live = []


class PowerWatch:
    """Core clock and socket power while the timed region runs (rocm-smi from a host thread, a sample every ~0.4 s): the SSFM
    kernels run the part at its power cap with the core clock throttled below 2.4 GHz (profiles/r05_notes.md, section 3b), which
    is part of what "peak" means for this line.  Best effort: any failure leaves the field null."""

    def __init__(self):
        import re
        import shutil
        import threading
        self.rows, self.cap = [], None
        self._stop = threading.Event()
        self._re = re.compile(r"sclk clock level: \w+: \((\d+)Mhz\).*?Current Socket Graphics Package Power \(W\): ([\d.]+)", re.S)
        self._recap = re.compile(r"Max Graphics Package Power \(W\): ([\d.]+)")
        self._smi = shutil.which("rocm-smi")
        # (not under a profiler: its preloaded library would travel into the child processes)
        if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            self._smi = None
        self._t = threading.Thread(target=self._run, daemon=True) if self._smi else None

    def _run(self):
        while not self._stop.is_set():
            try:
                out = subprocess.run([self._smi, "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
                m = self._re.search(out)
                if m:
                    self.rows.append((time.perf_counter(), int(m.group(1)), float(m.group(2))))
                c = self._recap.search(out)
                if c:
                    self.cap = float(c.group(1))
            except Exception:
                return
            self._stop.wait(0.25)

    def start(self):
        if self._t:
            self._t.start()
        return self

    def stop(self, t0, t1):
        if not self._t:
            return None
        self._stop.set()
        self._t.join(timeout=6)
        rows = [r for r in self.rows if t0 + 0.3 <= r[0] <= t1]            # (inside the timed region, past the ramp)
        if not rows:
            return None
        return {"sclk_mhz": sum(r[1] for r in rows) / len(rows), "socket_power_w": sum(r[2] for r in rows) / len(rows), "power_cap_w": self.cap,
                "samples": len(rows), "how": "rocm-smi --showclocks --showpower from a host thread during the timed region"}


def host_cores():
    """CPU cores this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box
    hands each GPU a share of the host, e.g. 16 of 256 logical CPUs), PLX_BENCH_CORES overrides."""
    if os.environ.get("PLX_BENCH_CORES"):
        return max(1, int(os.environ["PLX_BENCH_CORES"]))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                   # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = min(cores, max(1, int(quota)))
    return max(1, min(cores, 64))


def cpu_params(cfg, hp, noise):
    gam, betat, db1 = hp._keep
    p = dict(tx_x=hp.tx_host[0], tx_y=hp.tx_host[1], betat=betat, db1=db1, dzmax=min(cfg.dzmax, cfg.length), dphimax=cfg.dphimax,
             gam=gam, alphalin=hp.alphalin, length=cfg.length, fls=list(hp.fls), nt=cfg.nt, rx_scale=hp.rx_scale, noise=noise,
             symbolrate=cfg.symbolrate, lam=cfg.lam, disp=cfg.disp, slope=cfg.slope, fft_length=cfg.fft_length, cde_L=cfg.cde_L,
             power_mw=hp.power_mw, cma_mu=cfg.cma_mu, cma_taps=cfg.cma_taps, freqavg=cfg.freqavg, phasavg=cfg.phasavg,
             poworder=cfg.poworder, adcbits=cfg.adcbits, front=None, nspans=cfg.nspans, span_sigma=None)
    if cfg.nch > 1:
        p["tx_x"], p["tx_y"] = hp.tx_columns()
        p["nplates"] = hp.nplates
    if cfg.span_nf_db is not None and cfg.nspans > 1:
        from polmux_amd.ampliflat import ase_sigma
        hp.bind_gstate()
        sg = ase_sigma(cfg.span_nf_db, math.exp(hp.alphalin * cfg.length), cfg.nch)
        p["span_sigma"] = float(sg[0]) if cfg.nch == 1 else np.ascontiguousarray(sg, dtype=float)
    if hp.front is not None:
        p["front"] = hp.front_tables
        p["front_shifts"] = hp.front_shifts
    return p


def cpu_baseline(cfg, hp, nframes, noise, frame_steps=None, scales=None):
    """The reference algorithm restated on the CPU (oracle/, kind 'port') on a bounded sample of the SAME workload
    (frames of the batch, same receiver noise level): one core, then one frame stream per host core.
    frame_steps: SSFM steps per frame of the timed batch, all spans (the oracle takes the same steps: parity tests)."""
    from oracle import cpu_chain
    p = cpu_params(cfg, hp, noise)
    if cfg.nch == 1 and cfg.nspans * (cfg.nfft / 65536.0) * 0.4 * nframes > 60.0:
        # long-haul frames (BASELINE config[4]): ONE span of the batch's lowest launch power + one receiver pass are timed; the
        # batch's CPU time is taken as (seconds per step of that span) x (the steps the timed GPU batch really made, all frames
        # and spans) + (receiver seconds) x frames -- a step costs the same at every launch power
        sc = float(np.min(scales)) if scales is not None else 1.0
        tf, k, tr = cpu_chain.run_span_sample(p, sc, 999)
        F = len(scales) if scales is not None else 1
        steps_batch = float(frame_steps) if frame_steps else float(k * cfg.nspans * F)
        est = tf / k * steps_batch + tr * F
        return (F * cfg.nfft / est / 1e9, tf + tr, k, "one 2^%d-sample frame at the batch's lowest launch power through ONE of its %d spans "
                "(%d steps, %.1f s) + one receiver pass (%.1f s); value = samples of the batch / (seconds per step x the %.0f steps the "
                "timed batch made over all frames and spans + receiver seconds x %d frames)" % (int(np.log2(cfg.nfft)), cfg.nspans, k, tf, tr, steps_batch, F)), None
    if cfg.nch > 1:
        # a WDM frame of BASELINE config[2] is ~16 C1 frames x 10 spans of CPU work: the bounded sample is ONE frame through
        # the first `k` spans (with amplifiers) + all of its receivers; the spans being alike, the frame's time is taken
        # as t_fibre * nspans / k + t_rx
        k = max(1, min(cfg.nspans, 1 if cfg.nfft * cfg.nch >= (1 << 20) else 2))
        tf, tr, nc = cpu_chain.run_wdm_frame(p, k, 999)
        est = tf * cfg.nspans / k + tr
        return (cfg.nch * cfg.nfft / est / 1e9, tf + tr, nc, "one %d-channel frame: %d of its %d spans (%.1f s) + its %d receivers (%.1f s); "
                "value = samples of the frame / (fibre time x %d / %d + receiver time)" % (cfg.nch, k, cfg.nspans, tf, cfg.nch, tr, cfg.nspans, k)), None
    dt, nc = cpu_chain.run_frames(p, nframes, 999)
    one = (nframes * cfg.nfft / dt / 1e9, dt, nc)
    cores = host_cores()
    per = max(1, nframes // 4)
    res = cpu_chain.run_parallel(p, per, cores)
    allc = None if res is None else (cores * per * cfg.nfft / res[1] / 1e9, res[1], res[0], cores, per)
    return one, allc


def gateway_bench(cfg, hp):
    """Per-call cost of the DROP-IN tier (include/polmux_hip.h, tier A): what the unchanged MATLAB drivers would pay per MEX
    call -- host arrays in, host arrays out, PCIe both ways, the library's cached plan and buffers (plx_gateway.h) -- each
    beside oracle/ doing the same call on one host core.  Outside the timed region; never the reported value."""
    from oracle import plxo
    from polmux_amd import _abi
    lib = _abi.get()
    out = {}

    def stats():
        v = np.zeros(8, np.int64)
        lib.call("plx_gateway_stats", v.ctypes.data)
        return v

    def timed(fn, n):
        t0 = time.perf_counter()
        fn()
        first = (time.perf_counter() - t0) * 1e3
        s0 = stats()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        per = (time.perf_counter() - t0) * 1e3 / n
        s1 = stats()
        return first, per, int(s1[1] - s0[1] + s1[2] - s0[2] + s1[3] - s0[3])    # allocations + plan builds of the repeats

    lib.call("plx_release_all")
    # -- cmaadaptivefilter: ONE pass over 1024 dual-pol samples, 7 taps (the driver makes up to 299 per frame, DspPdmCohQpsk.m:176-191)
    r = np.random.default_rng(5)
    L, taps = 1024, 7
    xx = (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2))) / math.sqrt(2)
    xr, xi = np.asfortranarray(xx.real), np.asfortranarray(xx.imag)
    h0 = np.zeros((taps, 2)); h0[taps // 2, 0] = 1.0
    g0 = np.zeros((taps, 2)); g0[taps // 2, 1] = 1.0
    R = np.array([1.0, 1.0])
    yr, yi = np.zeros((L - taps + 1, 2), order="F"), np.zeros((L - taps + 1, 2), order="F")

    def cma():
        h1r, h1i, h2r, h2i = np.asfortranarray(h0), np.zeros((taps, 2), order="F"), np.asfortranarray(g0), np.zeros((taps, 2), order="F")
        lib.call("plx_cmaadaptivefilter", xr.ctypes.data, xi.ctypes.data, L, h1r.ctypes.data, h1i.ctypes.data, h2r.ctypes.data,
                 h2i.ctypes.data, float(taps), 1e-3, R.ctypes.data, 1.0, yr.ctypes.data, yi.ctypes.data)
    first, per, allocs = timed(cma, 100)
    t0 = time.perf_counter()
    for _ in range(20):
        plxo.cmaadaptivefilter(xx, h0.astype(complex), g0.astype(complex), taps, 1e-3, [1.0, 1.0], 1)
    out["plx_cmaadaptivefilter"] = {"what": "one pass, 1024 x 2 samples, 7 taps", "first_call_ms": first, "ms_per_call": per,
                                    "allocations_in_repeats": allocs, "oracle_ms_per_call": (time.perf_counter() - t0) * 1e3 / 20}
    # -- fastexp on 2^16 phases (fastexp.c:37-47)
    x = r.standard_normal(65536) * 100.0
    er, ei = np.empty_like(x), np.empty_like(x)
    first, per, allocs = timed(lambda: lib.call("plx_fastexp", x.ctypes.data, er.ctypes.data, ei.ctypes.data, x.size), 100)
    t0 = time.perf_counter()
    for _ in range(20):
        plxo.fastexp(x)
    out["plx_fastexp"] = {"what": "2^16 phases", "first_call_ms": first, "ms_per_call": per, "allocations_in_repeats": allocs,
                          "oracle_ms_per_call": (time.perf_counter() - t0) * 1e3 / 20}
    # -- matrix_ssfm: one frame of this run through one span (fiber.m:372-389)
    gam, betat, db1 = hp._keep
    d = _abi.SsfmDesc()
    d.nfft, d.nfc, d.dual_pol, d.max_frames = cfg.nfft, cfg.nch, 1, 1
    for i in range(4):
        d.fls[i] = hp.fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length = min(cfg.dzmax, cfg.length), cfg.dphimax, hp.alphalin, cfg.length
    d.nplates, d.manakov = 1, int(str(cfg.manakov).lower() == "yes")
    d.gam, d.betat, d.db1 = gam.ctypes.data, betat.ctypes.data, db1.ctypes.data
    z = np.zeros(1)
    fd, nc = C.c_double(), C.c_int32()
    tx, ty = hp.tx_columns() if cfg.nch > 1 else hp.tx_host

    def span():
        uxr, uxi = np.asfortranarray(tx.real), np.asfortranarray(tx.imag)
        uyr, uyi = np.asfortranarray(ty.real), np.asfortranarray(ty.imag)
        lib.call("plx_matrix_ssfm", uxr.ctypes.data, uxi.ctypes.data, uyr.ctypes.data, uyi.ctypes.data, C.byref(d), z.ctypes.data,
                 z.ctypes.data, z.ctypes.data, C.byref(fd), C.byref(nc))
    if not hp.pmd:
        first, per, allocs = timed(span, 10)
        t0 = time.perf_counter()
        plxo.matrix_ssfm(tx, ty, betat, db1, d.dzmaxt, d.dphimaxt, gam, hp.alphalin, cfg.length, 1, d.manakov, hp.fls, [0.0], [0.0], [0.0])
        out["plx_matrix_ssfm"] = {"what": "one 2^%d-sample frame, one span, %d steps" % (int(np.log2(cfg.nfft)), nc.value),
                                  "first_call_ms": first, "ms_per_call": per, "allocations_in_repeats": allocs,
                                  "oracle_ms_per_call": (time.perf_counter() - t0) * 1e3}
    # -- the whole pol-demux DRIVER of one frame (DspPdmCohQpsk.m:142-192): (a) ONE call of plx_cmapolardemux (pass loop on the
    #    device), (b) what the unchanged driver pays: one plx_cmaadaptivefilter MEX call per pass with the 5e-5 test on the host,
    #    (c) the oracle's driver on one host core.  Input: frame 0 of the last batch after CDE, 1 sps, /4 sqrt(P) (:12-23)
    try:
        eq = hp.eq[0].cpu().numpy()                                   # [2, 2 nsymb]
        xs = np.ascontiguousarray(eq[:, ::2].T) / (4.0 * math.sqrt(hp.power_mw))
        Ld, tp, mu = xs.shape[0], cfg.cma_taps, cfg.cma_mu
        xr, xi = np.asfortranarray(xs.real), np.asfortranarray(xs.imag)
        Mi = np.array([1.0, 0, 0, 0, 0, 0, 1.0, 0])                     # phizero = 0: M = I (:157-158)
        yr, yi = np.zeros((Ld, 2), order="F"), np.zeros((Ld, 2), order="F")
        hh = [np.zeros((tp, 2), order="F") for _ in range(4)]
        npass = C.c_int32()

        def one_call():
            lib.call("plx_cmapolardemux", xr.ctypes.data, xi.ctypes.data, Ld, tp, mu, R.ctypes.data, Mi.ctypes.data, yr.ctypes.data,
                     yi.ctypes.data, hh[0].ctypes.data, hh[1].ctypes.data, hh[2].ctypes.data, hh[3].ctypes.data, C.byref(npass))
        first, per, allocs = timed(one_call, 5)
        half = tp // 2
        ext = np.concatenate([xs[Ld - half:], xs, xs[:half]]) if half else xs      # :161-165
        er, ei = np.asfortranarray(ext.real), np.asfortranarray(ext.imag)
        budget = 50 * int(math.ceil(1.0 / (Ld * mu)))
        y2r, y2i = np.zeros((Ld, 2), order="F"), np.zeros((Ld, 2), order="F")

        def per_pass_driver():
            h1r = np.zeros((tp, 2), order="F"); h1r[half, 0] = 1.0
            h2r = np.zeros((tp, 2), order="F"); h2r[half, 1] = 1.0
            h1i, h2i = np.zeros((tp, 2), order="F"), np.zeros((tp, 2), order="F")
            c = 1
            while c < budget:                                                      # :176-191
                o = (h1r.copy(), h1i.copy(), h2r.copy(), h2i.copy())
                lib.call("plx_cmaadaptivefilter", er.ctypes.data, ei.ctypes.data, ext.shape[0], h1r.ctypes.data, h1i.ctypes.data,
                         h2r.ctypes.data, h2i.ctypes.data, float(tp), mu, R.ctypes.data, 1.0, y2r.ctypes.data, y2i.ctypes.data)
                d = max(np.abs((o[0] - h1r) + 1j * (o[1] - h1i)).max(), np.abs((o[2] - h2r) + 1j * (o[3] - h2i)).max())
                c += 1
                if d < 5e-5:
                    break
            return c - 1
        t0 = time.perf_counter()
        n_loop = per_pass_driver()
        loop_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        _, _, _, n_or = plxo.cmapolardemux(xs, np.eye(2), tp, mu, [1.0, 1.0])
        or_ms = (time.perf_counter() - t0) * 1e3
        out["plx_cmapolardemux"] = {"what": "the pol-demux driver of ONE frame (L = %d, %d taps, mu = 1/%.0f): %d passes" % (Ld, tp, 1 / mu, npass.value),
                                    "first_call_ms": first, "ms_per_call": per, "allocations_in_repeats": allocs, "passes": int(npass.value),
                                    "per_pass_mex_loop_ms": loop_ms, "per_pass_mex_loop_passes": int(n_loop),
                                    "oracle_ms_per_call": or_ms, "oracle_passes": int(n_or)}
    except Exception as exc:           # (a side measurement: never takes the line down)
        out["plx_cmapolardemux"] = {"error": repr(exc)}
    v = stats()
    out["library_state"] = {"device_bytes": int(v[5]), "pinned_bytes": int(v[6]), "plans_built": int(v[3]), "plans_found": int(v[4])}
    out["note"] = ("host arrays in and out (PCIe both ways) with the library's cached plans and buffers; first_call_ms includes "
                   "the plan build and the buffers' first growth; oracle = the CPU restatement doing the same call on one core")
    lib.call("plx_release_all")
    return out


def ladder_scales(nframes, pavg_mw, rank=0, world=1):
    """BASELINE config[4]: 64 launch powers -4...+8 dBm in equal dB steps; frame f of rank r sits at point
    (r + world * f) % 64, so that 8 GPUs x 8 frames cover the ladder once, each GPU with a similar mix of powers."""
    dbm = -4.0 + 12.0 * ((rank + world * np.arange(nframes)) % 64) / 63.0
    return 10 ** (dbm / 10) / pavg_mw


def workload_label(a, n):
    """config.workload from the ACTUAL arguments: which BASELINE configuration (if any) this line is."""
    what = ("28 Gbaud PDM-QPSK, 2^%d-sample dual-pol frame, %dx80 km SSMF span fiber('%s'), CDE_OFDE 256/128, CMA 7 taps + "
            "V&V carrier recovery; front end: %s" % (int(np.log2(n)), a.spans, a.flag, a.frontend))
    if a.nch > 1:
        what = "%d 'sepfields' channels %.2f nm apart per frame (shared dz, per-channel receivers), " % (a.nch, a.chspacing) + what
        tag = "ex03/ex10-style WDM PDM-QPSK (BASELINE config[2]: 16 channels, 10x80 km nonlinear SSFM%s)" % (
            "" if (a.nch == 16 and a.spans == 10) else "; this line: %d channels, %d span(s)" % (a.nch, a.spans))
        return tag + ": " + what
    if a.mc:
        tag = "ex24_pmd-style random-PMD batch (BASELINE config[3] realisations in the timed region: fresh waveplates per frame and step)"
    elif (n == 1 << 20 or a.spans > 1 or a.power_ladder) and a.flag == "g-s-":
        tag = "long-haul launch-power-sweep style (BASELINE config[4]: 2^20-sample frame x 40 spans x 64-point ladder%s)" % (
            "" if (n == 1 << 20 and a.spans == 40 and a.power_ladder) else
            "; this line: 2^%d samples, %d span(s), %s" % (int(np.log2(n)), a.spans, "power ladder" if a.power_ladder else "one launch power"))
    elif n == 65536 and a.flag == "g-s-" and a.nt == 64:
        tag = "Run_my_PDM_QPSK-style (BASELINE config[1])"
    elif n == 1 << 18 and a.flag == "g-s-" and a.nt == 64:
        tag = "custom configuration (no BASELINE config; the frame size Run_my_PDM_QPSK.m:21-24 ships with, 4096 symbols x 64 samples, on config[1]'s link)"
    else:
        tag = "custom configuration (no BASELINE config)"
    return tag + ": " + what


def offline_traffic(fused, F, n, nch=1, flag="g-s-"):
    """HBM bytes per launch of the dominant kernel from the PMC counters.  Counters need their own rocprofv3 passes
    (FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md, HBM section), so this is NOT measured in this run: it is read from the
    summary of scripts/traffic_pmc.sh on this round's build, if one is committed (2^16- and 2^20-sample frames, and the
    16-channel WDM frame of BASELINE config[2])."""
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        tj = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tj):
            continue
        tr = json.load(open(tj))
        if nch > 1:
            tr = tr.get("wdm_%dch" % nch, {}) if (n == 65536 and flag == "gps-") else {}
        elif flag != "g-s-":
            return None, None
        elif n == 1 << 20:
            tr = tr.get("frames_2pow20", {})
        elif n == 1 << 18:
            tr = tr.get("frames_2pow18", {})
        elif n != 65536:
            return None, None
        if not fused:
            tr = tr.get("plain_three_sweep", {})
        per = tr.get("bytes_per_sample_by_kernel")
        if not per:
            if nch > 1:
                continue
            return None, None
        k = "k_colx16" if fused else "k_col_fwd"
        return per[k] * F * nch * n, {"file": "profiles/" + name, "how": "offline rocprofv3 --pmc passes (not this run)",
                                      "bytes_per_sample_by_kernel": per}
    return None, None


class Env:
    """What every line of a run shares: the rank's place in the job, the library binding, where collectives' tensors live."""


# the bounded legs of the other BASELINE configurations AS STATED (arguments on top of the headline's)
CONFIG_LEGS = {
    "c2": dict(nch=16, spans=10, nf=5.0, frames=32, steps=2, warmup=1, flag=None),
    "c4": dict(nsymb=16384, frames=8, spans=40, power_ladder=True, ladder_world=8, variants=1, steps=2, warmup=1, flag=None),
}


FP64_PEAK_TFLOPS = 78.6       # 256 CUs x 4 SIMDs x 16 FP64 lanes x 2 flop per FMA x 2.4 GHz (vector FP64, no MFMA on this path)


def fused_mismatch(f64, names):
    """the FP64 summary was taken on the fused step: it says nothing about the three-sweep kernels"""
    return names[0] != "k_colx16" and "k_colx16" in f64["flop_per_sample_per_launch"] and "k_col_fwd" not in f64["flop_per_sample_per_launch"]


def offline_fp64(n, nch, flag, mc):
    """FP64 flop per dual-pol sample per launch of each step kernel, from the SQ instruction counters (scripts/fp64_pmc.sh: a
    rocprofv3 --pmc pass of its own, so NOT measured in this run; committed summary of this round's build).  None when the
    summary has no entry for this workload shape."""
    for name in ("r05_fp64.json",):
        fj = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(fj):
            continue
        tab = json.load(open(fj))
        pmd = flag == "gps-"
        if nch == 16 and n == 65536 and pmd:
            tag = "c2_frame"
        elif nch != 1:
            return None
        elif n == 65536:
            tag = "c1_mc" if (pmd and mc) else (None if pmd else "c1")
        elif n == 1 << 20:
            tag = "big_pmd" if pmd else "big"
        elif n == 1 << 18:
            tag = "mid_pmd" if pmd else "mid"
        else:
            tag = None
        w = tab.get("workloads", {}).get(tag) if tag else None
        if not w:
            return None
        return {"file": "profiles/" + name, "workload": tag, "what": w["what"], "how": "offline rocprofv3 --pmc pass (not this run)",
                "peak_TFLOPs": tab.get("peak_TFLOPs", FP64_PEAK_TFLOPS), "peak_source": tab.get("peak_source"),
                "flop_per_sample_per_launch": {k: v["flop_per_sample_per_launch"] for k, v in w["kernels"].items()}}
    return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    import torch.distributed as dist
    E = Env()
    E.rank = int(os.environ.get("RANK", "0"))
    E.world = int(os.environ.get("WORLD_SIZE", "1"))
    E.local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box (dev only): every rank on cuda:0, collectives over gloo
    E.rehearsal = os.environ.get("PLX_BENCH_REHEARSAL") == "1"
    if E.rehearsal:
        E.local = 0
        os.environ["PLX_SSFM_NO_FUSE"] = "1"     # several ranks share one GPU here: the fused sweep needs the chip to itself
    torch.cuda.set_device(E.local)
    if E.world > 1:
        if E.rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", E.local))
    from polmux_amd import _abi
    E.lib = _abi.get()
    E.lib.call("plx_set_device", E.local)
    E.cdev = "cpu" if (E.rehearsal or E.world == 1) else "cuda"        # where the collectives' tensors live
    headline_is_default = (a.nch == 1 and a.nsymb == 1024 and a.nt == 64 and a.spans == 1 and a.flag is None and not a.mc and
                           not a.power_ladder and a.frontend == "pick")
    out = run_line(a, E)
    if E.world == 1 and (a.configs == "yes" or (a.configs == "auto" and headline_is_default)):
        # BASELINE configs 2 and 4 as stated, bounded: a short timed region each (no Monte-Carlo, gateway or latency side legs)
        out["configs"] = {}
        for name, over in CONFIG_LEGS.items():
            b = argparse.Namespace(**vars(a))
            for k, v in over.items():
                setattr(b, k, v)
            b.mc_rounds, b.no_single_frame, b.no_gateway, b.no_cohmix_line, b.share_device = 0, True, True, True, "auto"
            t0 = time.perf_counter()
            leg = run_line(b, E)
            for k in ("mc", "gateway", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "cpu_baseline_all_cores"):
                leg.pop(k, None)
            leg["leg_seconds"] = time.perf_counter() - t0
            out["configs"][name] = leg
    if E.rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if E.world > 1:
        dist.destroy_process_group()


def run_line(a, E):
    """One bench line: the timed region for the workload the arguments describe, then the side legs they ask for.
    Returns the line as a dict on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from polmux_amd import mc, pipeline
    rank, world, local, rehearsal, cdev, lib = E.rank, E.world, E.local, E.rehearsal, E.cdev, E.lib
    a = argparse.Namespace(**vars(a))          # (defaults are filled in below: the caller's copy stays as parsed)

    if a.mc or (a.flag is None and a.nch > 1):
        a.flag = "gps-"
    if a.flag is None:
        a.flag = "g-s-"
    nch = a.nch
    cfg = pipeline.HotPathConfig(nsymb=a.nsymb, nt=a.nt, pavg_mw=a.pavg, flag=a.flag, frontend=a.frontend, nspans=a.spans,
                                 span_nf_db=a.nf, variants=a.variants, nch=nch, chspacing=a.chspacing)
    F = a.frames
    cfg.share_device = a.share_device == "yes"
    hp = pipeline.HotPath(cfg, max_frames=F)
    n = cfg.nfft
    lw = a.ladder_world if a.ladder_world > 0 else world
    scales = ladder_scales(F, a.pavg, rank if a.ladder_world <= 0 else 0, lw) if a.power_ladder else None
    share_why = "asked for" if cfg.share_device else None
    if a.share_device == "auto" and not a.no_overlap and not hp.overlap_ok():
        # A frame of this plan is the whole grid of the fused sweep: its receiver cannot run beside the next fibre.  One
        # calibration batch (not timed, not counted): is the receiver long enough to be worth the three-sweep step's ~15 %?
        cx, cy = hp.make_batch(F, scales)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hp.fibre(cx, cy)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        hp.receive(cx, cy, noise_sigma=a.noise, noise_seed=1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del cx, cy
        if (t2 - t1) > 0.18 * (t1 - t0):
            hp.close()
            cfg.share_device = True
            hp = pipeline.HotPath(cfg, max_frames=F)
            share_why = "auto: receiver %.0f ms against fibre %.0f ms on the fused step, serial" % ((t2 - t1) * 1e3, (t1 - t0) * 1e3)
    hp.profile(True)          # a HIP event between consecutive launches of the step loop: per-kernel durations, live
    # inputs for every step are staged in HBM before the timed region (fibre works in place)
    total = a.steps + a.warmup
    # (bounded by free HBM: with more steps than buffers a buffer is refilled from a pristine copy by one
    # device-to-device copy on the fibre stream -- inside the timed region, reported as "restaged_batches")
    batch_bytes = 2 * F * nch * n * 16
    free_b, _tot = torch.cuda.mem_get_info()
    nbuf = max(2, min(total, int(0.6 * free_b // batch_bytes)))
    if os.environ.get("PLX_BENCH_NBUF"):      # dev: force the restaging path
        nbuf = max(2, min(total, int(os.environ["PLX_BENCH_NBUF"])))
    batches = [hp.make_batch(F, scales) for _ in range(nbuf)]
    pristine = hp.make_batch(F, scales) if nbuf < total else None
    restaged = 0
    buf_free = [None] * nbuf      # event: the receiver that last read this buffer has finished

    def get_batch(i):
        nonlocal restaged
        b = batches[i % nbuf]
        if i >= nbuf:
            if buf_free[i % nbuf] is not None:
                torch.cuda.current_stream().wait_event(buf_free[i % nbuf])
            b[0].copy_(pristine[0]); b[1].copy_(pristine[1])
            if i >= a.warmup:
                restaged += 1
        return b
    torch.cuda.synchronize()
    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # receiver on its own HIP stream: Rx of batch i overlaps the fibre of batch i+1 (both inside the timed region)
    rx_stream = None if (a.no_overlap or not hp.overlap_ok()) else torch.cuda.Stream()
    err_total = torch.zeros(2, dtype=torch.int64, device="cuda")
    errs, resolved = [], []
    fib_ms, rx_ms, row_launches, sample_steps = [], [], 0, 0
    k_ms, k_n = np.zeros(4), np.zeros(4, np.int64)
    util = np.zeros(3, np.int64)
    for i in range(a.warmup):
        ux, uy = get_batch(i)
        hp.run(ux, uy, noise_sigma=a.noise, noise_seed=1000 * rank + i)
    sync_all()
    hp.kernel_times()                  # (drop the warm-up's kernel intervals)
    pwatch = PowerWatch().start() if rank == 0 else None
    t0 = time.perf_counter()
    # The receiver of a batch is ENQUEUED by a second host thread (its kernels go to the receiver's stream either way): the
    # fibre call blocks its host thread until the step loop has ended, and the ~3 ms it takes to enqueue the receiver's dozen
    # launches would otherwise sit between two fibres with the fibre's stream idle.  (Inline when the receiver shares the
    # fibre's stream or buffers are restaged.)
    import queue
    import threading
    rxq, rx_fail = None, []
    if rx_stream is not None and nbuf >= total and not a.no_rx_thread:
        rxq = queue.Queue()

        def rx_worker():
            try:
                torch.cuda.set_device(local)
                lib.call("plx_set_device", local)
                while True:
                    item = rxq.get()
                    if item is None:
                        return
                    i_, ux_, uy_, ready_, e2_ = item
                    rx_stream.wait_event(ready_)
                    with torch.cuda.stream(rx_stream):
                        err_ = hp.receive(ux_, uy_, noise_sigma=a.noise, noise_seed=1000 * rank + i_)
                        ev.record(e2_, rx_stream.cuda_stream)
                        errs.append(err_.sum(0))
                        if a.mc:
                            resolved.append(hp.errors_resolved(F * nch).sum())
            except BaseException as exc:       # (reported by the main thread)
                rx_fail.append(exc)
        rx_thread = threading.Thread(target=rx_worker, daemon=True)
        rx_thread.start()
    for i in range(a.warmup, total):
        ux, uy = get_batch(i)
        e0, e1, e2 = ev.create(), ev.create(), ev.create()
        ev.record(e0, stream)
        if a.mc:   # realisation index = (rank, step, frame): fresh waveplates, keyed so that sharding does not change them
            hp.set_random_pmd(range((rank * total + i) * F, (rank * total + i + 1) * F))
        hp.fibre(ux, uy)
        ev.record(e1, stream)
        if rxq is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            rxq.put((i, ux, uy, ready, e2))
        else:
            rs = rx_stream.cuda_stream if rx_stream is not None else stream
            err = hp.receive(ux, uy, noise_sigma=a.noise, noise_seed=1000 * rank + i, side_stream=rx_stream)
            # (receive() makes rx_stream wait for the fibre; the Rx interval is measured on the Rx stream)
            ev.record(e2, rs)
            if rx_stream is not None:
                with torch.cuda.stream(rx_stream):
                    errs.append(err.sum(0))
                    if a.mc:   # a blind receiver behind random birefringence: resolve pol swap + pi/2 ambiguity (ex20:160-173)
                        resolved.append(hp.errors_resolved(F * nch).sum())
            else:
                errs.append(err.sum(0))
                if a.mc:
                    resolved.append(hp.errors_resolved(F * nch).sum())
            if rx_stream is not None and nbuf < total:
                buf_free[i % nbuf] = torch.cuda.Event()
                buf_free[i % nbuf].record(rx_stream)
        fib_ms.append((e0, e1)); rx_ms.append((e1, e2))
        rl, ss = hp.ssfm_stats()
        row_launches += rl; sample_steps += ss
        util += np.array(hp.utilisation(), np.int64)
    k_ms, k_n = hp.kernel_times()      # (all timed steps; the warm-up's intervals were dropped before the loop)
    if rxq is not None:
        rxq.put(None)
        rx_thread.join()
        if rx_fail:
            raise rx_fail[0]
    sync_all()
    dt = time.perf_counter() - t0
    power_state = pwatch.stop(t0, t0 + dt) if pwatch is not None else None
    for e in errs:
        err_total += e
    res_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    for r_ in resolved:
        res_total += r_
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        err_total, res_total = err_total.to(cdev), res_total.to(cdev)
        dist.all_reduce(err_total, op=dist.ReduceOp.SUM)       # the one exchange step (RCCL over xGMI)
        dist.all_reduce(res_total, op=dist.ReduceOp.SUM)
    fib = sum(ev.elapsed_ms(x, y) for x, y in fib_ms)
    rxm = sum(ev.elapsed_ms(x, y) for x, y in rx_ms)
    ev.destroy_all()
    ncyc = hp.last_ncycle(F)

    # ---- Monte-Carlo leg (BASELINE config[3]; M2 of SURVEY 8d): outside the timed region above, timed on its own ----
    mc_out = None
    if a.mc_rounds > 0:
        mcfg = pipeline.HotPathConfig(nsymb=a.nsymb, nt=a.nt, pavg_mw=a.pavg, flag="gps-", frontend="cohmix", rx_amp=True,
                                      span_nf_db=a.mc_nf)
        camp = pipeline.McCampaignPool(mcfg, frames_per_call=a.mc_frames, n=a.mc_depth + 1)
        x = dict(stop=(0.01, 95.0), nmin=100)                   # 1 % relative accuracy at 95 % confidence (ber_estimate.m:128-139)
        for w_ in range(a.mc_depth + 1):                        # warm-up of every instance of the pool, not counted
            camp.simulate(list(range(10 ** 6 + rank * a.mc_frames, 10 ** 6 + (rank + 1) * a.mc_frames)))
        # mc_estimate (mc_estimate.m:133-212) on a continuous per-realisation sample, the EVM of the recovered symbols,
        # gathered with the counts: 0.1 % accuracy of the mean at 95 % confidence
        # the campaign is short (two rounds: ~0.16 s), so one host hiccup shows as 10 %: it is run TWICE (the same realisations,
        # the same statistics) and the faster repetition is the figure; both are listed in "runs"
        mc_runs = []
        for rep_ in range(2):
            sb_ = mc.ShardedBer(camp.simulate, camp.bits_per_realisation, x, per_rank_per_round=a.mc_frames, device=cdev,
                                x_samples=dict(stop=(1e-3, 95.0), nmin=50))
            sync_all()
            t1 = time.perf_counter()
            res_ = sb_.run(max_realisations=a.mc_rounds * a.mc_frames * world, depth=a.mc_depth)
            sync_all()
            dt_ = time.perf_counter() - t1
            if world > 1:
                tm = torch.tensor([dt_], dtype=torch.float64, device=cdev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dt_ = float(tm.item())
            mc_runs.append(dt_)
            if rep_ == 0 or dt_ <= min(mc_runs[:-1]):
                sb, res, mdt = sb_, res_, dt_
        done = len(sb.counts)
        mc_out = {"realisations_per_s": done / mdt, "realisations": done, "rounds": sb.rounds, "seconds": mdt,
                  "runs": {"seconds": mc_runs, "reported": "the faster of two repetitions of the same campaign"},
                  "per_gpu_per_round": a.mc_frames, "rounds_in_flight": a.mc_depth + 1, "bits_per_realisation": camp.bits_per_realisation,
                  "avgber": float(res[1][0]), "stdber": float(res[3][0]), "nruns_bits": float(res[2][0]),
                  "evm_mc_estimate": None if sb.samples_result is None else {
                      "mean": float(sb.samples_result[1]["mean"][0]), "stdmean": float(sb.samples_result[1]["stdmean"][0]),
                      "var": float(sb.samples_result[1]["var"][0]), "varlim": [float(v) for v in sb.samples_result[1]["varlim"][:, 0]],
                      "nruns": float(sb.samples_result[1]["nruns"][0]), "accurate_enough": bool(not sb.samples_result[0][0])},
                  "stopped_by_rule": bool(not res[0][0]), "stop_rule": "1 % at 95 % confidence, nmin 100 (ber_estimate.m:128-139)",
                  "exchange": "one all_reduce(SUM) of int64[%d] per round (%s), then the sequential ber_estimate replay on "
                              "every rank" % (a.mc_frames * world, "gloo rehearsal" if rehearsal else ("RCCL" if world > 1 else "single rank")),
                  "realisation": "C1 frame, fiber('gps-') with 100 fresh random waveplates (fiber.m:274-276), ampliflat gain "
                                 "%.1f dB with ASE of NF %.1f dB (ampliflat.m:91-136, device Philox keyed by realisation), "
                                 "receiver_cohmix + ADC + decimate, CDE_OFDE, CMA + CPE, decisions with pol-swap / pi/2 "
                                 "resolution (ex20_coherent_polmux.m:160-173)" % (10 * math.log10(math.exp(camp.hp.alphalin * mcfg.length)), a.mc_nf)}
        # the SAME realisation on the host cores (reported baseline; N = 1 only): oracle chain with fresh waveplates, the
        # amplifier's ASE, receiver_cohmix front end and the CMA's full pass budget, one stream of realisations per core
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            from oracle import cpu_chain
            from polmux_amd.ampliflat import ase_sigma
            pm = cpu_params(mcfg, camp.hp, 0.0)
            pm["mc"] = {"nplates": camp.hp.nplates, "amp_sigma": float(ase_sigma(a.mc_nf, math.exp(camp.hp.alphalin * mcfg.length), 1)[0])}
            cores, per = host_cores(), 2
            res_cpu = cpu_chain.run_parallel(pm, per, cores)
            if res_cpu is not None:
                mc_out["cpu_baseline"] = {"realisations_per_s": cores * per / res_cpu[1], "cores": cores, "kind": "port",
                                          "sample": "%d processes x %d realisations of the same kind through oracle/ (fibre 'gps-' with %d fresh "
                                                    "waveplates, amplifier ASE, receiver_cohmix + ADC + decimate, CDE, CMA + CPE), busiest "
                                                    "process %.1f s" % (cores, per, camp.hp.nplates, res_cpu[1])}
        camp.close()
        # ---- the same campaign as BASELINE config[3] STATES it: mc_total realisations in total, sharded r mod N -- strong
        # scaling: N GPUs take mc_total / N each in ONE round (one all-reduce).  Its floor is a latency, not a rate: the fibre of
        # mc_total / N realisations + ONE receiver (the CMA of a noise-loaded realisation runs all of its 299 passes, ~55 ms
        # whatever the batch size) + the front end, so the curve flattens as N grows.
        if a.mc_total > 0 and a.mc_total % world == 0:
            per = a.mc_total // world
            camp2 = pipeline.McCampaignPool(mcfg, frames_per_call=per, n=1)
            camp2.simulate(list(range(2 * 10 ** 6 + rank * per, 2 * 10 ** 6 + (rank + 1) * per)))       # warm-up, not counted
            ss_runs = []
            for rep_ in range(2):              # (the faster of two repetitions, as above)
                sb2_ = mc.ShardedBer(camp2.simulate, camp2.bits_per_realisation, x, per_rank_per_round=per, device=cdev)
                sync_all()
                t1 = time.perf_counter()
                res2_ = sb2_.run(max_realisations=a.mc_total, depth=1)
                sync_all()
                dt_ = time.perf_counter() - t1
                if world > 1:
                    tm = torch.tensor([dt_], dtype=torch.float64, device=cdev)
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    dt_ = float(tm.item())
                ss_runs.append(dt_)
                if rep_ == 0 or dt_ <= min(ss_runs[:-1]):
                    sb2, res2, sdt = sb2_, res2_, dt_
            mc_out["strong_scaling"] = {"realisations_total": a.mc_total, "realisations": len(sb2.counts), "per_gpu": per, "rounds": sb2.rounds,
                                        "exchanges": sb2.exchanges, "seconds": sdt, "runs_seconds": ss_runs, "realisations_per_s": len(sb2.counts) / sdt,
                                        "avgber": float(res2[1][0]), "stopped_by_rule": bool(not res2[0][0]),
                                        "shape": "BASELINE config[3] as stated: %d realisations in total, realisation r on GPU r mod %d, "
                                                 "one round, one all_reduce(SUM) of int64[%d]" % (a.mc_total, world, a.mc_total)}
            camp2.close()
            # ---- what ONE of eight ranks would do in that campaign, measured here on one GPU: its 1024 / 8 = 128 realisations
            # (r = 0, 8, 16, ...) as one round of 128, and as two / four parts in flight (own plans, buffers and receiver streams:
            # the receivers -- a latency, the CMA's 299 passes -- run beside each other) with still ONE exchange at the end.
            # The 8-GPU figure DESIGN.md section 7 predicts is 1024 / (this time + the all-reduce).
            if world == 1 and a.mc_total % 8 == 0 and a.mc_total >= 64:
                share = a.mc_total // 8
                shapes = {}
                for parts in (1, 2, 4):
                    pool = pipeline.McCampaignPool(mcfg, frames_per_call=share // parts, n=parts, split=True)
                    view = pipeline.McRankShare(pool, 0, 8)
                    view.simulate(list(range(3 * 10 ** 6, 3 * 10 ** 6 + share)))                    # warm-up, not counted
                    sb3 = mc.ShardedBer(view.simulate, pool.bits_per_realisation, x, per_rank_per_round=share, device=cdev)
                    best = None
                    for rep_ in range(3):
                        sb3 = mc.ShardedBer(view.simulate, pool.bits_per_realisation, x, per_rank_per_round=share, device=cdev)
                        sync_all()
                        t1 = time.perf_counter()
                        res3 = sb3.run(max_realisations=share, depth=1)
                        sync_all()
                        dt3 = time.perf_counter() - t1
                        best = dt3 if best is None else min(best, dt3)
                    shapes["%dx%d" % (parts, share // parts)] = {"seconds": best, "realisations": len(sb3.counts), "exchanges": sb3.exchanges,
                                                                "avgber": float(res3[1][0]), "errors": int(sum(sb3.counts))}
                    pool.close()
                ks = min(shapes, key=lambda k_: shapes[k_]["seconds"])
                mc_out["strong_scaling_rank_share"] = {
                    "what": "the share of rank 0 of 8 in BASELINE config[3] as stated (realisations 0, 8, ..., %d: %d of %d), on this one GPU; "
                            "parts x size in flight, ONE exchange at the end; best of 3 repetitions" % (a.mc_total - 8, share, a.mc_total),
                    "shapes": shapes, "identical_counts": len({v_["errors"] for v_ in shapes.values()}) == 1, "best_shape": ks,
                    "predicted_8gpu_realisations_per_s": a.mc_total / shapes[ks]["seconds"],
                    "prediction_note": "1024 realisations / the measured time of one rank's share (ranks are independent until the one "
                                       "all-reduce of int64[%d], whose latency is not in this figure)" % a.mc_total}

    # SURVEY 8d's M1 read literally -- ONE frame through fibre + receiver, nothing else on the GPU (outside the timed region)
    single = None
    if rank == 0 and not a.mc and not a.no_single_frame:
        sx, sy = hp.make_batch(1)
        hp.profile(False)              # (no HIP event between the launches)
        hp.fibre(sx.clone(), sy.clone())
        torch.cuda.synchronize()
        best = None
        for rep_ in range(3):          # (best of three: the first repetition after the campaign still finds the clocks low)
            wx, wy = sx.clone(), sy.clone()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hp.fibre(wx, wy)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hp.receive(wx, wy, noise_sigma=a.noise, noise_seed=4242)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if best is None or t3 - t1 < best[2]:
                best = (t2 - t1, t3 - t2, t3 - t1)
        single = {"fibre_ms": best[0] * 1e3, "rx_ms": best[1] * 1e3, "gsample_per_s": nch * n / best[2] / 1e9, "repetitions": 3}
    # the same workload with the reference's OWN front end (receiver_cohmix + 5-bit ADC + decimate on the device) instead of the
    # harness's 2-sps pick: a short timed region of its own, reported beside the headline (never instead of it)
    cohmix_line = None
    if rank == 0 and world == 1 and a.frontend == "pick" and not a.mc and not a.no_cohmix_line and not a.power_ladder and a.spans == 1:
        ccfg = pipeline.HotPathConfig(nsymb=a.nsymb, nt=a.nt, pavg_mw=a.pavg, flag=a.flag, frontend="cohmix", variants=a.variants,
                                      nch=nch, chspacing=a.chspacing)
        chp = pipeline.HotPath(ccfg, max_frames=F)
        csteps = max(2, min(4, a.steps))
        cb = [chp.make_batch(F) for _ in range(2)]
        cpristine = chp.make_batch(F)
        crx = None if (a.no_overlap or not chp.overlap_ok()) else torch.cuda.Stream()
        free_ev = [None, None]

        def cstep(i):
            ux_, uy_ = cb[i % 2]
            if i >= 2:         # (the front end overwrites the field with the photocurrents: restage from the pristine copy)
                if free_ev[i % 2] is not None:
                    torch.cuda.current_stream().wait_event(free_ev[i % 2])
                ux_.copy_(cpristine[0]); uy_.copy_(cpristine[1])
            chp.fibre(ux_, uy_)
            chp.receive(ux_, uy_, noise_sigma=a.noise, noise_seed=7000 + i, side_stream=crx)
            if crx is not None:
                free_ev[i % 2] = torch.cuda.Event(); free_ev[i % 2].record(crx)
        cstep(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(1, 1 + csteps):
            cstep(i)
        torch.cuda.synchronize()
        cdt = time.perf_counter() - t1
        cohmix_line = {"value": csteps * F * nch * n / cdt / 1e9, "unit": "Gsample/s", "steps": csteps, "ms_per_step": cdt / csteps * 1e3,
                       "front_end": "receiver_cohmix (optical filter, hybrids + balanced photodiodes, electrical filter) + %d-bit ADC + decimate "
                                    "to 2 sps on the device (RxPdmCohQpsk.m:36-72)" % ccfg.adcbits,
                       "note": "same batch size and fibre; two field buffers, restaged from a pristine copy inside the timed region"}
        chp.close()
        del cb, cpristine
    gateway = None
    if rank == 0 and world == 1 and not a.no_gateway:
        gateway = gateway_bench(cfg, hp)
    out = None
    if rank == 0:
        samples = float(world) * a.steps * F * nch * n
        value = samples / dt / 1e9
        fused = hp.fused()
        # ---- roofline: the dominant kernel, priced with the bytes it REALLY moves per launch (one sweep over the batch:
        # 32 B read + 32 B written per dual-pol sample; tables are shared by all frames and stay in L2) over its
        # average ACTIVE launch duration, measured live with HIP events between the launches on the fibre stream ----
        row_kernel = hp.row_kernel()
        names = ["k_colx16" if fused else "k_col_fwd", row_kernel, "k_col_inv", "control"]
        kern = {}
        for k in range(3):
            if k_n[k]:
                avg_ms = k_ms[k] / k_n[k]
                # with a power ladder the launches shrink as frames finish: bytes per launch = bytes of the frames still active
                act = float(np.mean([(ncyc > s).sum() for s in range(int(ncyc.max()))])) if len(ncyc) else float(F)
                gbs = SWEEP_BYTES * act * nch * n / (avg_ms * 1e-3) / 1e9
                kern[names[k]] = {"avg_launch_us": avg_ms * 1e3, "active_launches": int(k_n[k]), "achieved_GBs": gbs,
                                  "frac_of_8TBs": gbs / HBM_PEAK_GBS}
        dom = max(kern, key=lambda k_: kern[k_]["avg_launch_us"] * kern[k_]["active_launches"]) if kern else names[0]
        # the FP64 side of the same launches: flop per launch (SQ instruction counters, offline) over the SAME live HIP-event
        # durations, against the part's vector-FP64 peak -- beside the HBM fraction, for every kernel the summary covers
        f64 = offline_fp64(n, nch, a.flag, a.mc)
        fp64 = None
        if f64 is not None and not fused_mismatch(f64, names):
            fk = {}
            for k_, v_ in kern.items():
                fl = f64["flop_per_sample_per_launch"].get(k_.split("<")[0])
                if fl is None:
                    continue
                tf = fl * act * nch * n / (v_["avg_launch_us"] * 1e-6) / 1e12
                fk[k_] = {"flop_per_sample_per_launch": fl, "flops_per_launch": fl * act * nch * n, "achieved_TFLOPs": tf,
                          "frac": tf / f64["peak_TFLOPs"]}
            if fk:
                fp64 = {"peak": f64["peak_TFLOPs"], "unit": "TFLOP/s", "peak_source": f64["peak_source"], "source": {k_: f64[k_] for k_ in ("file", "workload", "what", "how")},
                        "kernels": fk}
                if dom in fk:
                    fp64.update({"kernel": dom, "flops_per_launch": fk[dom]["flops_per_launch"], "achieved_TFLOPs": fk[dom]["achieved_TFLOPs"], "frac": fk[dom]["frac"]})
        sweeps = 2 if fused else 3
        group_bytes = SWEEP_BYTES * sweeps
        group_gbs = group_bytes * sample_steps / (fib * 1e-3) / 1e9
        active_frames = float(np.mean([(ncyc > s).sum() for s in range(int(ncyc.max()))])) if len(ncyc) else float(F)
        # (per launch like `achieved`: the frames a launch covers on average -- fewer than F once frames of a ladder have finished)
        traffic, traffic_src = offline_traffic(fused, active_frames, n, nch, a.flag)
        # which roof is nearer: the larger of the two fractions names the bound; with both below one half the kernel is
        # bound by neither rate but by the latency structure of a workgroup's life (occupancy, barriers, exchanges)
        hbm_frac = kern.get(dom, {}).get("frac_of_8TBs") or 0.0
        f64_frac = (fp64 or {}).get("frac")
        if f64_frac is None:
            roof_bound, roof_note = "hbm", "no FP64 instruction count on file for this workload shape: priced against the HBM roof only"
        else:
            roof_bound = "hbm" if hbm_frac >= f64_frac else "fp64-valu"
            roof_note = ("HBM fraction %.3f, FP64 fraction %.3f of the dominant kernel: %s" % (hbm_frac, f64_frac,
                         "the nearer roof is %s" % ("HBM" if hbm_frac >= f64_frac else "the FP64 VALU rate") if max(hbm_frac, f64_frac) >= 0.5 else
                         "NEITHER roof is within a factor of two -- latency-bound (occupancy / barrier / exchange structure), see DESIGN.md"))
        out = {
            "metric": "dual-pol Gsample/s through SSFM+Rx-DSP", "value": value, "unit": "Gsample/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(a, n),
                       "frames_per_gpu_per_step": F, "channels_per_frame": nch, "nsymb": a.nsymb, "nt": a.nt, "pavg_mw": a.pavg,
                       "tx_variants": hp.nvar, "power_ladder": bool(a.power_ladder),
                       "ladder_dbm": [float(v) for v in 10 * np.log10(scales * a.pavg)] if (scales is not None and F <= 64) else None,
                       "ladder_share": ("rank 0's share of %d GPUs: frame f at ladder point (%d f) %% 64" % (lw, lw)) if (a.power_ladder and a.ladder_world > 0) else None,
                       "ssfm_steps_per_frame": sample_steps / (a.steps * F * nch * n),
                       "ssfm_steps_min_max": [int(ncyc.min()), int(ncyc.max())] if len(ncyc) else None,
                       # lock-step launches over frames with different trip counts (fiber.m:518): frame-steps with work / frame
                       # slots the workgroups iterated over (device active list, rebuilt before every step) / frame slots the
                       # host's grids covered (its view lags: surplus workgroups exit at once) / without any compaction
                       "active_frame_utilisation": float(util[0] / max(1, util[1])),
                       "launched_slot_utilisation": float(util[0] / max(1, util[2])),
                       "utilisation_without_compaction": float(ncyc.sum() / (F * ncyc.max())) if len(ncyc) else None,
                       "rx_noise_sigma": a.noise,
                       "fibre_ms_per_step": fib / a.steps, "rxdsp_ms_per_step": rxm / a.steps,
                       "frames_per_s": float(world) * a.steps * F / dt, "fresh_pmd_per_realisation": bool(a.mc),
                       "fused_grid_workgroups": hp.info()[3], "column_tiles_per_frame": hp.info()[4],
                       "fibre_step": ("three sweeps per step, barrier-free (PLX_SSFM_SHARE_DEVICE; %s): the receiver of batch i runs beside the "
                                      "fibre of batch i+1" % share_why) if cfg.share_device else
                                     ("two sweeps per step (fused column sweep)" if fused else "three sweeps per step"),
                       "receiver_beside_next_fibre": bool(rx_stream is not None),
                       "bit_errors_xy": err_total.cpu().tolist(),
                       "bit_errors_resolved": int(res_total.item()) if a.mc else None,
                       "bits": int(world) * a.steps * F * nch * 4 * a.nsymb, "restaged_batches": restaged,
                       "rehearsal_all_ranks_on_one_gpu": bool(rehearsal),
                       "power_state": power_state,
                       "single_frame": single,
                       "with_reference_front_end": cohmix_line},
            # a PMD plan's row pass does one exponential + 20 multiply-adds per waveplate trunk and frequency: FP64-VALU-bound
            # (exception 1), still priced in bytes against the HBM peak
            "roofline": {"bound": roof_bound, "kernel": dom, "fp64": fp64, "bound_note": roof_note,
                         "achieved": kern.get(dom, {}).get("achieved_GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kern.get(dom, {}).get("frac_of_8TBs"),
                         "algorithmic_bytes_per_launch": SWEEP_BYTES * active_frames * nch * n,
                         "bytes_per_sample_per_launch": SWEEP_BYTES,
                         "avg_launch_us": kern.get(dom, {}).get("avg_launch_us"),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernels": kern,
                         "step_group": {"kernels": [k_ for k_ in names[:3] if k_ in kern], "bytes_per_sample_step": group_bytes,
                                        "achieved_GBs": group_gbs, "frac_of_8TBs": group_gbs / HBM_PEAK_GBS,
                                        "sample_steps_per_s": sample_steps / (fib * 1e-3),
                                        "timed_by": "HIP events around plx_ssfm_propagate_dev on the fibre stream (launch gaps, "
                                                    "memsets and the chunked loop's idle launches included)"},
                         "achievable": {"hbm_copy_GBs": HBM_COPY_GBS, "inplace_read_modify_write_GBs": INPLACE_RW_GBS,
                                        "frac_of_inplace_rw": (kern.get(dom, {}).get("achieved_GBs") or 0.0) / INPLACE_RW_GBS},
                         "limiter": "below the HBM roof: %s holds 16 FP64 complex points per lane (256 VGPRs, 8 waves per CU: a lone wave "
                                    "issues one FP64 operation per ~8 clocks, so a tile's arithmetic is as long as its traffic) and a third "
                                    "of a workgroup's time is the skew of the 32 tiles of a frame meeting at the barrier of nextstep's maximum "
                                    "(profiles/r02_notes.md); the row pass (%s) streams at the in-place read-modify-write rate" % (names[0], names[1])
                                    if fused else "three plain sweeps at the in-place streaming rate",
                         "survey_accounting": {"bytes_per_sample_step": SURVEY_BYTES_PER_SAMPLE_STEP,
                                               "frac": SURVEY_BYTES_PER_SAMPLE_STEP * sample_steps / (fib * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "note": "SURVEY 8(d)'s four-sweep budget; this implementation moves %d B" % int(group_bytes)},
                         "launches": int(row_launches)},
            "mc": mc_out,
            "gateway": gateway,
        }
        if not a.no_cpu_baseline and world == 1:      # a reported baseline, timed at N = 1 only
            one, allc = cpu_baseline(cfg, hp, a.cpu_frames, a.noise, frame_steps=sample_steps / (a.steps * nch * n), scales=scales)
            v, cdt, nc = one[:3]
            out["cpu_baseline"] = {"value": v, "unit": "Gsample/s", "cores": 1, "kind": "port",
                                   "sample": one[3] if len(one) > 3 else
                                             "%d frame(s) of the same batch through oracle/ (fibre %d steps + front end + noise + "
                                             "CDE + CMA + CPE), %.1f s" % (a.cpu_frames, nc, cdt)}
            if allc is not None:
                va, busiest, wall, cores, per = allc
                out["cpu_baseline_all_cores"] = {"value": va, "unit": "Gsample/s", "cores": cores, "kind": "port",
                                                 "frames_per_s": cores * per / busiest,     # (frames of THIS line's workload; the Monte-Carlo realisation is timed in mc.cpu_baseline)
                                                 "sample": "%d processes x %d frame(s) each, busiest process %.1f s (%.1f s "
                                                           "wall incl. process start)" % (cores, per, busiest, wall)}
    hp.close()
    del batches, pristine
    torch.cuda.empty_cache()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
