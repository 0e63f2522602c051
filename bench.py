#!/usr/bin/env python
"""bench.py -- dual-pol Gsample/s through SSFM + Rx-DSP on MI355X.

One "step" = one pass of the hot path (fiber 'g-s-' split-step Fourier propagation
-> 2-sps pick -> CDE_OFDE overlap-save -> DspPdmCohQpsk with CMA + carrier recovery
-> decisions / error count) over ONE BATCH of synthetic PDM-QPSK frames that is
already resident in HBM when the timed region starts.  Workload at N=1: BASELINE
config[1] (Run_my_PDM_QPSK-style: 28 Gbaud PDM-QPSK, 2^16-sample frame, one 80 km
SSMF span + CMA demux), batched over --frames independent frames per GPU.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); frames are
independent units, so ranks shard them with no data-path collective (weak scaling);
the only exchange is the final all-reduce of the error counters.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
SSFM_BYTES_PER_SAMPLE_STEP = 272.0   # 4 sweeps x (32 R + 32 W) + 16 B of betat/db1, dual-pol


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--nsymb", type=int, default=1024)
    ap.add_argument("--nt", type=int, default=64)
    ap.add_argument("--pavg", type=float, default=2.0)
    ap.add_argument("--flag", default="g-s-")
    ap.add_argument("--noise", type=float, default=0.05, help="receiver noise sigma per quadrature (full scale 1)")
    ap.add_argument("--frontend", default="pick", choices=["pick", "cohmix"],
                    help="pick: 2-sps sampling supplied by the harness (SURVEY 8d C1); cohmix: receiver_cohmix + ADC + decimate on the device")
    ap.add_argument("--mc", action="store_true",
                    help="BASELINE config[3] style step: fiber('gps-') with a fresh random-birefringence draw per frame "
                         "and per step (Monte-Carlo PMD realisations), receiver noise as ASE stand-in")
    ap.add_argument("--spans", type=int, default=1, help="spans per step, with in-line amplifiers between them (config[4]: 40)")
    ap.add_argument("--nf", type=float, default=None, help="noise figure [dB] of the in-line amplifiers (default: noiseless)")
    ap.add_argument("--no-overlap", action="store_true", help="run the receiver on the fibre stream (no stream overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the one-frame latency measurement after the timed region")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the batch the one-core CPU baseline processes (~0.5 s each)")
    return ap.parse_args()


class HipEvents:
    """HIP events on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def create(self):
        e = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value


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
    if cfg.span_nf_db is not None and cfg.nspans > 1:
        from polmux_amd.ampliflat import ase_sigma
        p["span_sigma"] = float(ase_sigma(cfg.span_nf_db, math.exp(hp.alphalin * cfg.length), 1)[0])
    if hp.front is not None:
        p["front"] = hp.front_tables
        p["front_shifts"] = hp.front_shifts
    return p


def cpu_baseline(cfg, hp, nframes, noise):
    """The reference algorithm restated on the CPU (oracle/, kind 'port') on a bounded sample of the SAME workload
    (frames of the batch, same receiver noise level): one core, then one frame stream per host core."""
    from oracle import cpu_chain
    p = cpu_params(cfg, hp, noise)
    dt, nc = cpu_chain.run_frames(p, nframes, 999)
    one = (nframes * cfg.nfft / dt / 1e9, dt, nc)
    cores = host_cores()
    per = max(1, nframes // 4)
    res = cpu_chain.run_parallel(p, per, cores)
    allc = None if res is None else (cores * per * cfg.nfft / res[1] / 1e9, res[1], res[0], cores, per)
    return one, allc


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box (dev only): every rank on cuda:0, collectives over gloo
    rehearsal = os.environ.get("PLX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        os.environ["PLX_SSFM_NO_FUSE"] = "1"     # several ranks share one GPU here: the fused sweep assumes it owns the chip
    torch.cuda.set_device(local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from polmux_amd import _abi, pipeline
    _abi.get().call("plx_set_device", local)

    if a.mc:
        a.flag = "gps-"
    cfg = pipeline.HotPathConfig(nsymb=a.nsymb, nt=a.nt, pavg_mw=a.pavg, flag=a.flag, frontend=a.frontend, nspans=a.spans,
                                 span_nf_db=a.nf)
    F = a.frames
    hp = pipeline.HotPath(cfg, max_frames=F)
    n = cfg.nfft
    # inputs for every step are staged in HBM before the timed region (fibre works in place)
    total = a.steps + a.warmup
    # (bounded by free HBM: with more steps than buffers a buffer is refilled from a pristine copy by one
    # device-to-device copy on the fibre stream -- inside the timed region, reported as "restaged_batches")
    batch_bytes = 2 * F * n * 16
    free_b, _tot = torch.cuda.mem_get_info()
    nbuf = max(2, min(total, int(0.6 * free_b // batch_bytes)))
    if os.environ.get("PLX_BENCH_NBUF"):      # dev: force the restaging path
        nbuf = max(2, min(total, int(os.environ["PLX_BENCH_NBUF"])))
    batches = [hp.make_batch(F) for _ in range(nbuf)]
    pristine = hp.make_batch(F) if nbuf < total else None
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
    rx_stream = None if a.no_overlap else torch.cuda.Stream()
    err_total = torch.zeros(2, dtype=torch.int64, device="cuda")
    errs, resolved = [], []
    fib_ms, rx_ms, row_launches, sample_steps = [], [], 0, 0
    for i in range(a.warmup):
        ux, uy = get_batch(i)
        hp.run(ux, uy, noise_sigma=a.noise, noise_seed=1000 * rank + i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(a.warmup, total):
        ux, uy = get_batch(i)
        e0, e1, e2 = ev.create(), ev.create(), ev.create()
        ev.record(e0, stream)
        if a.mc:   # realisation index = (rank, step, frame): fresh waveplates, keyed so that sharding does not change them
            hp.set_random_pmd(range((rank * total + i) * F, (rank * total + i + 1) * F))
        hp.fibre(ux, uy)
        ev.record(e1, stream)
        rs = rx_stream.cuda_stream if rx_stream is not None else stream
        e1b = ev.create()
        err = hp.receive(ux, uy, noise_sigma=a.noise, noise_seed=1000 * rank + i, side_stream=rx_stream)
        # (receive() makes rx_stream wait for the fibre; the Rx interval is measured on the Rx stream)
        ev.record(e2, rs)
        if rx_stream is not None:
            with torch.cuda.stream(rx_stream):
                errs.append(err.sum(0))
                if a.mc:   # a blind receiver behind random birefringence: resolve pol swap + pi/2 ambiguity (ex20:160-173)
                    resolved.append(hp.errors_resolved(F).sum())
        else:
            errs.append(err.sum(0))
            if a.mc:
                resolved.append(hp.errors_resolved(F).sum())
        if rx_stream is not None and nbuf < total:
            buf_free[i % nbuf] = torch.cuda.Event()
            buf_free[i % nbuf].record(rx_stream)
        fib_ms.append((e0, e1)); rx_ms.append((e1, e2))
        rl, ss = hp.ssfm_stats()
        row_launches += rl; sample_steps += ss
    sync_all()
    dt = time.perf_counter() - t0
    for e in errs:
        err_total += e
    res_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    for r_ in resolved:
        res_total += r_
    if world > 1:
        cdev = "cpu" if rehearsal else "cuda"
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        err_total, res_total = err_total.to(cdev), res_total.to(cdev)
        dist.all_reduce(err_total, op=dist.ReduceOp.SUM)       # the one exchange step (RCCL over xGMI)
        dist.all_reduce(res_total, op=dist.ReduceOp.SUM)
    fib = sum(ev.elapsed_ms(x, y) for x, y in fib_ms)
    rxm = sum(ev.elapsed_ms(x, y) for x, y in rx_ms)

    # SURVEY 8d's M1 read literally -- ONE frame through fibre + receiver, nothing else on the GPU (outside the timed region)
    single = None
    if rank == 0 and not a.mc and not a.no_single_frame:
        sx, sy = hp.make_batch(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hp.fibre(sx, sy)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hp.receive(sx, sy, noise_sigma=a.noise, noise_seed=4242)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        single = {"fibre_ms": (t2 - t1) * 1e3, "rx_ms": (t3 - t2) * 1e3, "gsample_per_s": n / (t3 - t1) / 1e9}
    if rank == 0:
        samples = float(world) * a.steps * F * n
        value = samples / dt / 1e9
        # roofline of the dominant kernel group: the SSFM step (3 sweeps), HBM-bound.
        # achieved = algorithmic bytes (272 B per dual-pol sample-step) / measured fibre time (HIP events)
        achieved = SSFM_BYTES_PER_SAMPLE_STEP * sample_steps / (fib * 1e-3) / 1e9
        # HBM traffic per step-launch from the PMC counters (collected offline exactly as the micro-arch guide
        # prescribes: separate FETCH_SIZE / WRITE_SIZE passes, x2 read correction on gfx950), profiles/r01_traffic.json
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and a.flag == "g-s-" and a.nsymb * a.nt == 65536:
            tr = json.load(open(tj))
            if os.environ.get("PLX_SSFM_NO_FUSE"):
                tr = tr["plain_three_sweep"]
            traffic = tr["bytes_per_sample_step"] * F * n
        out = {
            "metric": "dual-pol Gsample/s through SSFM+Rx-DSP", "value": value, "unit": "Gsample/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Run_my_PDM_QPSK-style (BASELINE config[1]): 28 Gbaud PDM-QPSK, 2^%d-sample "
                                   "dual-pol frame, %dx80 km SSMF span fiber('%s'), CDE_OFDE 256/128, CMA 7 taps + "
                                   "V&V carrier recovery; front end: %s" % (int(np.log2(n)), a.spans, a.flag, a.frontend),
                       "frames_per_gpu_per_step": F, "nsymb": a.nsymb, "nt": a.nt, "pavg_mw": a.pavg,
                       "ssfm_steps_per_frame": sample_steps / (a.steps * F * n), "rx_noise_sigma": a.noise,
                       "fibre_ms_per_step": fib / a.steps, "rxdsp_ms_per_step": rxm / a.steps,
                       "mc_realisations_per_s": float(world) * a.steps * F / dt, "fresh_pmd_per_realisation": bool(a.mc),
                       "bit_errors_xy": err_total.cpu().tolist(),
                       "bit_errors_resolved": int(res_total.item()) if a.mc else None,
                       "bits": int(world) * a.steps * F * 4 * a.nsymb, "restaged_batches": restaged,
                       "single_frame": single},
            "roofline": {"bound": "hbm", "kernel": "SSFM step (k_colx16 [inverse + forward column pass, fused] + k_row)"
                         if not os.environ.get("PLX_SSFM_NO_FUSE") and a.nsymb * a.nt == 65536 else "SSFM step (k_col_fwd + k_row + k_col_inv)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": SSFM_BYTES_PER_SAMPLE_STEP * F * n,
                         "traffic": traffic, "sample_steps_per_s": sample_steps / (fib * 1e-3),
                         "launches": row_launches, "ms_per_step_launch": fib / max(1, row_launches),
                         "measured_inplace_rw_stream_GBs": 5100.0,   # scripts/micro/seg_copy.hip, profiles/r01_notes.md
                         "note": "one step-launch = the kernels of one SSFM step over the whole batch; 'launches' also counts "
                                 "the few no-op launches of the chunked step loop after every frame has finished"},
        }
        if not a.no_cpu_baseline and world == 1:      # a reported baseline, timed at N = 1 only
            (v, cdt, nc), allc = cpu_baseline(cfg, hp, a.cpu_frames, a.noise)
            out["cpu_baseline"] = {"value": v, "unit": "Gsample/s", "cores": 1, "kind": "port",
                                   "sample": "%d frame(s) of the same batch through oracle/ (fibre %d steps + front end + noise + "
                                             "CDE + CMA + CPE), %.1f s" % (a.cpu_frames, nc, cdt)}
            if allc is not None:
                va, busiest, wall, cores, per = allc
                out["cpu_baseline_all_cores"] = {"value": va, "unit": "Gsample/s", "cores": cores, "kind": "port",
                                                 "realisations_per_s": cores * per / busiest,
                                                 "sample": "%d processes x %d frame(s) each, busiest process %.1f s (%.1f s "
                                                           "wall incl. process start)" % (cores, per, busiest, wall)}
        print(json.dumps(out))
    hp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
