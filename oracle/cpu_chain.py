"""TEST INFRASTRUCTURE ONLY -- the whole hot path of one frame on the CPU oracle, for bench.py's cpu_baseline leg.

fibre (plxo.matrix_ssfm, fiber.m:459-554) -> front end (2-sps pick, or oracle/front.py) -> receiver noise ->
CDE_OFDE -> DspPdmCohQpsk (CMA + carrier recovery) -> samp2pat.  Importable without torch so that worker processes
(one frame per host core) start quickly.
"""
import time

import numpy as np


def run_frames(p, nframes, seed):
    """p: dict of plain numpy arrays / scalars (picklable).  Returns (seconds, ncycle of the last frame)."""
    from oracle import plxo
    r = np.random.default_rng(seed)
    nc = 0
    t0 = time.perf_counter()
    for _ in range(nframes):
        nc = 0
        ox, oy = p["tx_x"], p["tx_y"]
        nspans = int(p.get("nspans", 1))
        mc = p.get("mc")                 # a Monte-Carlo realisation (BASELINE config[3]): fresh waveplates, amplifier ASE after the span
        for span in range(nspans):
            nplates, db0, th, ep = 1, [0.0], [0.0], [0.0]
            if mc:                       # fiber.m:274-276
                nplates = int(mc["nplates"])
                db0 = r.random(nplates) * 2 * np.pi - np.pi
                th = r.random(nplates) * np.pi - 0.5 * np.pi
                ep = 0.5 * np.arcsin(r.random(nplates) * 2 - 1)
            rc, fd, k, ox, oy = plxo.matrix_ssfm(ox, oy, p["betat"], p["db1"], p["dzmax"], p["dphimax"], p["gam"],
                                                 p["alphalin"], p["length"], nplates, 0, p["fls"], db0, th, ep)
            nc += k
            if mc and span + 1 == nspans:                              # ampliflat(G,'gain',{f}) behind the span, ampliflat.m:91-143
                g = np.exp(p["alphalin"] * p["length"])
                ox = np.sqrt(g) * ox + mc["amp_sigma"] * (r.standard_normal(ox.shape) + 1j * r.standard_normal(ox.shape))
                oy = np.sqrt(g) * oy + mc["amp_sigma"] * (r.standard_normal(oy.shape) + 1j * r.standard_normal(oy.shape))
            if span + 1 < nspans:                                      # in-line amplifier, ampliflat.m:123-143
                g = np.exp(p["alphalin"] * p["length"])
                ox, oy = np.sqrt(g) * ox, np.sqrt(g) * oy
                if p.get("span_sigma"):
                    ox = ox + p["span_sigma"] * (r.standard_normal(ox.shape) + 1j * r.standard_normal(ox.shape))
                    oy = oy + p["span_sigma"] * (r.standard_normal(oy.shape) + 1j * r.standard_normal(oy.shape))
        if p.get("front") is not None:
            from oracle import front
            t = p["front"]
            cur = front.receiver_cohmix(ox[:, 0], oy[:, 0], t["hopt"], t["elo"], t["hel"], True)
            rx = front.rx_front(cur, True, p["adcbits"], p["front_shifts"], t["decim"], t["fir"])
        else:
            half = p["nt"] // 2
            rx = np.stack([ox[::half, 0], oy[::half, 0]], 1) * p["rx_scale"]
        if p["noise"]:
            rx = rx + p["noise"] * (r.standard_normal(rx.shape) + 1j * r.standard_normal(rx.shape))
        ex, ey, _ = plxo.cde_ofde(rx[:, 0], rx[:, 1], 2 * p["symbolrate"] * 1e9, p["lam"] * 1e-9, p["length"] * int(p.get("nspans", 1)), p["disp"] * 1e-6,
                                  p["slope"] * 1e-6, p["fft_length"], p["cde_L"])
        op = plxo.dsp_params(power_mw=p["power_mw"], applypol=True, polmethod="cma", cma_mu=p["cma_mu"], cma_taps=p["cma_taps"],
                             freqavg=p["freqavg"], phasavg=p["phasavg"], poworder=p["poworder"])
        sym = plxo.dsp_pdm_coh_qpsk(np.stack([ex, ey], 1), op)
        plxo.samp2pat_coherent(np.angle(sym))
    return time.perf_counter() - t0, nc


def run_wdm_frame(p, spans, seed):
    """One 'sepfields' WDM frame (BASELINE config[2]: tx_x / tx_y [nfft x nch], shared step length fiber.m:694-698) through
    `spans` of its p["nspans"] spans with the in-line amplifiers, then EVERY channel through the receiver chain.
    Returns (seconds in the fibre spans, seconds in the receivers, ncycle summed over the spans done)."""
    from oracle import plxo
    r = np.random.default_rng(seed)
    ox, oy = p["tx_x"], p["tx_y"]
    nch = ox.shape[1]
    nc = 0
    t0 = time.perf_counter()
    for span in range(spans):
        nplates, db0, th, ep = 1, [0.0], [0.0], [0.0]
        if p["fls"][1]:                  # fiber.m:274-276
            nplates = int(p["nplates"])
            db0 = r.random(nplates) * 2 * np.pi - np.pi
            th = r.random(nplates) * np.pi - 0.5 * np.pi
            ep = 0.5 * np.arcsin(r.random(nplates) * 2 - 1)
        rc, fd, k, ox, oy = plxo.matrix_ssfm(ox, oy, p["betat"], p["db1"], p["dzmax"], p["dphimax"], p["gam"],
                                             p["alphalin"], p["length"], nplates, 0, p["fls"], db0, th, ep)
        nc += k
        g = np.exp(p["alphalin"] * p["length"])                       # in-line amplifier, ampliflat.m:123-143
        ox, oy = np.sqrt(g) * ox, np.sqrt(g) * oy
        if p.get("span_sigma") is not None:
            sg = np.asarray(p["span_sigma"]).reshape(1, -1)
            ox = ox + sg * (r.standard_normal(ox.shape) + 1j * r.standard_normal(ox.shape))
            oy = oy + sg * (r.standard_normal(oy.shape) + 1j * r.standard_normal(oy.shape))
    t1 = time.perf_counter()
    half = p["nt"] // 2
    g = np.exp(-0.5 * p["alphalin"] * p["length"])                    # (the loop amplified the last span too: rx_scale undoes its loss itself)
    for c in range(nch):
        rx = np.stack([ox[::half, c], oy[::half, c]], 1) * (p["rx_scale"] * g)
        if p["noise"]:
            rx = rx + p["noise"] * (r.standard_normal(rx.shape) + 1j * r.standard_normal(rx.shape))
        ex, ey, _ = plxo.cde_ofde(rx[:, 0], rx[:, 1], 2 * p["symbolrate"] * 1e9, p["lam"] * 1e-9, p["length"] * int(p.get("nspans", 1)), p["disp"] * 1e-6,
                                  p["slope"] * 1e-6, p["fft_length"], p["cde_L"])
        op = plxo.dsp_params(power_mw=p["power_mw"], applypol=True, polmethod="cma", cma_mu=p["cma_mu"], cma_taps=p["cma_taps"],
                             freqavg=p["freqavg"], phasavg=p["phasavg"], poworder=p["poworder"])
        sym = plxo.dsp_pdm_coh_qpsk(np.stack([ex, ey], 1), op)
        plxo.samp2pat_coherent(np.angle(sym))
    return t1 - t0, time.perf_counter() - t1, nc


def run_span_sample(p, scale, seed):
    """Bounded sample of a LONG-HAUL frame (BASELINE config[4]: 2^20 samples x 40 spans is minutes of CPU work per frame): the
    Tx frame at launch-power scale `scale` through ONE span, then through the receiver chain once (a timing sample: the
    receiver sees the field after one span).  Returns (seconds in the span, its ncycle, seconds in the receiver)."""
    from oracle import plxo
    r = np.random.default_rng(seed)
    a = np.sqrt(scale)
    t0 = time.perf_counter()
    rc, fd, k, ox, oy = plxo.matrix_ssfm(p["tx_x"] * a, p["tx_y"] * a, p["betat"], p["db1"], p["dzmax"], p["dphimax"], p["gam"],
                                         p["alphalin"], p["length"], 1, 0, p["fls"], [0.0], [0.0], [0.0])
    t1 = time.perf_counter()
    half = p["nt"] // 2
    rx = np.stack([ox[::half, 0], oy[::half, 0]], 1) * (p["rx_scale"] / a)
    if p["noise"]:
        rx = rx + p["noise"] * (r.standard_normal(rx.shape) + 1j * r.standard_normal(rx.shape))
    ex, ey, _ = plxo.cde_ofde(rx[:, 0], rx[:, 1], 2 * p["symbolrate"] * 1e9, p["lam"] * 1e-9, p["length"], p["disp"] * 1e-6,
                              p["slope"] * 1e-6, p["fft_length"], p["cde_L"])
    op = plxo.dsp_params(power_mw=p["power_mw"], applypol=True, polmethod="cma", cma_mu=p["cma_mu"], cma_taps=p["cma_taps"],
                         freqavg=p["freqavg"], phasavg=p["phasavg"], poworder=p["poworder"])
    sym = plxo.dsp_pdm_coh_qpsk(np.stack([ex, ey], 1), op)
    plxo.samp2pat_coherent(np.angle(sym))
    return t1 - t0, k, time.perf_counter() - t1


def run_parallel(p, frames_per_core, cores, timeout_s=180.0):
    """One child process per core (`python -m oracle.cpu_chain params.npz n seed`: fresh interpreters that never see the
    parent's GPU state), each running frames_per_core frames.  Returns (wall seconds, busiest child's compute seconds)
    or None if a child fails or the time limit passes (children are then terminated)."""
    import os
    import pickle
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        pf = os.path.join(d, "params.pkl")
        with open(pf, "wb") as f:
            pickle.dump(p, f)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_chain", pf, str(frames_per_core), str(1000 + k)], cwd=root,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(cores)]
        secs, ok = [], True
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=max(1.0, timeout_s - (time.perf_counter() - t0)))
                secs.append(float(out.strip().split()[-1]))
            except Exception:
                ok = False
        if not ok or len(secs) != cores:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
            return None
        return time.perf_counter() - t0, max(secs)


if __name__ == "__main__":
    import pickle
    import sys
    with open(sys.argv[1], "rb") as f:
        params = pickle.load(f)
    dt, _ = run_frames(params, int(sys.argv[2]), int(sys.argv[3]))
    print(dt)
