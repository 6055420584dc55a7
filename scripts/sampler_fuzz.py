#!/usr/bin/env python3
"""Fuzz of the samplers through the front end: random inputs (dimension, particles, beads, 'bis' / 'sta', Lstag, Nlev, Nstag, CMFreq,
CWorm, Nobdm, Npw, trap, analytic or tabulated trial function, dt, walkers, shards) run three times -- host-driven sampler on the GPU,
device-resident sampler, and the CPU twin (host sampler over the scalar oracle = the reference's arithmetic) -- and compared bit for bit
(final worldlines; OBDM and permutation files byte for byte).   usage (GPU box): python scripts/sampler_fuzz.py [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pathintegralgroundstate_amd", "host", "pigs_vpi")
SHIM_EXE = os.path.join(ROOT, "tests", "shim", "_build", "pigs_vpi")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
bad = 0
t0 = time.time()
for case in range(ncase):
    dim = int(rng.integers(1, 4))
    trap = bool(rng.random() < 0.3) or os.environ.get("TRAP") == "1"        # TRAP=1: trapped systems only
    wide = os.environ.get("WIDE") == "1"       # larger systems, more walkers (slower per case)
    if wide:
        Np = int(rng.choice([2, 3, 4, 8, 17, 40, 65, 100, 130, 200, 256])) if not trap else int(rng.choice([2, 4, 7, 12, 20]))
        Nb = int(rng.choice([4, 5, 9, 16, 24, 40, 64]))
    else:
        Np = int(rng.choice([2, 3, 5, 9, 16, 21, 33, 64, 70])) if not trap else int(rng.choice([2, 3, 6, 9]))
        Nb = int(rng.choice([4, 6, 8, 10, 12, 16, 20, 33]))
    small = os.environ.get("SMALL") == "1"     # tiny periodic boxes with long free segments: separations of several box lengths
    if small:
        trap = False
        Np = int(rng.choice([2, 3, 4, 6]))
        Nb = int(rng.choice([24, 40, 64]))
    sampling = "bis" if rng.random() < 0.6 else "sta"
    cworm = float(rng.choice([0.0, 0.3, 0.6, 2.0]))
    nlev_max = min(4 if trap else 7, int(np.floor(np.log2(2 * Nb))))
    Nlev = int(rng.integers(1, nlev_max + 1))
    Lstag = int(rng.integers(2, (Nb if cworm > 0 else 2 * Nb) + 1))
    Nstag = int(rng.integers(1, 4))
    CMFreq = int(rng.integers(1, 4))
    Nobdm = int(rng.integers(0, 6))
    Npw = int(rng.integers(0, 3))
    wf_table = "T" if rng.random() < 0.75 else "F"
    dt = float(rng.choice([5e-3, 1e-2, 3e-2]))
    dens = float(rng.choice([0.05, 0.2, 0.365]))
    if small:
        dt = float(rng.choice([3e-2, 6e-2]))
        dens = float(rng.choice([0.2, 0.365, 0.6]))
    NW = int(rng.choice([1, 2, 5])) if not wide else int(rng.choice([1, 3, 7, 260], p=[0.4, 0.3, 0.25, 0.05]))
    if wide and NW > 100 and Np > 20:
        NW = 7
    G = 2 if (NW >= 2 and rng.random() < 0.3) else 1
    a_ho = " ".join(["1.0d0", "1.3d0", "0.8d0"][:dim])
    # later additions (drawn last, so that the earlier records' case numbers keep their inputs when MORE is unset): the width of
    # the TranslateChain shift (up to several box lengths in a small box: beads left outside the box by the ONE fold of
    # BoundaryConditions), the Jastrow range, swapping off (quirk Q9: only without a worm)
    more = os.environ.get("MORE") == "1"
    delta_cm, Rm, swapping = 0.2, 1.10, "T"
    pot = str(rng.choice(["aziz2", "aziz2", "lj", "dipolar"]))          # the reference compiles ONE potential in; the table kinds are an input here
    k1v = int(rng.choice([0, 0, 2]))                                     # Delta-S kernel of the host-driven arm: default short arithmetic / exact-term
    nstep = 40 if not wide else (12 if Np >= 100 else (20 if Np >= 40 else 40))
    if more:
        delta_cm = float(rng.choice([0.05, 0.2, 1.0, 4.0, 9.0]))
        Rm = float(rng.choice([0.9, 1.1, 1.4]))
        if cworm == 0.0 and rng.random() < 0.3:
            swapping = "F"
    inp = f"""&system
 dim = {dim}, Np = {Np}, density = {dens}d0, trap = {'T' if trap else 'F'}
/
&samp
 resume = F, dt = {dt}d0, Nb = {Nb}, seed = {1000 + case}, delta_cm = {delta_cm}d0, CMFreq = {CMFreq},
 sampling = '{sampling}', Lstag = {Lstag}, Nlev = {Nlev}, Nstag = {Nstag}, Nblock = 2, Nstep = {nstep // 2}, Nbin = 40, Nk = 6
/
&obdm
 swapping = {swapping}, Nobdm = {Nobdm}, Npw = {Npw}, CWorm = {cworm}d0
/
&wavefun
 Nmax = 4000, wf_table = {wf_table}, v_table = T
/
&jastrow
 Rm = {Rm}d0
/
&extpot
 a_ho = {a_ho}
/
"""
    tag = (f"case {case}: dim={dim} Np={Np} Nb={Nb} {sampling} Lstag={Lstag} Nlev={Nlev} Nstag={Nstag} CMFreq={CMFreq} CWorm={cworm} "
           f"Nobdm={Nobdm} Npw={Npw} trap={trap} wf_table={wf_table} dt={dt} rho={dens} walkers={NW} shards={G} potential={pot} k1_variant(F)={k1v}" + (f" delta_cm={delta_cm} Rm={Rm} swapping={swapping}" if more else ""))
    if os.environ.get("ONLY") and case != int(os.environ["ONLY"]):      # ONLY=<case>: that case of the sequence alone
        continue
    out = {}
    fail = None
    for arm in "FTC":
        d = tempfile.mkdtemp(dir=os.environ.get("KEEP"))                # KEEP=<dir>: the arms' working directories stay there
        if os.environ.get("KEEP"):
            print("arm", arm, d, flush=True)
        gpu = f"&gpu\n n_walkers = {NW}, device = 0, device_sampler = {'T' if arm == 'T' else 'F'}, checkpointing = F, potential = '{pot}'"
        if arm == "F" and k1v:
            gpu += f", k1_variant = {k1v}"
        if G > 1:
            gpu += f", n_gpus = {G}, same_device = T"
        with open(os.path.join(d, "vpi.in"), "w") as f:
            f.write(inp + gpu + "\n/\n")
        with open(os.path.join(d, "vpi.in")) as fin, open(os.path.join(d, "out.txt"), "w") as fo:
            r = subprocess.run([SHIM_EXE if arm == "C" else EXE], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=d, timeout=900)
        if r.returncode != 0:
            fail = f"arm {arm} rc={r.returncode}: " + open(os.path.join(d, "out.txt")).read()[-300:].replace("\n", " | ")
            break
        out[arm] = d
    if fail is None and os.environ.get("RESUME", "1") == "1":
        # checkpoint / resume of the device-resident sampler: block 1, stop, resume for block 2 == the straight two-block run
        d = tempfile.mkdtemp()
        gpu = f"&gpu\n n_walkers = {NW}, device = 0, device_sampler = T, checkpointing = T, potential = '{pot}'"
        if G > 1:
            gpu += f", n_gpus = {G}, same_device = T"
        for part, txt in enumerate((inp.replace("Nblock = 2", "Nblock = 1"), inp.replace("Nblock = 2", "Nblock = 1").replace("resume = F", "resume = T"))):
            with open(os.path.join(d, "vpi.in"), "w") as f:
                f.write(txt + gpu + "\n/\n")
            with open(os.path.join(d, "vpi.in")) as fin, open(os.path.join(d, f"out{part}.txt"), "w") as fo:
                r = subprocess.run([EXE], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=d, timeout=900)
            if r.returncode != 0:
                fail = f"resume arm part {part} rc={r.returncode}: " + open(os.path.join(d, f"out{part}.txt")).read()[-300:].replace("\n", " | ")
                break
        if fail is None:
            wr = np.fromfile(os.path.join(d, "worldlines_final.bin"))
            wt = np.fromfile(os.path.join(out["T"], "worldlines_final.bin"))
            if not np.array_equal(wr.view(np.uint64), wt.view(np.uint64)):
                fail = "device sampler: block 1 + resume for block 2 differs from the straight run"
    if fail is None:
        w = {a: np.fromfile(os.path.join(out[a], "worldlines_final.bin")) for a in out}
        if not (np.array_equal(w["F"].view(np.uint64), w["T"].view(np.uint64))):
            fail = "host-driven (GPU) vs device-resident: worldlines differ"
        elif not np.array_equal(w["F"].view(np.uint64), w["C"].view(np.uint64)):
            fail = "GPU vs CPU twin: worldlines differ"
        else:
            def hexrows(path):
                import struct
                rows = []
                for ln in open(path):
                    t = ln.split()
                    if t:
                        rows.append([float(int(t[0]))] + [struct.unpack(">d", bytes.fromhex(h))[0] for h in t[1:7]])
                return np.array(rows, float).reshape(-1, 7)
            for wk in range(NW):
                for nme in ("nr_vpi", "perm_vpi", "gr_vpi"):      # integer-valued histograms: byte for byte
                    fn = f"{nme}.w{wk:04d}.out" if NW > 1 else f"{nme}.out"
                    pa, pb = os.path.join(out["F"], fn), os.path.join(out["T"], fn)
                    if os.path.exists(pa) and open(pa, "rb").read() != open(pb, "rb").read():
                        fail = f"{fn} differs between the samplers"
                # block energies: the same kernels on the same worldlines in the two GPU arms -> the same bits; the CPU twin's
                # (the oracle's) estimators to 1e-10 (E, K against |K|+|V|)
                hn = f"e_vpi.w{wk:04d}.hex" if NW > 1 else "e_vpi.hex"
                hf, ht, hc = (hexrows(os.path.join(out[a], hn)) for a in "FTC")
                if hf.shape != ht.shape or not np.array_equal(hf.view(np.uint64), ht.view(np.uint64)):
                    fail = f"{hn}: block energies differ between the two GPU arms"
                elif hf.shape != hc.shape:
                    fail = f"{hn}: number of diagonal blocks differs from the CPU twin"
                elif len(hf):
                    sc_e = np.abs(hc[:, 2]) + np.abs(hc[:, 3]); sc_t = np.abs(hc[:, 5]) + np.abs(hc[:, 6])
                    scale = np.stack([sc_e, sc_e, np.abs(hc[:, 3]), sc_t, sc_t, np.abs(hc[:, 6])], 1)
                    ok = np.isfinite(hc[:, 1:]) & (scale > 0)
                    if np.any(np.abs(hf[:, 1:] - hc[:, 1:])[ok] > 1e-10 * scale[ok]):
                        fail = f"{hn}: block energies differ from the CPU twin's beyond 1e-10: {np.max((np.abs(hf[:, 1:] - hc[:, 1:]) / np.where(scale > 0, scale, 1))[ok]):.2e}"
                sn = f"sk_vpi.w{wk:04d}.out" if NW > 1 else "sk_vpi.out"
                if not trap and os.path.exists(os.path.join(out["F"], sn)):
                    a_, b_ = np.loadtxt(os.path.join(out["F"], sn)), np.loadtxt(os.path.join(out["T"], sn))
                    # host libm cos / sin (host-driven arm: sample_mod.f90:430-470 on the mirror) against the device library's in
                    # k_structure: S(k) itself to 1e-8; its error bar is the root of <S^2> - <S>^2 over two blocks -- a difference
                    # of 1e-11 between numbers of the size of S^2 -- and is compared on the scale of S (round 3: WIDE seed 4242 case 24)
                    ok_ = a_.shape == b_.shape
                    if ok_:
                        a3, b3 = a_.reshape(-1, 3), b_.reshape(-1, 3)           # (q, S, error bar) per direction and q
                        ok_ = np.allclose(a3[:, :2], b3[:, :2], rtol=1e-8, atol=1e-12, equal_nan=True) and \
                            bool(np.all((np.abs(a3[:, 2] - b3[:, 2]) <= 1e-8 * (np.abs(a3[:, 1]) + np.abs(a3[:, 2])) + 1e-12) | (np.isnan(a3[:, 2]) & np.isnan(b3[:, 2]))))
                    if not ok_:
                        fail = f"{sn}: S(k) differs between the samplers"
    if fail:
        bad += 1
        print("FAIL", tag, "|", fail, flush=True)
    elif case % 5 == 0:
        print("ok  ", tag, flush=True)
print(f"{ncase} cases, {bad} failing, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
