#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Run in the build container only (needs oracle/_ref/, i.e. /root/reference compiled by
oracle/Makefile with AMD flang -O2).  Every output below is computed by the reference's own
routines through oracle/ref_probe.f90 (full 64-bit values); nothing here comes from our
restatement.  The fixtures are data: inputs + expected outputs, no reference source text.

    python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import REF_VPI, Ref, System, build_oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

VPI_IN = """&system
 dim = {dim}, Np = {Np}, density = {density}, trap = {trap}
/
&samp
 resume = F, dt = {dt}, Nb = {Nb}, seed = {seed}, delta_cm = 0.12d0, CMFreq = 1,
 sampling = '{sampling}', Lstag = {Lstag}, Nlev = {Nlev}, Nstag = {Nstag},
 Nblock = {Nblock}, Nstep = {Nstep}, Nbin = 100, Nk = 50
/
&obdm
 swapping = T, CWorm = {CWorm}, Nobdm = {Nobdm}, Npw = {Npw}
/
&wavefun
 Nmax = 10000, wf_table = {wf_table}, v_table = T
/
&jastrow
 Rm = 1.20d0
/
&extpot
 a_ho = {a_ho}
/
"""


def run_vpi(workdir, **kw):
    """Run the stock reference program; returns the checkpointed worldline (M,Np,dim)."""
    p = dict(dim=3, Np=64, density="0.365d0", trap="F", dt="5.0d-3", Nb=40, seed=1982,
             sampling="bis", Lstag=16, Nlev=4, Nstag=5, Nblock=1, Nstep=10, CWorm="0.0d0",
             Nobdm=0, Npw=0, a_ho="1.0d0", wf_table="T")
    p.update(kw)
    with open(os.path.join(workdir, "vpi.in"), "w") as f:
        f.write(VPI_IN.format(**p))
    with open(os.path.join(workdir, "vpi.in")) as fin, open(os.path.join(workdir, "stdout"), "w") as fo:
        subprocess.run([REF_VPI], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=workdir,
                       check=False, timeout=600)
    dim, Np, M = int(p["dim"]), int(p["Np"]), 2 * int(p["Nb"]) + 1
    with open(os.path.join(workdir, "checkpoint.dat")) as f:
        lines = f.read().split("\n")
    vals = np.array([[float(x) for x in ln.split()] for ln in lines[3:3 + Np * M]])
    # checkpoint is particle-major: do ip; do ib  (vpi_mod.f90:289-295)
    return np.ascontiguousarray(vals.reshape(Np, M, dim).transpose(1, 0, 2))


def wrap(x, L):
    x = np.where(x > L / 2, x - L, x)
    return np.where(x < -L / 2, x + L, x)


def action_cases(ref, S, VT, WF, Path, n, rng, sigma):
    """n UpdateAction cases over even / odd / end beads; outputs from the reference."""
    ip = rng.integers(1, S.Np + 1, n).astype(np.int32)
    ib = rng.integers(0, S.M, n).astype(np.int32)
    ib[::7] = 0
    ib[3::7] = 2 * S.Nb
    xold = Path[ib, ip - 1].copy()
    xnew = xold + rng.normal(0.0, sigma, xold.shape)
    if not S.trap:
        xnew = wrap(xnew, S.Lbox[:S.dim])
    dS = np.empty(n)
    parts = np.empty((n, 3))
    for i in range(n):
        dS[i] = ref.update_action(WF, VT, Path, int(ip[i]), int(ib[i]), xnew[i], xold[i])
        odd = int(ib[i]) % 2 == 1
        dp, df = ref.update_pot(VT, int(ip[i]), Path[ib[i]], xnew[i], xold[i], odd)
        dw = ref.update_wf(WF, int(ip[i]), Path[ib[i]], xnew[i], xold[i]) \
            if ib[i] in (0, 2 * S.Nb) else 0.0
        parts[i] = (dp, df, dw)
    return dict(ip=ip, ib=ib, xnew=xnew, xold=xold, DeltaS=dS, parts=parts)


def energies(ref, S, VT, WF, Path):
    pe = np.array([ref.potential_energy(VT, Path[ib], True) for ib in range(S.M)])
    pe0 = np.array([ref.potential_energy(VT, Path[ib], False)[0] for ib in range(S.M)])
    le = np.array([ref.local_energy(WF, VT, Path[0]), ref.local_energy(WF, VT, Path[2 * S.Nb])])
    te = np.array(ref.therm_energy(VT, Path))
    return dict(pot_f2=pe, pot_only=pe0, local=le, therm=te)


def sysmeta(S):
    return dict(dim=S.dim, Np=S.Np, Nb=S.Nb, Nmax=S.Nmax, density=S.density, Rm=S.Rm, dt=S.dt,
                trap=int(S.trap), a_ho=S.a_ho, Lbox=S.Lbox, rcut=S.rcut, dr=S.dr)


def main():
    build_oracle()
    assert Ref.available(), "oracle/_ref/libvpiref.so missing: run `make -C oracle`"
    ref = Ref()
    rng = np.random.default_rng(20261004)

    # ---------------- C2: liquid 4He, N=64, 81 beads (reference namelist Nb=40) -------------
    S = System(dim=3, Np=64, Nb=40)
    VT, WF = ref.tables(S)
    np.savez_compressed(os.path.join(OUT, "tables_he4_n64.npz"), VTable=VT, LogWF=WF, **sysmeta(S))
    with tempfile.TemporaryDirectory() as td:
        Peq = run_vpi(td, Np=64, Nb=40, Nstep=12)      # stock program, 12 MC steps from seed 1982
    ref.set_system(S)
    Pinit, _ = ref.init(1982)
    Prnd = wrap(Pinit + rng.normal(0, 0.15, Pinit.shape), S.Lbox)   # overlapping beads
    for tag, Path, sig in (("eq", Peq, 0.08), ("rnd", Prnd, 0.3)):
        d = action_cases(ref, S, VT, WF, Path, 1500, rng, sig)
        d.update(energies(ref, S, VT, WF, Path))
        np.savez_compressed(os.path.join(OUT, f"he4_n64_{tag}.npz"), Path=Path, **d, **sysmeta(S))

    # ---------------- table boundary: Lennard-Jones and dipolar tables -----------
    # These potentials are not active code in the reference; parity is pinned at the table
    # boundary by feeding the same externally filled VTable to the reference routines.
    r = (np.arange(1, S.Nmax + 1) - 1) * S.dr
    with np.errstate(all="ignore"):
        lj = np.zeros(S.Nmax + 2)
        lj[1:S.Nmax + 1] = 22.0228 * (1.0 / r ** 6 - 1.0) / r ** 6
        lj[0], lj[S.Nmax + 1] = lj[2], lj[S.Nmax]
        dip = np.zeros(S.Nmax + 2)
        dip[1:S.Nmax + 1] = 1.0 / r ** 3
        dip[0], dip[S.Nmax + 1] = dip[2], dip[S.Nmax]
    for tag, tab in (("lj", lj), ("dipolar", dip)):
        d = action_cases(ref, S, tab, WF, Peq, 600, rng, 0.08)
        d.update(energies(ref, S, tab, WF, Peq))
        np.savez_compressed(os.path.join(OUT, f"he4_n64_table_{tag}.npz"), VTable=tab, **d)

    # ---------------- C3 tables + a few slices of an N=256 worldline -------------------------
    S3 = System(dim=3, Np=256, Nb=80)
    VT3, WF3 = ref.tables(S3)
    np.savez_compressed(os.path.join(OUT, "tables_he4_n256.npz"), VTable=VT3, LogWF=WF3, **sysmeta(S3))

    # ---------------- trap: 1D harmonic oscillator N=2 (config 1) and a 3D trap ---------------
    for tag, kw in (("ho1d_n2", dict(dim=1, Np=2, Nb=10, trap=True, a_ho=[1.0])),
                    ("trap3d_n8", dict(dim=3, Np=8, Nb=6, trap=True, a_ho=[1.0, 1.3, 0.8])),
                    ("pbc2d_n16", dict(dim=2, Np=16, Nb=8, density=0.2))):
        St = System(**kw)
        VTt, WFt = ref.tables(St)
        ref.set_system(St)
        P0, _ = ref.init(7)
        scale = 0.4 if St.trap else 0.25
        Pt = P0 + rng.normal(0, scale, P0.shape)
        if not St.trap:
            Pt = wrap(Pt, St.Lbox[:St.dim])
        d = action_cases(ref, St, VTt, WFt, Pt, 400, rng, 0.3)
        d.update(energies(ref, St, VTt, WFt, Pt))
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), Path=Pt, VTable=VTt, LogWF=WFt, **d, **sysmeta(St))

    # ---------------- RNG: MT19937 stream + polar Box-Muller (random_mod.f90) ------------------
    ref.sgrnd(1982)
    u = np.array([ref.grnd() for _ in range(2000)])
    mti, mt = ref.rng_get_state()
    g = np.array([ref.rangauss(1.0, 0.0) for _ in range(500)])
    mti2, mt2 = ref.rng_get_state()
    np.savez_compressed(os.path.join(OUT, "rng_seed1982.npz"), grnd=u, mti_after=mti, mt_after=mt,
                        rangauss=g, mti_after_gauss=mti2, mt_after_gauss=mt2)

    # ---------------- primitives ----------------------------------------------------------------
    xs = rng.uniform(0.0, S.rcut, 3000)
    prim = {f"interp{o}": np.array([ref.interpolate(o, S.Nmax, S.dr, VT, x) for x in xs]) for o in (0, 1, 2)}
    gf = np.array([[ref_gf(ref, S, o, ib) for ib in (0, 1, 2, 39, 40, 79, 80)] for o in (0, 1)])
    np.savez_compressed(os.path.join(OUT, "primitives_n64.npz"), x=xs, green=gf, **prim)
    print("golden vectors written to", OUT)


def ref_gf(ref, S, opt, ib):
    ref.set_system(S)
    return ref.green_function(opt, ib, 5e-3, 1.2345678901234, -9.87654321)




# ---------------------------------------------------------------------------------------------
# End-to-end runs of the stock reference PROGRAM (oracle/_ref/vpi): its output files are kept
# as data fixtures under tests/golden/vpi_runs/<name>/ for the front-end / sampler tests.
# Next to them driver.npz: the same run through tests/golden/ref_driver.py (the reference's own
# movers and estimators called in the program's schedule, validated here against the program:
# bit-identical final worldline) with 64-bit block energies, per-step energies, final generator
# state, counters and event log.  Runs with `potential` other than aziz2 exist only in driver form:
# the reference compiles ONE pair potential in, the table of another is fed in at the table boundary.
RUNS = {
    # worm sector active: open / close / swap / OBDM with partial waves (Npw = 1), bisection sampling
    "he4_worm_s1982": dict(dim=3, Np=16, Nb=8, seed=1982, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                           Nblock=6, Nstep=25, CWorm="0.5d0", Nobdm=4, Npw=1),
    "he4_worm_s1983": dict(dim=3, Np=16, Nb=8, seed=1983, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                           Nblock=6, Nstep=25, CWorm="0.5d0", Nobdm=4, Npw=1),
    "he4_worm_s1984": dict(dim=3, Np=16, Nb=8, seed=1984, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                           Nblock=6, Nstep=25, CWorm="0.5d0", Nobdm=4, Npw=1),
    # a busier worm (dt = 0.02, CWorm = 0.6): several opens, closes and dozens of accepted swaps; Npw = 2
    "he4_wormbusy_s7": dict(dim=3, Np=16, Nb=8, seed=7, dt="2.0d-2", sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                            Nblock=6, Nstep=20, CWorm="0.6d0", Nobdm=4, Npw=2),
    "he4_wormbusy_s8": dict(dim=3, Np=16, Nb=8, seed=8, dt="2.0d-2", sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                            Nblock=6, Nstep=20, CWorm="0.6d0", Nobdm=4, Npw=2),
    # BASELINE config 1: 1D harmonic oscillator, N=2, 21 beads, staging sampling (swapping=T: quirk Q9)
    "ho1d_n2": dict(dim=1, Np=2, Nb=10, seed=1982, trap="T", a_ho="1.0d0", sampling="sta", Lstag=6, Nlev=2,
                    Nstag=4, Nblock=5, Nstep=40, CWorm="0.3d0", Nobdm=2, Npw=0, dt="1.0d-2"),
    # stock vpi.in (N=64, 65 beads, bis, Nlev=4, Nstag=5, worm on), shortened
    "he4_stock_short": dict(dim=3, Np=64, Nb=32, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                            Nblock=2, Nstep=6, CWorm="0.5d0", Nobdm=10, Npw=0),
    # pure diagonal PIGS with bisection sampling (CWorm = 0): the device-resident sampler's scope
    "he4_bis_cworm0_s1982": dict(dim=3, Np=32, Nb=16, seed=1982, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                                 Nblock=3, Nstep=8, CWorm="0.0d0", Nobdm=0, Npw=0),
    "he4_bis_cworm0_s1983": dict(dim=3, Np=32, Nb=16, seed=1983, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                                 Nblock=3, Nstep=8, CWorm="0.0d0", Nobdm=0, Npw=0),
    "trap2d_bis_cworm0": dict(dim=2, Np=6, Nb=8, seed=11, trap="T", a_ho="1.0d0 1.3d0", sampling="bis", Lstag=4,
                              Nlev=2, Nstag=2, Nblock=2, Nstep=10, CWorm="0.0d0", Nobdm=0, Npw=0, dt="1.0d-2"),
    # CWorm = 0 (quirk Q11): an open proposal is generated and always rejected
    "he4_cworm0": dict(dim=2, Np=9, Nb=6, seed=7, density="0.25d0", sampling="sta", Lstag=4, Nlev=2, Nstag=2,
                       Nblock=3, Nstep=20, CWorm="0.0d0", Nobdm=0, Npw=0),
    # Lstag > Nb with CWorm = 0 (and six bisection levels): the never-accepted open proposal of quirk Q11 then indexes
    # beads below 0 in the reference (it reads and restores whatever lies there: harmless, deterministic); only its
    # random numbers matter.  Program files only: the driver would run the reference's OpenChain on a numpy array.
    "lstag_gt_nb_bis6": dict(dim=3, Np=20, Nb=40, density="0.3d0", seed=77, sampling="bis", Lstag=50, Nlev=6, Nstag=2,
                             Nblock=2, Nstep=6, CWorm="0.0d0", Nobdm=0, Npw=0, driver=False),
    "lstag_gt_nb_sta": dict(dim=2, Np=12, Nb=10, density="0.1d0", seed=77, sampling="sta", Lstag=15, Nlev=2, Nstag=2,
                            Nblock=2, Nstep=6, CWorm="0.0d0", Nobdm=0, Npw=0, driver=False),
    # wf_table = F, the reference's default: the McMillan trial function evaluated analytically (system_mod.f90:38-66)
    "he4_wf_analytic": dict(dim=3, Np=16, Nb=8, seed=1982, sampling="bis", Lstag=8, Nlev=3, Nstag=3,
                            Nblock=4, Nstep=20, CWorm="0.5d0", Nobdm=4, Npw=1, wf_table="F"),
    # Nlev = 1, the reference's DEFAULT (vpi_mod.f90:47): the head / tail moves still bisect 2^2 beads (`Nlev' = int((level-1)*grnd())+2`,
    # vpi_mod.f90:1023,1209), only Bisection itself works on 2^1.  Round 3's sampler fuzz found K6 clamping nl to Nlev there.
    "he4_nlev1": dict(dim=3, Np=16, Nb=8, seed=1982, sampling="bis", Lstag=6, Nlev=1, Nstag=3,
                      Nblock=4, Nstep=15, CWorm="3.0d0", Nobdm=3, Npw=1, dt="2.0d-2"),
    # ---- BASELINE sizes ------------------------------------------------------------------------
    # C3: liquid 4He N=256, 161 beads, stock schedule, CWorm = 0
    "c3_n256_s1982": dict(dim=3, Np=256, Nb=80, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                          Nblock=1, Nstep=3, CWorm="0.0d0", Nobdm=0, Npw=0, big=True),
    "c3_n256_s1983": dict(dim=3, Np=256, Nb=80, seed=1983, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                          Nblock=1, Nstep=3, CWorm="0.0d0", Nobdm=0, Npw=0, big=True),
    # C5: N=256, 321 beads, worm sector with swaps and partial waves.  The Aziz form is what the program runs;
    # the dipolar form (BASELINE config 5's potential) goes through the driver with the r^-3 table.
    # CWorm = 20: the first open attempt of so short a run (step 4 for this seed) is accepted
    "c5_n256_aziz_s1982": dict(dim=3, Np=256, Nb=160, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                               Nblock=2, Nstep=4, CWorm="20.0d0", Nobdm=10, Npw=2, big=True),
    "c5_n256_dipolar_s1982": dict(dim=3, Np=256, Nb=160, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                                  Nblock=2, Nstep=4, CWorm="20.0d0", Nobdm=10, Npw=2, big=True, potential="dipolar"),
    # ---- BASELINE sizes, LONG: past the cold start.  Every run starts from the reference's own `init`
    # (so no start state has to be stored: the test replays the warm-up too) and goes on for 50-60 MC steps: at C3 the
    # first 30 steps are the warm-up during which acceptance and bead spread settle (accepted moves per sweep rise from
    # ~1 450 to ~1 800), the last 20 are in the regime bench.py times.  C5 with the STOCK CWorm = 0.5: the Aziz run
    # contains 4 accepted opens and 4 accepted closes (no swap is accepted in liquid 4He at this dt within 200 steps:
    # 0 of 290 tries), the dipolar run 1 open and >100 accepted swaps (and no close within 200 steps: 0 of 106 tries) --
    # between them every worm event at N=256, 321 beads.  driver.npz also holds the state at the end of every block.
    "c3_n256_long_s1982": dict(dim=3, Np=256, Nb=80, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                               Nblock=5, Nstep=10, CWorm="0.0d0", Nobdm=0, Npw=0, big=True),
    "c3_n256_long_s1983": dict(dim=3, Np=256, Nb=80, seed=1983, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                               Nblock=5, Nstep=10, CWorm="0.0d0", Nobdm=0, Npw=0, big=True),
    "c5_n256_aziz_long_s1982": dict(dim=3, Np=256, Nb=160, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                                    Nblock=6, Nstep=10, CWorm="0.5d0", Nobdm=10, Npw=2, big=True),
    "c5_n256_dipolar_long_s1982": dict(dim=3, Np=256, Nb=160, seed=1982, sampling="bis", Lstag=32, Nlev=4, Nstag=5,
                                       Nblock=6, Nstep=10, CWorm="0.5d0", Nobdm=10, Npw=2, big=True, potential="dipolar"),
}
RUN_FILES = ["e_vpi.out", "et_vpi.out", "gr_vpi.out", "sk_vpi.out", "nr_vpi.out", "fort.99"]
BIG_STRIDE = 8           # fixtures of the N=256 runs keep every 8th bead + SHA-256 + per-bead sums


def _fnum(x):
    return float(str(x).replace("d", "e"))


def drive_run(ref, kw, VT=None):
    """The run `kw` through ref_driver.drive(); returns (System, result)."""
    import ref_driver as rd
    trap = kw.get("trap", "F") == "T"
    a_ho = [float(t.replace("d", "e")) for t in str(kw.get("a_ho", "1.0d0")).split()]
    density = _fnum(kw.get("density", "0.365d0"))
    if trap:
        from pathintegralgroundstate_amd import SystemConfig
        density = SystemConfig(dim=kw["dim"], Np=kw["Np"], Nb=kw["Nb"], trap=True, a_ho=a_ho).density   # vpi.f90:82-93
    S = System(dim=kw["dim"], Np=kw["Np"], Nb=kw["Nb"], density=density,
               dt=_fnum(kw.get("dt", "5.0d-3")), trap=trap, a_ho=a_ho if trap else None,
               CWorm=_fnum(kw["CWorm"]), Npw=kw["Npw"], Nbin=100, wf_table=kw.get("wf_table", "T") == "T")
    VTr, WF = ref.tables(S)
    if VT is None:
        VT = VTr
    res = rd.drive(ref, S, VT, WF, kw["seed"], Nblock=kw["Nblock"], Nstep=kw["Nstep"], sampling=kw["sampling"],
                   Lstag=kw["Lstag"], Nlev=kw["Nlev"], Nstag=kw["Nstag"], CMFreq=1, delta_cm=0.12,
                   CWorm=_fnum(kw["CWorm"]), Nobdm=kw["Nobdm"], swapping=True, Nk=50)
    return S, res


def make_runs(only=None):
    import shutil
    import ref_driver as rd
    from pathintegralgroundstate_amd import SystemConfig, api
    ref = Ref()
    base = os.path.join(OUT, "vpi_runs")
    for name, kw in RUNS.items():
        if only and name not in only:
            continue
        kw = dict(kw)
        big = kw.pop("big", False)
        potential = kw.pop("potential", "aziz2")
        with_driver = kw.pop("driver", True)
        dst = os.path.join(base, name)
        os.makedirs(dst, exist_ok=True)
        P = None
        with tempfile.TemporaryDirectory() as td:
            if potential == "aziz2":
                P = run_vpi(td, **kw)
                for f in RUN_FILES:
                    if os.path.exists(os.path.join(td, f)):
                        shutil.copy(os.path.join(td, f), os.path.join(dst, f))
            else:
                p = dict(dim=3, Np=64, density="0.365d0", trap="F", dt="5.0d-3", Nb=40, seed=1982, sampling="bis",
                         Lstag=16, Nlev=4, Nstag=5, Nblock=1, Nstep=10, CWorm="0.0d0", Nobdm=0, Npw=0, a_ho="1.0d0",
                         wf_table="T")
                p.update(kw)
                with open(os.path.join(td, "vpi.in"), "w") as f:
                    f.write(VPI_IN.format(**p))
            shutil.copy(os.path.join(td, "vpi.in"), os.path.join(dst, "vpi.in"))
        if not with_driver:
            np.savez_compressed(os.path.join(dst, "final_worldline.npz"), Path=P)
            print(name, "program files only", flush=True)
            continue
        VT = None
        if potential != "aziz2":
            # the table of a potential the reference does not compile in: filled by the product's own host-side
            # table builder (pigs_tables.cpp) -- input data for both sides of the comparison
            cfg = SystemConfig.from_namelists(open(os.path.join(dst, "vpi.in")).read())
            VT, _ = api.build_tables(cfg, potential)
        S, res = drive_run(ref, kw, VT)
        if P is not None:
            rd.validate_against_program(res, P, np.loadtxt(os.path.join(dst, "e_vpi.out")).reshape(-1, 4)
                                        if os.path.getsize(os.path.join(dst, "e_vpi.out")) else np.zeros((0, 4)),
                                        np.loadtxt(os.path.join(dst, "et_vpi.out")).reshape(-1, 4)
                                        if os.path.getsize(os.path.join(dst, "et_vpi.out")) else np.zeros((0, 4)))
        c = rd.compact(res, BIG_STRIDE if big else 1)
        if not big:
            np.savez_compressed(os.path.join(dst, "final_worldline.npz"), Path=res["Path"])
            del c["Path_sub"]
        elif os.path.exists(os.path.join(dst, "final_worldline.npz")):
            os.remove(os.path.join(dst, "final_worldline.npz"))
        c["potential"] = np.array(potential)
        np.savez_compressed(os.path.join(dst, "driver.npz"), **c)
        print(name, "counters", dict(zip(rd.COUNTER_NAMES, res["counters"].tolist())), "events", len(res["events"]),
              flush=True)
    print("reference program runs written to", base)


def make_resume_fixture():
    """Reference run A (2 blocks) leaves checkpoint.dat + rand_state; reference run B resumes from them
    for 2 more blocks (quirk Q10: it reads the FIRST rand_state record, i.e. the state after block 1)."""
    import shutil
    dst = os.path.join(OUT, "vpi_runs", "he4_resume")
    os.makedirs(dst, exist_ok=True)
    kw = dict(dim=3, Np=16, Nb=8, seed=1982, sampling="bis", Lstag=8, Nlev=3, Nstag=3, Nblock=2, Nstep=15,
              CWorm="0.5d0", Nobdm=4)
    with tempfile.TemporaryDirectory() as td:
        run_vpi(td, **kw)
        shutil.copy(os.path.join(td, "checkpoint.dat"), os.path.join(dst, "checkpoint.dat"))
        shutil.copy(os.path.join(td, "rand_state"), os.path.join(dst, "rand_state"))
        txt = open(os.path.join(td, "vpi.in")).read().replace("resume = F", "resume = T")
    # run B in a FRESH directory holding only the two restart files (stale output files of run A
    # would otherwise shine through: the reference does not truncate them)
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(os.path.join(dst, "checkpoint.dat"), os.path.join(td, "checkpoint.dat"))
        shutil.copy(os.path.join(dst, "rand_state"), os.path.join(td, "rand_state"))
        for f in RUN_FILES:
            if os.path.exists(os.path.join(dst, f)):
                os.remove(os.path.join(dst, f))
        with open(os.path.join(td, "vpi.in"), "w") as f:
            f.write(txt)
        with open(os.path.join(td, "vpi.in")) as fin, open(os.path.join(td, "stdout2"), "w") as fo:
            subprocess.run([REF_VPI], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=td, check=False, timeout=600)
        shutil.copy(os.path.join(td, "vpi.in"), os.path.join(dst, "vpi.in"))
        for f in RUN_FILES:
            if os.path.exists(os.path.join(td, f)):
                shutil.copy(os.path.join(td, f), os.path.join(dst, f))
        # the worldline the resumed run ends on (its checkpoint.dat), as a fixture of its own
        dim, Np, M = 3, 16, 17
        with open(os.path.join(td, "checkpoint.dat")) as f:
            lines = f.read().split("\n")
        vals = np.array([[float(x) for x in ln.split()] for ln in lines[3:3 + Np * M]])
        np.savez_compressed(os.path.join(dst, "final_worldline.npz"),
                            Path=np.ascontiguousarray(vals.reshape(Np, M, dim).transpose(1, 0, 2)))
    print("resume fixture written to", dst)


def make_crystal_fixture():
    """crystal = T (reference vpi.f90:99-107, vpi_mod.f90:218-230): particle number, box and density come from
    config_ini.in, whose remaining lines are the start configuration (every bead of a particle on its lattice site).
    A 3x3x3 simple-cubic lattice with a seeded jitter at density 0.45; worm sector on.  Reference PROGRAM files +
    its final worldline; config_ini.in is part of the fixture (it is input data)."""
    import shutil
    dst = os.path.join(OUT, "vpi_runs", "he4_crystal")
    os.makedirs(dst, exist_ok=True)
    n, dens = 3, 0.45
    Np = n ** 3
    L = (Np / dens) ** (1.0 / 3.0)
    rng = np.random.default_rng(27)
    sites = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) * (L / n) - L / 2
    sites = sites + rng.normal(0, 0.03, sites.shape)
    cfg = "%d\n%s\n%.17g\n" % (Np, " ".join("%.17g" % L for _ in range(3)), dens)
    cfg += "".join(" ".join("%.17g" % x for x in r) + "\n" for r in sites)
    # the namelist's Np and density are deliberately different: config_ini.in wins (vpi.f90:103-105)
    kw = dict(dim=3, Np=8, Nb=8, seed=1982, sampling="bis", Lstag=8, Nlev=3, Nstag=3, Nblock=4, Nstep=15,
              CWorm="0.5d0", Nobdm=4, Npw=1, density="0.2d0")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "config_ini.in"), "w") as f:
            f.write(cfg)
        p = dict(dim=3, Np=64, density="0.365d0", trap="F", dt="5.0d-3", Nb=40, seed=1982, sampling="bis", Lstag=16, Nlev=4,
                 Nstag=5, Nblock=1, Nstep=10, CWorm="0.0d0", Nobdm=0, Npw=0, a_ho="1.0d0", wf_table="T")
        p.update(kw)
        txt = VPI_IN.format(**p).replace("trap = F", "crystal = T, trap = F")
        with open(os.path.join(td, "vpi.in"), "w") as f:
            f.write(txt)
        with open(os.path.join(td, "vpi.in")) as fin, open(os.path.join(td, "stdout"), "w") as fo:
            subprocess.run([REF_VPI], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=td, check=False, timeout=600)
        M = 2 * kw["Nb"] + 1
        with open(os.path.join(td, "checkpoint.dat")) as f:
            lines = f.read().split("\n")
        vals = np.array([[float(x) for x in ln.split()] for ln in lines[3:3 + Np * M]])
        P = np.ascontiguousarray(vals.reshape(Np, M, 3).transpose(1, 0, 2))
        for f in RUN_FILES + ["vpi.in", "config_ini.in"]:
            if os.path.exists(os.path.join(td, f)):
                shutil.copy(os.path.join(td, f), os.path.join(dst, f))
    np.savez_compressed(os.path.join(dst, "final_worldline.npz"), Path=P)
    print("crystal fixture written to", dst, "files", sorted(os.listdir(dst)))


if __name__ == "__main__":
    sys.path.insert(0, OUT)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "vectors"):
        main()
    if what in ("all", "runs"):
        make_runs(set(sys.argv[2:]) or None)
    if what in ("all", "resume"):
        make_resume_fixture()
    if what in ("all", "crystal"):
        make_crystal_fixture()
