#!/usr/bin/env python3
"""Soak: the device-resident sampler against the host-driven one (host libm, bit-identical to the reference wherever a
fixture exists) through the front end, same input, many steps: final worldlines must be identical bit for bit, the
permutation and OBDM files byte for byte.  Exercises the device log (pigs_log_host.h) on ~1e7 Box-Muller radii drawn by
real trajectories.   usage (GPU box): python scripts/k6_vs_host_soak.py [Nstep_total]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pathintegralgroundstate_amd", "host", "pigs_vpi")
SHIM_EXE = os.path.join(ROOT, "tests", "shim", "_build", "pigs_vpi")
nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
# (sampling, &system fields, &samp fields, CWorm [+ obdm fields], extra: dict(dt, wavefun, extpot, gpu))
CASES = [("bis", "dim = 3, Np = 30, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 3", "0.4d0", {}),
         ("sta", "dim = 2, Np = 21, density = 0.1d0", "Nb = 12, Lstag = 6, Nlev = 2", "0.5d0", {}),
         ("bis", "dim = 3, Np = 30, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 3", "0.0d0", {}),
         # 1D harmonic trap, N = 2, staging movers + worm + swap (BASELINE config 1's shape; quirk Q9: swapping = T)
         ("sta", "dim = 1, Np = 2, density = 0.1d0, trap = T", "Nb = 10, Lstag = 6, Nlev = 2", "0.3d0", {"extpot": "a_ho = 1.0d0"}),
         # 2D trap, bisection, diagonal sector only
         ("bis", "dim = 2, Np = 6, density = 0.1d0, trap = T", "Nb = 8, Lstag = 4, Nlev = 2", "0.0d0", {"extpot": "a_ho = 1.0d0 1.3d0"}),
         # the reference's default: analytic trial function (wf_table = F)
         ("bis", "dim = 3, Np = 16, density = 0.365d0", "Nb = 8, Lstag = 8, Nlev = 3", "0.5d0", {"wavefun": "Nmax = 10000, wf_table = F, v_table = T"}),
         # five bisection levels: the stage-machine kernel (pigs_diag.hip)
         ("bis", "dim = 3, Np = 20, density = 0.3d0", "Nb = 20, Lstag = 8, Nlev = 5", "0.4d0", {}),
         # three 64-partner passes per bead, two partial waves, more OBDM iterations
         ("bis", "dim = 3, Np = 130, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 4", "0.4d0, Npw = 2, Nobdm = 5", {"steps": 0.25}),
         ("bis", "dim = 2, Np = 37, density = 0.06d0", "Nb = 12, Lstag = 6, Nlev = 3", "0.4d0", {}),
         # walkers sharded over two contexts on this one GPU (one host thread each, one all-reduce per block)
         ("bis", "dim = 3, Np = 30, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 3", "0.4d0", {"gpu": "n_gpus = 2, same_device = T, "})]
# BASELINE sizes (slow on the host-driven side: 0.7 / 1.3 s per MC step): indices 10-12, not part of the default list
BIG = [("bis", "dim = 3, Np = 256, density = 0.365d0", "Nb = 80, Lstag = 32, Nlev = 4", "0.0d0", {"big": 1, "walkers": 2, "dt": "5.0d-3"}),
       ("bis", "dim = 3, Np = 256, density = 0.365d0", "Nb = 160, Lstag = 32, Nlev = 4", "0.5d0, Npw = 2, Nobdm = 10", {"big": 1, "walkers": 2, "dt": "5.0d-3"}),
       ("bis", "dim = 3, Np = 256, density = 0.365d0", "Nb = 160, Lstag = 32, Nlev = 4", "0.5d0, Npw = 2, Nobdm = 10",
        {"big": 1, "walkers": 2, "dt": "5.0d-3", "gpu": "potential = 'dipolar', "})]
# more walkers than CUs: the sweep kernel's 4-wave form (three workgroups per CU), TranslateChain with one workgroup per walker
BIG.append(("bis", "dim = 3, Np = 16, density = 0.3d0", "Nb = 12, Lstag = 6, Nlev = 3", "0.4d0", {"walkers": 300, "steps": 0.1}))
# 64 lock-step walkers with busy worms (large staged batches and commit lists), staging and bisection sampling
BIG.append(("sta", "dim = 3, Np = 24, density = 0.3d0", "Nb = 14, Lstag = 8, Nlev = 3", "0.6d0", {"walkers": 64, "steps": 0.5}))
BIG.append(("bis", "dim = 3, Np = 24, density = 0.3d0", "Nb = 16, Lstag = 8, Nlev = 4", "0.6d0, Nobdm = 6", {"walkers": 64, "steps": 0.5}))
# one dimension with periodic boundaries; a 3D trap with bisection sampling and a busy worm; no TranslateChain at all (CMFreq
# beyond the run) and every-step OBDM with no swaps (swapping = F needs CWorm = 0 in the reference: quirk Q9 -- so CWorm = 0 here)
BIG.append(("bis", "dim = 1, Np = 9, density = 0.5d0", "Nb = 12, Lstag = 6, Nlev = 3", "0.4d0", {}))
BIG.append(("bis", "dim = 3, Np = 7, density = 0.1d0, trap = T", "Nb = 8, Lstag = 6, Nlev = 3", "0.5d0", {"extpot": "a_ho = 1.0d0 1.2d0 0.9d0"}))
BIG.append(("sta", "dim = 3, Np = 20, density = 0.3d0", "Nb = 32, Lstag = 32, Nlev = 3", "0.5d0, Nobdm = 12", {"steps": 0.5}))
if os.environ.get("CASE"):
    CASES = CASES + BIG
    CASES = [CASES[int(x)] for x in os.environ["CASE"].split(",")]
ok = True
for sampling, system, samp, cworm, extra in CASES:
    nst = int(nstep * extra.get("steps", 1.0))
    nst = nst // 100 * 100 if nst >= 100 else nst
    NWK = extra.get("walkers", 4)
    inp = f"""&system
 {system}{"" if "trap" in system else ", trap = F"}
/
&samp
 resume = F, dt = {extra.get('dt', '1.0d-2')}, {samp}, seed = 4242, delta_cm = {'0.12d0, CMFreq = 1' if extra.get('big') else '0.2d0, CMFreq = 2'},
 sampling = '{sampling}', Nstag = {5 if extra.get('big') else 2}, Nblock = {nst // 100 if nst >= 100 else 1}, Nstep = {100 if nst >= 100 else nst}, Nbin = 50, Nk = 10
/
&obdm
 swapping = T, {"" if "Nobdm" in cworm else "Nobdm = 3, "}{"" if "Npw" in cworm else "Npw = 1, "}CWorm = {cworm}
/
&wavefun
 {extra.get("wavefun", "Nmax = 10000, wf_table = T, v_table = T" if extra.get("big") else "Nmax = 4000, wf_table = T, v_table = T")}
/
&jastrow
 Rm = {"1.20d0" if extra.get("big") else "1.10d0"}
/
&extpot
 {extra.get("extpot", "a_ho = 1.0d0")}
/
"""
    res = {}
    arms = "FT" + ("C" if os.environ.get("SHIM") else "")     # C: the CPU twin (host sampler over the scalar oracle = the reference's arithmetic)
    for dev in arms:
        d = tempfile.mkdtemp()
        with open(os.path.join(d, "vpi.in"), "w") as f:
            f.write(inp + f"&gpu\n {extra.get('gpu', '')}n_walkers = {NWK}, device = 0, device_sampler = {'F' if dev == 'C' else dev}, checkpointing = F\n/\n")
        t0 = time.time()
        with open(os.path.join(d, "vpi.in")) as fin, open(os.path.join(d, "out.txt"), "w") as fo:
            r = subprocess.run([SHIM_EXE if dev == "C" else EXE], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=d, timeout=3000)
        assert r.returncode == 0, open(os.path.join(d, "out.txt")).read()[-2000:]
        res[dev] = (d, time.time() - t0)
    a, b = res["F"][0], res["T"][0]
    wa, wb = np.fromfile(os.path.join(a, "worldlines_final.bin")), np.fromfile(os.path.join(b, "worldlines_final.bin"))
    same = np.array_equal(wa.view(np.uint64), wb.view(np.uint64))
    files = all(open(os.path.join(a, f"{n}.w{w:04d}.out"), "rb").read() == open(os.path.join(b, f"{n}.w{w:04d}.out"), "rb").read()
                for w in range(NWK) for n in ("nr_vpi", "perm_vpi") if os.path.exists(os.path.join(a, f"{n}.w{w:04d}.out")))
    if not same:                     # where: first block whose 64-bit energies differ, per walker (blocks of 100 steps)
        for w in range(NWK):
            la = open(os.path.join(a, f"e_vpi.w{w:04d}.hex")).read().splitlines()
            lb = open(os.path.join(b, f"e_vpi.w{w:04d}.hex")).read().splitlines()
            first = next((i for i, (x, y) in enumerate(zip(la, lb)) if x != y), None)
            print(f"   walker {w}: first differing diagonal block line {first} of {len(la)}/{len(lb)}: "
                  f"{la[first].split()[0] if first is not None else '-'}", flush=True)
    print(f"{sampling} [{system}; {samp}; CWorm = {cworm}{'; ' + str(extra) if extra else ''}]: {nst} MC steps x {NWK} walkers: worldlines bit-identical = {same}, OBDM / permutation files identical = {files}"
          f"  (host-driven {res['F'][1]:.1f} s, device {res['T'][1]:.1f} s)", flush=True)
    if "C" in res:
        wc = np.fromfile(os.path.join(res["C"][0], "worldlines_final.bin"))
        same_c = np.array_equal(wa.view(np.uint64), wc.view(np.uint64))
        print(f"   CPU twin (reference arithmetic, {res['C'][1]:.1f} s) vs the GPU runs: worldlines bit-identical = {same_c}", flush=True)
        ok = ok and same_c
    ok = ok and same and files
sys.exit(0 if ok else 1)
