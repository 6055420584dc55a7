"""End-to-end on the MI355X: the Fortran front end (host/pigs_vpi, linked against libpigs_hip.so)
against the output files of the reference PROGRAM (tests/golden/vpi_runs).

Identical-seed parity (BASELINE.json north star): the worldline depends only on the random
stream and the accept/reject decisions, so with identical decisions the FINAL WORLDLINE IS
BIT-IDENTICAL to the reference's although every Delta S was summed in a different order on the GPU;
block energies (E, K, V, Et, Kt, Vt per particle) agree to 1e-10 relative."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from helpers import (MIXED_TOL, block_energy_errors, check_worldline_vs_driver, driver_blocks,
                     fold_maxnorm, read_hex_blocks, same_bits)
from pathintegralgroundstate_amd import SystemConfig

pytestmark = pytest.mark.gpu
RUNS = os.path.join(GOLDEN, "vpi_runs")
HOST = os.path.join(ROOT, "pathintegralgroundstate_amd", "host")


@pytest.fixture(scope="module")
def exe(gpu_lib):
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "pigs_vpi")


def _run(exe, txt, wd, env=None, extra_files=()):
    import shutil
    for f in extra_files:
        shutil.copy(f, os.path.join(wd, os.path.basename(f)))
    with open(os.path.join(wd, "vpi.in"), "w") as f:
        f.write(txt)
    with open(os.path.join(wd, "vpi.in")) as fin, open(os.path.join(wd, "stdout.txt"), "w") as fo:
        r = subprocess.run([exe], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=wd, timeout=900, env=env)
    assert r.returncode == 0, open(os.path.join(wd, "stdout.txt")).read()[-3000:]


def _close(mine, ref, rel=1e-10):
    """Text files of the reference program (10 significant digits): used for the histogram files only; block
    energies are compared as 64-bit values through _hex_close."""
    a, b = np.atleast_2d(np.loadtxt(mine)), np.atleast_2d(np.loadtxt(ref))
    assert a.shape == b.shape, (a.shape, b.shape)
    # files carry 10 significant digits: allow one unit in the last printed place on top of rel
    return np.all(np.abs(a - b) <= rel * np.abs(b) + 1.01e-9 * np.abs(b))


def _hex_close(hexfile, src, rel=1e-10, mixed=1e-10, nblocks=None):
    """Block energies E K V Et Kt Vt (per particle) of e_vpi*.hex against the 64-bit values of the reference's own
    estimators run in the program's schedule (driver.npz): 1e-10 relative (helpers.block_energy_errors), no printing
    floor.  `mixed` is the bound for the E, K columns: 1e-10 where the worldline is bit-identical (host-driven
    sampler), helpers.MIXED_TOL for the device-resident sampler (see there)."""
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    blocks, rows = read_hex_blocks(hexfile)
    wb, wrows = driver_blocks(drv)
    if nblocks is not None:                                 # a run of the first `nblocks` blocks of a longer reference run
        keep = wb <= nblocks
        wb, wrows = wb[keep], wrows[keep]
    assert np.array_equal(blocks, wb), (blocks, wb)
    if len(wb) == 0:
        return True
    em, er = block_energy_errors(rows, wrows)
    assert np.all(er <= rel), er.max()
    assert np.all(em <= mixed), em.max()
    return True


def _lbox(src):
    cfg = SystemConfig.from_namelists(open(os.path.join(src, "vpi.in")).read())
    return cfg.Lbox, cfg.trap


@pytest.mark.parametrize("name", ["he4_worm_s1982", "ho1d_n2", "he4_stock_short", "he4_cworm0", "he4_wormbusy_s7",
                                  "he4_wormbusy_s8", "he4_wf_analytic", "he4_nlev1"])
def test_gpu_front_end_matches_reference_program(exe, name, tmp_path):
    src = os.path.join(RUNS, name)
    _run(exe, open(os.path.join(src, "vpi.in")).read() + "&gpu\n device_sampler = F\n/\n", str(tmp_path))     # the host-driven sampler
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want), "trajectory diverged from the reference (a decision flipped)"
    assert _hex_close(tmp_path / "e_vpi.hex", src)
    # histograms depend on the worldline only: identical files
    for f in ("gr_vpi.out", "sk_vpi.out", "nr_vpi.out"):
        if os.path.exists(os.path.join(src, f)):
            assert open(os.path.join(src, f), "rb").read() == open(tmp_path / f, "rb").read(), f
    assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


def test_gpu_front_end_exact_term_kernel(exe, tmp_path):
    """&gpu k1_variant = 2: the Delta-S kernel that keeps the reference's rounding of every term gives the same
    trajectory (bit-identical final worldline) as the default short arithmetic and the reference program."""
    src = os.path.join(RUNS, "he4_worm_s1982")
    _run(exe, open(os.path.join(src, "vpi.in")).read() + "&gpu\n n_walkers = 1, device = 0, device_sampler = F, k1_variant = 2\n/\n", str(tmp_path))
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    assert _hex_close(tmp_path / "e_vpi.hex", src)


@pytest.mark.parametrize("name", ["c3_n256_s1982", "c5_n256_aziz_s1982", "c5_n256_dipolar_s1982"])
def test_gpu_front_end_at_baseline_sizes(exe, name, tmp_path):
    """BASELINE configs 3 and 5 through the host-driven sampler on the MI355X (K1 batches of 1..161 items, four
    partner passes per item): final worldline bit-identical to the reference (SHA-256 over all coordinates), block
    energies 1e-10, the program's files byte for byte where the program can run the input (Aziz)."""
    src = os.path.join(RUNS, name)
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    pot = str(drv["potential"])
    _run(exe, open(os.path.join(src, "vpi.in")).read() +
         f"&gpu\n n_walkers = 1, device = 0, device_sampler = F, potential = '{pot}', checkpointing = F\n/\n", str(tmp_path))
    shape = tuple(int(x) for x in drv["Path_shape"])
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(shape)
    check_worldline_vs_driver(got, drv, None, tol=0.0)
    assert _hex_close(tmp_path / "e_vpi.hex", src)
    for f in ("gr_vpi.out", "sk_vpi.out", "nr_vpi.out"):
        if os.path.exists(os.path.join(src, f)):
            assert open(os.path.join(src, f), "rb").read() == open(tmp_path / f, "rb").read(), f
    if os.path.exists(os.path.join(src, "fort.99")):
        assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


@pytest.mark.parametrize("name", ["c3_n256_s1982", "c5_n256_aziz_s1982", "c5_n256_dipolar_s1982"])
def test_gpu_front_end_device_sampler_at_baseline_sizes(exe, name, tmp_path):
    """The same inputs with &gpu device_sampler = T (K6 + K2/K3/K4/K7 estimators, files as the reference)."""
    src = os.path.join(RUNS, name)
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    pot = str(drv["potential"])
    _run(exe, open(os.path.join(src, "vpi.in")).read() +
         f"&gpu\n n_walkers = 1, device = 0, device_sampler = T, potential = '{pot}', checkpointing = F\n/\n", str(tmp_path))
    assert "using the host-driven sampler" not in open(tmp_path / "stdout.txt").read()
    shape = tuple(int(x) for x in drv["Path_shape"])
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(shape)
    Lbox, trap = _lbox(src)
    check_worldline_vs_driver(got, drv, Lbox, trap, tol=0.0)           # bit-identical: SHA-256 of every coordinate
    assert _hex_close(tmp_path / "e_vpi.hex", src, mixed=MIXED_TOL)
    if os.path.exists(os.path.join(src, "nr_vpi.out")):
        assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
    if os.path.exists(os.path.join(src, "gr_vpi.out")):
        assert _close(tmp_path / "gr_vpi.out", os.path.join(src, "gr_vpi.out"), rel=1e-9)
    if os.path.exists(os.path.join(src, "fort.99")):
        assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


def test_gpu_lockstep_walkers(exe, tmp_path):
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read()
    _run(exe, base + "&gpu\n n_walkers = 3, device = 0, device_sampler = F\n/\n", str(tmp_path))
    got = np.fromfile(tmp_path / "worldlines_final.bin")
    for w, seed in enumerate((1982, 1983, 1984)):
        src = os.path.join(RUNS, f"he4_worm_s{seed}")
        want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
        assert same_bits(got.reshape((3,) + want.shape)[w], want), w
        assert _hex_close(tmp_path / f"e_vpi.w{w:04d}.hex", src)


def test_gpu_front_end_with_device_resident_sampler(exe, tmp_path):
    """&gpu device_sampler = T: the whole MC step on the GPU (K6); two walkers == the reference runs
    with seeds 1982 and 1983 (worldlines to rounding, energies and histograms as printed)."""
    base = open(os.path.join(RUNS, "he4_bis_cworm0_s1982", "vpi.in")).read()
    _run(exe, base + "&gpu\n n_walkers = 2, device = 0, device_sampler = T\n/\n", str(tmp_path))
    got = np.fromfile(tmp_path / "worldlines_final.bin")
    for w, seed in enumerate((1982, 1983)):
        src = os.path.join(RUNS, f"he4_bis_cworm0_s{seed}")
        want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
        assert same_bits(got.reshape((2,) + want.shape)[w], want), w
        assert _hex_close(tmp_path / f"e_vpi.w{w:04d}.hex", src, mixed=MIXED_TOL)
        assert _close(tmp_path / f"gr_vpi.w{w:04d}.out", os.path.join(src, "gr_vpi.out"), rel=1e-9)
        assert _close(tmp_path / f"sk_vpi.w{w:04d}.out", os.path.join(src, "sk_vpi.out"), rel=1e-8)


def test_gpu_front_end_device_sampler_worm_sector(exe, tmp_path):
    """device_sampler = T with CWorm > 0, swapping = T: open / close / half-chain moves / swap / OBDM all on the
    GPU; three walkers == the reference runs with seeds 1982-1984.  The permutation histogram (replayed on
    the host from the kernel's event log) and the OBDM file must equal the reference's exactly."""
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read()
    _run(exe, base + "&gpu\n n_walkers = 3, device = 0, device_sampler = T\n/\n", str(tmp_path))
    got = np.fromfile(tmp_path / "worldlines_final.bin")
    for w, seed in enumerate((1982, 1983, 1984)):
        src = os.path.join(RUNS, f"he4_worm_s{seed}")
        want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
        assert same_bits(got.reshape((3,) + want.shape)[w], want), w
        assert _hex_close(tmp_path / f"e_vpi.w{w:04d}.hex", src, mixed=MIXED_TOL)
        assert _close(tmp_path / f"gr_vpi.w{w:04d}.out", os.path.join(src, "gr_vpi.out"), rel=1e-9)
        assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / f"nr_vpi.w{w:04d}.out", "rb").read(), w
        assert open(tmp_path / f"perm_vpi.w{w:04d}.out").read().split() == open(os.path.join(src, "fort.99")).read().split(), w


def test_gpu_front_end_device_sampler_stock_input(exe, tmp_path):
    """device_sampler = T on the reference's stock vpi.in (shortened): N=64, Nb=32, Lstag=32 -- open/close and
    half-chain staging proposals of up to 96 Gaussians, several look-ahead refills of the random stream per
    stage -- against the reference program's files."""
    src = os.path.join(RUNS, "he4_stock_short")
    _run(exe, open(os.path.join(src, "vpi.in")).read() + "&gpu\n n_walkers = 1, device = 0, device_sampler = T\n/\n",
         str(tmp_path))
    assert "using the host-driven sampler" not in open(tmp_path / "stdout.txt").read()
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    assert _hex_close(tmp_path / "e_vpi.hex", src, mixed=MIXED_TOL)
    assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
    assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


@pytest.mark.parametrize("name", ["he4_cworm0", "ho1d_n2"])
def test_gpu_front_end_device_sampler_staging_movers(exe, name, tmp_path):
    """device_sampler = T with sampling = 'sta' (MoveHead, MoveTail, Staging on the GPU): the 2D periodic run with
    CWorm = 0 and the 1D trapped N=2 run with worm + swap (quirk Q9) against the reference program's files."""
    src = os.path.join(RUNS, name)
    _run(exe, open(os.path.join(src, "vpi.in")).read() + "&gpu\n n_walkers = 1, device = 0, device_sampler = T\n/\n",
         str(tmp_path))
    assert "using the host-driven sampler" not in open(tmp_path / "stdout.txt").read()
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    assert _hex_close(tmp_path / "e_vpi.hex", src, mixed=MIXED_TOL)
    if os.path.exists(os.path.join(src, "nr_vpi.out")):
        assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
    assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


def test_device_sampler_checkpoint_round_trip(exe, tmp_path):
    """device_sampler = T: 4 blocks in one go == 2 blocks + resume for 2 more; exercises the block-form
    generator snapshot the sampler kernel keeps for checkpoints (pigs_sampler_get_rng)."""
    base = open(os.path.join(RUNS, "he4_bis_cworm0_s1982", "vpi.in")).read().replace("Nblock = 3", "Nblock = 4")
    gpu = "&gpu\n n_walkers = 2, device = 0, device_sampler = T\n/\n"
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    _run(exe, base + gpu, str(a))
    _run(exe, base.replace("Nblock = 4", "Nblock = 2") + gpu, str(b))
    first = [open(b / f"et_vpi.w{w:04d}.out").read().splitlines() for w in range(2)]
    _run(exe, base.replace("Nblock = 4", "Nblock = 2").replace("resume = F", "resume = T") + gpu, str(b))
    strip = lambda ls: [l.split()[1:] for l in ls]
    for w in range(2):
        second = open(b / f"et_vpi.w{w:04d}.out").read().splitlines()
        whole = open(a / f"et_vpi.w{w:04d}.out").read().splitlines()
        assert strip(first[w]) + strip(second) == strip(whole), w
    assert same_bits(np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin"))


@pytest.mark.parametrize("sampling,extra,samp,cworm", [
    ("bis", "dim = 2, Np = 37, density = 0.06d0", "Nb = 12, Lstag = 6, Nlev = 3", "0.4d0"),
    ("sta", "dim = 3, Np = 21, density = 0.2d0", "Nb = 12, Lstag = 6, Nlev = 3", "0.4d0"),
    # beyond the one-launch kernel's four levels: 2^5, 2^6 beads per bisection segment (pigs_diag.hip's stage machine)
    ("bis", "dim = 3, Np = 20, density = 0.3d0", "Nb = 40, Lstag = 30, Nlev = 6", "0.0d0"),
    # Lstag > Nb at CWorm = 0: the never-accepted open proposal leaves the chain (both samplers draw it, build nothing)
    ("bis", "dim = 3, Np = 20, density = 0.3d0", "Nb = 40, Lstag = 50, Nlev = 6", "0.0d0"),
    ("sta", "dim = 3, Np = 21, density = 0.2d0", "Nb = 10, Lstag = 15, Nlev = 3", "0.0d0"),
    ("bis", "dim = 3, Np = 70, density = 0.3d0", "Nb = 20, Lstag = 8, Nlev = 5", "0.4d0"),
    # more OBDM iterations per step than round 1's fixed 64-int event log could hold (Nobdm <= 30)
    ("bis", "dim = 3, Np = 12, density = 0.3d0", "Nb = 10, Lstag = 6, Nlev = 2", "2.0d0, Nobdm = 45"),
    # chains long enough for the speculative proposals of the one-launch kernel (>= 24 beads) with 2 and 4 levels
    ("bis", "dim = 3, Np = 30, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 2", "0.4d0"),
    ("bis", "dim = 3, Np = 40, density = 0.25d0", "Nb = 24, Lstag = 10, Nlev = 4", "0.4d0"),
    # three 64-partner passes per bead: the end bead's eight (new | old) x pass tasks, TranslateChain on four CUs per walker
    ("bis", "dim = 3, Np = 130, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 4", "0.4d0"),
    # found by scripts/sampler_fuzz.py (round 3): Nlev = 1, the reference's default (head / tail moves still bisect 2^2 beads);
    # two particles with an open worm in the stage-machine kernel (every visit to the same particle: no chain fetched ahead)
    ("bis", "dim = 3, Np = 33, density = 0.2d0", "Nb = 12, Lstag = 6, Nlev = 1", "0.0d0"),
    ("bis", "dim = 2, Np = 9, density = 0.2d0", "Nb = 8, Lstag = 6, Nlev = 1", "2.0d0"),
    ("bis", "dim = 2, Np = 2, density = 0.2d0", "Nb = 33, Lstag = 13, Nlev = 5", "0.6d0")])
def test_device_sampler_equals_host_driven_sampler_on_new_shapes(exe, tmp_path, sampling, extra, samp, cworm):
    """No reference run exists for these shapes; the host-driven sampler (bit-identical to the reference wherever
    a fixture exists) is the yardstick: same input, three walkers, device_sampler = F and T must give the same
    trajectories (worldlines to 1e-10), the same block energies and identical permutation / OBDM files."""
    inp = f"""&system
 {extra}, trap = F
/
&samp
 resume = F, dt = 5.0d-3, {samp}, seed = 77, delta_cm = 0.15d0, CMFreq = 2,
 sampling = '{sampling}', Nstag = 2,
 Nblock = 3, Nstep = 12, Nbin = 50, Nk = 10
/
&obdm
 swapping = T, Nobdm = 3, Npw = 1, CWorm = {cworm}
/
&wavefun
 Nmax = 4000, wf_table = T, v_table = T
/
&jastrow
 Rm = 1.10d0
/
&extpot
 a_ho = 1.0d0
/
"""
    a, b = tmp_path / "host", tmp_path / "dev"
    a.mkdir(); b.mkdir()
    _run(exe, inp + "&gpu\n n_walkers = 3, device = 0, device_sampler = F, checkpointing = F\n/\n", str(a))
    _run(exe, inp + "&gpu\n n_walkers = 3, device = 0, device_sampler = T, checkpointing = F\n/\n", str(b))
    assert "using the host-driven sampler" not in open(b / "stdout.txt").read()
    wa, wb = np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin")
    cfg = SystemConfig.from_namelists(inp)
    assert wa.shape == wb.shape
    assert same_bits(wa, wb), "device-resident and host-driven sampler: worldlines differ"
    for w in range(3):
        ba, ra = read_hex_blocks(a / f"e_vpi.w{w:04d}.hex")
        bb, rb = read_hex_blocks(b / f"e_vpi.w{w:04d}.hex")
        em, er = block_energy_errors(rb, ra)
        assert np.array_equal(ba, bb) and np.all(er <= 1e-10) and np.all(em <= MIXED_TOL), w
        assert open(a / f"nr_vpi.w{w:04d}.out", "rb").read() == open(b / f"nr_vpi.w{w:04d}.out", "rb").read(), w
        assert open(a / f"perm_vpi.w{w:04d}.out").read() == open(b / f"perm_vpi.w{w:04d}.out").read(), w


@pytest.mark.parametrize("dev", ["F", "T"])
def test_gpu_resume_from_reference_checkpoint(exe, dev, tmp_path):
    """checkpoint.dat / rand_state written by the REFERENCE program (tests/golden/vpi_runs/he4_resume; quirk Q10:
    the first rand_state record) resumed on the MI355X with the host-driven sampler and with the device-resident
    one (whose generator snapshot is uploaded in block form): the resumed run reproduces the reference's resumed run."""
    import shutil
    src = os.path.join(RUNS, "he4_resume")
    shutil.copy(os.path.join(src, "checkpoint.dat"), tmp_path / "checkpoint.dat")
    shutil.copy(os.path.join(src, "rand_state"), tmp_path / "rand_state")
    _run(exe, open(os.path.join(src, "vpi.in")).read() + f"&gpu\n n_walkers = 1, device = 0, device_sampler = {dev}\n/\n",
         str(tmp_path))
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    for f in ("e_vpi.out", "et_vpi.out"):
        assert _close(tmp_path / f, os.path.join(src, f)), f
    assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()


@pytest.mark.parametrize("dev,W", [("T", 1), ("F", 1), ("F", 3)])
@pytest.mark.parametrize("name", ["lstag_gt_nb_bis6", "lstag_gt_nb_sta"])
def test_gpu_samplers_lstag_beyond_nb(exe, name, dev, W, tmp_path):
    """Lstag > Nb with CWorm = 0 (bisection with six levels / staging sampling): the reference's never-accepted open
    proposal (quirk Q11) then reaches below bead 0 -- only its random numbers matter, and both samplers consume exactly
    those and build nothing (host-driven: bit-identical worldline, also with three walkers, where round 2's
    out-of-bounds stores corrupted the heap; glibc heap checking on)."""
    src = os.path.join(RUNS, name)
    _run(exe, open(os.path.join(src, "vpi.in")).read() + f"&gpu\n n_walkers = {W}, device = 0, device_sampler = {dev}\n/\n",
         str(tmp_path), env=dict(os.environ, MALLOC_CHECK_="3"))
    assert "using the host-driven sampler" not in open(tmp_path / "stdout.txt").read()
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape((W,) + want.shape)[0]
    assert same_bits(got, want)
    if W > 1:
        return
    for f in ("e_vpi.out", "et_vpi.out"):
        assert _close(tmp_path / f, os.path.join(src, f)), f
    assert _close(tmp_path / "gr_vpi.out", os.path.join(src, "gr_vpi.out"), rel=1e-9)


@pytest.mark.parametrize("dev", ["F", "T"])
def test_gpu_front_end_sharded_contexts_one_gpu(exe, dev, tmp_path):
    """&gpu n_gpus = 2, same_device = T: two contexts (two host threads, two shards of walkers) on this one GPU, the
    block-estimator vector all-reduced once per block (on duplicate devices the library's in-process rehearsal form
    stands in for RCCL).  Per-walker files and final worldlines equal the one-context run of the same walkers; the
    walker-summed files agree to summation order.  Host-driven and device-resident sampler."""
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read().replace("Nblock = 6", "Nblock = 3")
    a, b = tmp_path / "one", tmp_path / "sharded"
    a.mkdir(); b.mkdir()
    _run(exe, base + f"&gpu\n n_walkers = 4, device = 0, n_gpus = 1, device_sampler = {dev}\n/\n", str(a))
    _run(exe, base + f"&gpu\n n_walkers = 4, device = 0, n_gpus = 2, same_device = T, device_sampler = {dev}\n/\n", str(b))
    assert same_bits(np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin"))
    for w in range(4):
        for f in ("e_vpi", "et_vpi", "gr_vpi", "nr_vpi"):
            name = f"{f}.w{w:04d}.out"
            assert open(a / name, "rb").read() == open(b / name, "rb").read(), name
    for f in ("e_vpi.out", "et_vpi.out", "gr_vpi.out", "nr_vpi.out"):
        x, y = np.loadtxt(a / f), np.loadtxt(b / f)
        ok = np.isfinite(x)
        assert x.shape == y.shape and np.array_equal(ok, np.isfinite(y))
        assert np.all(np.abs(x - y)[ok] <= 1e-9 * np.abs(x[ok]) + 1e-300), f


# ---- long trajectories at the BASELINE sizes (tests/golden/vpi_runs/*_long_*: 50-60 MC steps from the reference's init) ----
@pytest.mark.parametrize("dev", ["F", "T"])
def test_gpu_front_end_config3_long_trajectory(exe, dev, tmp_path):
    """C3, 50 MC steps (30 of warm-up + 20 in the regime bench.py times), two lock-step walkers = the reference chains
    of seeds 1982 and 1983, host-driven sampler (~18 000 K1 batches per step) and device-resident sampler: final
    worldlines bit-identical (SHA-256), 64-bit block energies of all five blocks to 1e-10."""
    src = [os.path.join(RUNS, f"c3_n256_long_s{s_}") for s_ in (1982, 1983)]
    _run(exe, open(os.path.join(src[0], "vpi.in")).read() +
         f"&gpu\n n_walkers = 2, device = 0, device_sampler = {dev}, checkpointing = F\n/\n", str(tmp_path))
    got = np.fromfile(tmp_path / "worldlines_final.bin")
    for w in range(2):
        drv = dict(np.load(os.path.join(src[w], "driver.npz")))
        shape = tuple(int(x) for x in drv["Path_shape"])
        check_worldline_vs_driver(got.reshape((2,) + shape)[w], drv, None, tol=0.0)
        assert _hex_close(tmp_path / f"e_vpi.w{w:04d}.hex", src[w])


@pytest.mark.parametrize("name", ["c5_n256_aziz_long_s1982", "c5_n256_dipolar_long_s1982"])
@pytest.mark.parametrize("dev,nblocks", [("F", 2), ("T", 6)])
def test_gpu_front_end_config5_long_worm_trajectories(exe, name, dev, nblocks, tmp_path):
    """C5 with the stock CWorm = 0.5.  Device-resident sampler: all 60 steps (Aziz: 4 opens + 4 closes; dipolar: 1 open,
    > 100 swaps), final worldline bit-identical, OBDM file and permutation histogram as the reference program writes
    them.  Host-driven sampler (1.2 s per step at this size): the first two blocks = 20 steps, which must end on the
    reference's state at the end of its block 2 (Aziz: open at step 13, close at 15; dipolar: open at 7, dozens of swaps)."""
    src = os.path.join(RUNS, name)
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    pot = str(drv["potential"])
    txt = open(os.path.join(src, "vpi.in")).read().replace("Nblock = 6", f"Nblock = {nblocks}")
    assert f"Nblock = {nblocks}" in txt
    _run(exe, txt + f"&gpu\n n_walkers = 1, device = 0, device_sampler = {dev}, potential = '{pot}', checkpointing = F\n/\n",
         str(tmp_path))
    shape = tuple(int(x) for x in drv["Path_shape"])
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(shape)
    from helpers import sha256_of
    assert np.array_equal(sha256_of(got), drv["ckpt_sha"][nblocks - 1]), "worldline differs from the reference's at the end of block %d" % nblocks
    assert _hex_close(tmp_path / "e_vpi.hex", src, nblocks=nblocks)
    if nblocks == 6:
        if os.path.exists(os.path.join(src, "nr_vpi.out")):
            assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
        if os.path.exists(os.path.join(src, "fort.99")):
            assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


@pytest.mark.parametrize("dev", ["F", "T"])
def test_gpu_crystal_start_from_config_ini(exe, dev, tmp_path):
    """crystal = T on the MI355X, both samplers: Np / box / density and the start configuration from config_ini.in
    (reference vpi.f90:99-107, vpi_mod.f90:218-230) -- worldline bit-identical to the reference program's, block
    energies as printed, OBDM file and permutation histogram identical."""
    src = os.path.join(RUNS, "he4_crystal")
    _run(exe, open(os.path.join(src, "vpi.in")).read() + f"&gpu\n n_walkers = 1, device = 0, device_sampler = {dev}\n/\n",
         str(tmp_path), extra_files=[os.path.join(src, "config_ini.in")])
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    for f in ("e_vpi.out", "et_vpi.out"):
        assert _close(tmp_path / f, os.path.join(src, f)), f
    assert _close(tmp_path / "gr_vpi.out", os.path.join(src, "gr_vpi.out"), rel=1e-9)
    assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
    assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


@pytest.mark.parametrize("sampling,system,samp,cworm", [
    ("bis", "dim = 3, Np = 30, density = 0.3d0", "Nb = 16, Lstag = 6, Nlev = 3", "0.4d0"),
    ("sta", "dim = 2, Np = 21, density = 0.1d0", "Nb = 12, Lstag = 6, Nlev = 2", "0.5d0")])
def test_samplers_agree_bit_for_bit_over_hundreds_of_steps(exe, tmp_path, sampling, system, samp, cworm):
    """Soak (short form of scripts/k6_vs_host_soak.py, which runs 3 000 steps: identical): 300 MC steps x 4 walkers with a
    busy worm sector through the host-driven sampler and through the device-resident one.  Final worldlines bit-identical,
    OBDM and permutation files byte-identical, 64-bit block energies 1e-10.  Round 3 found with it that the host-driven
    sampler lost an accepted bead about once in 100 steps ON THE GPU ONLY: a Swap stage in which no walker reached a
    proposal flushed its (asynchronous) commits and returned without a synchronising call, and the next mover refilled the
    pinned staging arrays while the commit kernel was still on its way (host/pigs_sampler.f90 commit_in_flight); the
    front end's own device-vs-mirror check stops such a run (status 3)."""
    inp = f"""&system
 {system}, trap = F
/
&samp
 resume = F, dt = 1.0d-2, {samp}, seed = 4242, delta_cm = 0.2d0, CMFreq = 2,
 sampling = '{sampling}', Nstag = 2, Nblock = 3, Nstep = 100, Nbin = 50, Nk = 10
/
&obdm
 swapping = T, Nobdm = 3, Npw = 1, CWorm = {cworm}
/
&wavefun
 Nmax = 4000, wf_table = T, v_table = T
/
&jastrow
 Rm = 1.10d0
/
&extpot
 a_ho = 1.0d0
/
"""
    a, b = tmp_path / "host", tmp_path / "dev"
    a.mkdir(); b.mkdir()
    _run(exe, inp + "&gpu\n n_walkers = 4, device = 0, device_sampler = F, checkpointing = F\n/\n", str(a))
    _run(exe, inp + "&gpu\n n_walkers = 4, device = 0, device_sampler = T, checkpointing = F\n/\n", str(b))
    assert same_bits(np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin"))
    swaps = 0
    for w in range(4):
        for f in ("nr_vpi", "perm_vpi"):
            assert open(a / f"{f}.w{w:04d}.out", "rb").read() == open(b / f"{f}.w{w:04d}.out", "rb").read(), (f, w)
        ba, ra = read_hex_blocks(a / f"e_vpi.w{w:04d}.hex")
        bb, rb = read_hex_blocks(b / f"e_vpi.w{w:04d}.hex")
        assert np.array_equal(ba, bb)
        if len(ba):
            em, er = block_energy_errors(rb, ra)
            assert np.all(er <= 1e-10) and np.all(em <= MIXED_TOL), w
    out = open(a / "stdout.txt").read()
    assert "Swap acc" in out


def test_sampler_choice_is_automatic_when_left_out(exe, tmp_path):
    """&gpu without device_sampler: the front end takes the device-resident sampler where it serves the input (and says so)
    -- same files as the reference run -- and the host-driven one where it does not (a trapped system with Nlev = 5: K6's
    trap form stops at four levels)."""
    src = os.path.join(RUNS, "he4_worm_s1982")
    a = tmp_path / "auto"; a.mkdir()
    _run(exe, open(os.path.join(src, "vpi.in")).read(), str(a))
    assert "device-resident (K6" in open(a / "stdout.txt").read()
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    assert same_bits(np.fromfile(a / "worldlines_final.bin").reshape(want.shape), want)
    assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(a / "nr_vpi.out", "rb").read()
    b = tmp_path / "fallback"; b.mkdir()
    inp = """&system
 dim = 2, Np = 6, density = 0.1d0, trap = T
/
&samp
 resume = F, dt = 1.0d-2, Nb = 20, seed = 11, delta_cm = 0.2d0, CMFreq = 1, sampling = 'bis', Lstag = 4, Nlev = 5, Nstag = 2,
 Nblock = 1, Nstep = 5, Nbin = 50, Nk = 10
/
&obdm
 swapping = T, CWorm = 0.0d0, Nobdm = 0, Npw = 0
/
&wavefun
 Nmax = 4000, wf_table = T, v_table = T
/
&jastrow
 Rm = 1.10d0
/
&extpot
 a_ho = 1.0d0 1.3d0
/
"""
    _run(exe, inp, str(b))
    assert "host-driven (the device-resident sampler does not serve" in open(b / "stdout.txt").read()


def test_gpu_device_sampler_with_the_references_default_nlev_1(exe, tmp_path):
    """Nlev = 1 is the reference's DEFAULT (vpi_mod.f90:47).  Its head / tail moves draw their level as int((Nlev-1)*grnd())+2
    = 2 and bisect 2^2 beads all the same (vpi_mod.f90:1023,1209); only Bisection itself works on 2^1.  Round 3's sampler
    fuzz (scripts/sampler_fuzz.py) found the device-resident sampler clamping that level to Nlev: every Nlev = 1 input took
    another trajectory than the host-driven sampler.  Here against the reference PROGRAM (he4_nlev1: open, close, half-chain
    moves): worldline bit-identical, OBDM and permutation files identical, 64-bit block energies 1e-10."""
    src = os.path.join(RUNS, "he4_nlev1")
    _run(exe, open(os.path.join(src, "vpi.in")).read() + "&gpu\n n_walkers = 1, device = 0, device_sampler = T\n/\n", str(tmp_path))
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    got = np.fromfile(tmp_path / "worldlines_final.bin").reshape(want.shape)
    assert same_bits(got, want)
    assert _hex_close(tmp_path / "e_vpi.hex", src, mixed=MIXED_TOL)
    assert open(os.path.join(src, "nr_vpi.out"), "rb").read() == open(tmp_path / "nr_vpi.out", "rb").read()
    assert open(tmp_path / "perm_vpi.out").read().split() == open(os.path.join(src, "fort.99")).read().split()


def test_sampler_fuzz_first_cases(exe):
    """The first 16 cases of scripts/sampler_fuzz.py (seed 31337: the run that found the Nlev = 1 defect): random inputs through
    the host-driven sampler, the device-resident sampler and the CPU twin with the reference's arithmetic, all bit-identical."""
    import sys
    shim = os.path.join(ROOT, "tests", "shim", "_build", "pigs_vpi")
    if not os.path.exists(shim):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hostlib import build_cpu_host
        build_cpu_host()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sampler_fuzz.py"), "16", "31337"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "16 cases, 0 failing" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_small_box_with_long_free_end_segments_vs_the_cpu_twin(exe, tmp_path):
    """Case 241 of `WIDE=1 scripts/sampler_fuzz.py 260 777`: two particles in a box of L = 2.15, head / tail moves over up to
    2^7 links at dt = 0.03 (a free spread of ~ L).  Their proposals land several L away, BoundaryConditions folds them once
    (pbc_mod.f90:20-21) and MinimumImage folds a separation once (pbc_mod.f90:40-41): what is beyond 1.5 L stays outside the
    cutoff.  The short-arithmetic minimum image folded by rint(v/L) -- all the way -- and step 37 took another decision than
    the reference's arithmetic.  Host-driven sampler, device-resident sampler and the CPU twin (tests/shim + oracle) are
    bit-identical again after 40 MC steps."""
    import sys
    shim = os.path.join(ROOT, "tests", "shim", "_build", "pigs_vpi")
    if not os.path.exists(shim):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hostlib import build_cpu_host
        build_cpu_host()
    inp = """&system
 dim = 3, Np = 2, density = 0.2d0, trap = F
/
&samp
 resume = F, dt = 0.03d0, Nb = 64, seed = 1241, delta_cm = 0.2d0, CMFreq = 1,
 sampling = 'bis', Lstag = 36, Nlev = 7, Nstag = 2, Nblock = 2, Nstep = 20, Nbin = 40, Nk = 6
/
&obdm
 swapping = T, Nobdm = 3, Npw = 0, CWorm = 2.0d0
/
&wavefun
 Nmax = 4000, wf_table = T, v_table = T
/
&jastrow
 Rm = 1.10d0
/
&extpot
 a_ho = 1.0d0 1.3d0 0.8d0
/
"""
    out = {}
    for arm in "FTC":
        d = tmp_path / arm
        d.mkdir()
        _run(shim if arm == "C" else exe, inp + f"&gpu\n n_walkers = 1, device = 0, device_sampler = {'T' if arm == 'T' else 'F'}, "
             "checkpointing = F, potential = 'dipolar'\n/\n", str(d))
        out[arm] = np.fromfile(d / "worldlines_final.bin")
    assert same_bits(out["F"], out["C"]), "host-driven sampler vs CPU twin"
    assert same_bits(out["T"], out["C"]), "device-resident sampler vs CPU twin"
