"""The vpi.in-driven front end (host/pigs_vpi.f90) against output files of the reference PROGRAM
(tests/golden/vpi_runs, produced by oracle/_ref/vpi): byte-identical observable files and a
bit-identical final worldline, for diagonal + worm sectors, staging + bisection sampling, trap and
PBC, and for several lock-step walkers (walker w == the reference run with seed+w).
Host logic only here: the C ABI is served by tests/shim (CPU oracle).  tests/test_gpu_host.py runs
the same comparison through libpigs_hip.so on the MI355X."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import check_worldline_vs_driver, driver_blocks, read_hex_blocks, same_bits
from hostlib import build_cpu_host

RUNS = os.path.join(GOLDEN, "vpi_runs")
FILES = ["e_vpi.out", "et_vpi.out", "gr_vpi.out", "sk_vpi.out", "nr_vpi.out"]


def run_pigs_vpi(exe, vpi_in_text, workdir, env=None, extra_files=()):
    for f in extra_files:                                   # input files next to vpi.in (config_ini.in of a crystal start)
        shutil.copy(f, os.path.join(workdir, os.path.basename(f)))
    with open(os.path.join(workdir, "vpi.in"), "w") as f:
        f.write(vpi_in_text)
    with open(os.path.join(workdir, "vpi.in")) as fin, open(os.path.join(workdir, "stdout.txt"), "w") as fo:
        r = subprocess.run([exe], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=workdir, timeout=900, env=env)
    assert r.returncode == 0, open(os.path.join(workdir, "stdout.txt")).read()[-2000:]


def final_worldline(workdir, shape, W=1):
    a = np.fromfile(os.path.join(workdir, "worldlines_final.bin"))
    return a.reshape((W,) + shape)


@pytest.fixture(scope="module")
def exe():
    return build_cpu_host()[2]


@pytest.mark.parametrize("name", ["he4_worm_s1982", "ho1d_n2", "he4_stock_short", "he4_cworm0",
                                  "he4_wormbusy_s7", "he4_wormbusy_s8", "he4_wf_analytic", "he4_nlev1"])
def test_front_end_reproduces_reference_files(exe, name, tmp_path):
    """he4_wormbusy_*: Npw = 2 partial waves in nr_vpi.out and dozens of accepted swaps (fort.99)."""
    src = os.path.join(RUNS, name)
    run_pigs_vpi(exe, open(os.path.join(src, "vpi.in")).read(), str(tmp_path))
    for f in FILES:
        if os.path.exists(os.path.join(src, f)):
            assert open(os.path.join(src, f), "rb").read() == open(tmp_path / f, "rb").read(), f
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    assert same_bits(final_worldline(str(tmp_path), want.shape)[0], want)
    ref_perm = open(os.path.join(src, "fort.99")).read().split()
    assert open(tmp_path / "perm_vpi.out").read().split() == ref_perm
    # block energies beyond the 10 printed digits: e_vpi.hex against the reference's own estimators evaluated
    # in the program's schedule (driver.npz, tests/golden/ref_driver.py)
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    blocks, rows = read_hex_blocks(tmp_path / "e_vpi.hex")
    wb, wrows = driver_blocks(drv)
    assert np.array_equal(blocks, wb)
    assert np.all(np.abs(rows - wrows) <= 1e-13 * np.abs(wrows)), np.max(np.abs(rows - wrows) / np.abs(wrows))


@pytest.mark.parametrize("name", ["c3_n256_s1982", "c5_n256_dipolar_s1982"])
def test_front_end_at_baseline_sizes(exe, name, tmp_path):
    """BASELINE configs 3 and 5 (N=256, 161 / 321 beads; C5 with the dipolar table, worm sector and swaps) through
    the host-driven sampler: final worldline bit-identical to the reference's movers run in the program's
    schedule (SHA-256 of all N*M*3 coordinates), 64-bit block energies, and -- where the reference PROGRAM can run
    the input (Aziz potential) -- its files byte for byte."""
    src = os.path.join(RUNS, name)
    drv = dict(np.load(os.path.join(src, "driver.npz")))
    pot = str(drv["potential"])
    run_pigs_vpi(exe, open(os.path.join(src, "vpi.in")).read() +
                 f"&gpu\n n_walkers = 1, device = 0, potential = '{pot}', checkpointing = F\n/\n", str(tmp_path))
    shape = tuple(int(x) for x in drv["Path_shape"])
    got = final_worldline(str(tmp_path), shape)[0]
    check_worldline_vs_driver(got, drv, None, tol=0.0)
    blocks, rows = read_hex_blocks(tmp_path / "e_vpi.hex")
    wb, wrows = driver_blocks(drv)
    assert np.array_equal(blocks, wb)
    assert np.all(np.abs(rows - wrows) <= 1e-13 * np.abs(wrows)), np.max(np.abs(rows - wrows) / np.abs(wrows))
    for f in FILES:
        if os.path.exists(os.path.join(src, f)):
            assert open(os.path.join(src, f), "rb").read() == open(tmp_path / f, "rb").read(), f


@pytest.mark.parametrize("W", [1, 3])
@pytest.mark.parametrize("name", ["lstag_gt_nb_bis6", "lstag_gt_nb_sta"])
def test_host_driven_sampler_lstag_beyond_nb(exe, name, W, tmp_path):
    """Lstag > Nb at CWorm = 0 (input the reference runs: its never-accepted OpenChain proposal, quirk Q11, then
    indexes Path below bead 0 / above bead 2Nb, vpi_mod.f90:1853-1857).  The host-driven sampler draws exactly that
    proposal's random numbers and builds nothing (round 2 indexed its own arrays out of bounds there and corrupted
    the heap from three walkers on).  Files byte-identical, final worldline bit-identical to the reference run;
    glibc's heap checking on (MALLOC_CHECK_=3 aborts on the first corrupted chunk)."""
    src = os.path.join(RUNS, name)
    env = dict(os.environ, MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    run_pigs_vpi(exe, open(os.path.join(src, "vpi.in")).read() +
                 f"&gpu\n n_walkers = {W}, device = 0, device_sampler = F\n/\n", str(tmp_path), env=env)
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    assert same_bits(final_worldline(str(tmp_path), want.shape, W)[0], want)
    for f in FILES:
        mine = tmp_path / (f if W == 1 else f.replace(".out", ".w0000.out"))
        assert open(os.path.join(src, f), "rb").read() == open(mine, "rb").read(), f


@pytest.mark.parametrize("W", [1, 2])
def test_crystal_start_from_config_ini(exe, W, tmp_path):
    """crystal = T (vpi.f90:99-107, vpi_mod.f90:218-230): Np, box and density come from config_ini.in -- the
    namelist's Np = 8 and density = 0.2 are overridden by the file's 27 particles at 0.45 -- and every bead starts on
    the particle's lattice site; no random number is spent on the start.  Reference program files byte for byte, final
    worldline bit-identical; with two walkers each one reads the same lattice (walker 0 = the reference run)."""
    src = os.path.join(RUNS, "he4_crystal")
    txt = open(os.path.join(src, "vpi.in")).read()
    assert "crystal = T" in txt and "Np = 8" in txt
    run_pigs_vpi(exe, txt + f"&gpu\n n_walkers = {W}, device = 0\n/\n", str(tmp_path),
                 extra_files=[os.path.join(src, "config_ini.in")])
    want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
    assert want.shape == (17, 27, 3)
    assert same_bits(final_worldline(str(tmp_path), want.shape, W)[0], want)
    for f in FILES:
        mine = tmp_path / (f if W == 1 else f.replace(".out", ".w0000.out"))
        assert open(os.path.join(src, f), "rb").read() == open(mine, "rb").read(), f
    mine = tmp_path / ("perm_vpi.out" if W == 1 else "perm_vpi.w0000.out")
    assert open(mine).read().split() == open(os.path.join(src, "fort.99")).read().split()


def test_worm_sector_with_lstag_beyond_nb_is_refused(exe, tmp_path):
    """CWorm > 0 with Lstag > Nb: the reference's half-chain movers would read and write outside Path and could accept
    the result -- there is nothing defined to reproduce, so the front end stops with a message (both samplers)."""
    txt = open(os.path.join(RUNS, "lstag_gt_nb_sta", "vpi.in")).read().replace("CWorm = 0.0d0", "CWorm = 0.5d0")
    for dev in "FT":
        with open(tmp_path / "vpi.in", "w") as f:
            f.write(txt + f"&gpu\n n_walkers = 1, device = 0, device_sampler = {dev}\n/\n")
        r = subprocess.run([exe], stdin=open(tmp_path / "vpi.in"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           cwd=tmp_path, timeout=120)
        assert r.returncode == 2 and b"Lstag <= Nb" in r.stdout, r.stdout[-500:]


def test_lockstep_walkers_reproduce_per_seed_reference_runs(exe, tmp_path):
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read()
    run_pigs_vpi(exe, base + "&gpu\n n_walkers = 3, device = 0\n/\n", str(tmp_path))
    for w, seed in enumerate((1982, 1983, 1984)):
        src = os.path.join(RUNS, f"he4_worm_s{seed}")
        for f in FILES:
            mine = tmp_path / f.replace(".out", f".w{w:04d}.out")
            assert open(os.path.join(src, f), "rb").read() == open(mine, "rb").read(), (w, f)
        want = np.load(os.path.join(src, "final_worldline.npz"))["Path"]
        assert same_bits(final_worldline(str(tmp_path), want.shape, 3)[w], want)
    # the unsuffixed files hold the walker average of the block values
    av = np.loadtxt(tmp_path / "e_vpi.out")
    per = [np.loadtxt(tmp_path / f"e_vpi.w{w:04d}.out") for w in range(3)]
    blocks = sorted(set(int(b) for p in per for b in np.atleast_2d(p)[:, 0]))
    assert len(av) == len(blocks)


def test_reference_program_itself_if_present(exe, tmp_path):
    """Where the reference build exists, run it fresh on a new input and compare again."""
    from oracle.pyoracle import REF_VPI
    if not os.path.exists(REF_VPI):
        pytest.skip("oracle/_ref/vpi not built here")
    txt = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read().replace("seed = 1982", "seed = 4242") \
        .replace("Nstep = 25", "Nstep = 15")
    a, b = tmp_path / "ref", tmp_path / "mine"
    a.mkdir(); b.mkdir()
    run_pigs_vpi(REF_VPI, txt, str(a))
    run_pigs_vpi(exe, txt, str(b))
    for f in FILES:
        assert open(a / f, "rb").read() == open(b / f, "rb").read(), f


def test_resume_from_reference_checkpoint(exe, tmp_path):
    """checkpoint.dat / rand_state written by the reference are read back the way the reference reads
    them (incl. quirk Q10: the FIRST rand_state record) and the resumed run reproduces the
    reference's resumed run byte for byte."""
    import shutil
    src = os.path.join(RUNS, "he4_resume")
    shutil.copy(os.path.join(src, "checkpoint.dat"), tmp_path / "checkpoint.dat")
    shutil.copy(os.path.join(src, "rand_state"), tmp_path / "rand_state")
    run_pigs_vpi(exe, open(os.path.join(src, "vpi.in")).read(), str(tmp_path))
    for f in FILES:
        if os.path.exists(os.path.join(src, f)):
            assert open(os.path.join(src, f), "rb").read() == open(tmp_path / f, "rb").read(), f


def test_own_checkpoint_round_trip(exe, tmp_path):
    """4 blocks in one go == 2 blocks, stop, resume for 2 more (our rand_state holds the NEWEST state)."""
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read().replace("Nblock = 6", "Nblock = 4")
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    run_pigs_vpi(exe, base, str(a))
    run_pigs_vpi(exe, base.replace("Nblock = 4", "Nblock = 2"), str(b))
    first = open(b / "et_vpi.out").read().splitlines()
    run_pigs_vpi(exe, base.replace("Nblock = 4", "Nblock = 2").replace("resume = F", "resume = T"), str(b))
    second = open(b / "et_vpi.out").read().splitlines()
    whole = open(a / "et_vpi.out").read().splitlines()
    strip = lambda ls: [l.split()[1:] for l in ls]          # drop the block number column
    assert strip(first) + strip(second) == strip(whole)
    assert same_bits(np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin"))
    # the checkpoint is the reference's text format: trap, isopen, iworm, Np*(2Nb+1) bead lines, 2 blank, 2 xend
    lines = open(b / "checkpoint.dat").read().split("\n")
    assert lines[0].strip() == ".False." and len([l for l in lines if l.strip()]) == 3 + 16 * 17 + 2


def test_host_threads_change_nothing(exe, tmp_path):
    """The per-walker loops of a stage (proposal generation, Metropolis decisions) run on OpenMP threads in the host-driven
    sampler: walkers own their random streams, item slots and mirror rows, so every file is the same byte for byte."""
    base = open(os.path.join(RUNS, "he4_stock_short", "vpi.in")).read()
    outs = []
    for n in (1, 4):
        d = tmp_path / f"t{n}"
        d.mkdir()
        run_pigs_vpi(exe, base + "&gpu\n n_walkers = 6, device = 0\n/\n", str(d), env=dict(os.environ, PIGS_HOST_THREADS=str(n)))
        outs.append(d)
    assert f"host threads per stage:{4:6d}" in open(outs[1] / "stdout.txt").read()
    assert same_bits(np.fromfile(outs[0] / "worldlines_final.bin"), np.fromfile(outs[1] / "worldlines_final.bin"))
    names = FILES + [f"e_vpi.w{w:04d}.hex" for w in range(6)] + [f"e_vpi.w{w:04d}.out" for w in range(6)]
    for f in names:
        assert open(outs[0] / f, "rb").read() == open(outs[1] / f, "rb").read(), f


@pytest.mark.parametrize("G", [2, 3])
def test_walkers_sharded_over_contexts_equal_one_context(exe, G, tmp_path):
    """&gpu n_gpus = G (BASELINE config 4's structure: walkers in contiguous shards, one context + one host thread per
    shard, ONE all-reduce of the block-estimator vector per block -- here between CPU-shim contexts): every walker's
    files, the final worldlines and the walker-summed files equal the single-context run of the same 5 walkers (walker
    w runs the chain of seed+w-1 whatever the partition; sums over walkers differ by their order only)."""
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read().replace("Nblock = 6", "Nblock = 3")
    a, b = tmp_path / "one", tmp_path / "sharded"
    a.mkdir(); b.mkdir()
    run_pigs_vpi(exe, base + "&gpu\n n_walkers = 5, device = 0, n_gpus = 1\n/\n", str(a))
    run_pigs_vpi(exe, base + f"&gpu\n n_walkers = 5, device = 0, n_gpus = {G}, same_device = T\n/\n", str(b))
    assert f"GPUs (walker shards):{G:6d}" in open(b / "stdout.txt").read()
    assert same_bits(np.fromfile(a / "worldlines_final.bin"), np.fromfile(b / "worldlines_final.bin"))
    for w in range(5):
        for f in FILES + ["perm_vpi.out", "e_vpi.hex"]:
            name = f.replace(".out", f".w{w:04d}.out").replace(".hex", f".w{w:04d}.hex")
            assert open(a / name, "rb").read() == open(b / name, "rb").read(), name
    # walker w of the sharded run is the reference run of seed 1982 + w
    for w, seed in enumerate((1982, 1983, 1984)):
        src = os.path.join(RUNS, f"he4_worm_s{seed}")
        ref = np.atleast_2d(np.loadtxt(os.path.join(src, "e_vpi.out")))
        assert np.array_equal(np.atleast_2d(np.loadtxt(b / f"e_vpi.w{w:04d}.out")), ref[ref[:, 0] <= 3])
    # the walker-summed files (written by shard 1 from the reduced vector)
    for f in ("e_vpi.out", "et_vpi.out", "gr_vpi.out", "sk_vpi.out", "nr_vpi.out"):
        x, y = np.loadtxt(a / f), np.loadtxt(b / f)
        assert x.shape == y.shape and x.size > 0, f
        ok = np.isfinite(x)
        assert np.array_equal(ok, np.isfinite(y)) and np.all(np.abs(x - y)[ok] <= 1e-9 * np.abs(x[ok]) + 1e-300), f


def test_every_shard_reaches_the_block_all_reduce_exactly_once(exe, tmp_path):
    """Collective safety of the sharded front end (&gpu n_gpus = G: one host thread + one context per shard, ONE
    pigs_estimators_allreduce per block and thread, pigs_vpi.f90): every shard must reach the block's all-reduce exactly
    once whatever happened in its block -- also when none of its walkers had a diagonal step (all of them open for the
    whole block), where a guard like `if (nd > 0)` around the call would leave the other shards waiting forever on a
    real RCCL communicator.  CWorm = 30 keeps the worms open most of the time, blocks of 2 steps; the CPU shim counts
    the calls per context."""
    base = open(os.path.join(RUNS, "he4_worm_s1982", "vpi.in")).read()
    txt = base.replace("CWorm = 0.5d0", "CWorm = 30.0d0").replace("Nblock = 6", "Nblock = 12").replace("Nstep = 25", "Nstep = 2")
    assert "CWorm = 30.0d0" in txt and "Nstep = 2" in txt
    with open(tmp_path / "vpi.in", "w") as f:
        f.write(txt + "&gpu\n n_walkers = 3, device = 0, n_gpus = 3, same_device = T\n/\n")
    r = subprocess.run([exe], stdin=open(tmp_path / "vpi.in"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp_path,
                       timeout=300, env=dict(os.environ, PIGS_SHIM_TRACE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("shim: context rank")]
    assert len(lines) == 3, r.stderr[-2000:]
    calls = [int(ln.split(":")[2].split()[0]) for ln in lines]
    empty = [int(ln.split(",")[1].split()[0]) for ln in lines]
    assert calls == [12, 12, 12], lines                      # once per block and shard
    assert sum(empty) > 0, lines                             # ... including blocks in which a shard had no diagonal step
