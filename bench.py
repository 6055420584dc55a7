#!/usr/bin/env python3
"""bench.py -- the headline measurement of BASELINE.json on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): liquid 4He, N=256
particles, 161 beads (reference namelist Nb=80), 128 independent walkers resident per GPU.
One STEP = one pass of the hot path (K1, the batched Delta-S evaluator that replaces the
reference's `call UpdateAction`, reference vpi_mod.f90:2491-2841) over one full-chain stage:
for every walker one particle is displaced and the Delta S of each of its 161 beads is
evaluated, i.e. 128*161 = 20 608 items, each visiting its 255 partners -- every resident slice
is read exactly once per step, exactly the access pattern of one lock-step TranslateChain stage
of the sampler.  Proposals (synthetic, seeded) are resident in HBM before the timed region.

metric  : bead-pair action evals / s (one eval = one (proposal bead, partner) visit = both the
          old and the new distance + their table lookups)
roofline: HBM bound named by the north star; achieved = ALGORITHMIC bytes per launch
          (6 200 B per item at N=256) / average launch duration from HIP events on the
          context's own stream.
cpu_baseline: the pinned scalar C restatement of the same routine (oracle/, kind "port"),
          timed on this machine's host cores on a bounded sample of the same batch.

roofline_large: the same stage on 3x the walkers (380 MB of worldlines: beyond the 256 MiB Infinity Cache), so
          that the headline fraction cannot be a cache artefact.
roofline_k2 / roofline_mc: secondary figures -- K2 (slice energies of ThermEnergy) against the FP64 vector peak,
          and the device-resident sampler's bead-pair rate against K1's.

N>1 (launched by torch.distributed.run): one process per GPU, walkers shard across ranks (weak
scaling: 128 walkers per GPU), no data-path collective in the Delta-S leg; the `mc` leg ends its block with ONE
all-reduce (RCCL) of the real block-estimator vector (energies and their squares, g(r), S(k), counters:
sharding.EstimatorVector), as the sampler does once per block, and rank 0 checks the sum.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: FP64 vector peak (the bound of K2 / K4)
# algorithmic flops of one unordered pair of a slice in K2: distance + minimum image + r^2 (17),
# sqrt (8), V interpolation (8) on every slice; dV/dr interpolation (10), 1/r (4), force terms on both particles (9) on
# odd slices -> 33 / 56, 44.5 on average over a chain
K2_FLOPS_PER_PAIR = 44.5
TRAFFIC_FILE = os.path.join("profiles", "r03_k1_hbm_traffic.json")


# K1 kernel behind each k1_variant on the bench workload (0 = the library's choice: periodic system, Np <= 256,
# a launch of >= 16 items per CU -> the persistent LDS-table kernel with the short arithmetic)
K1_KERNELS = {0: "pigs::k_delta_action_pipe2<3>", 12: "pigs::k_delta_action_pipe2<3>", 13: "pigs::k_delta_action_grid<3>",
              2: "pigs::k_delta_action_v2<3,false,256,false,false>"}


def make_workload(cfg, W, nsets, seed):
    """Seeded synthetic worldlines + `nsets` full-chain proposal stages."""
    rng = np.random.default_rng(seed)
    L = cfg.Lbox[0]
    g = int(np.ceil(cfg.Np ** (1.0 / 3.0)))
    lat = (np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3)[:cfg.Np]
           + 0.5) * (L / g) - L / 2
    # bead spread ~ sqrt(dt*|ib-Nb|)-like noise around a jittered lattice (no hard overlaps)
    ibs = np.arange(cfg.M)
    sig = 0.03 + np.sqrt(cfg.dt * np.minimum(np.abs(ibs - cfg.Nb), 16))[None, :, None, None]
    Paths = lat[None, None] + rng.normal(0, 0.05, (W, 1, cfg.Np, 3)) + sig * rng.normal(0, 0.5, (W, cfg.M, cfg.Np, 3))
    Paths = np.where(Paths > L / 2, Paths - L, Paths)
    Paths = np.where(Paths < -L / 2, Paths + L, Paths)
    sets = []
    w = np.repeat(np.arange(W, dtype=np.int32), cfg.M)
    ib = np.tile(np.arange(cfg.M, dtype=np.int32), W)
    for _ in range(nsets):
        ip = np.repeat(rng.integers(1, cfg.Np + 1, W).astype(np.int32), cfg.M)
        xold = Paths[w, ib, ip - 1].copy()
        xnew = xold + rng.normal(0, np.sqrt(cfg.dt), xold.shape)
        xnew = np.where(xnew > L / 2, xnew - L, xnew)
        xnew = np.where(xnew < -L / 2, xnew + L, xnew)
        sets.append((w, ip, ib, xnew, xold))
    return Paths, sets


def cpu_baseline(cfg, VT, WF, Paths, sets, budget_s=12.0):
    """Time the pinned C restatement (oracle/) on this host: 1 core, bounded sample."""
    from oracle.pyoracle import Oracle, System
    S = System(dim=cfg.dim, Np=cfg.Np, Nb=cfg.Nb, density=cfg.density, dt=cfg.dt, Rm=cfg.Rm)
    o = Oracle()
    evals, t0, i = 0, time.perf_counter(), 0
    out = None
    while True:
        w, ip, ib, xnew, xold = sets[i % len(sets)]
        out = o.delta_action_batch(S, WF, VT, Paths, w, ip, ib, xnew, xold)
        evals += len(w) * (cfg.Np - 1)
        i += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    one = evals / el

    # all host cores, one thread per core over disjoint item ranges (ctypes drops the GIL)
    import threading
    nthr = max(1, min(os.cpu_count() or 1, 64))
    w, ip, ib, xnew, xold = sets[0]
    chunks = np.array_split(np.arange(len(w)), nthr)
    reps = max(1, int(4.0 * one * nthr / (len(w) * (cfg.Np - 1))))

    def work(idx):
        for _ in range(reps):
            o.delta_action_batch(S, WF, VT, Paths, w[idx], ip[idx], ib[idx], xnew[idx], xold[idx])

    th = [threading.Thread(target=work, args=(c,)) for c in chunks]
    t1 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    allc = reps * len(w) * (cfg.Np - 1) / (time.perf_counter() - t1)
    return dict(value=one, unit="bead-pair action evals/s", cores=1, kind="port",
                sample="%d full-chain stages (%d items, %.2e pair evals) of the bench batch in %.1f s"
                       % (i, i * len(sets[0][0]), evals, el),
                all_cores=dict(value=allc, cores=nthr)), out


def cpu_sweep_baseline(np_, nb):
    """Same-box CPU figure for the MC-sweeps half of the metric: the build's own CPU restatement of one walker's sweep
    -- the Fortran host sampler (host/pigs_sampler.f90, bit-identical to the reference's movers) over the C-ABI served by
    the scalar CPU oracle (tests/shim: test infrastructure, used here only as the thing TIMED for the baseline) -- run as
    the front end runs it: N=256, 161 beads, stock schedule, estimators every step, 1 core.  Two runs (2 and 14 MC steps)
    so that set-up (tables, init, files) cancels: sweeps/s = 12 / (t14 - t2)  (~10 s of CPU work)."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from hostlib import BUILD, build_cpu_host
        exe = os.path.join(BUILD, "pigs_vpi")
        if not os.path.exists(exe):
            exe = build_cpu_host()[2]
    finally:
        sys.path.pop(0)
    inp = ("&system\n dim = 3, Np = %d, density = 0.365d0, trap = F\n/\n&samp\n resume = F, dt = 5.0d-3, Nb = %d, seed = 1982, "
           "delta_cm = 0.12d0, CMFreq = 1,\n sampling = 'bis', Lstag = %d, Nlev = 4, Nstag = 5,\n Nblock = 1, Nstep = %%d, Nbin = 100, "
           "Nk = 50\n/\n&obdm\n swapping = T, CWorm = 0.0d0, Nobdm = 0, Npw = 0\n/\n&wavefun\n Nmax = 10000, wf_table = T, "
           "v_table = T\n/\n&jastrow\n Rm = 1.20d0\n/\n&extpot\n a_ho = 1.0d0\n/\n&gpu\n n_walkers = 1, device = 0, "
           "device_sampler = F, checkpointing = F\n/\n" % (np_, nb, min(32, nb)))
    env = dict(os.environ, OMP_NUM_THREADS="1", PIGS_HOST_THREADS="1")
    times = {}
    for nstep in (2, 14):
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "vpi.in"), "w") as f:
                f.write(inp % nstep)
            t0 = time.perf_counter()
            with open(os.path.join(td, "vpi.in")) as fin, open(os.path.join(td, "out.txt"), "w") as fo:
                subprocess.run([exe], stdin=fin, stdout=fo, stderr=subprocess.STDOUT, cwd=td, check=True, timeout=600, env=env)
            times[nstep] = time.perf_counter() - t0
    per_sweep = (times[14] - times[2]) / 12.0
    return dict(value=1.0 / per_sweep, unit="walker-sweeps/s", cores=1, kind="port",
                sample="12 MC steps of one N=%d, %d-bead walker (stock schedule + estimators): Fortran host sampler over the "
                       "scalar C oracle; %.2f s for 14 steps - %.2f s for 2 steps" % (np_, 2 * nb + 1, times[14], times[2]))


def launch_stats(ms):
    """median / min / max / mean of per-launch durations (ms)."""
    a = np.asarray(ms, float)
    return {"median_ms": float(np.median(a)), "min_ms": float(a.min()), "max_ms": float(a.max()), "mean_ms": float(a.mean()),
            "p10_ms": float(np.percentile(a, 10)), "p90_ms": float(np.percentile(a, 90)), "launches": int(a.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--walkers", type=int, default=128, help="walkers per GPU")
    ap.add_argument("--np", type=int, default=256)
    ap.add_argument("--nb", type=int, default=80, help="reference namelist Nb (beads = 2*Nb+1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--variant", type=int, default=0, help="K1 kernel variant (0 = library default)")
    ap.add_argument("--large-walkers", type=int, default=384, help="walkers of the beyond-Infinity-Cache leg (0: skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse "
                                                      "the multi-process path on a one-GPU machine)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-c5", action="store_true", help="skip the BASELINE-config-5 sampler leg")
    ap.add_argument("--no-mc", action="store_true", help="skip the device-resident sampler legs (profiling runs of K1 only)")
    ap.add_argument("--mc-steps", type=int, default=14, help="MC steps per timed leg of the device-resident sampler (>= 0.5 s at 128 walkers)")
    ap.add_argument("--mc-large-walkers", type=int, default=1024, help="walkers of the second sampler leg (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pathintegralgroundstate_amd import SystemConfig, api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_device:
            local = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = SystemConfig(dim=3, Np=args.np, Nb=args.nb, density=0.365, dt=5e-3, Rm=1.2)
    W = args.walkers
    VT, WF = api.build_tables(cfg)
    nsets = 8
    Paths, sets = make_workload(cfg, W, nsets, seed=1982 + rank)

    ctx = api.PigsContext(cfg, VT, WF, n_walkers=W, device_id=local)
    ctx.upload_all(Paths)
    if args.variant:
        ctx.set_tuning("k1_variant", args.variant)
    dsets = []
    for (w, ip, ib, xnew, xold) in sets:
        dsets.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (w, ip, ib, xnew, xold)))
    n_items = len(sets[0][0])
    d_out = [torch.empty(n_items, dtype=torch.float64, device=dev) for _ in range(nsets)]
    torch.cuda.synchronize()

    def step(i):
        w, ip, ib, xn, xo = dsets[i % nsets]
        ctx.delta_action_batch_dev(n_items, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(),
                                   xo.data_ptr(), d_out[i % nsets].data_ptr())

    kstream = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)

    # context warm-up, untimed and independent of --warmup: first touch of every item set and output buffer, kernel
    # attributes, clocks (part of setting the workload up, like the uploads above)
    for i in range(2 * nsets):
        step(i)
    ctx.sync()
    for i in range(args.warmup):
        step(i)
    ctx.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(kstream)
    for i in range(args.steps):
        step(i)
    ev1.record(kstream)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps

    # per-launch durations (a pass of its own after the timed region: one event between every two launches)
    def per_launch(fn, n, stream):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record(stream)
        for i in range(n):
            fn(i)
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    k1_launches = launch_stats(per_launch(step, min(200, max(20, args.steps)), kstream))

    red_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    pair_evals_per_step = n_items * (cfg.Np - 1)
    value = world * pair_evals_per_step * args.steps / elapsed
    alg_bytes = n_items * (cfg.dim * cfg.Np * 8 + 2 * cfg.dim * 8 + 8)      # 6 200 B/item
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, gfx950 FETCH_SIZE x2 correction: profiles/README.md) -- valid for this workload only
    traffic = None
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            tj = json.load(f)
        if tj.get("algorithmic_bytes_per_launch") == alg_bytes and tj.get("kernel") == K1_KERNELS.get(args.variant):
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    # what a kernel that ONLY reads the resident worldlines reaches on this chip (measurement aid of the library): context
    # for `frac`, which stays priced against the guide's 8 TB/s
    stream_read = None
    try:
        sb, st_s = ctx.stream_read(50)
        stream_read = {"GBs": sb / st_s / 1e9, "bytes_per_pass": sb, "us_per_pass": 1e6 * st_s,
                       "kernel": "pigs::k_stream_read (plain double2 loads of the same %d resident walkers)" % W}
    except Exception as e:                                   # noqa: BLE001 -- informative only
        stream_read = {"error": str(e)}

    # ---- the same stage on 3x the walkers: 380 MB of worldlines, beyond the 256 MiB Infinity Cache ----------------
    large = None
    if args.large_walkers > 0:
        WL = args.large_walkers
        PathsL, setsL = make_workload(cfg, WL, 2, seed=4000 + rank)
        ctxL = api.PigsContext(cfg, VT, WF, n_walkers=WL, device_id=local)
        ctxL.upload_all(PathsL)
        del PathsL
        if args.variant:
            ctxL.set_tuning("k1_variant", args.variant)
        dL = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in st) for st in setsL]
        nL = len(setsL[0][0])
        oL = torch.empty(nL, dtype=torch.float64, device=dev)
        ksL = torch.cuda.ExternalStream(ctxL.stream(), device=dev)

        def stepL(i):
            w, ip, ib, xn, xo = dL[i % 2]
            ctxL.delta_action_batch_dev(nL, w.data_ptr(), ip.data_ptr(), ib.data_ptr(), xn.data_ptr(), xo.data_ptr(),
                                        oL.data_ptr())
        # warm-up: this leg streams from HBM (380 MB, beyond the Infinity Cache) right after seconds of host work with an idle
        # GPU, and the memory side needs ~100 launches (~10 ms of sustained traffic) to reach its steady state: timed after 40
        # launches the same 100 launches average 96 us, after 400 or 2 000 they average 77.4 / 76.2 us with p10-p90 of
        # 75.7-79.6 us (gpurun r3: PIGS_BENCH_LARGE_WARMUP sweep).  Round 2 timed 6 launches after creation and saw 72-121 us.
        for i in range(int(os.environ.get("PIGS_BENCH_LARGE_WARMUP", "400"))):
            stepL(i)
        ctxL.sync()
        nstepL = max(10, min(args.steps, 100))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(ksL)
        for i in range(nstepL):
            stepL(i)
        e1.record(ksL)
        ctxL.sync()
        torch.cuda.synchronize()
        msL = e0.elapsed_time(e1) / nstepL
        large_launches = launch_stats(per_launch(stepL, 100, ksL))
        bytesL = nL * (cfg.dim * cfg.Np * 8 + 2 * cfg.dim * 8 + 8)
        trafficL = None
        try:
            tl = json.load(open(os.path.join(ROOT, TRAFFIC_FILE))).get("large", {})
            if tl.get("algorithmic_bytes_per_launch") == bytesL:
                trafficL = tl["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        large = {"bound": "hbm", "walkers": WL, "traffic": trafficL,
                 "traffic_source": (TRAFFIC_FILE + " (same PMC passes)") if trafficL is not None else None, "working_set_bytes": int(WL * cfg.M * cfg.dim * cfg.Np * 8),
                 "algorithmic_bytes_per_launch": bytesL, "kernel_ms": msL, "achieved": bytesL / (msL * 1e-3) / 1e9,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytesL / (msL * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "steps": nstepL, "warmup_launches": int(os.environ.get("PIGS_BENCH_LARGE_WARMUP", "400")), "per_launch": large_launches, "note": "same kernel and stage as `roofline`, worldlines 3x the 128-walker set: larger "
                                          "than the 256 MiB Infinity Cache, so every slice comes from HBM"}
        try:
            sb, st_s = ctxL.stream_read(30)
            large["streaming_read_same_bytes"] = {"GBs": sb / st_s / 1e9, "bytes_per_pass": sb, "us_per_pass": 1e6 * st_s}
        except Exception as e:                               # noqa: BLE001 -- informative only
            large["streaming_read_same_bytes"] = {"error": str(e)}
        ctxL.close()
        del dL, oL

    # ---- second half of BASELINE's metric: MC sweeps / s.  The device-resident sampler (K6) advances
    # every resident walker by whole MC steps (CM move of every particle + Nstag x Np x {head, tail,
    # bisection}, stock vpi.in schedule CMFreq=1 Nstag=5 Nlev=4) with no host in the loop.
    mc = None
    try:
        if args.no_mc:
            raise api.PigsError("skipped (--no-mc)")
        mcfg = SystemConfig(dim=3, Np=args.np, Nb=args.nb, density=0.365, dt=5e-3, Rm=1.2, Nlev=4, Nstag=5,
                            Lstag=32, CMFreq=1, delta_cm=0.12)
        ctx.sampler_init(Nlev=mcfg.Nlev, Nstag=mcfg.Nstag, CMFreq=1, Lstag=min(32, args.nb), delta_cm=mcfg.delta_cm_eff)
        if args.same_device and world > 1:      # rehearsal: several processes on one chip -- no cooperating workgroups (pigs_cm.hip)
            ctx.set_tuning("cm_split", 1)
        for w in range(W):
            ctx.sampler_seed(w, 1982 + rank * W + w)
        acc0 = ctx.sampler_counters()
        # warm-up: four whole MC steps from the jittered-lattice start (kernel attributes, first touch, the first sweeps'
        # many rejections) before the timed legs
        ctx.sampler_step(1)
        ctx.sync()
        # round 2 timed steps 2..4 (one warm-up step): kept as `first_steps` so that the rounds can be compared -- the
        # walkers are still settling there (fewer accepted head / tail moves, hence fewer bisection levels run) and a
        # step is ~1 ms cheaper than in the regime the legs below time
        t0f = time.perf_counter()
        for q in range(3):
            ctx.sampler_step(2 + q)
        ctx.sync()
        first_steps_s = (time.perf_counter() - t0f) / 3
        nmc = max(1, args.mc_steps)
        istep = 4

        def timed_steps(fn):
            """nmc calls of fn, each followed by a stream synchronisation: per-step wall times (s) and their sum.  The
            synchronisation costs ~10 us of a ~40 ms step and makes min / median / max of a step visible."""
            ts = []
            for _ in range(nmc):
                t = time.perf_counter()
                fn()
                ctx.sync()
                ts.append(time.perf_counter() - t)
            return ts
        if world > 1:
            dist.barrier()

        def moves_only():
            nonlocal istep
            istep += 1
            ctx.sampler_step(istep)
        t1 = time.perf_counter()
        ts_moves = timed_steps(moves_only)
        mc_el = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([mc_el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mc_el = float(tt.item())
        acc = (ctx.sampler_counters() - acc0).sum(0) / (W * (nmc + 4))
        # the same steps followed by the diagonal-sector estimators of vpi.f90:443-469 (2 x LocalEnergy K4,
        # ThermEnergy K2/K3, g(r) + S(k) K7) for every walker -- ONE library call and one synchronisation per step
        # (pigs_diagonal_estimators): SURVEY 8d's definition of a sweep.  The block's estimator vector (what the front end
        # accumulates per block: vpi.f90:456-469) is filled from them and all-reduced ONCE at the end of the block,
        # inside the timed region.
        from pathintegralgroundstate_amd.sharding import EstimatorVector
        ev = EstimatorVector(Nbin=100, Nk=50, dim=3, Npw=0)
        c16_0 = ctx.sampler_counters16()
        if world > 1:
            dist.barrier()

        # The estimators of step n run on the context's second stream, on a snapshot of the worldlines, while step n+1 is
        # sampled (pigs_diagonal_estimators_begin / _end): at 128 walkers the sampler leaves half of the CUs idle.
        est_pending = False

        def collect():
            nonlocal est_pending
            if not est_pending:
                return
            est_pending = False
            r = ctx.diagonal_estimators_end()
            E = 0.5 * (r["E1"] + r["E2"])
            Pt = r["Vt"]
            K = E - Pt
            ev.add("n_diag", W)
            for name, v in (("E", E), ("K", K), ("V", Pt), ("Et", r["Et"]), ("Kt", r["Kt"]), ("Vt", Pt)):
                ev.add(name, v.sum()); ev.add(name + "2", (v * v).sum())
            ev.add("gr", r["gr"].sum(0)); ev.add("Sk", r["Sk"].sum(0).T.ravel()); ev.add("ngr", W)

        def full_step():
            nonlocal istep, est_pending
            istep += 1
            ctx.sampler_step(istep)                         # queued; the previous step's estimators run beside it
            collect()
            ctx.diagonal_estimators_begin(100, cfg.rcut / 100.0, 50)
            est_pending = True
        t2 = time.perf_counter()
        ts_full = timed_steps(full_step)
        collect()                                           # the last step's estimators: inside the timed region
        c16 = (ctx.sampler_counters16() - c16_0).sum(0)
        for name, q in (("acc_cm", 0), ("acc_head", 1), ("acc_tail", 2), ("acc_bd", 3), ("try_open", 4), ("try_cm", 14),
                        ("try_stag", 15)):
            ev.add(name, float(c16[q]))
        mine = ev.data.copy()
        if world > 1:                                           # the one collective of a block: RCCL over xGMI
            t_est = torch.from_numpy(mine.copy()).to(red_dev)
            dist.all_reduce(t_est)
            reduced = t_est.cpu().numpy()
        else:
            reduced = mine.copy()
        ctx.sync()
        mc_full = time.perf_counter() - t2
        reduce_ok = None
        if world > 1:
            tt = torch.tensor([mc_full], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mc_full = float(tt.item())
            # untimed check of the reduction: rank 0 adds the ranks' vectors itself
            parts_ = [torch.zeros(ev.size, dtype=torch.float64, device=red_dev) for _ in range(world)]
            dist.all_gather(parts_, torch.from_numpy(mine).to(red_dev))
            want_sum = np.sum([q.cpu().numpy() for q in parts_], axis=0)
            reduce_ok = bool(np.allclose(reduced, want_sum, rtol=1e-12, atol=0))
            assert reduce_ok, "estimator all-reduce disagrees with the sum of the ranks"
        n_diag_got, n_diag_want = float(reduced[ev.fields["n_diag"]][0]), float(world * W * nmc)
        assert n_diag_got == n_diag_want, (n_diag_got, n_diag_want)
        est_summary = {"length": int(ev.size), "n_diag": n_diag_got,
                       # proof that every rank's shard went through the reduction: n_diag == ranks x walkers/rank x steps
                       "n_diag_expected": n_diag_want, "n_diag_check": n_diag_got == n_diag_want,
                       "ranks_reduced": world, "global_walkers": world * W, "sum_equals_sum_of_ranks": reduce_ok,
                       "E_per_particle": float(reduced[ev.fields["E"]][0] / reduced[ev.fields["n_diag"]][0] / cfg.Np),
                       "Et_per_particle": float(reduced[ev.fields["Et"]][0] / reduced[ev.fields["n_diag"]][0] / cfg.Np),
                       "collective": "all_reduce(sum, f64) x1 per block" if world > 1 else "none (one rank)"}
        # the sampler alone on the second walker count (1 024 walkers = 4 per CU): moves only
        mc_large = None
        if args.mc_large_walkers > 0 and world == 1:
            WL2 = args.mc_large_walkers
            ctx2 = api.PigsContext(cfg, VT, WF, n_walkers=WL2, device_id=local)
            try:
                P2, _ = make_workload(cfg, min(WL2, 64), 1, seed=777)
                ctx2.upload_all(np.concatenate([P2] * ((WL2 + len(P2) - 1) // len(P2)))[:WL2])
                del P2
                ctx2.sampler_init(Nlev=mcfg.Nlev, Nstag=mcfg.Nstag, CMFreq=1, Lstag=min(32, args.nb), delta_cm=mcfg.delta_cm_eff)
                for w in range(WL2):
                    ctx2.sampler_seed(w, 5000 + w)
                ctx2.sampler_step(1)
                ctx2.sync()
                n2 = max(2, min(5, nmc))
                ts2 = []
                for q in range(n2):
                    t = time.perf_counter()
                    ctx2.sampler_step(2 + q)
                    ctx2.sync()
                    ts2.append(time.perf_counter() - t)
                mc_large = {"walkers": WL2, "walker_sweeps_per_s": WL2 * n2 / sum(ts2), "ms_per_mc_step": 1e3 * sum(ts2) / n2,
                            "per_step": launch_stats([1e3 * t for t in ts2]), "note": "moves only; four walkers per CU"}
            finally:
                ctx2.close()
        # The same walkers as TWO shards of W/2 on this one GPU (the front end's `n_gpus = 2, same_device = T`; tuning key
        # "cm_shared"): the shards' TranslateChain kernels are chained device-wide and run on the CUs the other shard's
        # bisection phase leaves idle (H = 3 CUs per walker next to 64 sweeping walkers) -- a staggered schedule.  Steps are
        # queued without a synchronisation in between (each shard has its own host thread in the front end); moves only,
        # from the main leg's current worldlines.  Reported next to `moves_only`, which stays the one-context figure.
        mc_two = None
        if world == 1 and W % 2 == 0 and W >= 2:
            n_cu = torch.cuda.get_device_properties(local).multi_processor_count
            Hs = max(1, min(4, (n_cu - W // 2) // (W // 2)))
            P_now = ctx.download_all()

            def run_groups(groups):
                """nmc MC steps (after two warm-up steps) of the walkers in `groups`, one context per group: seconds per step"""
                cl = []
                try:
                    for lo, hi in groups:
                        cs = api.PigsContext(cfg, VT, WF, n_walkers=hi - lo, device_id=local)
                        cl.append(cs)
                        cs.upload_all(P_now[lo:hi])
                        cs.sampler_init(Nlev=mcfg.Nlev, Nstag=mcfg.Nstag, CMFreq=1, Lstag=min(32, args.nb), delta_cm=mcfg.delta_cm_eff)
                        if len(groups) > 1:
                            cs.set_tuning("cm_shared", 1)
                            cs.set_tuning("cm_split", Hs)
                        else:
                            cs.set_tuning("cm_exclusive", 1)       # the main context is idle during this leg
                        for w in range(lo, hi):
                            cs.sampler_seed(w - lo, 7000 + w)
                    for q in range(2):
                        for cs in cl:
                            cs.sampler_step(1 + q)
                    for cs in cl:
                        cs.sync()
                    t = time.perf_counter()
                    for q in range(nmc):
                        for cs in cl:
                            cs.sampler_step(3 + q)
                    for cs in cl:
                        cs.sync()
                    return (time.perf_counter() - t) / nmc
                finally:
                    for cs in cl:
                        cs.close()
            t_one = run_groups([(0, W)])
            t_two = run_groups([(0, W // 2), (W // 2, W)])
            del P_now
            mc_two = {"walkers": W, "shards": 2, "cm_split": Hs, "walker_sweeps_per_s": W / t_two, "ms_per_mc_step": 1e3 * t_two,
                      "one_context_same_start": {"walker_sweeps_per_s": W / t_one, "ms_per_mc_step": 1e3 * t_one},
                      "mc_steps_timed": nmc,
                      "note": "two contexts of W/2 walkers on this GPU, TranslateChain kernels chained device-wide (cm_shared): one "
                              "shard's TranslateChain beside the other's bisection phase; steps queued back to back; "
                              "`one_context_same_start`: the same walkers, seeds and steps in ONE context, timed the same way"}
        # BASELINE config 5's shape on this GPU: N=256, 321 beads, dipolar r^-3 table, worm sector with the stock CWorm = 0.5,
        # Nobdm = 10, swapping, two partial waves -- divergent worm control flow next to the long-tail pair kernel.  Moves only.
        mc_c5 = None
        if world == 1 and not args.no_c5:
            c5 = SystemConfig(dim=3, Np=args.np, Nb=2 * args.nb, density=0.365, dt=5e-3, Rm=1.2, Nlev=4, Nstag=5, Lstag=32,
                              CMFreq=1, delta_cm=0.12)
            VT5, WF5 = api.build_tables(c5, "dipolar")
            ctx5 = api.PigsContext(c5, VT5, WF5, n_walkers=W, device_id=local)
            try:
                P5, _ = make_workload(c5, min(W, 32), 1, seed=555)
                P5 = np.concatenate([P5] * ((W + len(P5) - 1) // len(P5)))[:W]
                ctx5.upload_all(P5)
                ctx5.sampler_init(Nlev=4, Nstag=5, CMFreq=1, Lstag=32, delta_cm=c5.delta_cm_eff, CWorm=0.5, swapping=True,
                                  Nobdm=10, Nbin=100, Npw=2)
                ctx5.set_tuning("cm_exclusive", 1)             # the main context is idle during this leg
                xe5 = np.repeat(P5[:, c5.Nb, c5.Np - 1][:, None, :], 2, axis=1)
                ctx5.sampler_set_worm(np.zeros(W, np.int32), np.zeros(W, np.int32), xe5)
                del P5
                for w in range(W):
                    ctx5.sampler_seed(w, 9000 + w)
                for q in range(6):                              # worms open during the warm-up
                    ctx5.sampler_step(1 + q)
                ctx5.sync()
                k0 = ctx5.sampler_counters16().sum(0)
                n5 = max(2, min(8, nmc))
                ts5 = []
                for q in range(n5):
                    t = time.perf_counter()
                    ctx5.sampler_step(7 + q)
                    ctx5.sync()
                    ts5.append(time.perf_counter() - t)
                k1 = ctx5.sampler_counters16().sum(0) - k0
                mc_c5 = {"workload": "N=%d, %d beads, dipolar table, CWorm=0.5 Nobdm=10 Npw=2 swapping, %d walkers" % (c5.Np, c5.M, W),
                         "walker_sweeps_per_s": W * n5 / sum(ts5), "ms_per_mc_step": 1e3 * sum(ts5) / n5,
                         "per_step": launch_stats([1e3 * t for t in ts5]),
                         "open_walkers_now": int(ctx5.sampler_get_worm()[0].sum()),
                         "worm_events_in_timed_steps": {"open": "%d/%d" % (k1[5], k1[4]), "close": "%d/%d" % (k1[7], k1[6]),
                                                        "swap": "%d/%d" % (k1[13], k1[12])}}
            finally:
                ctx5.close()
        # K2 alone (ThermEnergy of every walker: 2Nb slices x Np(Np-1)/2 pairs each) against the FP64 vector peak
        ctx.therm_energy_batch()
        t3 = time.perf_counter()
        for _ in range(5):
            ctx.therm_energy_batch()
        k2_ms = 1e3 * (time.perf_counter() - t3) / 5
        k2_pairs = W * 2 * cfg.Nb * (cfg.Np * (cfg.Np - 1) // 2)
        k2 = {"bound": "fp64-valu", "kernel": "pigs::k_slice_energy_lds (+ k_therm_combine; host-timed call incl. the result copy)",
              "ms_per_call": k2_ms, "pairs_per_call": k2_pairs, "flops_per_pair": K2_FLOPS_PER_PAIR,
              "achieved": k2_pairs * K2_FLOPS_PER_PAIR / (k2_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": k2_pairs * K2_FLOPS_PER_PAIR / (k2_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
        # Delta-S items of one sweep of one walker: Np*(2Nb+1) + the bisection stages actually run;
        # the exact count depends on early exits, so it is bounded by SURVEY 8d's schedule
        mc = {"walker_sweeps_per_s": world * W * nmc / mc_full, "ms_per_mc_step": 1e3 * mc_full / nmc,
              "mc_steps_timed": nmc, "per_step": launch_stats([1e3 * t for t in ts_full]),
              "moves_only": {"walker_sweeps_per_s": world * W * nmc / mc_el, "ms_per_mc_step": 1e3 * mc_el / nmc,
                             "per_step": launch_stats([1e3 * t for t in ts_moves]),
                             "first_steps": {"ms_per_mc_step": 1e3 * first_steps_s, "walker_sweeps_per_s": W / first_steps_s,
                                             "note": "steps 2-4 after one warm-up step, as round 2 timed them (this rank only)"}},
              "walkers_per_gpu": W, "moves_only_two_shards": mc_two, "moves_only_large": mc_large, "config5": mc_c5,
              "kernels": "pigs::k_sweep (open/close attempt; bisection + worm moves) around pigs::k_cm (TranslateChain on 2 CUs per "
                         "walker while CUs >= 2 x walkers) + k_local_energy x2, k_slice_energy, "
                         "k_therm_combine, k_structure per step",
              "accepted_moves_per_sweep_per_walker": {"cm": acc[0], "head": acc[1], "tail": acc[2], "bisection": acc[3]},
              "schedule": "CMFreq=1 Nstag=5 Nlev=4 sampling=bis CWorm=0 (stock vpi.in), estimators every step",
              "estimator_vector": est_summary,
              # measured in the survey container, NOT on this box: the same-box figure is `cpu_baseline` below
              "reference_program_sweeps_per_s_per_core_survey_container": 1.24,
              "cpu_baseline": None}
        # the sampler's bead-pair rate (16.36 M bead-pair Delta-S evaluations per sweep of one walker at this
        # size and schedule, measured on the reference) against K1's kernel-only rate on the same GPU
        mc_pairs = 16.36e6 * W * nmc / mc_el
        mc["roofline"] = {"bound": "K1 rate (hbm)", "pair_evals_per_sweep_per_walker": 16.36e6, "achieved": mc_pairs,
                          "peak": pair_evals_per_step / (kern_ms * 1e-3), "unit": "bead-pair action evals/s",
                          "frac": mc_pairs / (pair_evals_per_step / (kern_ms * 1e-3)),
                          "hbm_frac_equivalent": mc_pairs * alg_bytes / pair_evals_per_step / 1e9 / HBM_PEAK_GBS}
        mc["roofline_k2"] = k2
    except api.PigsError as exc:       # pragma: no cover
        mc = {"error": str(exc)}

    # correctness of what was timed: rank 0 checks one stage against the oracle
    got = d_out[0].cpu().numpy()
    result = None
    if rank == 0:
        cpu = None
        if not args.no_cpu and world == 1:            # cpu_baseline: rank 0 at N=1 only
            cpu, want = cpu_baseline(cfg, VT, WF, Paths, sets)
            # `want` is the oracle's result for the last stage it ran; recompute stage 0 if different
            from oracle.pyoracle import Oracle, System
            S = System(dim=cfg.dim, Np=cfg.Np, Nb=cfg.Nb)
            w, ip, ib, xnew, xold = sets[0]
            sel = np.arange(0, n_items, 37)
            want = Oracle().delta_action_batch(S, WF, VT, Paths, w[sel], ip[sel], ib[sel], xnew[sel], xold[sel])
            err = np.abs(got[sel] - want) / (np.abs(want) + 1e-9 * np.max(np.abs(want)))
            assert np.all(err < 1e-8), "bench output disagrees with the oracle: %g" % err.max()
            if isinstance(mc, dict) and "error" not in mc:
                try:
                    mc["cpu_baseline"] = cpu_sweep_baseline(cfg.Np, cfg.Nb)
                except Exception as e:                           # noqa: BLE001 -- the baseline is reported, never required
                    mc["cpu_baseline"] = {"error": repr(e)[:300]}
        result = {
            "metric": "bead-pair action evals/sec (K1 Delta-S, N=256 Nb=161 4He, 128 walkers/GPU)",
            "value": value, "unit": "bead-pair action evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "liquid 4He N=%d beads=%d (namelist Nb=%d), %d walkers/GPU x %d GPU(s) = %d walkers, one "
                                   "full-chain Delta-S stage per step (%d items x %d partners per GPU)"
                                   % (cfg.Np, cfg.M, cfg.Nb, W, world, W * world, n_items, cfg.Np - 1),
                       "walkers_per_gpu": W, "global_walkers": W * world, "items_per_step": n_items,
                       "items_per_step_global": n_items * world,
                       "stages_per_sweep_equiv": None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # measured live: kernel_ms (HIP events on the context's stream).  NOT measured in this run:
                         # `traffic` (PMC passes need rocprofv3) and the kernel's name -- both come from the file below
                         "traffic_source": TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
                                           if traffic is not None else None,
                         "kernel": K1_KERNELS.get(args.variant, "pigs::k_delta_action_v2<...> (variant %d)" % args.variant),
                         "kernel_source": "library dispatch rule (pigs_k1.hip launch_delta_action); confirmed by "
                                          "profiles/r03_bench_kernel_stats.csv",
                         "kernel_ms": kern_ms, "per_launch": k1_launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "streaming_read_same_bytes": stream_read},
            "roofline_large": large,
            "cpu_baseline": cpu,
            "kernel_only_evals_per_s": pair_evals_per_step / (kern_ms * 1e-3),
            "mc": mc,
        }
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
