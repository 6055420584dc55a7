"""The reference's MC step loop (vpi.f90:244-545) driven from Python over the reference's OWN movers
and estimators (oracle/ref_probe.f90 -> oracle/_ref/libvpiref.so).  BUILD CONTAINER ONLY; test
infrastructure used by make_golden.py to write fixtures.

Why: the reference PROGRAM prints block energies with 10 digits and builds its potential table from
the one pair potential compiled into it.  Through this driver the same routines (TranslateChain,
Bisection, OpenChain, Swap, LocalEnergy, ThermEnergy, OBDM, grnd ...) run the same schedule with
  * full 64-bit per-step / per-block energies (1e-10 comparisons without a printing floor),
  * an externally filled VTable (dipolar, Lennard-Jones: not active code in the reference),
  * the final generator state, worm state, counters and event log as integers.
Nothing below re-implements a mover or an estimator; only the schedule of calls is restated, and
`validate_against_program` checks that restatement against the stock program (bit-identical final
worldline, block energies equal to the printed digits)."""
import hashlib

import numpy as np


def drive(ref, S, VT, WF, seed, *, Nblock, Nstep, sampling="bis", Lstag=16, Nlev=4, Nstag=5, CMFreq=1,
          delta_cm=0.12, CWorm=0.0, Nobdm=0, swapping=True, Nk=50, progress=None):
    """Returns a dict of arrays.  `S` must carry CWorm / Npw / Nbin (they are module globals of the
    reference); delta_cm is the namelist value (scaled here like vpi.f90:93,126)."""
    assert abs(S.CWorm - CWorm) < 1e-300 or S.CWorm == CWorm
    ref.set_system(S)
    Path, xend = ref.init(seed)
    Np, Nb = S.Np, S.Nb
    if S.trap:
        delta = delta_cm * float(np.min(S.a_ho[:S.dim]))
    else:
        delta = delta_cm / S.density ** (1.0 / float(np.float32(S.dim)))
    isopen, iworm = False, 0
    cnt = dict(acc_cm=0, acc_head=0, acc_tail=0, acc_bd=0, try_open=0, acc_open=0, try_close=0, acc_close=0,
               acc_cm_half=0, acc_head_half=0, acc_tail_half=0, acc_bd_half=0, try_swap=0, acc_swap=0,
               try_cm=0, try_stag=0)
    steps = []          # per MC step: [diag, E, Kin, Pot, Et, Kt]
    events = []         # (global step, code, arg): 1 open accepted (iworm), 2 close accepted, 3 swap accepted (partner)
    rows_e, rows_t = [], []
    nrho_total = np.zeros((S.Nbin, S.Npw + 1))
    gr_total = np.zeros(S.Nbin)
    sk_total = np.zeros((Nk, S.dim))
    head, tail, mid = (("MoveHead", "MoveTail", "Staging") if sampling == "sta"
                       else ("MoveHeadBisection", "MoveTailBisection", "Bisection"))
    par = Lstag if sampling == "sta" else Nlev
    gstep = 0
    ckpt_sha, ckpt_mti, ckpt_cnt, ckpt_nev = [], [], [], []
    for iblock in range(1, Nblock + 1):
        bE = np.zeros(3)
        bT = np.zeros(3)
        idiag_block = 0
        for istep in range(1, Nstep + 1):
            gstep += 1
            iupdate = int(ref.grnd() * 2)
            if isopen:
                if iupdate == 0:
                    isopen, a = ref.close_chain(WF, VT, Lstag, iworm, Path, xend, isopen)
                    cnt["try_close"] += 1
                    cnt["acc_close"] += a
                    if a:
                        events.append((gstep, 2, 0))
            elif iupdate == 1:
                iworm = int(ref.grnd() * Np) + 1
                isopen, a = ref.open_chain(WF, VT, Lstag, iworm, Path, xend, isopen)
                cnt["try_open"] += 1
                cnt["acc_open"] += a
                if a:
                    events.append((gstep, 1, iworm))
            movers = [ip for ip in range(1, Np + 1) if not (isopen and ip == iworm)]
            if istep % CMFreq == 0:
                for ip in movers:
                    cnt["try_cm"] += 1
                    cnt["acc_cm"] += ref.translate_chain(delta, WF, VT, ip, Path)
            for _ in range(Nstag):
                for ip in movers:
                    cnt["try_stag"] += 1
                    cnt["acc_head"] += ref.diag_move(head, WF, VT, par, ip, Path)
                    cnt["acc_tail"] += ref.diag_move(tail, WF, VT, par, ip, Path)
                    cnt["acc_bd"] += ref.diag_move(mid, WF, VT, par, ip, Path)
            if isopen:
                for _ in range(Nobdm):
                    for j in (1, 2):
                        cnt["acc_cm_half"] += ref.half_move("TranslateHalfChain", j, delta, WF, VT, Lstag, iworm, Path, xend)
                    for j in (1, 2):
                        cnt["acc_head_half"] += ref.half_move("MoveHeadHalfChain", j, delta, WF, VT, Lstag, iworm, Path, xend)
                        cnt["acc_tail_half"] += ref.half_move("MoveTailHalfChain", j, delta, WF, VT, Lstag, iworm, Path, xend)
                        cnt["acc_bd_half"] += ref.half_move("StagingHalfChain", j, delta, WF, VT, Lstag, iworm, Path, xend)
                    if swapping:
                        cnt["try_swap"] += 1
                        iworm, ik, swapped, a = ref.swap(WF, VT, Lstag, iworm, Path, xend)
                        cnt["acc_swap"] += a
                        if swapped:
                            events.append((gstep, 3, ik))
                    if not S.trap:
                        nrho_total += ref.obdm(xend)
                steps.append([0, np.nan, np.nan, np.nan, np.nan, np.nan])
            else:
                idiag_block += 1
                E1 = ref.local_energy(WF, VT, Path[0])[0]
                E2 = ref.local_energy(WF, VT, Path[2 * Nb])[0]
                E = 0.5 * (E1 + E2)
                Et, Kt, Pot = ref.therm_energy(VT, Path)
                Kin = E - Pot
                bE += [E, Kin, Pot]
                bT += [Et, Kt, Pot]
                steps.append([1, E, Kin, Pot, Et, Kt])
                if not S.trap:
                    gr_total += ref.pair_correlation(Path[Nb])
                    sk_total += ref.structure_factor(Nk, Path[Nb])
            if progress:
                progress(gstep)
        if idiag_block:
            n = float(np.float32(idiag_block))                      # NormalizeAv divides by real(Nitem)
            rows_e.append([iblock, *(bE / n / Np)])
            rows_t.append([iblock, *(bT / n / Np)])
        # state at the end of every block: a run of the first k blocks must end exactly here (prefix tests of the
        # long trajectories: the host-driven sampler does not have to run all of them)
        ckpt_sha.append(np.frombuffer(hashlib.sha256(np.ascontiguousarray(Path).tobytes()).digest(), np.uint8).copy())
        ckpt_mti.append(int(ref.rng_get_state()[0]))
        ckpt_cnt.append([cnt[k] for k in COUNTER_NAMES])
        ckpt_nev.append(len(events))
    mti, mt = ref.rng_get_state()
    return dict(Path=Path.copy(), xend=xend.copy(), isopen=int(isopen), iworm=int(iworm),
                steps=np.array(steps), block_e=np.array(rows_e).reshape(-1, 4), block_t=np.array(rows_t).reshape(-1, 4),
                counters=np.array([cnt[k] for k in COUNTER_NAMES], np.int64), events=np.array(events, np.int64).reshape(-1, 3),
                nrho_total=nrho_total, gr_total=gr_total, sk_total=sk_total, mti=np.int32(mti), mt=mt.copy(),
                ckpt_sha=np.array(ckpt_sha, np.uint8), ckpt_mti=np.array(ckpt_mti, np.int32),
                ckpt_counters=np.array(ckpt_cnt, np.int64), ckpt_nevents=np.array(ckpt_nev, np.int64))


# order = the 16 counters of the device-resident sampler (pigs_sampler.hip)
COUNTER_NAMES = ["acc_cm", "acc_head", "acc_tail", "acc_bd", "try_open", "acc_open", "try_close", "acc_close",
                 "acc_cm_half", "acc_head_half", "acc_tail_half", "acc_bd_half", "try_swap", "acc_swap", "try_cm",
                 "try_stag"]


def compact(res, bead_stride):
    """Fixture form of a drive() result: the worldline as every bead_stride-th bead + SHA-256 of
    the full array + per-bead coordinate sums (so that no bead goes unchecked), everything else as is."""
    P = res["Path"]
    out = {k: v for k, v in res.items() if k != "Path"}
    out["Path_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(P).tobytes()).digest(), np.uint8)
    out["Path_shape"] = np.array(P.shape)
    out["bead_stride"] = np.int32(bead_stride)
    out["Path_sub"] = P[::bead_stride].copy()
    out["bead_sums"] = P.sum(axis=1)                                # (M, dim)
    out["bead_abs_sums"] = np.abs(P).sum(axis=1)
    return out


def validate_against_program(res, prog_path, e_rows, et_rows):
    """Driver result == stock program run with the same input: bit-identical final worldline, block
    energies equal to the digits the program printed."""
    assert np.array_equal(res["Path"].view(np.uint64), prog_path.view(np.uint64)), "driver diverged from the program"
    for mine, theirs in ((res["block_e"], e_rows), (res["block_t"], et_rows)):
        theirs = np.atleast_2d(theirs)
        assert mine.shape[0] == theirs.shape[0], (mine.shape, theirs.shape)
        if len(mine):
            assert np.all(np.abs(mine[:, 1:] - theirs[:, 1:4]) <= 0.6e-9 * np.abs(theirs[:, 1:4])), (mine, theirs)
