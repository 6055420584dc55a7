/* pigs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See pigs_oracle.h.
 *
 * Scalar fp64 restatement of the reference hot path, same loop order and the same
 * left-to-right expression order as the Fortran (compile with -ffp-contract=off:
 * the reference x86-64 build has no FMA).  Every function cites the reference
 * file:line it follows.  Quirk numbers (Q1..Q15) refer to SURVEY.md §5.
 */
#include "pigs_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* interpolate.f90:1-45.  F(0:N+1); F(i) is *treated* as the value at r=i*dx
 * although the builders fill it at (i-1)*dx (Q1: kept, not fixed).          */
double po_interpolate(int opt, int N, double dx, const double *F, double x)
{
    int    ix   = (int)(x / dx) + 1;            /* interpolate.f90:13 */
    double aux1 = x - (double)(ix - 1) * dx;    /* :14 */
    double aux2 = dx - aux1;                    /* :15 */
    /* Beyond the table's last cell the reference reads F out of bounds: undefined.  Only a TRAPPED system gets there (no
     * cutoff: a pair farther apart than rcut = 30 a_ho, vpi.f90:84-92 -- round 3's fuzz with TranslateChain shifts of 9 a_ho).
     * The product clamps the cell index (pigs_device.h lerp_setup / flerp_setup) and so does this checker: an identity for
     * every r <= rcut, i.e. wherever the reference is defined. */
    if (ix > N) ix = N;

    if (opt == 0) {
        return (aux1 * F[ix] + aux2 * F[ix - 1]) / dx;                 /* :21 */
    } else if (opt == 1) {
        double Fbefore = (aux1 * F[ix - 1] + aux2 * F[ix - 2]) / dx;   /* :25 */
        double Fafter  = (aux1 * F[ix + 1] + aux2 * F[ix]) / dx;       /* :26 */
        return 0.5 * (Fafter - Fbefore) / dx;                          /* :28 */
    } else if (opt == 2) {
        double Fbefore = (aux1 * F[ix - 1] + aux2 * F[ix - 2]) / dx;   /* :32 */
        double Fcurr   = (aux1 * F[ix] + aux2 * F[ix - 1]) / dx;       /* :33 */
        double Fafter  = (aux1 * F[ix + 1] + aux2 * F[ix]) / dx;       /* :34 */
        return (Fafter - 2.0 * Fcurr + Fbefore) / (dx * dx);           /* :36 */
    }
    return 0.0;
}

/* pbc_mod.f90:29-52: single wrap per coordinate, mutates xij (Q14). */
void po_minimum_image(const po_sys *s, double *xij, double *rij2)
{
    double r2 = 0.0;
    for (int k = 0; k < s->dim; ++k) {
        if (xij[k] >  s->LboxHalf[k]) xij[k] = xij[k] - s->Lbox[k];
        if (xij[k] < -s->LboxHalf[k]) xij[k] = xij[k] + s->Lbox[k];
        r2 = r2 + xij[k] * xij[k];
    }
    *rij2 = r2;
}

/* pbc_mod.f90:11-25 */
void po_boundary_conditions(const po_sys *s, int k, double *x)
{
    if (*x >  s->LboxHalf[k]) *x = *x - s->Lbox[k];
    if (*x < -s->LboxHalf[k]) *x = *x + s->Lbox[k];
}

/* global_mod.f90:19-72: Chin-action bead weights (opt 0) and their d/dt form (opt 1). */
double po_green_function(int opt, int ib, int Nb, double dt, double Pot, double F2)
{
    double g = 0.0;
    if (opt == 0) {
        double Ve = Pot;
        double Vc = Pot + dt * dt * F2 / 6.0;                       /* :34 */
        if (ib == 0)             g = dt * Ve / 3.0;                 /* :37 */
        else if (ib == 2 * Nb)   g = dt * Ve / 3.0;                 /* :39 */
        else if (ib % 2 == 0)    g = 2.0 * dt * Ve / 3.0;           /* :42 */
        else                     g = 4.0 * dt * Vc / 3.0;           /* :44 */
    } else if (opt == 1) {
        double dVe = Pot;
        double dVc = Pot + dt * dt * F2 / 2.0;                      /* :53 */
        if (ib == 0)             g = dVe / 3.0;
        else if (ib == 2 * Nb)   g = dVe / 3.0;
        else if (ib % 2 == 0)    g = 2.0 * dVe / 3.0;
        else                     g = 4.0 * dVc / 3.0;
    }
    return g;
}

/* ------------------------------------------------------------------------ */
/* system_mod.f90:136-182: Aziz-II HFD-B(HE), reduced units (sigma=2.556 A). */
double po_potential(double rij)
{
    const double E_0 = 10.948, rm = 2.963, A = 1.8443101e5;
    const double alpha = 10.43329537, beta = -2.27965105;
    const double C6 = 1.36745214, C8 = 0.42123807, C10 = 0.17473318, D = 1.4826;
    const double V0 = E_0 / 1.85505153154686;

    double dij  = rij * 2.556 / rm;
    double dij2 = dij * dij;
    double dij4 = dij2 * dij2;
    double dij6 = dij4 * dij2;
    double Hx;
    if (dij <= D) {
        double t = D / dij - 1.0;
        Hx = exp(-(t * t));
    } else {
        Hx = 1.0;
    }
    return V0 * (A * exp(-alpha * dij + beta * dij2) - (C6 + C8 / dij2 + C10 / dij4) * Hx / dij6);
}

static double ipow5(double q)
{
    /* x**5 as the reference's flang -O2 build evaluates it: a left-to-right product
     * (pinned by tests/test_oracle_vs_ref.py; square-and-multiply differs in the last ulp) */
    return q * q * q * q * q;
}

/* system_mod.f90:38-66: McMillan u(r) = -0.5 (Rm/r)^5 and derivatives. */
double po_logpsi(int opt, double Rm, double rij)
{
    if (opt == 0) return -0.5 * ipow5(Rm / rij);
    if (opt == 1) return 2.5 * ipow5(Rm / rij) / rij;
    if (opt == 2) return -15.0 * ipow5(Rm / rij) / (rij * rij);
    return 0.0;
}

/* system_mod.f90:238-252 */
double po_trap_pot(int opt, double a, double x)
{
    /* a_osc**4 as the reference's flang -O2 build evaluates it: left-to-right product */
    double a4 = a * a * a * a;
    if (opt == 0) return 0.5 * (x * x) / a4;
    if (opt == 1) return x / a4;
    return 0.0;
}

/* system_mod.f90:213-234 */
double po_trap_psi(int opt, double a, double x)
{
    if (opt == 0) { double q = x / a; return -0.5 * (q * q); }
    if (opt == 1) return -(x / (a * a));
    if (opt == 2) return -1.0 / (a * a);
    return 0.0;
}

/* vpi_mod.f90:94,127: dr = rmax/real(Nmax-1); real() is single precision but the
 * value is an exactly representable small integer. */
double po_table_dr(double rmax, int Nmax)
{
    return rmax / (double)(float)(Nmax - 1);
}

/* vpi_mod.f90:116-145: r=(i-1)*dr, i=1..Nmax; ghost cells F(0)=F(2), F(Nmax+1)=F(Nmax) (Q4). */
void po_potential_table(int Nmax, double rmax, double *VT)
{
    double dr = po_table_dr(rmax, Nmax);
    for (int i = 1; i <= Nmax; ++i) {
        double r = (double)(i - 1) * dr;
        VT[i] = po_potential(r);
    }
    VT[0]        = VT[2];
    VT[Nmax + 1] = VT[Nmax];
}

/* vpi_mod.f90:84-112 */
void po_jastrow_table(int Nmax, double Rm, double rmax, double *WF)
{
    double dr = po_table_dr(rmax, Nmax);
    for (int i = 1; i <= Nmax; ++i) {
        double r = (double)(i - 1) * dr;
        WF[i] = po_logpsi(0, Rm, r);
    }
    WF[0]        = WF[2];
    WF[Nmax + 1] = WF[Nmax];
}

/* vpi.f90:112: Lbox = (real(Np)/density)**(1.d0/real(dim)) */
double po_box_length(int Np, int dim, double density)
{
    return pow((double)(float)Np / density, 1.0 / (double)(float)dim);
}

/* ------------------------------------------------------------------------ */
/* vpi_mod.f90:2660-2841.  ip 1-based; R(dim,Np) column-major.  Row ip of R is
 * never read (aliasing contract).  Only the moved particle's force
 * enters DeltaF2 (Q2); trap branch sums PotNew with no cutoff (Q5); the
 * derivative lookup is re-evaluated per k (Q6: same value each time).          */
void po_update_pot(const po_sys *s, const double *VT, int ip, const double *R,
                   const double *xnew, const double *xold, double *DeltaPot, double *DeltaF2)
{
    const int dim = s->dim, Np = s->Np, Nmax = s->Nmax;
    const double dr = s->dr, rcut2 = s->rcut2;
    double PotNew = 0.0, PotOld = 0.0;
    double Fnew[PO_MAXDIM] = {0, 0, 0}, Fold[PO_MAXDIM] = {0, 0, 0};
    double xijnew[PO_MAXDIM], xijold[PO_MAXDIM];

    if (s->trap) {                                            /* :2688-2695 */
        for (int k = 0; k < dim; ++k) {
            PotNew  = PotNew + po_trap_pot(0, s->a_ho[k], xnew[k]);
            PotOld  = PotOld + po_trap_pot(0, s->a_ho[k], xold[k]);
            Fold[k] = po_trap_pot(1, s->a_ho[k], xold[k]);
            Fnew[k] = po_trap_pot(1, s->a_ho[k], xnew[k]);
        }
    }

    for (int jp = 1; jp <= Np; ++jp) {                        /* :2697 */
        if (jp == ip) continue;                               /* :2699 */
        const double *Rj = R + (size_t)(jp - 1) * dim;
        double rijnew2 = 0.0, rijold2 = 0.0;
        for (int k = 0; k < dim; ++k) {                       /* :2704-2709 */
            xijnew[k] = xnew[k] - Rj[k];
            xijold[k] = xold[k] - Rj[k];
        }
        if (s->trap) {                                        /* :2711-2717 */
            for (int k = 0; k < dim; ++k) {
                rijold2 = rijold2 + xijold[k] * xijold[k];
                rijnew2 = rijnew2 + xijnew[k] * xijnew[k];
            }
        } else {                                              /* :2719-2720 */
            po_minimum_image(s, xijold, &rijold2);
            po_minimum_image(s, xijnew, &rijnew2);
        }

        int do_new = s->trap ? 1 : (rijnew2 <= rcut2);        /* :2723 (Q5) / :2771 */
        if (do_new) {
            double rijnew = sqrt(rijnew2);
            if (s->v_table) PotNew = PotNew + po_interpolate(0, Nmax, dr, VT, rijnew);
            else            PotNew = PotNew + po_potential(rijnew);
            if (DeltaF2) {
                for (int k = 0; k < dim; ++k)                 /* :2783-2785 */
                    Fnew[k] = Fnew[k] + po_interpolate(1, Nmax, dr, VT, rijnew) * xijnew[k] / rijnew;
            }
        }
        if (rijold2 <= rcut2) {                               /* :2745 / :2795 */
            double rijold = sqrt(rijold2);
            if (s->v_table) PotOld = PotOld + po_interpolate(0, Nmax, dr, VT, rijold);
            else            PotOld = PotOld + po_potential(rijold);
            if (DeltaF2) {
                for (int k = 0; k < dim; ++k)                 /* :2807-2809 */
                    Fold[k] = Fold[k] + po_interpolate(1, Nmax, dr, VT, rijold) * xijold[k] / rijold;
            }
        }
    }

    if (DeltaF2) {                                            /* :2825-2836 */
        double Fnew2 = 0.0, Fold2 = 0.0;
        for (int k = 0; k < dim; ++k) {
            Fnew2 = Fnew2 + Fnew[k] * Fnew[k];
            Fold2 = Fold2 + Fold[k] * Fold[k];
        }
        *DeltaF2 = Fnew2 - Fold2;
    }
    *DeltaPot = PotNew - PotOld;                              /* :2838 */
}

/* vpi_mod.f90:2534-2656 */
void po_update_wf(const po_sys *s, const double *LogWF, int ip, const double *R,
                  const double *xnew, const double *xold, double *DeltaPsi)
{
    const int dim = s->dim, Np = s->Np, Nmax = s->Nmax;
    const double dr = s->dr, rcut2 = s->rcut2;
    double PsiOld = 0.0, PsiNew = 0.0;
    double xijnew[PO_MAXDIM], xijold[PO_MAXDIM];

    if (s->trap) {                                            /* :2555-2560 */
        for (int k = 0; k < dim; ++k) {
            PsiOld = PsiOld + po_trap_psi(0, s->a_ho[k], xold[k]);
            PsiNew = PsiNew + po_trap_psi(0, s->a_ho[k], xnew[k]);
        }
    }

    for (int jp = 1; jp <= Np; ++jp) {
        if (jp == ip) continue;
        const double *Rj = R + (size_t)(jp - 1) * dim;
        double rijold2 = 0.0, rijnew2 = 0.0;
        for (int k = 0; k < dim; ++k) {
            xijold[k] = xold[k] - Rj[k];
            xijnew[k] = xnew[k] - Rj[k];
        }
        if (s->trap) {
            for (int k = 0; k < dim; ++k) {
                rijold2 = rijold2 + xijold[k] * xijold[k];
                rijnew2 = rijnew2 + xijnew[k] * xijnew[k];
            }
        } else {
            po_minimum_image(s, xijnew, &rijnew2);            /* :2587-2588 */
            po_minimum_image(s, xijold, &rijold2);
        }
        if (s->trap || rijold2 <= rcut2) {                    /* :2595 / :2619 */
            double rijold = sqrt(rijold2);
            double ur = s->wf_table ? po_interpolate(0, Nmax, dr, LogWF, rijold)
                                    : po_logpsi(0, s->Rm, rijold);
            PsiOld = PsiOld + ur;
        }
        if (s->trap || rijnew2 <= rcut2) {                    /* :2607 / :2633 */
            double rijnew = sqrt(rijnew2);
            double ur = s->wf_table ? po_interpolate(0, Nmax, dr, LogWF, rijnew)
                                    : po_logpsi(0, s->Rm, rijnew);
            PsiNew = PsiNew + ur;
        }
    }
    *DeltaPsi = PsiNew - PsiOld;                              /* :2653 */
}

/* vpi_mod.f90:2491-2530 */
void po_update_action(const po_sys *s, const double *LogWF, const double *VT,
                      const double *Path, int ip, int ib,
                      const double *xnew, const double *xold, double dt, double *DeltaS)
{
    const double *R = Path + (size_t)ib * s->dim * s->Np;     /* Path(:,:,ib) */
    double DeltaPot, DeltaF2, DeltaLogPsi;

    if (ib % 2 == 0) {                                        /* :2509-2514 */
        po_update_pot(s, VT, ip, R, xnew, xold, &DeltaPot, NULL);
        DeltaF2 = 0.0;
    } else {
        po_update_pot(s, VT, ip, R, xnew, xold, &DeltaPot, &DeltaF2);
    }
    if (ib == 0 || ib == 2 * s->Nb)                           /* :2519-2525 */
        po_update_wf(s, LogWF, ip, R, xnew, xold, &DeltaLogPsi);
    else
        DeltaLogPsi = 0.0;

    *DeltaS = -DeltaLogPsi + po_green_function(0, ib, s->Nb, dt, DeltaPot, DeltaF2); /* :2527 */
}

/* ------------------------------------------------------------------------ */
/* sample_mod.f90:13-150.  F is the full per-particle force field here (Q2).  */
void po_potential_energy(const po_sys *s, const double *VT, const double *R,
                         double *Pot_out, double *F2)
{
    const int dim = s->dim, Np = s->Np, Nmax = s->Nmax;
    const double dr = s->dr, rcut2 = s->rcut2;
    double Pot = 0.0;
    double F[PO_MAXDIM * 4096];
    double *Fp = F;
    double xij[PO_MAXDIM];
    /* Np is bounded by the fixed scratch above in this test-only code */
    if (Np > 4096) { *Pot_out = NAN; if (F2) *F2 = NAN; return; }

    for (int ip = 0; ip < Np; ++ip)                           /* :33-42 */
        for (int k = 0; k < dim; ++k) {
            if (s->trap) {
                Fp[ip * dim + k] = po_trap_pot(1, s->a_ho[k], R[ip * dim + k]);
                Pot = Pot + po_trap_pot(0, s->a_ho[k], R[ip * dim + k]);
            } else {
                Fp[ip * dim + k] = 0.0;
            }
        }

    for (int ip = 0; ip < Np - 1; ++ip) {                     /* :44-135 */
        for (int jp = ip + 1; jp < Np; ++jp) {
            double rij2 = 0.0;
            for (int k = 0; k < dim; ++k) xij[k] = R[ip * dim + k] - R[jp * dim + k];
            if (s->trap) {
                for (int k = 0; k < dim; ++k) rij2 = rij2 + xij[k] * xij[k];
            } else {
                po_minimum_image(s, xij, &rij2);
            }
            if (s->trap || rij2 <= rcut2) {                   /* :64 / :98 */
                double rij = sqrt(rij2);
                if (s->v_table) Pot = Pot + po_interpolate(0, Nmax, dr, VT, rij);
                else            Pot = Pot + po_potential(rij);
                if (F2) {
                    for (int k = 0; k < dim; ++k) {           /* :112-116 */
                        double fij = po_interpolate(1, Nmax, dr, VT, rij) * xij[k] / rij;
                        Fp[ip * dim + k] = Fp[ip * dim + k] + fij;
                        Fp[jp * dim + k] = Fp[jp * dim + k] - fij;
                    }
                }
            }
        }
    }

    if (F2) {                                                 /* :137-147 */
        double f2 = 0.0;
        for (int ip = 0; ip < Np; ++ip)
            for (int k = 0; k < dim; ++k) f2 = f2 + Fp[ip * dim + k] * Fp[ip * dim + k];
        *F2 = f2;
    }
    *Pot_out = Pot;
}

/* sample_mod.f90:154-319 */
void po_local_energy(const po_sys *s, const double *LogWF, const double *VT,
                     const double *R, double *E, double *Kin_out, double *Pot_out)
{
    const int dim = s->dim, Np = s->Np, Nmax = s->Nmax;
    const double dr = s->dr, rcut2 = s->rcut2;
    double Kin = 0.0, Pot = 0.0, LapLogPsi = 0.0;
    double F[PO_MAXDIM * 4096];
    double xij[PO_MAXDIM];
    if (Np > 4096) { *E = *Kin_out = *Pot_out = NAN; return; }
    /* (real(dim)-1): single-precision arithmetic on small integers, exact */
    const double dimm1 = (double)((float)dim - 1.0f);

    for (int i = 0; i < Np; ++i)                              /* :177-187 */
        for (int k = 0; k < dim; ++k) {
            if (s->trap) {
                F[i * dim + k] = po_trap_psi(1, s->a_ho[k], R[i * dim + k]);
                Pot            = Pot + po_trap_pot(0, s->a_ho[k], R[i * dim + k]);
                LapLogPsi      = LapLogPsi + po_trap_psi(2, s->a_ho[k], R[i * dim + k]);
            } else {
                F[i * dim + k] = 0.0;
            }
        }
    LapLogPsi = 0.5 * LapLogPsi;                              /* :189 */

    for (int i = 0; i < Np - 1; ++i) {
        for (int j = i + 1; j < Np; ++j) {
            double rij2 = 0.0;
            for (int k = 0; k < dim; ++k) xij[k] = R[i * dim + k] - R[j * dim + k];
            if (s->trap) {
                for (int k = 0; k < dim; ++k) rij2 = rij2 + xij[k] * xij[k];
            } else {
                po_minimum_image(s, xij, &rij2);
            }
            if (s->trap || rij2 <= rcut2) {                   /* :230 / :264 */
                double rij = sqrt(rij2);
                double dudr, d2udr2;
                if (s->wf_table) {
                    dudr   = po_interpolate(1, Nmax, dr, LogWF, rij);
                    d2udr2 = po_interpolate(2, Nmax, dr, LogWF, rij);
                } else {
                    dudr   = po_logpsi(1, s->Rm, rij);
                    d2udr2 = po_logpsi(2, s->Rm, rij);
                }
                LapLogPsi = LapLogPsi + (dimm1 * dudr / rij + d2udr2);   /* :280 */
                for (int k = 0; k < dim; ++k) {                          /* :282-286 */
                    double fij = dudr * xij[k] / rij;
                    F[i * dim + k] = F[i * dim + k] + fij;
                    F[j * dim + k] = F[j * dim + k] - fij;
                }
                if (s->v_table) Pot = Pot + po_interpolate(0, Nmax, dr, VT, rij);
                else            Pot = Pot + po_potential(rij);
            }
        }
    }

    Kin = 2.0 * LapLogPsi;                                    /* :305 */
    for (int i = 0; i < Np; ++i)
        for (int k = 0; k < dim; ++k) Kin = Kin + F[i * dim + k] * F[i * dim + k];
    Kin = -0.5 * Kin;                                         /* :315 */
    *E = Kin + Pot;
    *Kin_out = Kin;
    *Pot_out = Pot;
}

/* sample_mod.f90:323-388: slice 2Nb is skipped; spring term guarded by rcut2 (Q8). */
void po_therm_energy(const po_sys *s, const double *VT, const double *Path, double dt,
                     double *E_out, double *Ec, double *Ep_out)
{
    const int dim = s->dim, Np = s->Np, Nb = s->Nb;
    const size_t slice = (size_t)dim * Np;
    double E = 0.0, Ep = 0.0;
    double xij[PO_MAXDIM];

    for (int ib = 0; ib <= 2 * Nb - 1; ++ib) {                /* :344 */
        const double *R  = Path + slice * ib;
        const double *R1 = Path + slice * (ib + 1);
        double Pot, F2;
        if (ib % 2 == 0) { po_potential_energy(s, VT, R, &Pot, NULL); F2 = 0.0; }
        else             { po_potential_energy(s, VT, R, &Pot, &F2); }
        if (ib == Nb) Ep = Pot;                               /* :353 */
        E = E + po_green_function(1, ib, Nb, dt, Pot, F2);    /* :357 */
        for (int ip = 0; ip < Np; ++ip) {                     /* :359-380 */
            double rij2 = 0.0;
            for (int k = 0; k < dim; ++k) xij[k] = R[ip * dim + k] - R1[ip * dim + k];
            if (s->trap) {
                for (int k = 0; k < dim; ++k) rij2 = rij2 + xij[k] * xij[k];
                E = E - 0.5 * rij2 / (dt * dt);
            } else {
                po_minimum_image(s, xij, &rij2);
                if (rij2 <= s->rcut2) E = E - 0.5 * rij2 / (dt * dt);
            }
        }
    }
    /* :384: real(Nb), real(dim*Np) are single precision, exact for these integers */
    E = 0.5 * (E / (double)(float)Nb + (double)(float)(dim * Np) / dt);
    *E_out  = E;
    *Ec     = E - Ep;
    *Ep_out = Ep;
}

/* ------------------------------------------------------------------------ */
/* sample_mod.f90:392-428 */
void po_pair_correlation(const po_sys *s, int Nbin, double rbin, const double *R, double *gr)
{
    const int dim = s->dim, Np = s->Np;
    double xij[PO_MAXDIM];
    (void)Nbin;
    for (int ip = 0; ip < Np - 1; ++ip)
        for (int jp = ip + 1; jp < Np; ++jp) {
            double rij2;
            for (int k = 0; k < dim; ++k) xij[k] = R[ip * dim + k] - R[jp * dim + k];
            po_minimum_image(s, xij, &rij2);
            if (rij2 <= s->rcut2) {
                double rij = sqrt(rij2);
                int ibin = (int)(rij / rbin) + 1;             /* 1-based bin */
                gr[ibin - 1] = gr[ibin - 1] + 2.0;
            }
        }
}

/* sample_mod.f90:432-473: qbin(k)=2*pi/Lbox(k) (vpi.f90:119), pi=acos(-1). */
void po_structure_factor(const po_sys *s, int Nk, const double *R, double *Sk)
{
    const int dim = s->dim, Np = s->Np;
    const double pi = acos(-1.0);
    for (int iq = 1; iq <= Nk; ++iq)
        for (int k = 0; k < dim; ++k) {
            double qbin = 2.0 * pi / s->Lbox[k];
            double SumCos = 0.0, SumSin = 0.0;
            for (int ip = 0; ip < Np; ++ip) {
                double qr = (double)(float)iq * qbin * R[ip * dim + k];
                SumCos = SumCos + cos(qr);
                SumSin = SumSin + sin(qr);
            }
            Sk[(iq - 1) * dim + k] = Sk[(iq - 1) * dim + k] + (SumCos * SumCos + SumSin * SumSin);
        }
}

/* sample_mod.f90:477-526: nrho(0:Npw,Nbin) column-major. */
void po_obdm(const po_sys *s, int Nbin, int Npw, double rbin, const double *xend, double *nrho)
{
    const int dim = s->dim;
    double xij[PO_MAXDIM] = {0, 0, 0};
    double rij2;
    (void)Nbin;
    for (int k = 0; k < dim; ++k) xij[k] = xend[k] - xend[dim + k];
    po_minimum_image(s, xij, &rij2);
    if (rij2 <= s->rcut2) {
        double rij  = sqrt(rij2);
        int    ibin = (int)(rij / rbin) + 1;
        double sint = (dim > 1 ? xij[1] : 0.0) / rij;
        double cost = xij[0] / rij;
        /* exp2theta = exptheta*exptheta (complex multiply) */
        double e2r = cost * cost - sint * sint;
        double e2i = cost * sint + sint * cost;
        double er = 1.0, ei = 0.0;
        for (int m = 0; m <= Npw; ++m) {
            nrho[(size_t)(ibin - 1) * (Npw + 1) + m] += er;
            double nr = er * e2r - ei * e2i;
            double ni = er * e2i + ei * e2r;
            er = nr; ei = ni;
        }
    }
}

/* ------------------------------------------------------------------------ */
int64_t po_delta_action_batch(const po_sys *s, const double *LogWF, const double *VT,
                              const double *Paths, int64_t n_items,
                              const int32_t *walker, const int32_t *ip, const int32_t *ib,
                              const double *xnew, const double *xold, double dt, double *DeltaS)
{
    const size_t wl = (size_t)s->dim * s->Np * (2 * (size_t)s->Nb + 1);
    for (int64_t i = 0; i < n_items; ++i) {
        po_update_action(s, LogWF, VT, Paths + wl * (size_t)walker[i], ip[i], ib[i],
                         xnew + (size_t)i * s->dim, xold + (size_t)i * s->dim, dt, &DeltaS[i]);
    }
    return n_items * (int64_t)(s->Np - 1);
}

/* ------------------------------------------------------------------------ */
/* random_mod.f90:5-31: seeding by the 69069 LCG, 32-bit wrap-around. */
void po_sgrnd(po_rng *g, int32_t seed)
{
    g->mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 69069u * g->mt[i - 1];
    g->mti = 624;
}

/* random_mod.f90:35-115: MT19937 word; real in [0,1] = y/(2^32-1) (Q15). */
double po_grnd(po_rng *g)
{
    enum { N = 624, M = 397 };
    const uint32_t MATA = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;
    uint32_t y;
    if (g->mti >= N) {
        int kk;
        for (kk = 0; kk < N - M; ++kk) {
            y = (g->mt[kk] & UMASK) | (g->mt[kk + 1] & LMASK);
            g->mt[kk] = g->mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? MATA : 0u);
        }
        for (; kk < N - 1; ++kk) {
            y = (g->mt[kk] & UMASK) | (g->mt[kk + 1] & LMASK);
            g->mt[kk] = g->mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MATA : 0u);
        }
        y = (g->mt[N - 1] & UMASK) | (g->mt[0] & LMASK);
        g->mt[N - 1] = g->mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATA : 0u);
        g->mti = 0;
    }
    y = g->mt[g->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    /* signed y<0 branch of the Fortran adds 2^32: both branches = unsigned value */
    return (double)y / (4294967296.0 - 1.0);
}

/* random_mod.f90:195-219: polar Box-Muller, two deviates. */
void po_rangauss(po_rng *g, double sigma, double mu, double *x1, double *x2)
{
    double u1, u2, w;
    do {
        u1 = 2.0 * po_grnd(g) - 1.0;
        u2 = 2.0 * po_grnd(g) - 1.0;
        w  = u1 * u1 + u2 * u2;
    } while (!(w <= 1.0));
    w   = sqrt((-2.0 * log(w)) / w);
    *x1 = mu + sigma * u1 * w;
    *x2 = mu + sigma * u2 * w;
}
