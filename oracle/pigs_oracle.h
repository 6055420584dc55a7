/* pigs_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement (plain C) of the PIGS action / energy hot path of
 * amaciarey/PathIntegralGroundState, written from the algorithm's description in
 * SURVEY.md §8a, same loop and summation order as the reference so that it can be
 * compared bit for bit.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (pathintegralgroundstate_amd)
 * never does.
 *
 * Parity pin: tests/test_oracle_vs_ref.py compares every function below with the
 * unmodified reference compiled by oracle/Makefile into oracle/_ref/libvpiref.so
 * (bit-exact, where that file exists), and tests/test_oracle_golden.py compares it
 * with the committed fixtures under tests/golden/ that were generated from that
 * same reference build by tests/golden/make_golden.py.
 *
 * Conventions follow the reference: arrays are Fortran column-major
 * (Path(dim,Np,0:2*Nb), R(dim,Np)), particle indices ip are 1-based, bead indices
 * ib are 0-based, tables are F(0:Nmax+1) with F pointing at element 0.
 */
#ifndef PIGS_ORACLE_H
#define PIGS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_MAXDIM 3

/* The module globals the hot path reads implicitly
 * (reference global_mod.f90:5-12, system_mod.f90:8-9). */
typedef struct po_sys {
    int32_t dim, Np, Nb, Nmax;
    int32_t trap, wf_table, v_table, pad_;
    double  dr, rcut2, Rm;
    double  Lbox[PO_MAXDIM], LboxHalf[PO_MAXDIM], a_ho[PO_MAXDIM];
} po_sys;

/* ---- numeric primitives ------------------------------------------------ */
double po_interpolate(int opt, int N, double dx, const double *F, double x); /* interpolate.f90:1-45 */
void   po_minimum_image(const po_sys *s, double *xij, double *rij2);         /* pbc_mod.f90:29-52   */
void   po_boundary_conditions(const po_sys *s, int k, double *x);            /* pbc_mod.f90:11-25 (k 0-based) */
double po_green_function(int opt, int ib, int Nb, double dt, double Pot, double F2); /* global_mod.f90:19-72 */

/* ---- physical system (host-side table fill) ----------------------------- */
double po_potential(double rij);                       /* system_mod.f90:136-182 (Aziz-II HFD-B(HE)) */
double po_logpsi(int opt, double Rm, double rij);      /* system_mod.f90:38-66  (McMillan)          */
double po_trap_pot(int opt, double a, double x);       /* system_mod.f90:238-252 */
double po_trap_psi(int opt, double a, double x);       /* system_mod.f90:213-234 */
double po_table_dr(double rmax, int Nmax);             /* vpi_mod.f90:94,127: rmax/real(Nmax-1) */
void   po_potential_table(int Nmax, double rmax, double *VTable); /* vpi_mod.f90:116-145 */
void   po_jastrow_table(int Nmax, double Rm, double rmax, double *WF); /* vpi_mod.f90:84-112 */
double po_box_length(int Np, int dim, double density);  /* vpi.f90:112 */

/* ---- hot path, sampling side (vpi_mod.f90:2491-2841) --------------------- */
void po_update_pot(const po_sys *s, const double *VTable, int ip, const double *R,
                   const double *xnew, const double *xold,
                   double *DeltaPot, double *DeltaF2 /* NULL = absent optional */);
void po_update_wf(const po_sys *s, const double *LogWF, int ip, const double *R,
                  const double *xnew, const double *xold, double *DeltaPsi);
void po_update_action(const po_sys *s, const double *LogWF, const double *VTable,
                      const double *Path, int ip, int ib,
                      const double *xnew, const double *xold, double dt, double *DeltaS);

/* ---- hot path, estimator side (sample_mod.f90:13-388) -------------------- */
void po_potential_energy(const po_sys *s, const double *VTable, const double *R,
                         double *Pot, double *F2 /* NULL = absent optional */);
void po_local_energy(const po_sys *s, const double *LogWF, const double *VTable,
                     const double *R, double *E, double *Kin, double *Pot);
void po_therm_energy(const po_sys *s, const double *VTable, const double *Path, double dt,
                     double *E, double *Ec, double *Ep);

/* ---- structural estimators (sample_mod.f90:392-526) ---------------------- */
void po_pair_correlation(const po_sys *s, int Nbin, double rbin, const double *R, double *gr);
void po_structure_factor(const po_sys *s, int Nk, const double *R, double *Sk);
void po_obdm(const po_sys *s, int Nbin, int Npw, double rbin, const double *xend, double *nrho);

/* ---- batched forms (what the C-ABI boundary of the product computes) ------ */
/* items: walker[i] (0-based), ip[i] (1-based), ib[i] (0-based), xnew/xold (dim,n) col-major.
 * Paths = W worldlines back to back, each (dim,Np,0:2Nb).  Returns pair-evals done. */
int64_t po_delta_action_batch(const po_sys *s, const double *LogWF, const double *VTable,
                              const double *Paths, int64_t n_items,
                              const int32_t *walker, const int32_t *ip, const int32_t *ib,
                              const double *xnew, const double *xold, double dt, double *DeltaS);

/* ---- RNG (random_mod.f90:5-219): MT19937 'mt19937.f' + polar Box-Muller ---- */
typedef struct po_rng { int32_t mti; uint32_t mt[624]; } po_rng;
void   po_sgrnd(po_rng *g, int32_t seed);
double po_grnd(po_rng *g);
void   po_rangauss(po_rng *g, double sigma, double mu, double *x1, double *x2);

#ifdef __cplusplus
}
#endif
#endif
