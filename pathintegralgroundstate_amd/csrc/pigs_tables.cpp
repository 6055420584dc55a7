// pigs_tables.cpp -- host-side table fill (stays on the host: SURVEY §8a row a12).
// Our own statement of the pair functions the reference tabulates:
//   Aziz-II HFD-B(HE) He-He potential    (reference system_mod.f90:136-182)
//   McMillan pseudopotential u(r)         (reference system_mod.f90:38-66)
// on the reference's grid r=(i-1)*dr, dr=rmax/real(Nmax-1), with its ghost cells
// (reference vpi_mod.f90:84-145; quirks Q1/Q4 of SURVEY.md §5 are kept, not fixed).
#include "../../include/pigs_hip.h"

#include <cmath>

namespace {

// Reduced units: lengths in sigma = 2.556 A, energies in hbar^2/(m sigma^2).
double aziz2(double r)
{
    const double eps_over_k = 10.948, r_m = 2.963, A = 1.8443101e5;
    const double alpha = 10.43329537, beta = -2.27965105;
    const double c6 = 1.36745214, c8 = 0.42123807, c10 = 0.17473318, D = 1.4826;
    const double v0 = eps_over_k / 1.85505153154686;

    const double x  = r * 2.556 / r_m;
    const double x2 = x * x, x4 = x2 * x2, x6 = x4 * x2;
    double damp = 1.0;
    if (x <= D) {
        const double t = D / x - 1.0;
        damp = std::exp(-(t * t));
    }
    return v0 * (A * std::exp(-alpha * x + beta * x2) - (c6 + c8 / x2 + c10 / x4) * damp / x6);
}

double mcmillan(double Rm, double r)
{
    const double q = Rm / r;
    return -0.5 * (q * q * q * q * q);
}

// Lennard-Jones in units of sigma and hbar^2/(m sigma^2) (reference system_mod.f90:70-83, commented out there)
double lennard_jones(double r)
{
    const double v0 = 22.0228;
    const double r6 = std::pow(r, 6);
    return v0 * (1.0 / r6 - 1.0) / r6;
}

double dipolar(double r) { return 1.0 / (r * r * r); }

} // namespace

extern "C" int pigs_build_tables(int32_t Nmax, double Rm, double rmax, double *VTable, double *LogWF,
                                 double *dr_out)
{
    return pigs_build_tables_kind(PIGS_POT_AZIZ2, Nmax, Rm, rmax, VTable, LogWF, dr_out);
}

extern "C" int pigs_build_tables_kind(int32_t kind, int32_t Nmax, double Rm, double rmax, double *VTable,
                                      double *LogWF, double *dr_out)
{
    if (Nmax < 4 || !(rmax > 0.0) || kind < 0 || kind > 2) return PIGS_ERR_ARG;
    const double dr = rmax / (double)(float)(Nmax - 1);
    for (int i = 1; i <= Nmax; ++i) {
        const double r = (double)(i - 1) * dr;
        if (VTable) VTable[i] = kind == PIGS_POT_LJ ? lennard_jones(r) : kind == PIGS_POT_DIPOLAR ? dipolar(r) : aziz2(r);
        if (LogWF) LogWF[i] = mcmillan(Rm, r);
    }
    if (VTable) { VTable[0] = VTable[2]; VTable[Nmax + 1] = VTable[Nmax]; }
    if (LogWF) { LogWF[0] = LogWF[2]; LogWF[Nmax + 1] = LogWF[Nmax]; }
    if (dr_out) *dr_out = dr;
    return PIGS_OK;
}
