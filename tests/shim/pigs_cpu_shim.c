/* pigs_cpu_shim.c -- TEST INFRASTRUCTURE ONLY (lives under tests/, never shipped, never
 * loaded by the product).  Implements the C ABI of include/pigs_hip.h on the CPU with the
 * pinned oracle (oracle/pigs_oracle.c) so that the HOST LOGIC of the Fortran sampler
 * (RNG streams, proposal generation, accept/reject bookkeeping, output files) can be
 * tested in a container without a GPU against runs of the reference program.
 * It is not a fallback: the product library libpigs_hip.so has no CPU path, and nothing
 * outside tests/ links this file.  Parity claims for the kernels never rest on it.      */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pigs_hip.h"
#include "../../oracle/pigs_oracle.h"

struct pigs_ctx {
    po_sys  s;
    double  dt;
    int     W;
    size_t  wl;        /* doubles per worldline */
    double *VT, *WF, *paths;
    int32_t *st_w, *st_ip, *st_ib, *cs_w, *cs_ip, *cs_ib;
    double  *st_xn, *st_xo, *st_out, *cs_x;
    int64_t st_cap, cs_cap;
    void   *group;     /* shim_group of pigs_comm_init_all */
    int     grank;
    long    n_allreduce;   /* calls of pigs_estimators_allreduce on this context (PIGS_SHIM_TRACE) */
    long    n_allreduce_nd0; /* ... of which with vec[0] == 0: a block in which this shard had no diagonal step */
    double *pend_en, *pend_gr, *pend_sk; int pend_n, pend_nb, pend_nk, pend_st;   /* pigs_diagonal_estimators_begin / _end */
};

static char g_err[256] = "";
const char *pigs_last_error(void) { return g_err; }
int pigs_abi_version(void) { return PIGS_ABI_VERSION; }
int pigs_device_count(int32_t *n) { *n = 1; return PIGS_OK; }
int pigs_sync(pigs_ctx *c) { (void)c; return PIGS_OK; }
int pigs_stream(pigs_ctx *c, void **s) { (void)c; *s = NULL; return PIGS_OK; }
int pigs_set_tuning(pigs_ctx *c, const char *k, int32_t v) { (void)c; (void)k; (void)v; return PIGS_OK; }

int pigs_ctx_create(const pigs_params *p, const double *VT, const double *WF, int32_t W, int32_t dev, pigs_ctx **out)
{
    (void)dev;
    pigs_ctx *c = calloc(1, sizeof *c);
    c->s.dim = p->dim; c->s.Np = p->Np; c->s.Nb = p->Nb; c->s.Nmax = p->Nmax;
    c->s.trap = p->trap; c->s.wf_table = p->wf_table; c->s.v_table = p->v_table;
    c->s.dr = p->dr; c->s.rcut2 = p->rcut2; c->s.Rm = p->Rm;
    for (int k = 0; k < 3; ++k) {
        c->s.Lbox[k] = k < p->dim ? p->Lbox[k] : 1.0;
        c->s.LboxHalf[k] = 0.5 * c->s.Lbox[k];
        c->s.a_ho[k] = k < p->dim ? p->a_ho[k] : 1.0;
    }
    c->dt = p->dt; c->W = W;
    c->wl = (size_t)p->dim * p->Np * (2 * (size_t)p->Nb + 1);
    size_t tb = (size_t)(p->Nmax + 2) * sizeof(double);
    c->VT = malloc(tb); c->WF = malloc(tb);
    memcpy(c->VT, VT, tb); memcpy(c->WF, WF, tb);
    c->paths = calloc(c->wl * W, sizeof(double));
    *out = c;
    return PIGS_OK;
}

int pigs_ctx_destroy(pigs_ctx *c)
{
    if (!c) return PIGS_OK;
    if (getenv("PIGS_SHIM_TRACE"))
        fprintf(stderr, "shim: context rank %d: %ld all-reduce calls, %ld with no diagonal step in the shard\n", c->grank,
                c->n_allreduce, c->n_allreduce_nd0);
    free(c->pend_en); free(c->pend_gr); free(c->pend_sk);
    free(c->VT); free(c->WF); free(c->paths); free(c);
    return PIGS_OK;
}

int pigs_build_tables(int32_t Nmax, double Rm, double rmax, double *VT, double *WF, double *dr)
{
    if (VT) po_potential_table(Nmax, rmax, VT);
    if (WF) po_jastrow_table(Nmax, Rm, rmax, WF);
    if (dr) *dr = po_table_dr(rmax, Nmax);
    return PIGS_OK;
}

int pigs_build_tables_kind(int32_t kind, int32_t Nmax, double Rm, double rmax, double *VT, double *WF, double *dr)
{
    int rc = pigs_build_tables(Nmax, Rm, rmax, VT, WF, dr);
    if (kind != 0 && VT) {
        const double h = po_table_dr(rmax, Nmax);
        for (int i = 1; i <= Nmax; ++i) {
            const double r = (double)(i - 1) * h;
            VT[i] = kind == 1 ? 22.0228 * (1.0 / pow(r, 6) - 1.0) / pow(r, 6) : 1.0 / (r * r * r);
        }
        VT[0] = VT[2]; VT[Nmax + 1] = VT[Nmax];
    }
    return rc;
}

int pigs_path_upload(pigs_ctx *c, int32_t w, const double *P) { memcpy(c->paths + c->wl * w, P, c->wl * sizeof(double)); return PIGS_OK; }
int pigs_path_download(pigs_ctx *c, int32_t w, double *P) { memcpy(P, c->paths + c->wl * w, c->wl * sizeof(double)); return PIGS_OK; }
int pigs_path_upload_all(pigs_ctx *c, const double *P) { memcpy(c->paths, P, c->wl * c->W * sizeof(double)); return PIGS_OK; }
int pigs_path_download_all(pigs_ctx *c, double *P) { memcpy(P, c->paths, c->wl * c->W * sizeof(double)); return PIGS_OK; }

int pigs_delta_action_batch(pigs_ctx *c, int64_t n, const int32_t *w, const int32_t *ip, const int32_t *ib,
                            const double *xn, const double *xo, double *dS)
{
    po_delta_action_batch(&c->s, c->WF, c->VT, c->paths, n, w, ip, ib, xn, xo, c->dt, dS);
    return PIGS_OK;
}

int pigs_delta_action_batch_dev(pigs_ctx *c, int64_t n, const int32_t *w, const int32_t *ip, const int32_t *ib,
                                const double *xn, const double *xo, double *dS)
{
    return pigs_delta_action_batch(c, n, w, ip, ib, xn, xo, dS);
}

int pigs_delta_action_parts(pigs_ctx *c, int64_t n, const int32_t *w, const int32_t *ip, const int32_t *ib,
                            const double *xn, const double *xo, double *parts)
{
    const int d = c->s.dim;
    for (int64_t i = 0; i < n; ++i) {
        const double *R = c->paths + c->wl * w[i] + (size_t)ib[i] * d * c->s.Np;
        double dp, df = 0.0, dw = 0.0;
        if (ib[i] % 2) po_update_pot(&c->s, c->VT, ip[i], R, xn + i * d, xo + i * d, &dp, &df);
        else po_update_pot(&c->s, c->VT, ip[i], R, xn + i * d, xo + i * d, &dp, NULL);
        if (ib[i] == 0 || ib[i] == 2 * c->s.Nb) po_update_wf(&c->s, c->WF, ip[i], R, xn + i * d, xo + i * d, &dw);
        parts[3 * i] = dp; parts[3 * i + 1] = df; parts[3 * i + 2] = dw;
    }
    return PIGS_OK;
}

int pigs_commit_beads(pigs_ctx *c, int64_t n, const int32_t *w, const int32_t *ip, const int32_t *ib, const double *x)
{
    const int d = c->s.dim;
    for (int64_t i = 0; i < n; ++i)
        memcpy(c->paths + c->wl * w[i] + ((size_t)ib[i] * c->s.Np + (ip[i] - 1)) * d, x + i * d, d * sizeof(double));
    return PIGS_OK;
}

int pigs_swap_tails(pigs_ctx *c, int32_t w, int32_t iw, int32_t ik)
{
    const int d = c->s.dim;
    for (int ib = c->s.Nb; ib <= 2 * c->s.Nb; ++ib)
        for (int k = 0; k < d; ++k) {
            double *a = c->paths + c->wl * w + ((size_t)ib * c->s.Np + (iw - 1)) * d + k;
            double *b = c->paths + c->wl * w + ((size_t)ib * c->s.Np + (ik - 1)) * d + k;
            double t = *a; *a = *b; *b = t;
        }
    return PIGS_OK;
}

int pigs_potential_energy_slice(pigs_ctx *c, int32_t w, int32_t ib, int32_t want, double *Pot, double *F2)
{
    const double *R = c->paths + c->wl * w + (size_t)ib * c->s.dim * c->s.Np;
    po_potential_energy(&c->s, c->VT, R, Pot, want ? F2 : NULL);
    if (!want && F2) *F2 = 0.0;
    return PIGS_OK;
}

int pigs_therm_energy_batch(pigs_ctx *c, int32_t n, const int32_t *ws, double *E, double *Ec, double *Ep)
{
    for (int i = 0; i < n; ++i)
        po_therm_energy(&c->s, c->VT, c->paths + c->wl * (ws ? ws[i] : i), c->dt, &E[i], &Ec[i], &Ep[i]);
    return PIGS_OK;
}

int pigs_local_energy_batch(pigs_ctx *c, int32_t n, const int32_t *ws, int32_t ib, double *E, double *K, double *P)
{
    for (int i = 0; i < n; ++i)
        po_local_energy(&c->s, c->WF, c->VT, c->paths + c->wl * (ws ? ws[i] : i) + (size_t)ib * c->s.dim * c->s.Np,
                        &E[i], &K[i], &P[i]);
    return PIGS_OK;
}

int pigs_stage_reserve(pigs_ctx *c, int64_t cap, int64_t keep, int32_t **w, int32_t **ip, int32_t **ib,
                       double **xn, double **xo, double **dS)
{
    (void)keep;
    if (cap > c->st_cap) {
        const size_t d = c->s.dim;
        c->st_w = realloc(c->st_w, cap * 4); c->st_ip = realloc(c->st_ip, cap * 4); c->st_ib = realloc(c->st_ib, cap * 4);
        c->st_xn = realloc(c->st_xn, cap * d * 8); c->st_xo = realloc(c->st_xo, cap * d * 8);
        c->st_out = realloc(c->st_out, cap * 8);
        c->st_cap = cap;
    }
    *w = c->st_w; *ip = c->st_ip; *ib = c->st_ib; *xn = c->st_xn; *xo = c->st_xo; *dS = c->st_out;
    return PIGS_OK;
}

int pigs_delta_action_staged(pigs_ctx *c, int64_t n)
{
    return pigs_delta_action_batch(c, n, c->st_w, c->st_ip, c->st_ib, c->st_xn, c->st_xo, c->st_out);
}

int pigs_commit_reserve(pigs_ctx *c, int64_t cap, int64_t keep, int32_t **w, int32_t **ip, int32_t **ib, double **x)
{
    (void)keep;
    if (cap > c->cs_cap) {
        c->cs_w = realloc(c->cs_w, cap * 4); c->cs_ip = realloc(c->cs_ip, cap * 4); c->cs_ib = realloc(c->cs_ib, cap * 4);
        c->cs_x = realloc(c->cs_x, cap * c->s.dim * 8);
        c->cs_cap = cap;
    }
    *w = c->cs_w; *ip = c->cs_ip; *ib = c->cs_ib; *x = c->cs_x;
    return PIGS_OK;
}

int pigs_commit_staged(pigs_ctx *c, int64_t n) { return pigs_commit_beads(c, n, c->cs_w, c->cs_ip, c->cs_ib, c->cs_x); }

/* K6 exists on the GPU only */
int pigs_sampler_init(pigs_ctx *c, const pigs_sweep_params *sp) { (void)c; (void)sp; snprintf(g_err, sizeof g_err, "device sampler needs a GPU"); return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_seed(pigs_ctx *c, int32_t w, int32_t s) { (void)c; (void)w; (void)s; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_set_rng(pigs_ctx *c, int32_t w, int32_t m, const int32_t mt[624]) { (void)c; (void)w; (void)m; (void)mt; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_get_rng(pigs_ctx *c, int32_t w, int32_t *m, int32_t mt[624]) { (void)c; (void)w; (void)m; (void)mt; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_step(pigs_ctx *c, int32_t i) { (void)c; (void)i; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_counters(pigs_ctx *c, int64_t *a) { (void)c; (void)a; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_counters16(pigs_ctx *c, int64_t *a) { (void)c; (void)a; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_get_worm(pigs_ctx *c, int32_t *o, int32_t *i, double *x) { (void)c; (void)o; (void)i; (void)x; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_set_worm(pigs_ctx *c, const int32_t *o, const int32_t *i, const double *x) { (void)c; (void)o; (void)i; (void)x; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_events(pigs_ctx *c, int32_t *e) { (void)c; (void)e; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_event_ints(pigs_ctx *c, int32_t *n) { (void)c; (void)n; return PIGS_ERR_UNSUPPORTED; }
int pigs_sampler_nrho(pigs_ctx *c, double *n, const int32_t *r) { (void)c; (void)n; (void)r; return PIGS_ERR_UNSUPPORTED; }
int pigs_slice_download(pigs_ctx *c, int32_t ib, double *R)
{
    const size_t d = c->s.dim, n = c->s.Np;
    for (int w = 0; w < c->W; ++w) memcpy(R + (size_t)w * d * n, c->paths + c->wl * w + (size_t)ib * d * n, d * n * sizeof(double));
    return PIGS_OK;
}

int pigs_structure_batch(pigs_ctx *c, int32_t n, const int32_t *ws, int32_t ib, int32_t Nbin, double rbin, int32_t Nk,
                         double *gr, double *Sk)
{
    const size_t d = c->s.dim;
    memset(gr, 0, (size_t)n * Nbin * sizeof(double));
    memset(Sk, 0, (size_t)n * Nk * d * sizeof(double));
    for (int i = 0; i < n; ++i) {
        const double *R = c->paths + c->wl * (ws ? ws[i] : i) + (size_t)ib * d * c->s.Np;
        po_pair_correlation(&c->s, Nbin, rbin, R, gr + (size_t)i * Nbin);
        po_structure_factor(&c->s, Nk, R, Sk + (size_t)i * Nk * d);
    }
    return PIGS_OK;
}

int pigs_comm_unique_id(char id[128]) { memset(id, 0, 128); return PIGS_OK; }
int pigs_comm_init_rank(pigs_ctx *c, int32_t n, int32_t r, const char id[128]) { (void)c; (void)n; (void)r; (void)id; return PIGS_OK; }
/* several contexts of one process (the front end's &gpu n_gpus > 1: one host thread per context): the vectors meet in
 * host memory and every rank adds them in rank order -- what RCCL does between GPUs */
typedef struct shim_group {
    int n, arrived, generation, len;
    pthread_mutex_t m;
    pthread_cond_t cv;
    double *slot, *sum;
} shim_group;

int pigs_comm_init_all(pigs_ctx **c, int32_t n)
{
    if (!c || n < 1) return PIGS_ERR_ARG;
    shim_group *g = (shim_group *)calloc(1, sizeof *g);
    g->n = n;
    pthread_mutex_init(&g->m, NULL);
    pthread_cond_init(&g->cv, NULL);
    for (int i = 0; i < n; ++i) { c[i]->group = g; c[i]->grank = i; }
    return PIGS_OK;
}

int pigs_estimators_allreduce(pigs_ctx *c, double *v, int32_t n)
{
    if (!c || !v || n < 0) return PIGS_ERR_ARG;
    ++c->n_allreduce;
    if (n > 0 && v[0] == 0.0) ++c->n_allreduce_nd0;
    shim_group *g = (shim_group *)c->group;
    if (!g || g->n == 1) return PIGS_OK;
    pthread_mutex_lock(&g->m);
    if (g->len < n) {
        g->slot = (double *)realloc(g->slot, (size_t)g->n * n * sizeof(double));
        g->sum  = (double *)realloc(g->sum, (size_t)n * sizeof(double));
        g->len = n;
    }
    memcpy(g->slot + (size_t)c->grank * n, v, (size_t)n * sizeof(double));
    const int gen = g->generation;
    if (++g->arrived == g->n) {
        for (int k = 0; k < n; ++k) {
            double t = 0.0;
            for (int r = 0; r < g->n; ++r) t += g->slot[(size_t)r * n + k];
            g->sum[k] = t;
        }
        g->arrived = 0;
        ++g->generation;
        pthread_cond_broadcast(&g->cv);
    } else {
        while (g->generation == gen) pthread_cond_wait(&g->cv, &g->m);
    }
    memcpy(v, g->sum, (size_t)n * sizeof(double));
    pthread_mutex_unlock(&g->m);
    return PIGS_OK;
}
int pigs_selftest_fastmath(pigs_ctx *c, int32_t b, int32_t i, uint64_t bad[4]) { (void)c; (void)b; (void)i; memset(bad, 0, 32); return PIGS_OK; }
int pigs_selftest_stream_read(pigs_ctx *c, int32_t reps, double *bytes, double *seconds) { (void)c; (void)reps; *bytes = 0.0; *seconds = 1.0; return PIGS_OK; }
int pigs_selftest_log(pigs_ctx *c, int64_t n, uint64_t seed, uint64_t *bad, double *x) { (void)c; (void)n; (void)seed; *bad = 0; if (x) *x = 0.0; return PIGS_OK; }

int pigs_diagonal_estimators(pigs_ctx *c, int32_t n, const int32_t *ws, int32_t Nbin, double rbin, int32_t Nk,
                             double *en, double *gr, double *Sk)
{
    for (int i = 0; i < n; ++i) {
        const int32_t w = ws ? ws[i] : i;
        pigs_local_energy_batch(c, 1, &w, 0, &en[9 * i], &en[9 * i + 1], &en[9 * i + 2]);
        pigs_local_energy_batch(c, 1, &w, 2 * c->s.Nb, &en[9 * i + 3], &en[9 * i + 4], &en[9 * i + 5]);
        pigs_therm_energy_batch(c, 1, &w, &en[9 * i + 6], &en[9 * i + 7], &en[9 * i + 8]);
    }
    if (gr && Sk) return pigs_structure_batch(c, n, ws, c->s.Nb, Nbin, rbin, Nk, gr, Sk);
    return PIGS_OK;
}

/* asynchronous pair: the shim evaluates at _begin (a snapshot in the literal sense) and hands the results out at _end */
int pigs_diagonal_estimators_begin(pigs_ctx *c, int32_t n, const int32_t *ws, int32_t Nbin, double rbin, int32_t Nk, int32_t structure)
{
    free(c->pend_en); free(c->pend_gr); free(c->pend_sk);
    c->pend_n = n; c->pend_nb = Nbin; c->pend_nk = Nk; c->pend_st = structure;
    c->pend_en = (double *)malloc(sizeof(double) * 9 * (size_t)(n > 0 ? n : 1));
    c->pend_gr = structure ? (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (Nbin > 0 ? Nbin : 1)) : NULL;
    c->pend_sk = structure ? (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (Nk > 0 ? Nk : 1) * c->s.dim) : NULL;
    return pigs_diagonal_estimators(c, n, ws, Nbin, rbin, Nk, c->pend_en, c->pend_gr, c->pend_sk);
}
int pigs_diagonal_estimators_end(pigs_ctx *c, double *en, double *gr, double *Sk)
{
    memcpy(en, c->pend_en, sizeof(double) * 9 * (size_t)c->pend_n);
    if (c->pend_st && gr && Sk) {
        memcpy(gr, c->pend_gr, sizeof(double) * (size_t)c->pend_n * c->pend_nb);
        memcpy(Sk, c->pend_sk, sizeof(double) * (size_t)c->pend_n * c->pend_nk * c->s.dim);
    }
    return PIGS_OK;
}
