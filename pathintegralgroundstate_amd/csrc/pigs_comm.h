// pigs_comm.h -- RCCL (xGMI) reduction of the block-estimator vector.
// librccl is loaded lazily with dlopen so that single-GPU users never pay for it.
// Functions return nullptr on success or a static error string.
#pragma once

#include <hip/hip_runtime.h>

struct pigs_comm;

const char *pigs_comm_get_unique_id(char id[128]);
const char *pigs_comm_create_rank(pigs_comm **out, int nranks, int rank, const char id[128]);
const char *pigs_comm_create_all(pigs_comm **out, int nranks, const int *devices);
const char *pigs_comm_allreduce_sum_f64(pigs_comm *c, double *d_buf, int n, hipStream_t s);
void        pigs_comm_destroy(pigs_comm *c);
