// Ranks of a multi-GPU run (one process per GPU of one node): control plane (TCP mesh) and data plane (RCCL
// point-to-point over xGMI, or the same messages staged through the mesh for ranks that share a GPU).
// Internal interface of comm.hip, used by slab.hip.
#pragma once
#include "common.h"
#include <vector>

namespace hcm {

bool active();                 // hc_comm_init* done and world > 1 ... or a periodic one-rank world on a transport
int rank();
int world();
int transport();
// x-neighbours of this rank in a ring (periodic) or a chain; -1 = none.  A one-rank periodic world is its own neighbour.
void neighbours(bool periodic, int &lo, int &hi);

// Stream-ordered neighbour exchange of DEVICE buffers: on return the transfers are enqueued on `s` (RCCL) or complete
// (TCP: the host waits for `s`, moves the bytes, and enqueues the uploads on `s`); either way work enqueued on `s`
// afterwards sees the received data.  Counts are bytes; a zero count (or a missing neighbour) skips that message on
// both sides.  Routing: what I send to my low neighbour is what it receives from its high side.
int exchange(hipStream_t s, bool periodic, const void *send_lo, size_t n_lo, const void *send_hi, size_t n_hi, void *recv_lo, size_t m_lo,
             void *recv_hi, size_t m_hi);

// control plane (host memory, blocking; set-up and output cadence only)
int barrier();
int allreduce(double *v, int n, int op);   // 0 sum, 1 min, 2 max; folded in rank order on rank 0
int bcast(void *buf, size_t bytes, int root);
int allgatherv(const void *mine, size_t bytes, std::vector<std::vector<char>> &all);   // all[r] = rank r's block, on every rank

}  // namespace hcm
