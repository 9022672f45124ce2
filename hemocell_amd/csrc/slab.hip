// HemoCell::iterate and collideAndStream on one x-slab of a multi-GPU run: the schedule that overlaps the neighbour
// exchange with the interior collide, and the particle-envelope synchronisation.
//
// Replaces (file:line in the HemoCell tree):
//   core/hemoCell.cpp:299-376                     iterate() as the reference runs it on every MPI rank
//   core/hemoCell.cpp:317 (Palabos duplicateOverlaps, envelope width core/hemoCell.cpp:142)   lattice faces
//   core/hemoCellFields.cpp:377-499, core/hemoCellParticleDataTransfer.cpp:33-466              syncEnvelopes
//   core/hemoCellParticleField.cpp:173-235        addParticle merge rule ("a local particle wins")
//   core/hemoCellFields.cpp:676-688               deleteNonLocalParticles, at cell granularity
//
// One process per GPU; rank r owns the x-planes [x0, x0 + nx) and talks to its two x-neighbours only (comm.hip).  No
// collective is on the stepping path.  Per step the 5 populations with c_x = +-1 of each face plane cross (width-1
// message); at a velocity update the neighbour also needs what lets it evaluate node velocities on its first halo plane
// (width-2 message: the face plane's 19 populations and the 5 of the plane behind it that stream onto that halo plane),
// and the cells within hc_cells::e_share lattice units of a face (hcp_set_envelope) are replicated on the neighbour as the reference's envelope copies
// are.  Every rank interpolates only the particles whose nearest node it owns; a record of 9 (12 with a repulsion) doubles per
// particle carries position, velocity and force to the other holder, where "a local particle wins" decides what is kept.
//
// Streams: the main stream never waits for a transfer.  It collides the planes that read no halo data; the side stream
// (hc::fork / route / join) unpacks the faces that arrived, collides the planes next to the faces, packs what those
// planes just produced, then runs advance, mechanics and the next spread (between velocity updates).  Every transfer runs
// on a third stream (hc::comm_stream) between two events -- "packed" on the packing stream, "arrived" for whoever unpacks -- so
// that a message in flight holds up no kernel: with the transfers on the side stream itself a 2.6 MB face message (0.7 ms
// as an RCCL send-to-self, rocprofv3 timeline profiles/r02_c_*) kept advance and spread waiting until the collide was over.
#include "cells.h"
#include "comm.h"

#include <chrono>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <unordered_set>

namespace {

constexpr int MAX_HDR = 8192;     // longs per cell type and face in the id header of the envelope synchronisation

double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Slab {
  hc_lattice *L = nullptr;
  hc_cells *C = nullptr;                     // null: fluid only
  double *hs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *hr[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [width - 1][side] device buffers
  int pending_width = 0;                     // a face exchange of this width is on its way and not yet unpacked
  bool halo_fresh = false;                   // the halo planes of f[cur] hold what the next collide needs
  bool spread_done = false;                  // the spread of the coming iteration already ran beside the last collide
  bool planned = false;                      // the cell extents for the coming envelope sync are on their way to the host
  // envelope synchronisation: id headers [side][type][MAX_HDR] (pinned staging, device send / receive, pinned landing)
  long *h_hdr_s = nullptr, *h_hdr_r = nullptr, *d_hdr_s = nullptr, *d_hdr_r = nullptr; int hdr_types = 0;
  hipEvent_t hdr_ev = nullptr, rec_ready = nullptr, rec_done = nullptr, halo_packed = nullptr, halo_arrived = nullptr, u_packed = nullptr, u_arrived = nullptr;
  double *us[2] = {nullptr, nullptr};         // face-plane velocities on their way to the neighbours ([side][3][plane]; they land in L->halo_u)
  double *d_rec_s[2] = {nullptr, nullptr}, *d_rec_r[2] = {nullptr, nullptr}; size_t rec_cap_s[2] = {0, 0}, rec_cap_r[2] = {0, 0};
  double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
std::map<hc_lattice *, Slab *> g_slabs;

bool periodic_x(const hc_lattice *L) { return L->periodic[0] != 0; }

int make_slab(hc_lattice *L, hc_cells *C, Slab **out) {
  auto it = g_slabs.find(L);
  Slab *S = it == g_slabs.end() ? nullptr : it->second;
  if (!S) {
    if (!hcm::active()) { hc::set_error("a lattice with n_slabs > 1 needs the ranks connected first (hc_comm_init_env / hc_comm_init)"); return HC_ERR_STATE; }
    if (hcm::world() != L->n_slabs && !(hcm::world() == 1 && L->n_slabs >= 1)) { hc::set_error("the lattice was created for " + std::to_string(L->n_slabs) + " slabs but the world has " + std::to_string(hcm::world()) + " ranks"); return HC_ERR_STATE; }
    S = new Slab(); S->L = L;
    for (int w = 0; w < 2; w++)
      for (int side = 0; side < 2; side++) {
        HC_HIP(hipMalloc((void **)&S->hs[w][side], hcl_halo_doubles(L, w + 1) * sizeof(double)));
        HC_HIP(hipMalloc((void **)&S->hr[w][side], hcl_halo_doubles(L, w + 1) * sizeof(double)));
      }
    HC_HIP(hipEventCreateWithFlags(&S->hdr_ev, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->rec_ready, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->rec_done, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->halo_packed, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->halo_arrived, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->u_packed, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&S->u_arrived, hipEventDisableTiming));
    for (int side = 0; side < 2; side++) {
      HC_HIP(hipMalloc((void **)&S->us[side], 3 * L->plane * sizeof(double)));
      HC_HIP(hipMalloc((void **)&L->halo_u[side], 3 * L->plane * sizeof(double)));
      HC_HIP(hipMemset(L->halo_u[side], 0, 3 * L->plane * sizeof(double)));
    }
    L->halo_u_valid = false;
    g_slabs[L] = S;
  }
  if (C) {
    double dmax = 0.0;
    for (int t = 0; t < C->ntypes; t++) dmax = std::max(dmax, C->types[t]->host.diameter);
    const int need = (int)std::ceil(2.0 * dmax + 2.0 * C->e_share);   // 40 planes for an RBC at dx = 0.5 um with the default envelope
    if (C->ntypes > 0 && L->nx < need) { hc::set_error("a slab that carries cells must be at least " + std::to_string(need) + " planes wide (two cell diameters plus the particle envelope on both faces), got " + std::to_string(L->nx)); return HC_ERR_ARG; }
    S->C = C;
    if (S->hdr_types < C->ntypes) {
      HC_HIP(hipDeviceSynchronize());
      if (S->h_hdr_s) HC_HIP(hipHostFree(S->h_hdr_s));
      if (S->h_hdr_r) HC_HIP(hipHostFree(S->h_hdr_r));
      if (S->d_hdr_s) HC_HIP(hipFree(S->d_hdr_s));
      if (S->d_hdr_r) HC_HIP(hipFree(S->d_hdr_r));
      const size_t bytes = (size_t)2 * C->ntypes * MAX_HDR * sizeof(long);
      HC_HIP(hipHostMalloc((void **)&S->h_hdr_s, bytes, hipHostMallocDefault));
      HC_HIP(hipHostMalloc((void **)&S->h_hdr_r, bytes, hipHostMallocDefault));
      HC_HIP(hipMalloc((void **)&S->d_hdr_s, bytes));
      HC_HIP(hipMalloc((void **)&S->d_hdr_r, bytes));
      S->hdr_types = C->ntypes;
    }
  }
  *out = S;
  return HC_OK;
}

// ---------------------------------------------------------------------------- lattice faces
// the stream transfers run on: the library's third stream, or -- strictly-one-stream mode -- the stream in use
hipStream_t transfer_stream();

// pack my faces (from the buffer the collide in progress is writing when next != 0) on the stream in use and hand them to the
// data plane on the transfer stream; halo_finish() later unpacks what arrived, on the stream in use then
int halo_begin(Slab *S, int width, int next) {
  hc_lattice *L = S->L;
  int lo, hi; hcm::neighbours(periodic_x(L), lo, hi);
  const int w = width - 1;
  int rc;
  if (width == 1) { rc = hcl_halo_pack_both(L, lo >= 0 ? S->hs[0][0] : nullptr, hi >= 0 ? S->hs[0][1] : nullptr, next); if (rc != HC_OK) return rc; }   // one launch
  else {
    if (lo >= 0) { rc = next ? hcl_halo_pack_next(L, 0, width, S->hs[w][0]) : hcl_halo_pack(L, 0, width, S->hs[w][0]); if (rc != HC_OK) return rc; }
    if (hi >= 0) { rc = next ? hcl_halo_pack_next(L, 1, width, S->hs[w][1]) : hcl_halo_pack(L, 1, width, S->hs[w][1]); if (rc != HC_OK) return rc; }
  }
  const hipStream_t X = transfer_stream();
  if (X != hc::stream()) {
    // the unpack of the previous message precedes this pack in the packing stream, so the receive buffers are free again
    HC_HIP(hipEventRecord(S->halo_packed, hc::stream()));
    HC_HIP(hipStreamWaitEvent(X, S->halo_packed, 0));
  }
  const size_t bytes = hcl_halo_doubles(L, width) * sizeof(double);
  rc = hcm::exchange(X, periodic_x(L), S->hs[w][0], bytes, S->hs[w][1], bytes, S->hr[w][0], bytes, S->hr[w][1], bytes);
  if (rc != HC_OK) return rc;
  if (X != hc::stream()) HC_HIP(hipEventRecord(S->halo_arrived, X));
  S->pending_width = width;
  return HC_OK;
}

int halo_finish(Slab *S) {
  if (!S->pending_width) return HC_OK;
  hc_lattice *L = S->L;
  int lo, hi; hcm::neighbours(periodic_x(L), lo, hi);
  const int width = S->pending_width, w = width - 1;
  int rc;
  if (transfer_stream() != hc::stream()) HC_HIP(hipStreamWaitEvent(hc::stream(), S->halo_arrived, 0));
  if (width == 1) { rc = hcl_halo_unpack_both(L, lo >= 0 ? S->hr[0][0] : nullptr, hi >= 0 ? S->hr[0][1] : nullptr); if (rc != HC_OK) return rc; }   // one launch
  else {
    if (lo >= 0) { rc = hcl_halo_unpack(L, 0, width, S->hr[w][0]); if (rc != HC_OK) return rc; }
    if (hi >= 0) { rc = hcl_halo_unpack(L, 1, width, S->hr[w][1]); if (rc != HC_OK) return rc; }
  }
  S->pending_width = 0;
  S->halo_fresh = true;
  return HC_OK;
}

// faces in flight are completed; if the halos are still stale a blocking exchange of the current state follows
int halo_make_fresh(Slab *S, int width) {
  int rc = halo_finish(S); if (rc != HC_OK) return rc;
  if (S->halo_fresh && width == 1) return HC_OK;
  rc = halo_begin(S, width, 0); if (rc != HC_OK) return rc;
  return halo_finish(S);
}

// The message of a velocity update.  The interpolation blends node velocities u = j/rho + F/2 (Cell::computeVelocity) over a
// particle's stencil, and a particle next to a face has stencil nodes on the neighbour's face plane.  Each rank evaluates u on
// its two face planes -- it owns every population that streams there once the crossing populations of the step have arrived --
// and sends the result: 3 planes of doubles per face instead of the 19 population planes from which the receiver could
// form the same expression itself (round 2; 1.5 MB instead of 10 MB per direction at 256 x 256).  On the stream in use: pack;
// on the transfer stream: the exchange into L->halo_u; the caller waits for u_arrived before it interpolates.
int velocity_exchange(Slab *S) {
  hc_lattice *L = S->L;
  int lo, hi; hcm::neighbours(periodic_x(L), lo, hi);
  int rc;
  rc = hcl_face_velocity_pack_both(L, lo >= 0 ? S->us[0] : nullptr, hi >= 0 ? S->us[1] : nullptr); if (rc != HC_OK) return rc;   // one launch
  const hipStream_t X = transfer_stream();
  if (X != hc::stream()) {
    HC_HIP(hipEventRecord(S->u_packed, hc::stream()));   // also: every kernel that read the previous velocities precedes this point
    HC_HIP(hipStreamWaitEvent(X, S->u_packed, 0));
  }
  const size_t bytes = 3 * L->plane * sizeof(double);
  rc = hcm::exchange(X, periodic_x(L), S->us[0], bytes, S->us[1], bytes, L->halo_u[0], bytes, L->halo_u[1], bytes);
  if (rc != HC_OK) return rc;
  if (X != hc::stream()) HC_HIP(hipEventRecord(S->u_arrived, X));
  L->halo_u_valid = true;
  return HC_OK;
}
int velocity_wait(Slab *S) {   // on the stream in use, before anything interpolates
  if (transfer_stream() != hc::stream()) HC_HIP(hipStreamWaitEvent(hc::stream(), S->u_arrived, 0));
  return HC_OK;
}

// ---------------------------------------------------------------------------- particle envelopes
struct Plan {                                // one cell type at one envelope synchronisation
  std::vector<int> send[2];                  // slots of the cells that cross the low / high face, ordered by cell id
  std::vector<int> gone;                     // slots of local cells that were deleted at a wall since the last synchronisation
  std::vector<long> gone_before;             // ids of cells this rank already compacted away since then (hc_cells::slab_gone)
  std::vector<long> ids_r[2], tag_r[2];      // what the neighbours announced: crossing cells, deleted cells
  std::vector<double> ext;                   // [n][4] extents of the local cells
  long n = 0;
};

int plan_cells(Slab *S) {
  for (int t = 0; t < S->C->ntypes; t++) { const int rc = hcp_cell_extents_begin(S->C, t); if (rc != HC_OK) return rc; }
  S->planned = true;
  return HC_OK;
}

// first part, needs positions only: which cells cross which face, which were deleted; the id headers leave on the transfer
// stream (ahead of the wide face message of the same step) and land in pinned memory behind an event
int sync_begin(Slab *S, std::vector<Plan> &plans) {
  hc_cells *C = S->C; hc_lattice *L = S->L;
  int lo, hi; hcm::neighbours(periodic_x(L), lo, hi);
  const double x0 = (double)L->x0, x1 = (double)(L->x0 + L->nx);
  plans.assign((size_t)C->ntypes, Plan());
  const double t_wait = wall_s();
  for (int t = 0; t < C->ntypes; t++) {
    Plan &P = plans[(size_t)t];
    if (!C->ext_pending[t] || C->ext_n[t] != C->ncells[t]) { const int rc = hcp_cell_extents_begin(C, t); if (rc != HC_OK) return rc; }
    C->ext_pending[t] = false;
    P.n = C->ext_n[t];
    if (P.n) HC_HIP(hipEventSynchronize(C->ext_done[t]));
    P.ext.assign(C->h_ext[t], C->h_ext[t] + 4 * P.n);
    const std::vector<long> &ids = C->hids[t];
    for (long c = 0; c < P.n; c++) {
      const double *e = &P.ext[(size_t)(4 * c)];
      if (e[3] != 0.0) { P.gone.push_back((int)c); continue; }
      if (e[2] <= 0.0) continue;   // only a rank that owns part of the cell forwards it (a pure ghost is its owner's business)
      if (lo >= 0 && e[0] < x0 + C->e_share) P.send[0].push_back((int)c);
      if (hi >= 0 && e[1] >= x1 - C->e_share) P.send[1].push_back((int)c);
    }
    P.gone_before.swap(C->slab_gone[t]); C->slab_gone[t].clear();
    for (int side = 0; side < 2; side++) {
      std::stable_sort(P.send[side].begin(), P.send[side].end(), [&](int a, int b) { return ids[(size_t)a] < ids[(size_t)b]; });
      const size_t k = P.send[side].size(), g = P.gone.size(), gb = P.gone_before.size();
      if (k + g + gb + 2 > (size_t)MAX_HDR) { hc::set_error("more cells cross one slab face (" + std::to_string(k) + ") than the id header holds"); return HC_ERR_STATE; }
      long *h = S->h_hdr_s + ((size_t)side * C->ntypes + t) * MAX_HDR;
      h[0] = (long)k;
      for (size_t i = 0; i < k; i++) h[1 + i] = ids[(size_t)P.send[side][i]];
      h[1 + k] = (long)(g + gb);
      for (size_t i = 0; i < g; i++) h[2 + k + i] = ids[(size_t)P.gone[i]];
      for (size_t i = 0; i < gb; i++) h[2 + k + g + i] = P.gone_before[i];
    }
  }
  S->stats[6] += wall_s() - t_wait;
  const size_t side_bytes = (size_t)C->ntypes * MAX_HDR * sizeof(long);
  const hipStream_t X = transfer_stream();   // the headers depend on host data only: nothing to wait for
  HC_HIP(hipMemcpyAsync(S->d_hdr_s, S->h_hdr_s, 2 * side_bytes, hipMemcpyHostToDevice, X));
  int rc = hcm::exchange(X, periodic_x(L), S->d_hdr_s, side_bytes, (char *)S->d_hdr_s + side_bytes, side_bytes, S->d_hdr_r, side_bytes,
                         (char *)S->d_hdr_r + side_bytes, side_bytes);
  if (rc != HC_OK) return rc;
  HC_HIP(hipMemcpyAsync(S->h_hdr_r, S->d_hdr_r, 2 * side_bytes, hipMemcpyDeviceToHost, X));
  HC_HIP(hipEventRecord(S->hdr_ev, X));
  return HC_OK;
}

int grow(double **buf, size_t *cap, size_t doubles) {
  if (*cap >= doubles) return HC_OK;
  HC_HIP(hipDeviceSynchronize());   // rare: the old block may still be in use on either stream
  if (*buf) HC_HIP(hipFree(*buf));
  *buf = nullptr; *cap = 0;
  const size_t n = doubles + doubles / 2 + 4096;
  HC_HIP(hipMalloc((void **)buf, n * sizeof(double)));
  *cap = n;
  return HC_OK;
}

// second part (main stream): the records of the crossing cells carry interpolated velocities, so those cells are
// interpolated first; their records then travel on the side stream while the caller interpolates all cells
int sync_records(Slab *S, std::vector<Plan> &plans, hipStream_t comm_stream) {
  hc_cells *C = S->C; hc_lattice *L = S->L;
  int lo, hi; hcm::neighbours(periodic_x(L), lo, hi);
  const double t_wait = wall_s();
  HC_HIP(hipEventSynchronize(S->hdr_ev));
  S->stats[6] += wall_s() - t_wait;
  size_t send_d[2] = {0, 0}, recv_d[2] = {0, 0};
  for (int t = 0; t < C->ntypes; t++) {
    Plan &P = plans[(size_t)t];
    const size_t rec = hcp_record_doubles(C, t);
    for (int side = 0; side < 2; side++) {
      P.ids_r[side].clear(); P.tag_r[side].clear();
      if ((side == 0 ? lo : hi) < 0) continue;
      const long *h = S->h_hdr_r + ((size_t)side * C->ntypes + t) * MAX_HDR;
      const long k = h[0];
      if (k < 0 || k + 2 > MAX_HDR) { hc::set_error("corrupt id header from a neighbour"); return HC_ERR_STATE; }
      const long g = h[1 + k];
      if (g < 0 || k + g + 2 > MAX_HDR) { hc::set_error("corrupt id header from a neighbour"); return HC_ERR_STATE; }
      P.ids_r[side].assign(h + 1, h + 1 + k);
      P.tag_r[side].assign(h + 2 + k, h + 2 + k + g);
      send_d[side] += P.send[side].size() * rec; recv_d[side] += (size_t)k * rec;
    }
    std::vector<int> both(P.send[0]);
    both.insert(both.end(), P.send[1].begin(), P.send[1].end());
    std::sort(both.begin(), both.end()); both.erase(std::unique(both.begin(), both.end()), both.end());
    if (!both.empty()) { const int rc = hcp_interpolate_cells(C, t, both.data(), (int)both.size()); if (rc != HC_OK) return rc; }
  }
  for (int side = 0; side < 2; side++) {
    int rc = grow(&S->d_rec_s[side], &S->rec_cap_s[side], send_d[side]); if (rc != HC_OK) return rc;
    rc = grow(&S->d_rec_r[side], &S->rec_cap_r[side], recv_d[side]); if (rc != HC_OK) return rc;
  }
  // periodic images are shifted by the domain length when they cross the seam (core/hemoCellParticleDataTransfer.cpp:33-65)
  const double shift[2] = {(periodic_x(L) && L->x0 == 0) ? (double)L->nx_global : 0.0,
                           (periodic_x(L) && L->x0 + L->nx == L->nx_global) ? -(double)L->nx_global : 0.0};
  size_t off[2] = {0, 0};
  for (int t = 0; t < C->ntypes; t++) {
    const size_t rec = hcp_record_doubles(C, t);
    for (int side = 0; side < 2; side++) {
      const std::vector<int> &sl = plans[(size_t)t].send[side];
      if (sl.empty()) continue;
      const int rc = hcp_pack_cells(C, t, sl.data(), (int)sl.size(), shift[side], S->d_rec_s[side] + off[side]); if (rc != HC_OK) return rc;
      off[side] += sl.size() * rec;
      S->stats[0] += (double)sl.size();
    }
  }
  if (comm_stream != hc::stream()) {
    HC_HIP(hipEventRecord(S->rec_ready, hc::stream()));
    HC_HIP(hipStreamWaitEvent(comm_stream, S->rec_ready, 0));
  }
  const int rc = hcm::exchange(comm_stream, periodic_x(L), S->d_rec_s[0], send_d[0] * sizeof(double), S->d_rec_s[1], send_d[1] * sizeof(double),
                               S->d_rec_r[0], recv_d[0] * sizeof(double), S->d_rec_r[1], recv_d[1] * sizeof(double));
  if (rc != HC_OK) return rc;
  if (comm_stream != hc::stream()) HC_HIP(hipEventRecord(S->rec_done, comm_stream));
  return HC_OK;
}

// last part (main stream): merge the records (a local particle wins), append new copies, drop the copies nobody refreshed
// and the cells that were deleted at a wall here or on a neighbour
int sync_merge(Slab *S, std::vector<Plan> &plans, hipStream_t comm_stream) {
  hc_cells *C = S->C;
  if (comm_stream != hc::stream()) HC_HIP(hipStreamWaitEvent(hc::stream(), S->rec_done, 0));
  size_t off[2] = {0, 0};
  bool any_gone = false;
  for (int t = 0; t < C->ntypes; t++) {
    Plan &P = plans[(size_t)t];
    const size_t rec = hcp_record_doubles(C, t);
    const long n = P.n;
    std::unordered_map<long, long> known;
    known.reserve((size_t)n * 2 + 16);
    for (long c = 0; c < n; c++) known[C->hids[t][(size_t)c]] = c;
    std::vector<char> refreshed((size_t)n, 0);
    long next_new = n;
    // a neighbour that has not heard yet still sends its copy of a cell this rank compacted away: it lands in a new slot
    // (the records of one message are contiguous) and leaves again with the drop list below
    const std::unordered_set<long> gone_before(P.gone_before.begin(), P.gone_before.end());
    std::vector<int> drop;
    for (int side = 0; side < 2; side++) {
      const std::vector<long> &rid = P.ids_r[side];
      if (rid.empty()) continue;
      std::vector<int> slots(rid.size()), is_new(rid.size());
      for (size_t i = 0; i < rid.size(); i++) {
        auto f = known.find(rid[i]);
        if (f != known.end()) { slots[i] = (int)f->second; is_new[i] = 0; if (f->second < n) refreshed[(size_t)f->second] = 1; }
        else {
          slots[i] = (int)next_new; is_new[i] = 1; known[rid[i]] = next_new++;
          if (!gone_before.empty() && gone_before.count(rid[i])) drop.push_back(slots[i]); else S->stats[1] += 1.0;
        }
      }
      const int rc = hcp_unpack_cells(C, t, slots.data(), rid.data(), is_new.data(), (int)rid.size(), S->d_rec_r[side] + off[side]);
      if (rc != HC_OK) return rc;
      off[side] += rid.size() * rec;
    }
    // a copy without any local particle survives only while its owner keeps refreshing it (deleteNonLocalParticles,
    // core/hemoCellFields.cpp:676-688); a cell that reached a wall -- here or on the neighbour that also holds it -- goes everywhere
    std::unordered_set<long> dead_ids;
    for (int side = 0; side < 2; side++) for (long id : P.tag_r[side]) dead_ids.insert(id);
    for (long c = 0; c < n; c++) {
      const double *e = &P.ext[(size_t)(4 * c)];
      const bool gone = e[3] != 0.0, told = !dead_ids.empty() && dead_ids.count(C->hids[t][(size_t)c]) > 0;
      if (gone || told) { drop.push_back((int)c); S->stats[3] += 1.0; C->n_deleted++; any_gone = any_gone || gone; }
      else if (e[2] <= 0.0 && !refreshed[(size_t)c]) { drop.push_back((int)c); S->stats[2] += 1.0; }
    }
    if (!drop.empty()) { const int rc = hcp_remove_cells(C, t, drop.data(), (int)drop.size()); if (rc != HC_OK) return rc; }
  }
  if (any_gone) {   // every cell the counters know about has just been removed (no advance ran since the extents were taken)
    HC_HIP(hipMemsetAsync(C->d_ntag, 0, 4 * sizeof(int), hc::stream()));
  }
  C->maybe_tagged = false; C->ntag_pending = false;
  return HC_OK;
}

int repulsion_at(hc_cells *C, long it) {
  int rc;
  if (C->rep_enabled && it % C->rep_timescale == 0) { if ((rc = hcp_repulsion(C)) != HC_OK) return rc; }             // core/hemoCell.cpp:307-309
  if (C->brep_enabled && it % C->brep_timescale == 0) { if ((rc = hcp_boundary_repulsion(C)) != HC_OK) return rc; }   // :310-312
  return HC_OK;
}

int g_slab_overlap = 1;
hipStream_t transfer_stream() { return g_slab_overlap ? hc::comm_stream() : hc::stream(); }

// one HemoCell::iterate on this slab.  more: another iteration follows in the same call, so the next spread may run beside
// this collide (never across the end of a call: the caller may edit vertex forces in between).
int step(Slab *S, hc_cells *C, long it, int k_p, int force_limit, bool more) {
  hc_lattice *L = S->L;   // C: null for a fluid-only call (lattice->collideAndStream() of a driver's warm-up loop), whatever is bound
  const bool cells = C != nullptr && C->ntypes > 0;
  const bool particle_step = cells && it % k_p == 0;
  const bool overlap = g_slab_overlap != 0 && L->nx >= 4;
  int rc;
#define TRY(call) do { if ((rc = (call)) != HC_OK) return rc; } while (0)
  if (particle_step && !S->planned) TRY(plan_cells(S));   // extents for the envelope sync at the end of this step
  if (cells && !S->spread_done) { TRY(repulsion_at(C, it)); TRY(hcp_spread(C, force_limit)); }   // :307-313
  S->spread_done = false;
  std::vector<Plan> plans;
  if (!overlap) {   // strictly one stream, nothing in flight across phases (A/B runs and slabs thinner than 4 planes)
    TRY(halo_make_fresh(S, 1));
    TRY(hcl_collide_stream_part(L, 0));                                          // :317
    hcl_step_end(L); S->halo_fresh = false;
    if (particle_step) {
      TRY(halo_make_fresh(S, 1));
      TRY(velocity_exchange(S)); TRY(velocity_wait(S));
      TRY(sync_begin(S, plans));
      TRY(sync_records(S, plans, hc::stream()));
      TRY(hcp_interpolate(C));                                                   // :327-332
      TRY(sync_merge(S, plans, hc::stream()));
    }
    if (cells) { TRY(hcp_advance(C, 0)); TRY(hcp_mechanics(C, it, 0)); }         // :342, :345
  } else if (particle_step) {
    TRY(sync_begin(S, plans));                     // id headers first (the host waits for the extents taken after the last advance): they are tiny
    TRY(hc::fork());
    hc::route(1);
    TRY(halo_make_fresh(S, 1));                    // the neighbours' faces (on their way since the previous step)
    TRY(hcl_collide_stream_part(L, 6));            // the two planes next to each face (one launch), beside the interior ...
    TRY(halo_begin(S, 1, 1));                      // ... so that the crossing populations travel during the interior collide
    TRY(hcl_zero_force_halos(L));                  // off the chain the message waits for
    hc::route(0);
    TRY(hcl_collide_stream_part(L, 3));            // main stream: planes 2 .. nx-3
    hcl_step_end(L);
    // :327-332.  Only vertices within a node of a face read the neighbours' face velocities; they belong to cells within the
    // envelope of the face -- the cells sync_records interpolates and sends -- or to pure envelope copies, which take their
    // owner's record.  All OTHER cells depend on nothing that is still on its way, and there are two waits to fill: the chain on
    // the side stream below (a message that only arrives when the collide is over, an unpack, a kernel, a second message), and
    // the records of the crossing cells further down.  So the other cells are interpolated in two halves, one behind the
    // interior collide, one behind the departure of the records.  (A vertex of such a cell never has a stencil node on a halo
    // plane; a pure envelope copy may, its values are formed from stale planes and replaced by the merge.)
    // The planes next to the faces were collided on the side stream: its "packed" event (recorded behind them) comes first.
    std::vector<std::vector<int>> rest((size_t)C->ntypes);
    for (int t = 0; t < C->ntypes; t++) {
      const Plan &P = plans[(size_t)t];
      std::vector<char> near((size_t)P.n, 0);
      for (int side = 0; side < 2; side++) for (int c : P.send[side]) near[(size_t)c] = 1;
      for (long c = 0; c < P.n; c++) if (!near[(size_t)c] && P.ext[(size_t)(4 * c + 3)] != 1.0) rest[(size_t)t].push_back((int)c);
    }
    if (transfer_stream() != hc::stream()) HC_HIP(hipStreamWaitEvent(hc::stream(), S->halo_packed, 0));
    // first part: half of them, but at least what covers the ~0.1 ms of that chain (the kernel does 6-7 cells of 642 vertices per microsecond)
    auto first_part = [&](int t) { const long n = (long)rest[(size_t)t].size(), v = 640L * 642 / C->types[t]->host.nv; return (int)std::min(n, std::max((n + 1) / 2, v)); };
    for (int t = 0; t < C->ntypes; t++) { const int h = first_part(t); if (h) TRY(interpolate_cells_staged(C, t, rest[(size_t)t].data(), h, 3 + 2 * t)); }
    hc::route(1);
    TRY(halo_finish(S));                           // the neighbours' crossing populations -> halo planes of the new state
    TRY(velocity_exchange(S));                     // node velocities of my face planes (they read planes -1 .. 1) -> the neighbours' first halo plane
    TRY(hc::join());
    TRY(velocity_wait(S));
    TRY(sync_records(S, plans, transfer_stream()));   // the cells near the faces, with the neighbours' velocities; their records leave on the transfer stream
    for (int t = 0; t < C->ntypes; t++) {             // the second half of the others, while the records travel
      const int h = first_part(t), n2 = (int)rest[(size_t)t].size() - h;
      if (n2 > 0) TRY(interpolate_cells_staged(C, t, rest[(size_t)t].data() + h, n2, 4 + 2 * t));
    }
    TRY(sync_merge(S, plans, transfer_stream()));
    TRY(hcp_advance(C, 0));                        // :342
    TRY(hcp_mechanics(C, it, 0));                  // :345
    S->halo_fresh = true;                          // the crossing populations are what the next collide reads
  } else {
    TRY(hc::fork());
    TRY(hcl_collide_stream_part(L, 1));            // main stream: the planes that read no halo data
    hc::route(1);
    TRY(halo_make_fresh(S, 1));                    // faces of the neighbours (already here after a velocity update)
    TRY(hcl_collide_stream_part(L, 5));            // the two face planes (one launch)
    TRY(halo_begin(S, 1, 1));                      // my faces of the state being written leave for the NEXT step
    TRY(hcl_zero_force_halos(L));                  // off the chain the message waits for
    hc::route(0);
    hcl_step_end(L);
    S->halo_fresh = false;
    hc::route(1);
    if (cells) {
      TRY(hcp_advance(C, 0));                                                   // :342
      if ((it + 1) % k_p == 0) TRY(plan_cells(S));                              // positions are final for the sync of the next step
      TRY(hcp_mechanics(C, it, 0));                                             // :345
      if (more) { TRY(repulsion_at(C, it + 1)); TRY(hcp_spread(C, force_limit)); S->spread_done = true; }   // :313 of iteration it + 1
    }
    TRY(hc::join());
  }
#undef TRY
  if (particle_step) { S->planned = false; S->stats[7] += 1.0; }
  return HC_OK;
}

int run(hc_lattice *L, hc_cells *C, long *iter, int n, int k_p, int force_limit) {
  Slab *S = nullptr;
  int rc = make_slab(L, C, &S); if (rc != HC_OK) return rc;
  if (C) { rc = sync_to_device(C); if (rc != HC_OK) return rc; }
  struct ForkGuard { ~ForkGuard() { if (hc::forked()) hc::join(); else hc::route(0); } } guard;   // error paths leave one timeline behind
  const double t0 = wall_s();
  long it = iter ? *iter : 0;
  for (int s = 0; s < n; s++, it++) {
    rc = step(S, C, it, k_p, force_limit, s + 1 < n);
    if (rc != HC_OK) return rc;
    if (iter) *iter = it + 1;
  }
  S->stats[4] += (double)n; S->stats[5] += wall_s() - t0;
  if (C && C->h_env_viol && *C->h_env_viol != 0) {   // written by the merge kernels that have completed so far; a late one is seen by the next call
    hc::set_error("particle envelope too small: " + std::to_string(*C->h_env_viol) + " particle(s) had crossed a slab face before a copy of their cell existed on the "
                  "neighbour (envelope " + std::to_string(C->e_share) + " lu per velocity update; raise <particleEnvelope> or lower stepParticleEvery)");
    return HC_ERR_STATE;
  }
  return HC_OK;
}

}  // namespace

namespace hcs {

int iterate_slab(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit) {
  return run(L, C, iter, n, particle_timescale, force_limit);
}
int collide_stream_slab(hc_lattice *L, int nsteps) { return run(L, nullptr, nullptr, nsteps, 1, 0); }

void set_overlap(int on) { g_slab_overlap = on != 0; }

void halos_stale(hc_lattice *L) {
  auto it = g_slabs.find(L);
  if (it != g_slabs.end()) { it->second->halo_fresh = false; it->second->pending_width = 0; }
}

void lattice_destroyed(hc_lattice *L) {
  auto it = g_slabs.find(L);
  if (it == g_slabs.end()) return;
  Slab *S = it->second;
  hipDeviceSynchronize();
  for (int w = 0; w < 2; w++) for (int side = 0; side < 2; side++) { if (S->hs[w][side]) hipFree(S->hs[w][side]); if (S->hr[w][side]) hipFree(S->hr[w][side]); }
  if (S->h_hdr_s) hipHostFree(S->h_hdr_s);
  if (S->h_hdr_r) hipHostFree(S->h_hdr_r);
  if (S->d_hdr_s) hipFree(S->d_hdr_s);
  if (S->d_hdr_r) hipFree(S->d_hdr_r);
  for (int side = 0; side < 2; side++) { if (S->d_rec_s[side]) hipFree(S->d_rec_s[side]); if (S->d_rec_r[side]) hipFree(S->d_rec_r[side]); }
  for (hipEvent_t e : {S->hdr_ev, S->rec_ready, S->rec_done, S->halo_packed, S->halo_arrived, S->u_packed, S->u_arrived}) if (e) hipEventDestroy(e);
  for (int side = 0; side < 2; side++) { if (S->us[side]) hipFree(S->us[side]); if (L->halo_u[side]) hipFree(L->halo_u[side]); L->halo_u[side] = nullptr; }
  L->halo_u_valid = false;
  delete S;
  g_slabs.erase(it);
}

}  // namespace hcs

extern "C" {

int hcl_slab_refresh_halos(hc_lattice *L, int width) {
  HC_REQUIRE(L && (width == 1 || width == 2), "hcl_slab_refresh_halos: width must be 1 or 2");
  if (L->n_slabs == 1) return HC_OK;
  Slab *S = nullptr;
  int rc = make_slab(L, nullptr, &S); if (rc != HC_OK) return rc;
  S->halo_fresh = false;   // the caller wants the halos of the state as it is now, at this width
  return halo_make_fresh(S, width);
}

int hc_slab_stats(hc_lattice *L, double out[8], int reset) {
  HC_REQUIRE(L && out, "hc_slab_stats: null pointer");
  auto it = g_slabs.find(L);
  for (int k = 0; k < 8; k++) out[k] = it == g_slabs.end() ? 0.0 : it->second->stats[k];
  if (reset && it != g_slabs.end()) for (double &v : it->second->stats) v = 0.0;
  return HC_OK;
}

// After the last hcp_add_cell of a slab run: the cells any rank rejected (a particle too close to a wall it can see) go
// everywhere, and the number of distinct cells per type over all slabs comes back (a cell counts where its particle 0 lives).
int hcp_slab_sync_placement(hc_cells *C, long *global_cells_per_type) {
  HC_REQUIRE(C, "hcp_slab_sync_placement: null pointer");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  std::vector<std::vector<char>> all;
  rc = hcm::allgatherv(C->slab_rejected.data(), C->slab_rejected.size() * sizeof(long), all); if (rc != HC_OK) return rc;
  std::unordered_set<long> rejected[8];
  for (auto &blk : all) {
    const long *p = reinterpret_cast<const long *>(blk.data());
    for (size_t i = 0; i + 1 < blk.size() / sizeof(long); i += 2) if (p[i] >= 0 && p[i] < 8) rejected[p[i]].insert(p[i + 1]);
  }
  C->slab_rejected.clear();
  const hc_lattice *L = C->L;
  std::vector<double> counts((size_t)C->ntypes, 0.0);
  for (int t = 0; t < C->ntypes; t++) {
    const size_t nv = (size_t)C->types[t]->host.nv, nc = C->hids[t].size();
    const bool has_rep = C->hrep[t].size() == 3 * nc * nv;
    size_t w = 0;
    for (size_t c = 0; c < nc; c++) {
      if (rejected[t].count(C->hids[t][c])) continue;
      if (w != c) {
        std::copy(C->hpos[t].begin() + 3 * c * nv, C->hpos[t].begin() + 3 * (c + 1) * nv, C->hpos[t].begin() + 3 * w * nv);
        std::copy(C->hvel[t].begin() + 3 * c * nv, C->hvel[t].begin() + 3 * (c + 1) * nv, C->hvel[t].begin() + 3 * w * nv);
        std::copy(C->hfrc[t].begin() + 3 * c * nv, C->hfrc[t].begin() + 3 * (c + 1) * nv, C->hfrc[t].begin() + 3 * w * nv);
        if (has_rep) std::copy(C->hrep[t].begin() + 3 * c * nv, C->hrep[t].begin() + 3 * (c + 1) * nv, C->hrep[t].begin() + 3 * w * nv);
        C->hids[t][w] = C->hids[t][c];
      }
      const long gx = (long)std::floor(C->hpos[t][3 * w * nv] + 0.5) - L->x0;
      if (L->n_slabs == 1 || (gx >= 0 && gx < L->nx)) counts[(size_t)t] += 1.0;
      w++;
    }
    if (w != nc) {
      C->hpos[t].resize(3 * w * nv); C->hvel[t].resize(3 * w * nv); C->hfrc[t].resize(3 * w * nv);
      if (has_rep) C->hrep[t].resize(3 * w * nv);
      C->hids[t].resize(w); C->htag[t].assign(w, 0); C->hdead[t].assign(w * nv, 0);
      C->host_dirty = true;
    }
  }
  rc = hcm::allreduce(counts.data(), C->ntypes, 0); if (rc != HC_OK) return rc;
  if (global_cells_per_type) for (int t = 0; t < C->ntypes; t++) global_cells_per_type[t] = (long)(counts[(size_t)t] + 0.5);
  return HC_OK;
}

}  // extern "C"
