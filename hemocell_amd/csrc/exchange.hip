// Slab runs: cell extents, envelope records (pack / unpack / remove), ownership counts and the
// ParticleInfo statistics over owned vertices.
//
// Replaces (file:line in the HemoCell tree):
//   core/hemoCellFields.cpp:377-499, core/hemoCellParticleDataTransfer.cpp:33-466   syncEnvelopes (device side)
//   core/hemoCellParticleField.cpp:173-235        addParticle merge rule
//   helper/particleInfo.cpp:30-140                force / velocity statistics
#include "cells.h"

namespace {

// ---------------------------------------------------------------------------- multi-slab cell exchange
// per-cell [min_x, max_x, number of particles whose nearest node lies in this slab, deletion state (0 complete, 1 gone,
// 2 incomplete)]; removed particles of an incomplete cell do not count
__global__ __launch_bounds__(256) void cell_extent_kernel(int nv, const double *px, double *out, int x0, int nx, const int *tag, const unsigned char *dead) {
  __shared__ double lo[256], hi[256];
  __shared__ int own[256];
  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * nv;
  const int state = tag[blockIdx.x];
  double a = 1e300, b = -1e300; int o = 0;
  if (state != 1)
    for (int i = tid; i < nv; i += 256) {
      if (state == 2 && dead[base + i]) continue;
      const double x = px[base + i]; a = fmin(a, x); b = fmax(b, x);
      const long gx = nearest_node(x) - x0;
      o += (gx >= 0 && gx < nx) ? 1 : 0;
    }
  lo[tid] = a; hi[tid] = b; own[tid] = o;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { lo[tid] = fmin(lo[tid], lo[tid + s]); hi[tid] = fmax(hi[tid], hi[tid + s]); own[tid] += own[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { double *r = out + 4 * (long)blockIdx.x; r[0] = lo[0]; r[1] = hi[0]; r[2] = (double)own[0]; r[3] = (double)state; }
}


// record layout per vertex: pos[3] vel[3] force[3] and, once a repulsion is enabled (rec == 12), force_repulsion[3]
// (the fields of serializeValues_t that change, core/hemoCellParticle.h:45-63)
__global__ __launch_bounds__(256) void pack_cells_kernel(int nv, int rec, const int *slots, VertArrays a, double *buf, double x_shift) {
  const long src = (long)slots[blockIdx.x] * nv, dst = (long)blockIdx.x * nv;
  for (int i = threadIdx.x; i < nv; i += 256) {
    double *r = buf + (dst + i) * rec;
    r[0] = a.p[0][src + i] + x_shift; r[1] = a.p[1][src + i]; r[2] = a.p[2][src + i];
    r[3] = a.v[0][src + i]; r[4] = a.v[1][src + i]; r[5] = a.v[2][src + i];
    r[6] = a.f[0][src + i]; r[7] = a.f[1][src + i]; r[8] = a.f[2][src + i];
    if (rec == 12) { r[9] = a.r[0][src + i]; r[10] = a.r[1][src + i]; r[11] = a.r[2][src + i]; }
  }
}

// merge rule of HemoCellParticleField::addParticle (core/hemoCellParticleField.cpp:173-235): a local
// particle wins over an incoming copy; "local" = its nearest lattice node lies in this slab
__global__ __launch_bounds__(256) void unpack_cells_kernel(int nv, int rec, const int *slots, const int *is_new, VertArrays a, const double *buf,
                                                           int x0, int nx, int *late) {
  const long dst = (long)slots[blockIdx.x] * nv, src = (long)blockIdx.x * nv;
  const bool fresh = is_new[blockIdx.x] != 0;
  if (fresh && threadIdx.x == 0) a.tag[slots[blockIdx.x]] = 0;   // a new copy is a complete cell, whatever lived in this slot before
  for (int i = threadIdx.x; i < nv; i += 256) {
    if (fresh) {
      a.dead[dst + i] = 0;
      // the envelope check: when a cell first arrives none of its particles may already reach this slab's nodes with its
      // stencil (x in (x0 - 1, x0 + nx)) -- it would have spread and been interpolated here before the copy existed
      const double x = buf[(src + i) * rec];
      if (x > (double)x0 - 1.0 && x < (double)(x0 + nx)) atomicAdd_system(late, 1);
    }
    bool take = fresh;
    if (!take) {
      const long gx = nearest_node(a.p[0][dst + i]) - x0;
      take = !(gx >= 0 && gx < nx);
    }
    if (take) {
      const double *r = buf + (src + i) * rec;
      a.p[0][dst + i] = r[0]; a.p[1][dst + i] = r[1]; a.p[2][dst + i] = r[2];
      a.v[0][dst + i] = r[3]; a.v[1][dst + i] = r[4]; a.v[2][dst + i] = r[5];
      a.f[0][dst + i] = r[6]; a.f[1][dst + i] = r[7]; a.f[2][dst + i] = r[8];
      if (rec == 12) { a.r[0][dst + i] = r[9]; a.r[1][dst + i] = r[10]; a.r[2][dst + i] = r[11]; }
    }
  }
}

__global__ __launch_bounds__(256) void move_cells_kernel(int nv, const int *src_slots, const int *dst_slots, VertArrays a) {
  const long src = (long)src_slots[blockIdx.x] * nv, dst = (long)dst_slots[blockIdx.x] * nv;
  if (threadIdx.x == 0) a.tag[dst_slots[blockIdx.x]] = a.tag[src_slots[blockIdx.x]];
  for (int i = threadIdx.x; i < nv; i += 256) {
    a.dead[dst + i] = a.dead[src + i];
    for (int d = 0; d < 3; d++) {
      a.p[d][dst + i] = a.p[d][src + i]; a.v[d][dst + i] = a.v[d][src + i]; a.f[d][dst + i] = a.f[d][src + i];
      if (a.r[d]) a.r[d][dst + i] = a.r[d][src + i];
    }
  }
}
// slots [first, first + n) no longer hold a cell: neutral state for whoever moves in next
__global__ void clear_state_kernel(int nv, long first, long n, VertArrays a) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k < n) a.tag[first + k] = 0;
  if (k < n * nv) a.dead[first * nv + k] = 0;
}

// ParticleInfo statistics (helper/particleInfo.cpp:30-95): magnitude of v (what 1) or of force + force_repulsion (what 2)
// over the vertices this slab owns (findParticles(localDomain))
__global__ __launch_bounds__(256) void vertex_stats_kernel(long n, int what, int all_owned, int x0, int nx, const double *px, const double *a0,
                                                           const double *a1, const double *a2, const double *r0, const double *r1, const double *r2,
                                                           double *partial, int accumulate, const int *vert_cell, const int *tag, const unsigned char *dead) {
  StatAcc acc{1e300, -1e300, 0.0, 0};
  if (accumulate) { const double *o = partial + 4 * blockIdx.x; if (threadIdx.x == 0 && o[3] > 0) { acc.mn = o[0]; acc.mx = o[1]; acc.sum = o[2]; acc.n = (long)o[3]; } }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)STAT_BLOCKS * 256) {
    if (dead[i] || tag[vert_cell[i]] == 1) continue;
    if (!all_owned) { const long gx = nearest_node(px[i]) - x0; if (gx < 0 || gx >= nx) continue; }
    double v0 = a0[i], v1 = a1[i], v2 = a2[i];
    if (what == 2 && r0) { v0 = v0 + r0[i]; v1 = v1 + r1[i]; v2 = v2 + r2[i]; }
    stat_add(acc, sqrt(v0 * v0 + v1 * v1 + v2 * v2));
  }
  __syncthreads();   // every thread has read the previous partial before it is overwritten
  stat_block_store(acc, partial);
}

__global__ __launch_bounds__(256) void owned_count_kernel(long n, const double *px, int x0, int nx, unsigned long long *count, const int *vert_cell,
                                                          const int *tag, const unsigned char *dead) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  int mine = 0;
  if (i < n && !dead[i] && tag[vert_cell[i]] != 1) { const long gx = nearest_node(px[i]) - x0; mine = (gx >= 0 && gx < nx) ? 1 : 0; }
  const unsigned long long b = __ballot(mine);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

}  // namespace

// persistent device / pinned host block of the statistics reductions (no allocation per call)
static int stat_scratch(hc_cells *C) {
  if (C->d_stat) return HC_OK;
  HC_HIP(hipMalloc((void **)&C->d_stat, (size_t)STAT_BLOCKS * 4 * sizeof(double)));
  HC_HIP(hipHostMalloc((void **)&C->h_stat, (size_t)STAT_BLOCKS * 4 * sizeof(double), hipHostMallocDefault));
  return HC_OK;
}

extern "C" {

// The kernel and the copy into pinned staging are enqueued and the call returns; _end waits for that copy alone (an
// event), not for whatever was enqueued on the stream afterwards, so a caller can start the extents early in a step and
// pick them up later without draining the queue.
int hcp_cell_extents_begin(hc_cells *C, int type) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes, "hcp_cell_extents_begin: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long nc = C->ncells[type];
  C->ext_n[type] = nc; C->ext_pending[type] = true;
  if (nc == 0) return HC_OK;
  if (nc > C->ext_cap[type]) {
    if (C->ext_done[type]) HC_HIP(hipEventSynchronize(C->ext_done[type]));
    if (C->h_ext[type]) HC_HIP(hipHostFree(C->h_ext[type]));
    C->d_ext[type] = C->h_ext[type] = nullptr; C->ext_cap[type] = 0;
    const long cap = nc + nc / 4 + 64;
    // the kernel stores its few KB straight into pinned host memory: no copy operation sits in the stream between this
    // kernel and the next one (an asynchronous device-to-host copy there held the following spread back by ~0.1 ms)
    HC_HIP(hipHostMalloc((void **)&C->h_ext[type], (size_t)(4 * cap) * sizeof(double), hipHostMallocMapped));
    HC_HIP(hipHostGetDevicePointer((void **)&C->d_ext[type], C->h_ext[type], 0));
    C->ext_cap[type] = cap;
  }
  if (!C->ext_done[type]) HC_HIP(hipEventCreateWithFlags(&C->ext_done[type], hipEventDisableTiming));
  hipLaunchKernelGGL(cell_extent_kernel, dim3((unsigned)nc), dim3(256), 0, hc::stream(), C->types[type]->host.nv,
                     (const double *)(C->pos[0] + C->first[type]), C->d_ext[type], C->L->x0, C->L->nx, (const int *)(C->d_tag + C->cell0[type]),
                     (const unsigned char *)(C->d_vdead + C->first[type]));
  HC_HIP(hipGetLastError());
  HC_HIP(hipEventRecord(C->ext_done[type], hc::stream()));
  return HC_OK;
}

int hcp_cell_extents_end(hc_cells *C, int type, double *minmax) {
  HC_REQUIRE(C && minmax && type >= 0 && type < C->ntypes, "hcp_cell_extents_end: bad arguments");
  HC_REQUIRE(C->ext_pending[type], "hcp_cell_extents_end: no hcp_cell_extents_begin is pending for this type");
  HC_REQUIRE(C->ext_n[type] == C->ncells[type], "hcp_cell_extents_end: the cell set changed since hcp_cell_extents_begin");
  C->ext_pending[type] = false;
  if (C->ext_n[type] == 0) return HC_OK;
  HC_HIP(hipEventSynchronize(C->ext_done[type]));
  for (long c = 0; c < C->ext_n[type]; c++) for (int k = 0; k < 3; k++) minmax[3 * c + k] = C->h_ext[type][4 * c + k];
  return HC_OK;
}

int hcp_cell_extents(hc_cells *C, int type, double *minmax) {
  HC_REQUIRE(C && minmax && type >= 0 && type < C->ntypes, "hcp_cell_extents: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = hcp_cell_extents_begin(C, type); if (rc != HC_OK) return rc;
  return hcp_cell_extents_end(C, type, minmax);
}

size_t hcp_record_doubles(const hc_cells *C, int type) {
  if (!C || type < 0 || type >= C->ntypes) return 0;
  return (size_t)C->types[type]->host.nv * (C->rep_on() ? 12 : 9);
}

int hcp_pack_cells(hc_cells *C, int type, const int *slots, int n, double x_shift, double *dev_buf) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes && n >= 0, "hcp_pack_cells: bad arguments");
  if (n == 0) return HC_OK;
  HC_REQUIRE(slots && dev_buf, "hcp_pack_cells: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  for (int i = 0; i < n; i++) HC_REQUIRE(slots[i] >= 0 && slots[i] < C->ncells[type], "hcp_pack_cells: slot out of range");
  int *d_slots = nullptr;
  rc = stage_ints(C, 0, &d_slots, slots, n); if (rc != HC_OK) return rc;
  hipLaunchKernelGGL(pack_cells_kernel, dim3((unsigned)n), dim3(256), 0, hc::stream(), C->types[type]->host.nv, C->rep_on() ? 12 : 9, (const int *)d_slots, vert_arrays(C, type), dev_buf, x_shift);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

int hcp_unpack_cells(hc_cells *C, int type, const int *slots, const long *cell_ids, const int *is_new, int n, const double *dev_buf) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes && n >= 0, "hcp_unpack_cells: bad arguments");
  if (n == 0) return HC_OK;
  HC_REQUIRE(slots && cell_ids && is_new && dev_buf, "hcp_unpack_cells: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  long n_new = 0;
  for (int i = 0; i < n; i++) {
    if (is_new[i]) { HC_REQUIRE(slots[i] == C->ncells[type] + n_new, "hcp_unpack_cells: new cells must be appended in slot order"); n_new++; }
    else HC_REQUIRE(slots[i] >= 0 && slots[i] < C->ncells[type], "hcp_unpack_cells: slot out of range");
  }
  if (C->ncells[type] + n_new > C->capc[type]) {
    // slow path: grow the device regions through the host staging
    rc = sync_to_host(C); if (rc != HC_OK) return rc;
    const size_t add = (size_t)C->types[type]->host.nv * 3;
    for (int i = 0; i < n; i++) {
      if (!is_new[i]) continue;
      C->hpos[type].resize(C->hpos[type].size() + add, 0.0); C->hvel[type].resize(C->hvel[type].size() + add, 0.0); C->hfrc[type].resize(C->hfrc[type].size() + add, 0.0);
      host_append_state(C, type, cell_ids[i]);   // id, deletion state and force_repulsion grow with the vertices: the state of the cells already here survives
    }
    C->host_dirty = true;
    rc = sync_to_device(C); if (rc != HC_OK) return rc;
  } else {
    for (int i = 0; i < n; i++) if (is_new[i]) C->hids[type].push_back(cell_ids[i]);
    C->ncells[type] += n_new;
    C->nverts += n_new * C->types[type]->host.nv;
  }
  int *d_slots = nullptr, *d_new = nullptr;
  rc = stage_ints(C, 0, &d_slots, slots, n); if (rc != HC_OK) return rc;
  rc = stage_ints(C, 1, &d_new, is_new, n); if (rc != HC_OK) return rc;
  hipLaunchKernelGGL(unpack_cells_kernel, dim3((unsigned)n), dim3(256), 0, hc::stream(), C->types[type]->host.nv, C->rep_on() ? 12 : 9, (const int *)d_slots, (const int *)d_new,
                     vert_arrays(C, type), dev_buf, C->L->x0, C->L->nx, C->d_env_viol);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

int hcp_remove_cells(hc_cells *C, int type, const int *slots, int n) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes && n >= 0, "hcp_remove_cells: bad arguments");
  if (n == 0) return HC_OK;
  HC_REQUIRE(slots, "hcp_remove_cells: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long nc = C->ncells[type], new_nc = nc - n;
  std::vector<char> dead((size_t)nc, 0);
  for (int i = 0; i < n; i++) { HC_REQUIRE(slots[i] >= 0 && slots[i] < nc && !dead[(size_t)slots[i]], "hcp_remove_cells: bad slot list"); dead[(size_t)slots[i]] = 1; }
  // holes below new_nc are filled with the live cells at and above new_nc (disjoint ranges: no hazard)
  std::vector<int> src, dst;
  long tail = new_nc;
  for (long h = 0; h < new_nc; h++) {
    if (!dead[(size_t)h]) continue;
    while (tail < nc && dead[(size_t)tail]) tail++;
    src.push_back((int)tail); dst.push_back((int)h);
    C->hids[type][(size_t)h] = C->hids[type][(size_t)tail];
    tail++;
  }
  C->hids[type].resize((size_t)new_nc);
  if (!src.empty()) {
    int *d_src = nullptr, *d_dst = nullptr;
    rc = stage_ints(C, 0, &d_src, src.data(), (int)src.size()); if (rc != HC_OK) return rc;
    rc = stage_ints(C, 1, &d_dst, dst.data(), (int)dst.size()); if (rc != HC_OK) return rc;
    hipLaunchKernelGGL(move_cells_kernel, dim3((unsigned)src.size()), dim3(256), 0, hc::stream(), C->types[type]->host.nv, (const int *)d_src, (const int *)d_dst, vert_arrays(C, type));
    HC_HIP(hipGetLastError());
  }
  {
    const int nv = C->types[type]->host.nv;
    hipLaunchKernelGGL(clear_state_kernel, dim3((unsigned)(((long)n * nv + 255) / 256)), dim3(256), 0, hc::stream(), nv, new_nc, (long)n, vert_arrays(C, type));
    HC_HIP(hipGetLastError());
  }
  C->ncells[type] = new_nc;
  C->nverts -= (long)n * C->types[type]->host.nv;
  return HC_OK;
}

int hcp_owned_vertices(hc_cells *C, long *n_owned) {
  HC_REQUIRE(C && n_owned, "hcp_owned_vertices: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  rc = stat_scratch(C); if (rc != HC_OK) return rc;
  unsigned long long *d = reinterpret_cast<unsigned long long *>(C->d_stat);
  HC_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), hc::stream()));
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    hipLaunchKernelGGL(owned_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), n, (const double *)(C->pos[0] + f), C->L->x0, C->L->nx, d,
                       (const int *)(C->d_vert_cell + f), (const int *)C->d_tag, (const unsigned char *)(C->d_vdead + f));
  }
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(C->h_stat, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  *n_owned = (long)*reinterpret_cast<unsigned long long *>(C->h_stat);
  return HC_OK;
}


// ParticleInfo::calculate{Velocity,Force}Statistics (helper/particleInfo.cpp:30-140) as a device reduction
int hcp_vertex_stats(hc_cells *C, int what, double out[3], long *n) {
  HC_REQUIRE(C && out && n && (what == 1 || what == 2), "hcp_vertex_stats: bad arguments (what: 1 velocity, 2 force)");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  rc = stat_scratch(C); if (rc != HC_OK) return rc;
  double *d_partial = C->d_stat;
  HC_HIP(hipMemsetAsync(d_partial, 0, (size_t)STAT_BLOCKS * 4 * sizeof(double), hc::stream()));
  const hc_lattice *L = C->L;
  int launched = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const long nt = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (nt == 0) continue;
    double **src = what == 1 ? C->vel : C->frc;
    hipLaunchKernelGGL(vertex_stats_kernel, dim3(STAT_BLOCKS), dim3(256), 0, hc::stream(), nt, what, L->n_slabs == 1 ? 1 : 0, L->x0, L->nx,
                       (const double *)(C->pos[0] + f), (const double *)(src[0] + f), (const double *)(src[1] + f), (const double *)(src[2] + f),
                       C->rep[0] ? (const double *)(C->rep[0] + f) : nullptr, C->rep[1] ? (const double *)(C->rep[1] + f) : nullptr,
                       C->rep[2] ? (const double *)(C->rep[2] + f) : nullptr, d_partial, launched, (const int *)(C->d_vert_cell + f), (const int *)C->d_tag,
                       (const unsigned char *)(C->d_vdead + f));
    launched = 1;
  }
  HC_HIP(hipGetLastError());
  return hc::stat_finish(d_partial, out, n, C->h_stat);
}


}  // extern "C"
