// Membrane vertices on the GPU: storage (host staging <-> device regions), cell placement, Euler advance with
// boundary tagging, and the whole HemoCell::iterate.
//
// Replaces (file:line in the HemoCell tree):
//   io/readPositionsBloodCells.cpp:113-169, 303-353                           cell placement (hcp_add_cell)
//   core/hemoCellParticle.h:188-203, core/hemoCellParticleField.cpp:566-588   advance
//   core/hemoCell.cpp:299-376                                                 iterate
#include "cells.h"
#include <cstddef>
#include <utility>

namespace {

// ----------------------------------------------------------------------------
// advance + boundary tagging
__global__ __launch_bounds__(256) void advance_kernel(LatView v, long n, double *px, double *py, double *pz, const double *vx,
                                                      const double *vy, const double *vz, const int *vert_cell, int *tag, unsigned char *dead,
                                                      int *counters, int whole_cell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int cell = vert_cell[i];
  if (dead[i] || tag[cell] == 1) return;   // removed particle / cell already gone
  const double x = px[i] + vx[i], y = py[i] + vy[i], z = pz[i] + vz[i];
  px[i] = x; py[i] = y; pz[i] = z;
  // nearest node is a boundary -> tag (core/hemoCellParticleField.cpp:571-583)
  long gx = nearest_node(x) - v.x0, gy = nearest_node(y), gz = nearest_node(z);
  bool inside = true;
  if (v.wrap_x) gx = pmod(gx, v.nx); else inside = inside && (gx >= (v.halo_x ? -HALO : 0) && gx < (v.halo_x ? v.nx + HALO : v.nx));
  if (gy < 0 || gy >= v.ny) { if (v.per_y) gy = pmod(gy, v.ny); else inside = false; }
  if (gz < 0 || gz >= v.nz) { if (v.per_z) gz = pmod(gz, v.nz); else inside = false; }
  if (inside && v.mask[(gx + HALO) * (long)v.plane + gy * v.nz + gz] != 0) {
    if (whole_cell) {   // HC_DELETE_CELL: the cell is gone at once
      if (atomicExch(&tag[cell], 1) != 1) atomicAdd(&counters[0], 1);
    } else {            // HC_DELETE_PARTICLE: removeParticles(1) takes this particle out, the cell is incomplete from now on (:584, :304-321)
      dead[i] = 1;
      atomicAdd(&counters[2], 1);
      if (atomicCAS(&tag[cell], 0, 2) == 0) atomicAdd(&counters[1], 1);
    }
  }
}

// the counters go to pinned host memory by a one-thread kernel: a copy operation in the stream would hold the next kernel back
__global__ void publish_counters_kernel(const int *counters, int *host_view) { for (int k = 0; k < 3; k++) host_view[k] = counters[k]; }
__global__ void sub_counters_kernel(int *counters, int a, int b, int c) { atomicSub(&counters[0], a); atomicSub(&counters[1], b); atomicSub(&counters[2], c); }

// deleteIncompleteCells (core/hemoCellParticleField.cpp:512-553): every incomplete cell goes
__global__ void delete_incomplete_kernel(long ncells, int *tag, int *counters) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  if (tag[c] == 2) { tag[c] = 1; atomicAdd(&counters[0], 1); }
}

__global__ void add_vertex_force_kernel(int n, const long *idx, const double *f, double *fx, double *fy, double *fz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long v = idx[i];
  fx[v] += f[3 * i]; fy[v] += f[3 * i + 1]; fz[v] += f[3 * i + 2];
}

__global__ void fill_vert_cell_kernel(long n, int nv, int cell0, long first, int *vert_cell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  vert_cell[first + i] = cell0 + (int)(i / nv);
}

}  // namespace

namespace hcc {

int free_device_arrays(hc_cells *C) {
  for (int d = 0; d < 3; d++) {
    if (C->pos[d]) hipFree(C->pos[d]);
    if (C->vel[d]) hipFree(C->vel[d]);
    if (C->frc[d]) hipFree(C->frc[d]);
    C->pos[d] = C->vel[d] = C->frc[d] = nullptr;
  }
  if (C->d_tag) hipFree(C->d_tag);
  if (C->d_vdead) hipFree(C->d_vdead);
  C->d_tag = nullptr; C->d_vdead = nullptr; C->cap = 0; C->tag_cap = 0;
  for (int t = 0; t < 8; t++) C->capc[t] = 0;
  if (C->d_vert_cell) hipFree(C->d_vert_cell);
  C->d_vert_cell = nullptr;
  for (int d = 0; d < 3; d++) { if (C->rep[d]) hipFree(C->rep[d]); C->rep[d] = nullptr; }
  for (int k = 0; k < 2; k++) { if (C->d_keys[k]) hipFree(C->d_keys[k]); if (C->d_vals[k]) hipFree(C->d_vals[k]); C->d_keys[k] = nullptr; C->d_vals[k] = nullptr; }
  if (C->d_sort_tmp) hipFree(C->d_sort_tmp);
  C->d_sort_tmp = nullptr; C->sort_tmp_bytes = 0; C->sort_cap = 0;
  return HC_OK;
}

// host staging -> device (after placement, upload or a deletion)
int sync_to_device(hc_cells *C) {
  if (!C->host_dirty) return HC_OK;
  bool grow = false;
  long nverts = 0;
  for (int t = 0; t < C->ntypes; t++) {
    C->ncells[t] = (long)C->hids[t].size();
    nverts += C->ncells[t] * C->types[t]->host.nv;
    if (C->ncells[t] > C->capc[t]) grow = true;
  }
  if (grow || C->cap == 0) {
    free_device_arrays(C);
    long cap = 0, capcells = 0;
    for (int t = 0; t < C->ntypes; t++) {
      C->capc[t] = C->ncells[t] + C->ncells[t] / 4 + 64;
      C->first[t] = cap; C->cell0[t] = capcells;
      cap += C->capc[t] * C->types[t]->host.nv; capcells += C->capc[t];
    }
    C->cap = cap > 0 ? cap : 1;
    for (int d = 0; d < 3; d++) {
      HC_HIP(hipMalloc((void **)&C->pos[d], C->cap * sizeof(double)));
      HC_HIP(hipMalloc((void **)&C->vel[d], C->cap * sizeof(double)));
      HC_HIP(hipMalloc((void **)&C->frc[d], C->cap * sizeof(double)));
    }
    HC_HIP(hipMalloc((void **)&C->d_vert_cell, C->cap * sizeof(int)));
    if (C->rep_on()) for (int d = 0; d < 3; d++) { HC_HIP(hipMalloc((void **)&C->rep[d], C->cap * sizeof(double))); HC_HIP(hipMemset(C->rep[d], 0, C->cap * sizeof(double))); }
    C->tag_cap = capcells + 1;
    HC_HIP(hipMalloc((void **)&C->d_tag, C->tag_cap * sizeof(int)));
    HC_HIP(hipMalloc((void **)&C->d_vdead, (size_t)C->cap));
    for (int t = 0; t < C->ntypes; t++) {
      const long n = C->capc[t] * C->types[t]->host.nv;
      hipLaunchKernelGGL(fill_vert_cell_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), n, C->types[t]->host.nv, (int)C->cell0[t], C->first[t], C->d_vert_cell);
      HC_HIP(hipGetLastError());
    }
  }
  C->nverts = nverts;
  std::vector<double> tmp;
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv;
    if (n == 0) continue;
    std::vector<double> *src[3] = {&C->hpos[t], &C->hvel[t], &C->hfrc[t]};
    double **dst[3] = {C->pos, C->vel, C->frc};
    tmp.resize((size_t)n);
    for (int w = 0; w < 3; w++)
      for (int d = 0; d < 3; d++) {
        for (long i = 0; i < n; i++) tmp[(size_t)i] = (*src[w])[(size_t)(3 * i + d)];
        HC_HIP(hipMemcpy(dst[w][d] + C->first[t], tmp.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      }
  }
  HC_HIP(hipMemsetAsync(C->d_tag, 0, C->tag_cap * sizeof(int), hc::stream()));
  HC_HIP(hipMemsetAsync(C->d_vdead, 0, (size_t)C->cap, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  for (int t = 0; t < C->ntypes; t++) {
    const long nc = C->ncells[t], n = nc * C->types[t]->host.nv;
    if (nc == 0) continue;
    // deletion state and force_repulsion travel with the cells (a staging written before they existed has none: all live, zero)
    if ((long)C->htag[t].size() == nc) HC_HIP(hipMemcpy(C->d_tag + C->cell0[t], C->htag[t].data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice));
    if ((long)C->hdead[t].size() == n) HC_HIP(hipMemcpy(C->d_vdead + C->first[t], C->hdead[t].data(), (size_t)n, hipMemcpyHostToDevice));
    if (C->rep[0] && (long)C->hrep[t].size() == 3 * n) {
      tmp.resize((size_t)n);
      for (int d = 0; d < 3; d++) {
        for (long i = 0; i < n; i++) tmp[(size_t)i] = C->hrep[t][(size_t)(3 * i + d)];
        HC_HIP(hipMemcpy(C->rep[d] + C->first[t], tmp.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      }
    }
  }
  C->host_dirty = false;
  return HC_OK;
}

// device -> host staging (before host-side edits)
int sync_to_host(hc_cells *C) {
  if (C->host_dirty) return HC_OK;  // host already authoritative
  HC_HIP(hipStreamSynchronize(hc::stream()));
  std::vector<double> tmp;
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv;
    std::vector<double> *dstv[3] = {&C->hpos[t], &C->hvel[t], &C->hfrc[t]};
    double **src[3] = {C->pos, C->vel, C->frc};
    tmp.resize((size_t)n);
    for (int w = 0; w < 3; w++) {
      dstv[w]->resize((size_t)(3 * n));
      for (int d = 0; d < 3; d++) {
        if (n) HC_HIP(hipMemcpy(tmp.data(), src[w][d] + C->first[t], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        for (long i = 0; i < n; i++) (*dstv[w])[(size_t)(3 * i + d)] = tmp[(size_t)i];
      }
    }
    C->htag[t].assign((size_t)C->ncells[t], 0); C->hdead[t].assign((size_t)n, 0);
    if (C->ncells[t] && C->d_tag) {
      HC_HIP(hipMemcpy(C->htag[t].data(), C->d_tag + C->cell0[t], (size_t)C->ncells[t] * sizeof(int), hipMemcpyDeviceToHost));
      HC_HIP(hipMemcpy(C->hdead[t].data(), C->d_vdead + C->first[t], (size_t)n, hipMemcpyDeviceToHost));
    }
    C->hrep[t].clear();
    if (C->rep[0] && n) {
      C->hrep[t].resize((size_t)(3 * n));
      for (int d = 0; d < 3; d++) {
        HC_HIP(hipMemcpy(tmp.data(), C->rep[d] + C->first[t], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        for (long i = 0; i < n; i++) C->hrep[t][(size_t)(3 * i + d)] = tmp[(size_t)i];
      }
    }
  }
  return HC_OK;
}

VertArrays vert_arrays(hc_cells *C, int t) {
  VertArrays a;
  for (int d = 0; d < 3; d++) {
    a.p[d] = C->pos[d] + C->first[t]; a.v[d] = C->vel[d] + C->first[t]; a.f[d] = C->frc[d] + C->first[t];
    a.r[d] = C->rep[d] ? C->rep[d] + C->first[t] : nullptr;
  }
  a.dead = C->d_vdead + C->first[t]; a.tag = C->d_tag + C->cell0[t];
  return a;
}
// stage a small host int array on the device in a persistent scratch slot.  The source is copied into a pinned block
// first, so the caller's array may be a temporary and the copy really is asynchronous; the copy and every later use are
// ordered on the stream in use.  The slot's event tells when the pinned block may be rewritten.
int stage_ints(hc_cells *C, int which, int **d, const int *h, int n) {
  if (!C->iscratch_ev[which]) HC_HIP(hipEventCreateWithFlags(&C->iscratch_ev[which], hipEventDisableTiming));
  else HC_HIP(hipEventSynchronize(C->iscratch_ev[which]));   // the previous copy has left the pinned block (normally long ago)
  if ((size_t)n > C->iscratch_cap[which]) {
    HC_HIP(hipDeviceSynchronize());   // the old device block may still be in use on either stream
    if (C->d_iscratch[which]) HC_HIP(hipFree(C->d_iscratch[which]));
    if (C->h_iscratch[which]) HC_HIP(hipHostFree(C->h_iscratch[which]));
    C->d_iscratch[which] = C->h_iscratch[which] = nullptr;
    C->iscratch_cap[which] = (size_t)n * 2 + 256;
    HC_HIP(hipMalloc((void **)&C->d_iscratch[which], C->iscratch_cap[which] * sizeof(int)));
    HC_HIP(hipHostMalloc((void **)&C->h_iscratch[which], C->iscratch_cap[which] * sizeof(int), hipHostMallocDefault));
  }
  if (n > 0) {
    std::memcpy(C->h_iscratch[which], h, (size_t)n * sizeof(int));
    HC_HIP(hipMemcpyAsync(C->d_iscratch[which], C->h_iscratch[which], (size_t)n * sizeof(int), hipMemcpyHostToDevice, hc::stream()));
    HC_HIP(hipEventRecord(C->iscratch_ev[which], hc::stream()));
  }
  *d = C->d_iscratch[which];
  return HC_OK;
}

// id, deletion state and force_repulsion of a cell whose vertices were just appended to the host staging of its type
void host_append_state(hc_cells *C, int type, long cell_id) {
  const size_t nv = (size_t)C->types[type]->host.nv, nc = C->hids[type].size();
  C->htag[type].resize(nc, 0); C->hdead[type].resize(nc * nv, 0);
  if (C->rep_on()) C->hrep[type].resize(3 * nc * nv, 0.0);
  C->hids[type].push_back(cell_id);
  C->htag[type].push_back(0); C->hdead[type].resize((nc + 1) * nv, 0);
  if (C->rep_on()) C->hrep[type].resize(3 * (nc + 1) * nv, 0.0);
  C->host_dirty = true;
}

}  // namespace hcc

extern "C" {

int hcp_celltype_create(hc_celltype **out, int model, int shape, const hc_params *P, const hc_material *M) {
  HC_REQUIRE(out && P && M, "hcp_celltype_create: null pointer");
  if (hc::stream() == nullptr) { hc::set_error("hcp_celltype_create: hc_init() has not been called"); return HC_ERR_STATE; }
  hc_celltype *T = new hc_celltype();
  std::string err = build_cell_tables(T->host, model, shape, *P, *M);
  if (!err.empty()) { delete T; hc::set_error("hcp_celltype_create: " + err); return HC_ERR_ARG; }
  const CellTables &H = T->host;
  int rc = HC_OK;
  auto up_i = [&](int **d, const std::vector<int> &v) { if (rc == HC_OK) rc = upload_vec(d, v); };
  auto up_d = [&](double **d, const std::vector<double> &v) { if (rc == HC_OK) rc = upload_vec(d, v); };
  up_i(&T->d_tri, flatten(H.triangles)); up_i(&T->d_edge, flatten(H.edges));
  up_i(&T->d_ebt, flatten(H.edge_bending_triangles)); up_i(&T->d_ebo, flatten(H.edge_bending_outer));
  up_i(&T->d_iedge, flatten(H.inner_edges));
  up_i(&T->d_vtri, H.vtri); up_i(&T->d_vtri_k, H.vtri_k); up_i(&T->d_vedge, H.vedge); up_i(&T->d_vedge_s, H.vedge_s);
  up_i(&T->d_bsrc, H.bsrc); up_i(&T->d_vouter, H.vouter); up_i(&T->d_vinner, H.vinner); up_i(&T->d_vinner_s, H.vinner_s);
  up_i(&T->d_ring, flatten(H.vertex_vertexes)); up_i(&T->d_nring, H.vertex_n_vertexes);
  up_d(&T->d_tri_area_eq, H.triangle_area_eq); up_d(&T->d_edge_len_eq, H.edge_length_eq);
  up_d(&T->d_edge_angle_eq, H.edge_angle_eq); up_d(&T->d_patch_eq, H.patch_dist_eq); up_d(&T->d_iedge_len_eq, H.inner_edge_length_eq);
  if (rc != HC_OK) return rc;
  *out = T;
  return HC_OK;
}

int hcp_celltype_destroy(hc_celltype *T) {
  if (!T) return HC_OK;
  int *ip[] = {T->d_tri, T->d_edge, T->d_ebt, T->d_ebo, T->d_iedge, T->d_vtri, T->d_vtri_k, T->d_vedge, T->d_vedge_s, T->d_bsrc,
               T->d_vouter, T->d_vinner, T->d_vinner_s, T->d_ring, T->d_nring};
  double *dp[] = {T->d_tri_area_eq, T->d_edge_len_eq, T->d_edge_angle_eq, T->d_patch_eq, T->d_iedge_len_eq};
  for (int *p : ip) if (p) hipFree(p);
  for (double *p : dp) if (p) hipFree(p);
  delete T;
  return HC_OK;
}

int hcp_celltype_sizes(const hc_celltype *T, int out[4]) {
  HC_REQUIRE(T && out, "hcp_celltype_sizes: null pointer");
  out[0] = T->host.nv; out[1] = T->host.nt; out[2] = T->host.ne; out[3] = T->host.nie;
  return HC_OK;
}

int hcp_celltype_tables(const hc_celltype *T, double *vertices, long *triangles, long *edges, double *edge_length_eq,
                        double *edge_angle_eq, double *triangle_area_eq, long *vertex_vertexes, double *patch_dist_eq,
                        double scalars[9]) {
  HC_REQUIRE(T, "hcp_celltype_tables: null cell type");
  const CellTables &H = T->host;
  if (vertices) for (int i = 0; i < H.nv; i++) for (int d = 0; d < 3; d++) vertices[3 * i + d] = H.vertices[i][d];
  if (triangles) for (int i = 0; i < H.nt; i++) for (int d = 0; d < 3; d++) triangles[3 * i + d] = H.triangles[i][d];
  if (edges) for (int i = 0; i < H.ne; i++) for (int d = 0; d < 2; d++) edges[2 * i + d] = H.edges[i][d];
  if (edge_length_eq) std::copy(H.edge_length_eq.begin(), H.edge_length_eq.end(), edge_length_eq);
  if (edge_angle_eq) std::copy(H.edge_angle_eq.begin(), H.edge_angle_eq.end(), edge_angle_eq);
  if (triangle_area_eq) std::copy(H.triangle_area_eq.begin(), H.triangle_area_eq.end(), triangle_area_eq);
  if (vertex_vertexes) for (int i = 0; i < H.nv; i++) for (int d = 0; d < 6; d++) vertex_vertexes[6 * i + d] = H.vertex_vertexes[i][d];
  if (patch_dist_eq) std::copy(H.patch_dist_eq.begin(), H.patch_dist_eq.end(), patch_dist_eq);
  if (scalars) {
    const double s[9] = {H.volume_eq, H.area_mean_eq, H.edge_mean_eq, H.angle_mean_eq, H.k_volume, H.k_area, H.k_link, H.k_bend, H.eta_m};
    std::copy(s, s + 9, scalars);
  }
  return HC_OK;
}

int hcp_celltype_tables2(const hc_celltype *T, long *edge_bending_triangles, long *edge_bending_outer_points, long *inner_edges,
                         double *inner_edge_length_eq, int *vertex_n_vertexes) {
  HC_REQUIRE(T, "hcp_celltype_tables2: null cell type");
  const CellTables &H = T->host;
  if (edge_bending_triangles) for (int i = 0; i < H.ne; i++) for (int d = 0; d < 2; d++) edge_bending_triangles[2 * i + d] = H.edge_bending_triangles[i][d];
  if (edge_bending_outer_points) for (int i = 0; i < H.ne; i++) for (int d = 0; d < 2; d++) edge_bending_outer_points[2 * i + d] = H.edge_bending_outer[i][d];
  if (inner_edges) for (int i = 0; i < H.nie; i++) for (int d = 0; d < 2; d++) inner_edges[2 * i + d] = H.inner_edges[i][d];
  if (inner_edge_length_eq) std::copy(H.inner_edge_length_eq.begin(), H.inner_edge_length_eq.end(), inner_edge_length_eq);
  if (vertex_n_vertexes) std::copy(H.vertex_n_vertexes.begin(), H.vertex_n_vertexes.end(), vertex_n_vertexes);
  return HC_OK;
}

int hcp_create(hc_cells **out, hc_lattice *L, const hc_params *P) {
  HC_REQUIRE(out && L && P, "hcp_create: null pointer");
  hc_cells *C = new hc_cells();
  C->L = L; C->P = *P;
  L->ibm = 1;
  HC_HIP(hipHostMalloc((void **)&C->h_ntag, 4 * sizeof(int), hipHostMallocMapped));
  HC_HIP(hipHostGetDevicePointer((void **)&C->h_ntag_dev, C->h_ntag, 0));
  for (int k = 0; k < 4; k++) C->h_ntag[k] = 0;
  HC_HIP(hipMalloc((void **)&C->d_ntag, 4 * sizeof(int)));
  HC_HIP(hipMemset(C->d_ntag, 0, 4 * sizeof(int)));
  HC_HIP(hipEventCreateWithFlags(&C->ntag_ev, hipEventDisableTiming));
  HC_HIP(hipHostMalloc((void **)&C->h_env_viol, sizeof(int), hipHostMallocMapped));
  HC_HIP(hipHostGetDevicePointer((void **)&C->d_env_viol, C->h_env_viol, 0));
  *C->h_env_viol = 0;
  *out = C;
  return HC_OK;
}

// The particle envelope of a slab run: <particleEnvelope> of the configuration, in lattice units.  The reference replicates
// single particles within that distance of a block face on the neighbour (core/hemoCell.cpp:139, core/hemoCellFields.cpp:39-43);
// here whole cells are replicated, so an envelope E acts as share = E - (diameter of the largest cell type): the distance a cell
// may still travel towards the face, between two velocity updates, before a complete copy must exist on the other side.  The
// slab has to keep the two faces' envelopes and a cell apart (nx >= 2 * diameter + 2 * share, the rule make_slab enforces): a
// request beyond that is CLAMPED to what the slab supports and the share in use is returned; a slab that cannot even hold
// E_SHARE_MIN is refused.
int hcp_set_envelope(hc_cells *C, double particle_envelope_lu, double *share_in_use) {
  HC_REQUIRE(C, "hcp_set_envelope: null pointer");
  HC_REQUIRE(particle_envelope_lu > 0.0 && particle_envelope_lu < 1e6, "hcp_set_envelope: the envelope must be a positive number of lattice units");
  HC_REQUIRE(C->ntypes > 0, "hcp_set_envelope: add the cell types first (the envelope is measured against the largest cell)");
  for (int t = 0; t < C->ntypes; t++) HC_REQUIRE(C->hids[t].empty(), "hcp_set_envelope: set the envelope before the first cell is placed (placement already uses it)");
  double dmax = 0.0;
  for (int t = 0; t < C->ntypes; t++) dmax = std::max(dmax, C->types[t]->host.diameter);
  double share = std::max(particle_envelope_lu - dmax, E_SHARE_MIN);
  const hc_lattice *L = C->L;
  if (L->n_slabs > 1) {
    const double room = 0.5 * ((double)L->nx - 2.0 * dmax);
    if (room < E_SHARE_MIN) { hc::set_error("hcp_set_envelope: a slab of " + std::to_string(L->nx) + " planes cannot hold two cell diameters (" + std::to_string(2.0 * dmax) + " lu) and a particle envelope"); return HC_ERR_ARG; }
    share = std::min(share, std::floor(room * 16.0) / 16.0);
  }
  C->e_share = share;
  if (share_in_use) *share_in_use = share;
  return HC_OK;
}

int hcp_envelope(const hc_cells *C, double *share_in_use, long *late_copies) {
  HC_REQUIRE(C, "hcp_envelope: null pointer");
  if (share_in_use) *share_in_use = C->e_share;
  if (late_copies) *late_copies = C->h_env_viol ? (long)*C->h_env_viol : 0;
  return HC_OK;
}

int hcp_destroy(hc_cells *C) {
  if (!C) return HC_OK;
  hipStreamSynchronize(hc::stream());
  free_device_arrays(C);
  if (C->h_ntag) hipHostFree(C->h_ntag);
  if (C->d_ntag) hipFree(C->d_ntag);
  if (C->ntag_ev) hipEventDestroy(C->ntag_ev);
  if (C->h_env_viol) hipHostFree(C->h_env_viol);
  if (C->d_bflag) hipFree(C->d_bflag);
  for (int k = 0; k < 19; k++) { if (C->h_iscratch[k]) hipHostFree(C->h_iscratch[k]); if (C->iscratch_ev[k]) hipEventDestroy(C->iscratch_ev[k]); }
  for (int k = 0; k < 2; k++) { if (C->det_keys[k]) hipFree(C->det_keys[k]); if (C->det_vals[k]) hipFree(C->det_vals[k]); }
  for (int k = 0; k < 3; k++) if (C->det_val[k]) hipFree(C->det_val[k]);
  if (C->det_tmp) hipFree(C->det_tmp);
  for (int t = 0; t < 8; t++) { if (C->h_ext[t]) hipHostFree(C->h_ext[t]); if (C->ext_done[t]) hipEventDestroy(C->ext_done[t]); }   // d_ext is the device view of h_ext
  if (C->d_stat) hipFree(C->d_stat);
  if (C->h_stat) hipHostFree(C->h_stat);
  if (C->d_info) hipFree(C->d_info);
  if (C->h_info) hipHostFree(C->h_info);
  if (C->h_vf) hipHostFree(C->h_vf);
  if (C->d_vf) hipFree(C->d_vf);
  if (C->vf_done) hipEventDestroy(C->vf_done);
  for (int k = 0; k < 19; k++) if (C->d_iscratch[k]) hipFree(C->d_iscratch[k]);
  delete C;
  return HC_OK;
}

int hcp_add_type(hc_cells *C, hc_celltype *T, int material_timescale, int *type_index) {
  HC_REQUIRE(C && T, "hcp_add_type: null pointer");
  HC_REQUIRE(C->ntypes < 8, "hcp_add_type: at most 8 cell types");
  HC_REQUIRE(material_timescale >= 1, "hcp_add_type: material timescale must be >= 1");
  C->types[C->ntypes] = T; C->timescale[C->ntypes] = material_timescale;
  if (type_index) *type_index = C->ntypes;
  C->ntypes++;
  return HC_OK;
}

int hcp_add_cell(hc_cells *C, int type, long cell_id, const double centre_lu[3], const double angles[3], double min_dist_um, int *placed) {
  HC_REQUIRE(C && centre_lu && angles, "hcp_add_cell: null pointer");
  HC_REQUIRE(type >= 0 && type < C->ntypes, "hcp_add_cell: unknown cell type");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  const CellTables &T = C->types[type]->host;
  const hc_lattice *L = C->L;
  const int nv = T.nv;
  // centre the mesh on its bounding box, rotate about that centre (X, Y, Z order), translate
  // (io/readPositionsBloodCells.cpp:113-123, :316-318, :349)
  auto bbox_centre = [&](const std::vector<Vec3> &v) {
    Vec3 lo = v[0], hi = v[0];
    for (auto &p : v) for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[d]); }
    return Vec3{(hi[0] + lo[0]) * 0.5, (hi[1] + lo[1]) * 0.5, (hi[2] + lo[2]) * 0.5};
  };
  std::vector<Vec3> m = T.vertices;
  const Vec3 c0 = bbox_centre(m);
  for (auto &p : m) for (int d = 0; d < 3; d++) p[d] -= c0[d];
  const Vec3 mc = bbox_centre(m);
  double R[3][3];
  rotation_matrix_xyz(angles[0], angles[1], angles[2], R);
  for (auto &p : m) {
    const Vec3 x{p[0] + -1.0 * mc[0], p[1] + -1.0 * mc[1], p[2] + -1.0 * mc[2]};
    for (int a = 0; a < 3; a++) { double s = 0; for (int b = 0; b < 3; b++) s += R[a][b] * x[b]; p[a] = s + mc[a]; }
  }
  // rejection test against the (host copy of the) mask, :139-164
  const std::vector<uint8_t> &mask = L->hmask;
  auto is_boundary = [&](long gx, long gy, long gz) -> bool {
    long lx = gx - L->x0, ly = gy, lz = gz;
    if (L->n_slabs == 1) { if (lx < 0 || lx >= L->nx) { if (L->periodic[0]) lx = ((lx % L->nx) + L->nx) % L->nx; else return false; } }
    else if (lx < -HALO || lx >= L->nx + HALO) return false;
    if (ly < 0 || ly >= L->ny) { if (L->periodic[1]) ly = ((ly % L->ny) + L->ny) % L->ny; else return false; }
    if (lz < 0 || lz >= L->nz) { if (L->periodic[2]) lz = ((lz % L->nz) + L->nz) % L->nz; else return false; }
    return mask[(size_t)(lx + HALO) * L->xs + (size_t)ly * L->nz + lz] != 0;
  };
  // slab of a multi-rank run: keep the cell (or its periodic image, shifted by the domain length as
  // core/hemoCellParticleDataTransfer.cpp:33-65 does) when one of its particles lives here or it reaches within the
  // envelope of a face; otherwise it is another rank's
  double cx = centre_lu[0];
  if (L->n_slabs > 1) {
    const double x0 = (double)L->x0, x1 = (double)(L->x0 + L->nx);
    const double shifts[3] = {0.0, -(double)L->nx_global, (double)L->nx_global};
    bool mine = false;
    for (int k = 0; k < (L->periodic[0] ? 3 : 1) && !mine; k++) {
      double lo = 1e300, hi = -1e300; bool own = false;
      for (int i = 0; i < nv; i++) {
        const double x = centre_lu[0] + shifts[k] + m[i][0];
        lo = std::min(lo, x); hi = std::max(hi, x);
        const double g = std::floor(x + 0.5);
        own = own || (g >= x0 && g < x1);
      }
      if (own || (hi >= x0 - C->e_share && lo < x1 + C->e_share)) { mine = true; cx = centre_lu[0] + shifts[k]; }
    }
    if (!mine) { if (placed) *placed = 0; return HC_OK; }
  }
  const int deny = (int)((min_dist_um * 1e-6) / C->P.dx);
  bool ok = true;
  for (int i = 0; i < nv && ok; i++) {
    const double v[3] = {cx + m[i][0], centre_lu[1] + m[i][1], centre_lu[2] + m[i][2]};
    const long n[3] = {(long)std::floor(v[0] + 0.5), (long)std::floor(v[1] + 0.5), (long)std::floor(v[2] + 0.5)};  // int(vertex+0.5), :135
    if (is_boundary(n[0], n[1], n[2])) { ok = false; break; }
    for (int a = -deny; a <= deny && ok; a++) for (int b = -deny; b <= deny && ok; b++) for (int c = -deny; c <= deny; c++)
      if (is_boundary(n[0] + a, n[1] + b, n[2] + c)) { ok = false; break; }
  }
  if (placed) *placed = ok ? 1 : 0;
  if (!ok) {
    if (L->n_slabs > 1) { C->slab_rejected.push_back(type); C->slab_rejected.push_back(cell_id); }   // the other holders must drop it too
    return HC_OK;
  }
  for (int i = 0; i < nv; i++) {
    C->hpos[type].push_back(cx + m[i][0]); C->hpos[type].push_back(centre_lu[1] + m[i][1]); C->hpos[type].push_back(centre_lu[2] + m[i][2]);
    for (int d = 0; d < 3; d++) { C->hvel[type].push_back(0.0); C->hfrc[type].push_back(0.0); }
  }
  host_append_state(C, type, cell_id);
  return HC_OK;
}

int hcp_add_cell_unchecked(hc_cells *C, int type, long cell_id, const double centre_lu[3], const double angles[3]) {
  HC_REQUIRE(C && centre_lu && angles, "hcp_add_cell_unchecked: null pointer");
  HC_REQUIRE(type >= 0 && type < C->ntypes, "hcp_add_cell_unchecked: unknown cell type");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  const CellTables &T = C->types[type]->host;
  (void)angles;
  for (int i = 0; i < T.nv; i++)
    for (int d = 0; d < 3; d++) { C->hpos[type].push_back(centre_lu[d] + T.vertices[i][d]); C->hvel[type].push_back(0.0); C->hfrc[type].push_back(0.0); }
  host_append_state(C, type, cell_id);
  return HC_OK;
}

int hcp_counts(hc_cells *C, long *n_vertices, long *n_cells, long *n_deleted) {
  HC_REQUIRE(C, "hcp_counts: null pointer");
  { int rc = settle(C); if (rc != HC_OK) return rc; }
  long nv = 0, nc = 0;
  for (int t = 0; t < C->ntypes; t++) { nc += (long)C->hids[t].size(); nv += (long)C->hids[t].size() * C->types[t]->host.nv; }
  if (n_vertices) *n_vertices = nv;
  if (n_cells) *n_cells = nc;
  if (n_deleted) *n_deleted = C->n_deleted;
  return HC_OK;
}

int hcp_type_range(hc_cells *C, int type, long *first_vertex, long *n_cells) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes, "hcp_type_range: bad arguments");
  { int rc = settle(C); if (rc != HC_OK) return rc; }
  long f = 0;
  for (int t = 0; t < type; t++) f += (long)C->hids[t].size() * C->types[t]->host.nv;
  if (first_vertex) *first_vertex = f;
  if (n_cells) *n_cells = (long)C->hids[type].size();
  return HC_OK;
}

int hcp_download(hc_cells *C, int what, double *out) {
  HC_REQUIRE(C && out && what >= 0 && what <= 2, "hcp_download: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  size_t o = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const std::vector<double> &src = what == 0 ? C->hpos[t] : what == 1 ? C->hvel[t] : C->hfrc[t];
    std::copy(src.begin(), src.end(), out + o);
    o += src.size();
  }
  return HC_OK;
}

int hcp_upload(hc_cells *C, int what, const double *in) {
  HC_REQUIRE(C && in && what >= 0 && what <= 2, "hcp_upload: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  size_t o = 0;
  for (int t = 0; t < C->ntypes; t++) {
    std::vector<double> &dst = what == 0 ? C->hpos[t] : what == 1 ? C->hvel[t] : C->hfrc[t];
    std::copy(in + o, in + o + dst.size(), dst.begin());
    o += dst.size();
  }
  C->host_dirty = true;
  return HC_OK;
}

// ---- the reference's particle record, HemoCellParticle::serializeValues_t (core/hemoCellParticle.h:45-63): 120 bytes,
// v @0, position @24, force @48, force_repulsion @72 (3 doubles each), plint cellId @96, uint16 vertexId @104,
// uint restime @108, uchar celltype @112.  A binding that keeps libhemocell's std::vector<HemoCellParticle> can hand
// its records over and get them back in this layout.
namespace {
struct SvRecord {
  double v[3], position[3], force[3], force_repulsion[3];
  long cellId; unsigned short vertexId; unsigned int restime; unsigned char celltype;
};
static_assert(sizeof(SvRecord) == 120, "serializeValues_t is 120 bytes");
static_assert(offsetof(SvRecord, cellId) == 96 && offsetof(SvRecord, vertexId) == 104 && offsetof(SvRecord, restime) == 108 &&
              offsetof(SvRecord, celltype) == 112, "serializeValues_t field offsets");
}  // namespace

int hcp_download_records(hc_cells *C, void *records, long n_records) {
  HC_REQUIRE(C && records && n_records >= 0, "hcp_download_records: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  long total = 0;
  for (int t = 0; t < C->ntypes; t++) { total += (long)C->hids[t].size() * C->types[t]->host.nv; for (unsigned char dd : C->hdead[t]) total -= dd ? 1 : 0; }
  HC_REQUIRE(n_records == total, "hcp_download_records: the buffer must hold exactly one record per particle (hcp_counts vertices minus hcp_deletion_counts removed-and-still-listed particles)");
  SvRecord *out = static_cast<SvRecord *>(records);
  long o = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const int nv = C->types[t]->host.nv;
    const bool has_rep = C->hrep[t].size() == C->hpos[t].size();
    for (size_t c = 0; c < C->hids[t].size(); c++)
      for (int i = 0; i < nv; i++) {
        if (C->hdead[t][c * (size_t)nv + (size_t)i]) continue;   // a particle removeParticles(1) took out (core/hemoCellParticleField.cpp:304-321)
        SvRecord &r = out[o++];
        std::memset(&r, 0, sizeof(r));
        const size_t k = 3 * (c * (size_t)nv + (size_t)i);
        for (int d = 0; d < 3; d++) { r.v[d] = C->hvel[t][k + d]; r.position[d] = C->hpos[t][k + d]; r.force[d] = C->hfrc[t][k + d]; r.force_repulsion[d] = has_rep ? C->hrep[t][k + d] : 0.0; }
        r.cellId = C->hids[t][c]; r.vertexId = (unsigned short)i; r.restime = 0; r.celltype = (unsigned char)t;
      }
  }
  return HC_OK;
}

// Replaces the whole vertex population by the given records (any order).  A cell that lacks records of some of its
// vertexIds arrives incomplete (no mechanics until hcp_delete_incomplete_cells removes it, as in the reference); cells
// keep the order of their first record within their type.
int hcp_upload_records(hc_cells *C, const void *records, long n_records) {
  HC_REQUIRE(C && (records || n_records == 0) && n_records >= 0, "hcp_upload_records: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  const SvRecord *in = static_cast<const SvRecord *>(records);
  std::vector<long> ids[8]; std::vector<double> pos[8], vel[8], frc[8], rep[8]; std::vector<int> seen[8];
  std::vector<std::pair<long, long>> index[8];   // (cellId, slot), sorted on demand
  for (long k = 0; k < n_records; k++) {
    const SvRecord &r = in[k];
    const int t = r.celltype;
    HC_REQUIRE(t < C->ntypes, "hcp_upload_records: record with an unknown celltype");
    const int nv = C->types[t]->host.nv;
    HC_REQUIRE(r.vertexId < nv, "hcp_upload_records: vertexId out of range for its cell type");
    long slot = -1;
    for (size_t c = ids[t].size(); c-- > 0;) if (ids[t][c] == r.cellId) { slot = (long)c; break; }   // records of a cell usually arrive together
    if (slot < 0) {
      slot = (long)ids[t].size(); ids[t].push_back(r.cellId);
      pos[t].resize(pos[t].size() + 3 * (size_t)nv, 0.0); vel[t].resize(pos[t].size(), 0.0); frc[t].resize(pos[t].size(), 0.0); rep[t].resize(pos[t].size(), 0.0);
      seen[t].resize(seen[t].size() + (size_t)nv, 0);
    }
    const size_t v = (size_t)slot * nv + r.vertexId;
    HC_REQUIRE(!seen[t][v], "hcp_upload_records: duplicate (cellId, vertexId)");
    seen[t][v] = 1;
    for (int d = 0; d < 3; d++) { pos[t][3 * v + d] = r.position[d]; vel[t][3 * v + d] = r.v[d]; frc[t][3 * v + d] = r.force[d]; rep[t][3 * v + d] = r.force_repulsion[d]; }
  }
  // A cell with a vertexId missing is what removeParticles(1) leaves behind (core/hemoCellParticleField.cpp:304-321) and what a
  // checkpoint written between two deleteIncompleteCells holds: it arrives as the incomplete cell it was (tag 2, the missing
  // particles flagged as removed), exactly the state hcp_download_records serialised; the reference's load path then runs
  // deleteIncompleteCells (core/hemoCellFields.cpp:272-274), which is the caller's next call.
  for (int t = 0; t < C->ntypes; t++) {
    const size_t nv = (size_t)C->types[t]->host.nv;
    C->hids[t].swap(ids[t]); C->hpos[t].swap(pos[t]); C->hvel[t].swap(vel[t]); C->hfrc[t].swap(frc[t]);
    C->htag[t].assign(C->hids[t].size(), 0); C->hdead[t].assign(C->hids[t].size() * nv, 0);
    for (size_t v = 0; v < seen[t].size(); v++) if (!seen[t][v]) { C->hdead[t][v] = 1; C->htag[t][v / nv] = 2; }
    if (C->rep_on()) C->hrep[t].swap(rep[t]); else C->hrep[t].clear();   // force_repulsion travels through the staging
  }
  C->host_dirty = true;
  return sync_to_device(C);
}

int hcp_download_cell_ids(hc_cells *C, long *ids) {
  HC_REQUIRE(C && ids, "hcp_download_cell_ids: null pointer");
  { int rc = settle(C); if (rc != HC_OK) return rc; }
  size_t o = 0;
  for (int t = 0; t < C->ntypes; t++) { std::copy(C->hids[t].begin(), C->hids[t].end(), ids + o); o += C->hids[t].size(); }
  return HC_OK;
}

int hcp_add_vertex_force(hc_cells *C, const long *vertex_index, int n, const double *f) {
  HC_REQUIRE(C && vertex_index && f && n >= 0, "hcp_add_vertex_force: bad arguments");
  if (n == 0) return HC_OK;
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const size_t bytes = (size_t)n * (sizeof(long) + 3 * sizeof(double));
  if (bytes > C->vf_cap) {
    HC_HIP(hipStreamSynchronize(hc::stream()));
    if (C->h_vf) HC_HIP(hipHostFree(C->h_vf));
    if (C->d_vf) HC_HIP(hipFree(C->d_vf));
    C->h_vf = C->d_vf = nullptr; C->vf_cap = 0;
    HC_HIP(hipHostMalloc((void **)&C->h_vf, 2 * bytes, hipHostMallocDefault));
    HC_HIP(hipMalloc((void **)&C->d_vf, 2 * bytes));
    C->vf_cap = 2 * bytes;
    if (!C->vf_done) HC_HIP(hipEventCreateWithFlags(&C->vf_done, hipEventDisableTiming));
  } else {
    HC_HIP(hipEventSynchronize(C->vf_done));   // the previous call's copy has left the pinned block
  }
  // vertex_index counts vertices in download order (types packed back to back); device regions have gaps
  long *h_idx = reinterpret_cast<long *>(C->h_vf);
  double *h_f = reinterpret_cast<double *>(C->h_vf + (size_t)n * sizeof(long));
  for (int i = 0; i < n; i++) {
    long v = vertex_index[i], packed0 = 0; bool found = false;
    for (int t = 0; t < C->ntypes && !found; t++) {
      const long nt = C->ncells[t] * C->types[t]->host.nv;
      if (v >= packed0 && v < packed0 + nt) { h_idx[i] = C->first[t] + (v - packed0); found = true; }
      packed0 += nt;
    }
    HC_REQUIRE(found, "hcp_add_vertex_force: vertex index out of range");
  }
  std::memcpy(h_f, f, (size_t)3 * n * sizeof(double));
  HC_HIP(hipMemcpyAsync(C->d_vf, C->h_vf, bytes, hipMemcpyHostToDevice, hc::stream()));
  HC_HIP(hipEventRecord(C->vf_done, hc::stream()));
  hipLaunchKernelGGL(add_vertex_force_kernel, dim3((n + 255) / 256), dim3(256), 0, hc::stream(), n, (const long *)C->d_vf,
                     (const double *)(C->d_vf + (size_t)n * sizeof(long)), C->frc[0], C->frc[1], C->frc[2]);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

}  // extern "C"

// ---- deletion bookkeeping.  Cells and particles are deleted ON THE DEVICE by the advance kernel (tag / dead flags that
// every kernel honours), so a step never waits for the host.  The host's view (cell counts, ids, compact storage) is
// brought up to date lazily: settle() when somebody asks, poll_deletions() from hc_iterate without ever blocking.
static int compact_gone(hc_cells *C) {
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
  for (int t = 0; t < C->ntypes; t++) {
    const size_t nc = C->hids[t].size(), nv = (size_t)C->types[t]->host.nv;
    const bool has_rep = C->hrep[t].size() == 3 * nc * nv;
    size_t w = 0;
    for (size_t c = 0; c < nc; c++) {
      if (C->htag[t][c] == 1) { C->n_deleted++; if (C->L && C->L->n_slabs > 1) C->slab_gone[t].push_back(C->hids[t][c]); continue; }
      if (w != c) {
        std::copy(C->hpos[t].begin() + 3 * c * nv, C->hpos[t].begin() + 3 * (c + 1) * nv, C->hpos[t].begin() + 3 * w * nv);
        std::copy(C->hvel[t].begin() + 3 * c * nv, C->hvel[t].begin() + 3 * (c + 1) * nv, C->hvel[t].begin() + 3 * w * nv);
        std::copy(C->hfrc[t].begin() + 3 * c * nv, C->hfrc[t].begin() + 3 * (c + 1) * nv, C->hfrc[t].begin() + 3 * w * nv);
        if (has_rep) std::copy(C->hrep[t].begin() + 3 * c * nv, C->hrep[t].begin() + 3 * (c + 1) * nv, C->hrep[t].begin() + 3 * w * nv);
        std::copy(C->hdead[t].begin() + c * nv, C->hdead[t].begin() + (c + 1) * nv, C->hdead[t].begin() + w * nv);
        C->htag[t][w] = C->htag[t][c]; C->hids[t][w] = C->hids[t][c];
      }
      w++;
    }
    C->hpos[t].resize(3 * w * nv); C->hvel[t].resize(3 * w * nv); C->hfrc[t].resize(3 * w * nv);
    if (has_rep) C->hrep[t].resize(3 * w * nv);
    C->hdead[t].resize(w * nv); C->htag[t].resize(w); C->hids[t].resize(w);
  }
  C->host_dirty = true;
  return sync_to_device(C);
}

// the counters have landed in the pinned block: account for them, compact when cells are gone
static int apply_counters(hc_cells *C) {
  const int gone = C->h_ntag[0];
  C->n_particles_deleted += C->h_ntag[2];
  if (C->h_ntag[0] == 0 && C->h_ntag[2] == 0) return HC_OK;
  // take off what was read, not more: advance kernels enqueued after that read may have counted further deletions already
  hipLaunchKernelGGL(sub_counters_kernel, dim3(1), dim3(1), 0, hc::stream(), C->d_ntag, C->h_ntag[0], C->h_ntag[1], C->h_ntag[2]);
  HC_HIP(hipGetLastError());
  C->h_ntag[0] = C->h_ntag[1] = C->h_ntag[2] = 0;
  return gone ? compact_gone(C) : HC_OK;
}

namespace hcc {
int settle(hc_cells *C) {
  if (C->host_dirty || !C->d_ntag) return HC_OK;      // the host staging is authoritative: nothing ran on the device since
  if (!C->maybe_tagged && !C->ntag_pending) return HC_OK;
  C->maybe_tagged = false; C->ntag_pending = false;
  hipLaunchKernelGGL(publish_counters_kernel, dim3(1), dim3(1), 0, hc::stream(), (const int *)C->d_ntag, C->h_ntag_dev);
  HC_HIP(hipGetLastError());
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return apply_counters(C);
}
}  // namespace hcc

// hc_iterate: pick up an asynchronous counter read if it has completed, start the next one; never waits
static int poll_deletions(hc_cells *C, bool start_next) {
  if (C->host_dirty || !C->d_ntag) return HC_OK;
  if (C->ntag_pending) {
    if (hipEventQuery(C->ntag_ev) != hipSuccess) return HC_OK;   // still in flight: look again next time
    C->ntag_pending = false;
    int rc = apply_counters(C); if (rc != HC_OK) return rc;
  }
  if (start_next && C->maybe_tagged) {
    hipLaunchKernelGGL(publish_counters_kernel, dim3(1), dim3(1), 0, hc::stream(), (const int *)C->d_ntag, C->h_ntag_dev);
    HC_HIP(hipGetLastError());
    HC_HIP(hipEventRecord(C->ntag_ev, hc::stream()));
    C->ntag_pending = true; C->maybe_tagged = false;
  }
  return HC_OK;
}

extern "C" {

int hcp_set_deletion_mode(hc_cells *C, int mode) {
  HC_REQUIRE(C && (mode == HC_DELETE_PARTICLE || mode == HC_DELETE_CELL), "hcp_set_deletion_mode: mode must be HC_DELETE_PARTICLE or HC_DELETE_CELL");
  C->del_mode = mode;
  return HC_OK;
}

int hcp_delete_incomplete_cells(hc_cells *C, long *n_cells_removed) {
  HC_REQUIRE(C, "hcp_delete_incomplete_cells: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long before = C->n_deleted;
  if (C->tag_cap > 0 && C->nverts > 0) {
    hipLaunchKernelGGL(delete_incomplete_kernel, dim3((unsigned)((C->tag_cap + 255) / 256)), dim3(256), 0, hc::stream(), C->tag_cap, C->d_tag, C->d_ntag);
    HC_HIP(hipGetLastError());
    C->maybe_tagged = true;
  }
  rc = settle(C); if (rc != HC_OK) return rc;
  if (n_cells_removed) *n_cells_removed = C->n_deleted - before;
  return HC_OK;
}

int hcp_deletion_counts(hc_cells *C, long *cells_removed, long *particles_removed, long *incomplete_cells, long *missing_particles) {
  HC_REQUIRE(C, "hcp_deletion_counts: null pointer");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  long inc = 0, miss = 0;
  for (int t = 0; t < C->ntypes; t++) { for (int g : C->htag[t]) inc += g == 2; for (unsigned char d : C->hdead[t]) miss += d ? 1 : 0; }
  if (cells_removed) *cells_removed = C->n_deleted;
  if (particles_removed) *particles_removed = C->n_particles_deleted;
  if (incomplete_cells) *incomplete_cells = inc;
  if (missing_particles) *missing_particles = miss;
  return HC_OK;
}

int hcp_download_alive(hc_cells *C, unsigned char *alive) {
  HC_REQUIRE(C && alive, "hcp_download_alive: null pointer");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_host(C); if (rc != HC_OK) return rc;
  size_t o = 0;
  for (int t = 0; t < C->ntypes; t++) for (unsigned char d : C->hdead[t]) alive[o++] = d ? 0 : 1;
  return HC_OK;
}

int hcp_advance(hc_cells *C, int check_deletions) {
  HC_REQUIRE(C, "hcp_advance: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  {
    hc::ProfScope prof(hc::PK_ADVANCE);
    const LatView v = make_view(C->L);
    for (int t = 0; t < C->ntypes; t++) {
      const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
      if (n == 0) continue;
      hipLaunchKernelGGL(advance_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, n, C->pos[0] + f, C->pos[1] + f,
                         C->pos[2] + f, (const double *)(C->vel[0] + f), (const double *)(C->vel[1] + f), (const double *)(C->vel[2] + f),
                         (const int *)(C->d_vert_cell + f), C->d_tag, C->d_vdead + f, C->d_ntag, C->del_mode == HC_DELETE_CELL ? 1 : 0);
      HC_HIP(hipGetLastError());
    }
    C->maybe_tagged = true;
  }
  if (check_deletions) return settle(C);   // phase-by-phase callers that want the compact storage at once (blocking)
  return HC_OK;
}

static int g_overlap = 1;   // hc_iterate: run advance + mechanics + the next spread beside the collide on steps without a particle update
static int g_spread_beside = 1;   // ... 2 in hc_set_overlap: advance and mechanics only, the next spread follows the collide on the main stream (A/B)
int hc_set_overlap(int on) { g_overlap = on != 0; g_spread_beside = on != 2; hcs::set_overlap(on); return HC_OK; }

int hc_iterate(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit, int deletion_check_every) {
  HC_REQUIRE(L && C && iter, "hc_iterate: null pointer");
  HC_REQUIRE(C->L == L, "hc_iterate: cells are bound to a different lattice");
  HC_REQUIRE(particle_timescale >= 1 && deletion_check_every >= 1, "hc_iterate: timescales must be >= 1");
  if (L->n_slabs > 1) return hcs::iterate_slab(L, C, iter, n, particle_timescale, force_limit);
  // Same phases in the same order as HemoCell::iterate.  Between two velocity updates advance(it), mechanics(it) and
  // spread(it+1) depend only on vertex data, not on collide(it): they run on the side stream beside it (the spread
  // adds into the force buffer of the next step, which the previous collide left clean).  Never across the end of
  // the call: the caller may edit vertex forces between calls (HemoCellStretch does).
  // Deletions (core/hemoCellParticleField.cpp:566-588) happen on the device inside the advance kernel at EVERY iteration,
  // whatever deletion_check_every says; that argument is only the cadence at which the host looks (without waiting) whether
  // gone cells can be compacted away.
  struct ForkGuard { ~ForkGuard() { if (hc::forked()) hc::join(); else hc::route(0); } } guard;   // error paths leave one timeline behind
  const bool may_overlap = g_overlap && !C->rep_enabled && !C->brep_enabled;
  bool spread_done = false;
  int rc;
  if ((rc = sync_to_device(C)) != HC_OK) return rc;
  if ((rc = poll_deletions(C, false)) != HC_OK) return rc;
  for (int s = 0; s < n; s++) {
    const long it = *iter;
    if (!spread_done) {
      if (C->rep_enabled && it % C->rep_timescale == 0) { if ((rc = hcp_repulsion(C)) != HC_OK) return rc; }   // core/hemoCell.cpp:307-309
      if (C->brep_enabled && it % C->brep_timescale == 0) { if ((rc = hcp_boundary_repulsion(C)) != HC_OK) return rc; }   // :310-312
      if ((rc = hcp_spread(C, force_limit)) != HC_OK) return rc;                // :313
    }
    spread_done = false;
    const bool particle_step = it % particle_timescale == 0;
    const bool overlap = may_overlap && !particle_step && s + 1 < n;
    if (overlap && (rc = hc::fork()) != HC_OK) return rc;
    if ((rc = hcl_collide_stream_part(L, 0)) != HC_OK) return rc;               // :317
    hcl_step_end(L);
    if (particle_step) { if ((rc = hcp_interpolate(C)) != HC_OK) return rc; }   // :327-332
    if (overlap) hc::route(1);
    if ((rc = hcp_advance(C, 0)) != HC_OK) return rc;                           // :342
    if ((rc = hcp_mechanics(C, it, 0)) != HC_OK) return rc;                     // :345
    if (overlap) {
      if (g_spread_beside) { if ((rc = hcp_spread(C, force_limit)) != HC_OK) return rc; spread_done = true; }   // :313 of iteration it + 1
      if ((rc = hc::join()) != HC_OK) return rc;
    }
    *iter = it + 1;                                                             // :374 (force zeroing is fused into the collide kernel)
    if (!overlap && (it + 1) % deletion_check_every == 0 && s + 1 < n) { if ((rc = poll_deletions(C, true)) != HC_OK) return rc; }   // on the main timeline only
  }
  return poll_deletions(C, true);
}

}  // extern "C"
