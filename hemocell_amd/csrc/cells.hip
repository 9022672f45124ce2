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
                                                      const double *vy, const double *vz, const int *vert_cell, int *tag, int *ntag) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double x = px[i] + vx[i], y = py[i] + vy[i], z = pz[i] + vz[i];
  px[i] = x; py[i] = y; pz[i] = z;
  // nearest node is a boundary -> tag (core/hemoCellParticleField.cpp:571-583)
  long gx = nearest_node(x) - v.x0, gy = nearest_node(y), gz = nearest_node(z);
  bool inside = true;
  if (v.wrap_x) gx = pmod(gx, v.nx); else inside = inside && (gx >= (v.halo_x ? -HALO : 0) && gx < (v.halo_x ? v.nx + HALO : v.nx));
  if (gy < 0 || gy >= v.ny) { if (v.per_y) gy = pmod(gy, v.ny); else inside = false; }
  if (gz < 0 || gz >= v.nz) { if (v.per_z) gz = pmod(gz, v.nz); else inside = false; }
  if (inside && v.mask[(gx + HALO) * (long)v.plane + gy * v.nz + gz] != 0) {
    if (atomicExch(&tag[vert_cell[i]], 1) == 0) atomicAdd(ntag, 1);
  }
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
  C->d_tag = nullptr; C->cap = 0; C->tag_cap = 0;
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
  HC_HIP(hipStreamSynchronize(hc::stream()));
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
  }
  return HC_OK;
}

VertArrays vert_arrays(hc_cells *C, int t) {
  VertArrays a;
  for (int d = 0; d < 3; d++) {
    a.p[d] = C->pos[d] + C->first[t]; a.v[d] = C->vel[d] + C->first[t]; a.f[d] = C->frc[d] + C->first[t];
    a.r[d] = C->rep[d] ? C->rep[d] + C->first[t] : nullptr;
  }
  return a;
}
// stage a small host int array on the device in a persistent scratch slot; the copy and every later use are
// ordered on the library stream, so no host synchronisation is needed
int stage_ints(hc_cells *C, int which, int **d, const int *h, int n) {
  if ((size_t)n > C->iscratch_cap[which]) {
    HC_HIP(hipStreamSynchronize(hc::stream()));
    if (C->d_iscratch[which]) HC_HIP(hipFree(C->d_iscratch[which]));
    C->iscratch_cap[which] = (size_t)n * 2 + 256;
    HC_HIP(hipMalloc((void **)&C->d_iscratch[which], C->iscratch_cap[which] * sizeof(int)));
  }
  if (n > 0) HC_HIP(hipMemcpyAsync(C->d_iscratch[which], h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, hc::stream()));
  *d = C->d_iscratch[which];
  return HC_OK;
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

int hcp_create(hc_cells **out, hc_lattice *L, const hc_params *P) {
  HC_REQUIRE(out && L && P, "hcp_create: null pointer");
  hc_cells *C = new hc_cells();
  C->L = L; C->P = *P;
  L->ibm = 1;
  HC_HIP(hipHostMalloc((void **)&C->h_ntag, sizeof(int), hipHostMallocDefault));
  *C->h_ntag = 0;
  HC_HIP(hipMalloc((void **)&C->d_ntag, sizeof(int)));
  HC_HIP(hipMemset(C->d_ntag, 0, sizeof(int)));
  *out = C;
  return HC_OK;
}

int hcp_destroy(hc_cells *C) {
  if (!C) return HC_OK;
  hipStreamSynchronize(hc::stream());
  free_device_arrays(C);
  if (C->h_ntag) hipHostFree(C->h_ntag);
  if (C->d_ntag) hipFree(C->d_ntag);
  if (C->d_bflag) hipFree(C->d_bflag);
  for (int t = 0; t < 8; t++) { if (C->d_ext[t]) hipFree(C->d_ext[t]); if (C->h_ext[t]) hipHostFree(C->h_ext[t]); if (C->ext_done[t]) hipEventDestroy(C->ext_done[t]); }
  if (C->h_vf) hipHostFree(C->h_vf);
  if (C->d_vf) hipFree(C->d_vf);
  if (C->vf_done) hipEventDestroy(C->vf_done);
  for (int k = 0; k < 2; k++) if (C->d_iscratch[k]) hipFree(C->d_iscratch[k]);
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
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
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
  const int deny = (int)((min_dist_um * 1e-6) / C->P.dx);
  bool ok = true;
  for (int i = 0; i < nv && ok; i++) {
    const double v[3] = {centre_lu[0] + m[i][0], centre_lu[1] + m[i][1], centre_lu[2] + m[i][2]};
    const long n[3] = {(long)std::floor(v[0] + 0.5), (long)std::floor(v[1] + 0.5), (long)std::floor(v[2] + 0.5)};  // int(vertex+0.5), :135
    if (is_boundary(n[0], n[1], n[2])) { ok = false; break; }
    for (int a = -deny; a <= deny && ok; a++) for (int b = -deny; b <= deny && ok; b++) for (int c = -deny; c <= deny; c++)
      if (is_boundary(n[0] + a, n[1] + b, n[2] + c)) { ok = false; break; }
  }
  if (placed) *placed = ok ? 1 : 0;
  if (!ok) return HC_OK;
  for (int i = 0; i < nv; i++) {
    for (int d = 0; d < 3; d++) { C->hpos[type].push_back(centre_lu[d] + m[i][d]); C->hvel[type].push_back(0.0); C->hfrc[type].push_back(0.0); }
  }
  C->hids[type].push_back(cell_id);
  C->host_dirty = true;
  return HC_OK;
}

int hcp_add_cell_unchecked(hc_cells *C, int type, long cell_id, const double centre_lu[3], const double angles[3]) {
  HC_REQUIRE(C && centre_lu && angles, "hcp_add_cell_unchecked: null pointer");
  HC_REQUIRE(type >= 0 && type < C->ntypes, "hcp_add_cell_unchecked: unknown cell type");
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
  const CellTables &T = C->types[type]->host;
  (void)angles;
  for (int i = 0; i < T.nv; i++)
    for (int d = 0; d < 3; d++) { C->hpos[type].push_back(centre_lu[d] + T.vertices[i][d]); C->hvel[type].push_back(0.0); C->hfrc[type].push_back(0.0); }
  C->hids[type].push_back(cell_id);
  C->host_dirty = true;
  return HC_OK;
}

int hcp_counts(const hc_cells *C, long *n_vertices, long *n_cells, long *n_deleted) {
  HC_REQUIRE(C, "hcp_counts: null pointer");
  long nv = 0, nc = 0;
  for (int t = 0; t < C->ntypes; t++) { nc += (long)C->hids[t].size(); nv += (long)C->hids[t].size() * C->types[t]->host.nv; }
  if (n_vertices) *n_vertices = nv;
  if (n_cells) *n_cells = nc;
  if (n_deleted) *n_deleted = C->n_deleted;
  return HC_OK;
}

int hcp_type_range(const hc_cells *C, int type, long *first_vertex, long *n_cells) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes, "hcp_type_range: bad arguments");
  long f = 0;
  for (int t = 0; t < type; t++) f += (long)C->hids[t].size() * C->types[t]->host.nv;
  if (first_vertex) *first_vertex = f;
  if (n_cells) *n_cells = (long)C->hids[type].size();
  return HC_OK;
}

int hcp_download(hc_cells *C, int what, double *out) {
  HC_REQUIRE(C && out && what >= 0 && what <= 2, "hcp_download: bad arguments");
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
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
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
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
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
  long total = 0;
  for (int t = 0; t < C->ntypes; t++) total += (long)C->hids[t].size() * C->types[t]->host.nv;
  HC_REQUIRE(n_records == total, "hcp_download_records: the buffer must hold exactly one record per vertex (hcp_counts)");
  std::vector<double> rep((size_t)(3 * total), 0.0);
  if (C->rep[0] && total) { rc = hcp_download_repulsion(C, rep.data()); if (rc != HC_OK) return rc; }
  SvRecord *out = static_cast<SvRecord *>(records);
  long o = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const int nv = C->types[t]->host.nv;
    for (size_t c = 0; c < C->hids[t].size(); c++)
      for (int i = 0; i < nv; i++, o++) {
        SvRecord &r = out[o];
        std::memset(&r, 0, sizeof(r));
        const size_t k = 3 * (c * (size_t)nv + (size_t)i);
        for (int d = 0; d < 3; d++) { r.v[d] = C->hvel[t][k + d]; r.position[d] = C->hpos[t][k + d]; r.force[d] = C->hfrc[t][k + d]; r.force_repulsion[d] = rep[(size_t)(3 * o + d)]; }
        r.cellId = C->hids[t][c]; r.vertexId = (unsigned short)i; r.restime = 0; r.celltype = (unsigned char)t;
      }
  }
  return HC_OK;
}

// Replaces the whole vertex population by the given records (any order).  Every cell must be complete -- one record
// per vertexId of its type -- as the reference requires before mechanics (deleteIncompleteCells); cells keep the order
// of their first record within their type.
int hcp_upload_records(hc_cells *C, const void *records, long n_records) {
  HC_REQUIRE(C && (records || n_records == 0) && n_records >= 0, "hcp_upload_records: bad arguments");
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
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
  for (int t = 0; t < C->ntypes; t++) for (int s : seen[t]) HC_REQUIRE(s, "hcp_upload_records: incomplete cell (a vertexId is missing)");
  for (int t = 0; t < C->ntypes; t++) { C->hids[t].swap(ids[t]); C->hpos[t].swap(pos[t]); C->hvel[t].swap(vel[t]); C->hfrc[t].swap(frc[t]); }
  C->host_dirty = true;
  rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->rep[0]) {   // force_repulsion lives on the device only
    std::vector<double> tmp;
    for (int t = 0; t < C->ntypes; t++) {
      const size_t n = rep[t].size() / 3;
      if (!n) continue;
      tmp.resize(n);
      for (int d = 0; d < 3; d++) {
        for (size_t i = 0; i < n; i++) tmp[i] = rep[t][3 * i + d];
        HC_HIP(hipMemcpy(C->rep[d] + C->first[t], tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
      }
    }
  }
  return HC_OK;
}

int hcp_download_cell_ids(hc_cells *C, long *ids) {
  HC_REQUIRE(C && ids, "hcp_download_cell_ids: null pointer");
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

// remove tagged cells (host round trip; rare)
static int purge_tagged(hc_cells *C) {
  HC_HIP(hipMemcpyAsync(C->h_ntag, C->d_ntag, sizeof(int), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  if (*C->h_ntag == 0) return HC_OK;
  std::vector<int> tags((size_t)C->tag_cap);
  HC_HIP(hipMemcpy(tags.data(), C->d_tag, (size_t)C->tag_cap * sizeof(int), hipMemcpyDeviceToHost));
  int rc = sync_to_host(C); if (rc != HC_OK) return rc;
  for (int t = 0; t < C->ntypes; t++) {
    const long nc = C->ncells[t]; const int nv = C->types[t]->host.nv;
    std::vector<double> np, nvl, nf; std::vector<long> nid;
    for (long c = 0; c < nc; c++) {
      if (tags[(size_t)(C->cell0[t] + c)]) { C->n_deleted++; continue; }
      np.insert(np.end(), C->hpos[t].begin() + 3 * c * nv, C->hpos[t].begin() + 3 * (c + 1) * nv);
      nvl.insert(nvl.end(), C->hvel[t].begin() + 3 * c * nv, C->hvel[t].begin() + 3 * (c + 1) * nv);
      nf.insert(nf.end(), C->hfrc[t].begin() + 3 * c * nv, C->hfrc[t].begin() + 3 * (c + 1) * nv);
      nid.push_back(C->hids[t][(size_t)c]);
    }
    C->hpos[t].swap(np); C->hvel[t].swap(nvl); C->hfrc[t].swap(nf); C->hids[t].swap(nid);
  }
  *C->h_ntag = 0;
  HC_HIP(hipMemset(C->d_ntag, 0, sizeof(int)));
  C->host_dirty = true;
  return sync_to_device(C);
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
                         (const int *)(C->d_vert_cell + f), C->d_tag, C->d_ntag);
      HC_HIP(hipGetLastError());
    }
  }
  if (check_deletions) return purge_tagged(C);
  return HC_OK;
}

static int g_overlap = 1;   // hc_iterate: run advance + mechanics + the next spread beside the collide on steps without a particle update
int hc_set_overlap(int on) { g_overlap = on != 0; return HC_OK; }

int hc_iterate(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit, int deletion_check_every) {
  HC_REQUIRE(L && C && iter, "hc_iterate: null pointer");
  HC_REQUIRE(C->L == L, "hc_iterate: cells are bound to a different lattice");
  HC_REQUIRE(L->n_slabs == 1, "hc_iterate: single-slab stepping only; multi-slab runs are driven phase by phase with halo exchange");
  HC_REQUIRE(particle_timescale >= 1 && deletion_check_every >= 1, "hc_iterate: timescales must be >= 1");
  // Same phases in the same order as HemoCell::iterate.  Between two velocity updates advance(it), mechanics(it) and
  // spread(it+1) depend only on vertex data, not on collide(it): they run on the side stream beside it (the spread
  // adds into the force buffer of the next step, which the previous collide left clean).  Never across the end of
  // the call: the caller may edit vertex forces between calls (HemoCellStretch does).
  struct ForkGuard { ~ForkGuard() { if (hc::forked()) hc::join(); else hc::route(0); } } guard;   // error paths leave one timeline behind
  const bool may_overlap = g_overlap && !C->rep_enabled && !C->brep_enabled;
  bool spread_done = false;
  int rc;
  for (int s = 0; s < n; s++) {
    const long it = *iter;
    if (!spread_done) {
      if (C->rep_enabled && it % C->rep_timescale == 0) { if ((rc = hcp_repulsion(C)) != HC_OK) return rc; }   // core/hemoCell.cpp:307-309
      if (C->brep_enabled && it % C->brep_timescale == 0) { if ((rc = hcp_boundary_repulsion(C)) != HC_OK) return rc; }   // :310-312
      if ((rc = hcp_spread(C, force_limit)) != HC_OK) return rc;                // :313
    }
    spread_done = false;
    const bool particle_step = it % particle_timescale == 0, check = (it % deletion_check_every) == 0;
    const bool overlap = may_overlap && !particle_step && !check && s + 1 < n;
    if (overlap && (rc = hc::fork()) != HC_OK) return rc;
    if ((rc = hcl_collide_stream_part(L, 0)) != HC_OK) return rc;               // :317
    hcl_step_end(L);
    if (particle_step) { if ((rc = hcp_interpolate(C)) != HC_OK) return rc; }   // :327-332
    if (overlap) hc::route(1);
    if ((rc = hcp_advance(C, check)) != HC_OK) return rc;                       // :342
    if ((rc = hcp_mechanics(C, it, 0)) != HC_OK) return rc;                     // :345
    if (overlap) {
      if ((rc = hcp_spread(C, force_limit)) != HC_OK) return rc;                // :313 of iteration it + 1
      if ((rc = hc::join()) != HC_OK) return rc;
      spread_done = true;
    }
    *iter = it + 1;                                                             // :374 (force zeroing is fused into the collide kernel)
  }
  return HC_OK;
}

}  // extern "C"
