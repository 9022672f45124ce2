// Vertex-vertex and boundary-particle repulsion on the GPU.
//
// Replaces (file:line in the HemoCell tree):
//   core/hemoCellParticleField.cpp:677-743        applyRepulsionForce (+ update_pg :137-168)
//   core/hemoCellParticleField.cpp:865-918        populateBoundaryParticles, applyBoundaryRepulsionForce
#include "cells.h"
#include <hipcub/hipcub.hpp>

namespace {

// ---------------------------------------------------------------------------- vertex-vertex repulsion
// applyRepulsionForce (core/hemoCellParticleField.cpp:677-743): vertices are binned by their nearest lattice
// node (update_pg, :137-168); two vertices of DIFFERENT cells in the same or in adjacent bins that are closer
// than r_cutoff repel each other with r_const * (r_cutoff / d) along their separation.  The reference visits a
// same-bin pair twice (its inner loop runs over ordered pairs there), so those pairs count double.  Gather form:
// every vertex sums over the 27 bins around its own; the bins come from a radix sort of (bin, vertex).
__global__ void rep_keys_kernel(LatView v, long n, long first, long packed0, const double *px, const double *py, const double *pz, unsigned int *keys, int *vals,
                                const int *vert_cell, const int *tag, const unsigned char *dead) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long lx = nearest_node(px[i]) - v.x0, ly = nearest_node(py[i]), lz = nearest_node(pz[i]);
  bool ok = !dead[i] && tag[vert_cell[i]] != 1;   // removed particles and gone cells are in no bin
  if (v.wrap_x) lx = pmod(lx, v.nx); else ok = ok && (lx >= -HALO && lx < v.nx + HALO);
  if (ly < 0 || ly >= v.ny) { if (v.per_y) ly = pmod(ly, v.ny); else ok = false; }
  if (lz < 0 || lz >= v.nz) { if (v.per_z) lz = pmod(lz, v.nz); else ok = false; }
  keys[packed0 + i] = ok ? (unsigned int)((lx + HALO) * (long)v.plane + ly * v.nz + lz) : 0xffffffffu;
  vals[packed0 + i] = (int)(first + i);
}

__device__ __forceinline__ long lower_bound_u32(const unsigned int *a, long n, unsigned int key) {
  long lo = 0, hi = n;
  while (lo < hi) { const long mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(256) void rep_force_kernel(LatView v, long cap, long nsorted, const unsigned int *keys, const int *vals, const int *vert_cell,
                                                        const double *px, const double *py, const double *pz, double *rx, double *ry, double *rz,
                                                        double r_const, double r_cutoff) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;   // position in the sorted order
  if (s >= nsorted) return;
  const unsigned int key = keys[s];
  const int i = vals[s];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (key != 0xffffffffu) {
    const int lxp = (int)(key / v.plane), rem = (int)(key - (long)lxp * v.plane), ly = rem / v.nz, lz = rem - ly * v.nz;   // lxp = padded x
    const double x = px[i], y = py[i], z = pz[i];
    const int ci = vert_cell[i];
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          long bx = lxp + dx, by = ly + dy, bz = lz + dz;
          if (v.wrap_x) { if (bx < HALO) bx += v.nx; else if (bx >= v.nx + HALO) bx -= v.nx; }
          else if (bx < 0 || bx >= v.nx + 2 * HALO) continue;
          if (by < 0) { if (!v.per_y) continue; by += v.ny; } else if (by >= v.ny) { if (!v.per_y) continue; by -= v.ny; }
          if (bz < 0) { if (!v.per_z) continue; bz += v.nz; } else if (bz >= v.nz) { if (!v.per_z) continue; bz -= v.nz; }
          const unsigned int nkey = (unsigned int)(bx * (long)v.plane + by * v.nz + bz);
          const double fac = (dx == 0 && dy == 0 && dz == 0) ? 2.0 : 1.0;
          for (long q = lower_bound_u32(keys, nsorted, nkey); q < nsorted && keys[q] == nkey; q++) {
            const int j = vals[q];
            if (j == i || vert_cell[j] == ci) continue;
            // positions are not re-wrapped when a cell crosses a periodic face: minimum image of the separation
            double d0 = x - px[j], d1 = y - py[j], d2 = z - pz[j];
            if (v.wrap_x) d0 = d0 - (double)v.nx * rint(d0 / (double)v.nx);
            if (v.per_y) d1 = d1 - (double)v.ny * rint(d1 / (double)v.ny);
            if (v.per_z) d2 = d2 - (double)v.nz * rint(d2 / (double)v.nz);
            const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (dist < r_cutoff) {
              const double m = fac * (r_const * (1 / (dist / r_cutoff)));
              a0 += m * (d0 / dist); a1 += m * (d1 / dist); a2 += m * (d2 / dist);
            }
          }
        }
  }
  rx[i] = a0; ry[i] = a1; rz[i] = a2;
}

// Boundary particles (core/hemoCellParticleField.cpp:865-918): every flagged wall node pushes the vertices binned in
// the 27 bins around it with k * (cutoff / d) along their separation.  Gather form: each vertex visits the 27 nodes
// around its own bin in ascending (x, y, z) order -- the order in which the reference's x-major list of boundary
// particles reaches it -- and ADDS to force_repulsion (only applyRepulsionForce ever zeroes it, :703).
__global__ __launch_bounds__(256) void boundary_rep_kernel(LatView v, long n, const uint8_t *bflag, const double *px, const double *py, const double *pz,
                                                           double *rx, double *ry, double *rz, double br_const, double br_cutoff,
                                                           const int *vert_cell, const int *tag, const unsigned char *dead) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (dead[i] || tag[vert_cell[i]] == 1) return;
  const double x = px[i], y = py[i], z = pz[i];
  const long cx = nearest_node(x), cy = nearest_node(y), cz = nearest_node(z);
  if ((cy < 0 || cy >= v.ny) && !v.per_y) return;   // not in the particle grid (update_pg, :158-161)
  if ((cz < 0 || cz >= v.nz) && !v.per_z) return;
  if (!v.wrap_x) { const long lx = cx - v.x0; if (v.halo_x ? (lx < -HALO || lx >= v.nx + HALO) : (lx < 0 || lx >= v.nx)) return; }
  double a0 = rx[i], a1 = ry[i], a2 = rz[i];
  for (int dx = -1; dx <= 1; dx++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dz = -1; dz <= 1; dz++) {
        const long gx = cx + dx, gy = cy + dy, gz = cz + dz;
        long lx = gx - v.x0, ly = gy, lz = gz;
        if (v.wrap_x) lx = pmod(lx, v.nx);
        else if (v.halo_x) { if (lx < -HALO || lx >= v.nx + HALO) continue; }
        else if (lx < 0 || lx >= v.nx) continue;
        if (ly < 0 || ly >= v.ny) { if (v.per_y) ly = pmod(ly, v.ny); else continue; }
        if (lz < 0 || lz >= v.nz) { if (v.per_z) lz = pmod(lz, v.nz); else continue; }
        if (!bflag[(lx + HALO) * (long)v.plane + ly * v.nz + lz]) continue;
        const double d0 = x - (double)gx, d1 = y - (double)gy, d2 = z - (double)gz;
        const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if (dist < br_cutoff) {
          const double m = br_const * (1 / (dist / br_cutoff));
          a0 = a0 + m * (d0 / dist); a1 = a1 + m * (d1 / dist); a2 = a2 + m * (d2 / dist);
        }
      }
  rx[i] = a0; ry[i] = a1; rz[i] = a2;
}

}  // namespace

extern "C" {

// hemocell.setRepulsion(k, cutoff_um) + setRepulsionTimeScaleSeperation (core/hemoCell.cpp:394-397,420-426)
int hcp_set_repulsion(hc_cells *C, double r_const, double r_cutoff_lu, int timescale) {
  HC_REQUIRE(C && r_cutoff_lu > 0 && timescale >= 1, "hcp_set_repulsion: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  C->rep_const = r_const; C->rep_cutoff = r_cutoff_lu; C->rep_timescale = timescale;
  if (!C->rep_enabled) {
    C->rep_enabled = 1;
    if (C->cap > 0) for (int d = 0; d < 3; d++) { HC_HIP(hipMalloc((void **)&C->rep[d], C->cap * sizeof(double))); HC_HIP(hipMemset(C->rep[d], 0, C->cap * sizeof(double))); }
  }
  return HC_OK;
}

// cellfields->applyRepulsionForce() (core/hemoCell.cpp:307-309 -> core/hemoCellParticleField.cpp:696-743)
int hcp_repulsion(hc_cells *C) {
  HC_REQUIRE(C, "hcp_repulsion: null pointer");
  HC_REQUIRE(C->rep_enabled, "hcp_repulsion: call hcp_set_repulsion first");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  if (!C->rep[0]) for (int d = 0; d < 3; d++) { HC_HIP(hipMalloc((void **)&C->rep[d], C->cap * sizeof(double))); HC_HIP(hipMemset(C->rep[d], 0, C->cap * sizeof(double))); }
  const long n = C->nverts;
  if (n > C->sort_cap) {
    HC_HIP(hipStreamSynchronize(hc::stream()));
    for (int k = 0; k < 2; k++) { if (C->d_keys[k]) HC_HIP(hipFree(C->d_keys[k])); if (C->d_vals[k]) HC_HIP(hipFree(C->d_vals[k])); }
    if (C->d_sort_tmp) HC_HIP(hipFree(C->d_sort_tmp));
    C->sort_cap = n + n / 4 + 1024;
    for (int k = 0; k < 2; k++) { HC_HIP(hipMalloc((void **)&C->d_keys[k], C->sort_cap * sizeof(unsigned int))); HC_HIP(hipMalloc((void **)&C->d_vals[k], C->sort_cap * sizeof(int))); }
    C->sort_tmp_bytes = 0; C->d_sort_tmp = nullptr;
    HC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, C->sort_tmp_bytes, C->d_keys[0], C->d_keys[1], C->d_vals[0], C->d_vals[1], (int)C->sort_cap, 0, 32, hc::stream()));
    HC_HIP(hipMalloc(&C->d_sort_tmp, C->sort_tmp_bytes));
  }
  const LatView v = make_view(C->L);
  long packed0 = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const long nt = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (nt == 0) continue;
    hipLaunchKernelGGL(rep_keys_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, hc::stream(), v, nt, f, packed0,
                       (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f), C->d_keys[0], C->d_vals[0],
                       (const int *)(C->d_vert_cell + f), (const int *)C->d_tag, (const unsigned char *)(C->d_vdead + f));
    HC_HIP(hipGetLastError());
    packed0 += nt;
  }
  size_t tmp = C->sort_tmp_bytes;
  HC_HIP(hipcub::DeviceRadixSort::SortPairs(C->d_sort_tmp, tmp, C->d_keys[0], C->d_keys[1], C->d_vals[0], C->d_vals[1], (int)n, 0, 32, hc::stream()));
  hipLaunchKernelGGL(rep_force_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, C->cap, n, (const unsigned int *)C->d_keys[1],
                     (const int *)C->d_vals[1], (const int *)C->d_vert_cell, (const double *)C->pos[0], (const double *)C->pos[1], (const double *)C->pos[2],
                     C->rep[0], C->rep[1], C->rep[2], C->rep_const, C->rep_cutoff);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

// hemocell.enableBoundaryParticles(k, cutoff_um, timestep) (core/hemoCell.cpp:428-436): populateBoundaryParticles
// (core/hemoCellParticleField.cpp:865-890) becomes a flag map -- wall nodes with a non-wall node among their 26
// neighbours -- built from the host mask (halo planes included; neighbours beyond them count as unknown = wall)
int hcp_set_boundary_repulsion(hc_cells *C, double br_const, double br_cutoff_lu, int timescale) {
  HC_REQUIRE(C && br_cutoff_lu > 0 && timescale >= 1, "hcp_set_boundary_repulsion: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const hc_lattice *L = C->L;
  const int NX = L->nx + 2 * HALO, ny = L->ny, nz = L->nz;
  const bool wrap_x = L->n_slabs == 1 && L->periodic[0];
  std::vector<uint8_t> flag(L->npad, 0);
  auto solid = [&](int xp, int y, int z, bool &known) -> bool {   // xp = padded x
    known = true;
    if (wrap_x) { int lx = xp - HALO; lx = ((lx % L->nx) + L->nx) % L->nx; xp = lx + HALO; }
    else if (xp < 0 || xp >= NX) { known = false; return true; }
    if (L->n_slabs == 1 && !L->periodic[0] && (xp < HALO || xp >= HALO + L->nx)) { known = false; return true; }   // outside the domain
    if (y < 0 || y >= ny) { if (L->periodic[1]) y = (y + ny) % ny; else { known = false; return true; } }
    if (z < 0 || z >= nz) { if (L->periodic[2]) z = (z + nz) % nz; else { known = false; return true; } }
    return L->hmask[(size_t)xp * L->xs + (size_t)y * nz + z] != 0;
  };
  for (int xp = 0; xp < NX; xp++)
    for (int y = 0; y < ny; y++)
      for (int z = 0; z < nz; z++) {
        bool known;
        if (L->n_slabs == 1 && (xp < HALO || xp >= HALO + L->nx)) continue;   // single slab: halo planes are never addressed
        if (!solid(xp, y, z, known)) continue;
        bool near = false;
        for (int a = -1; a <= 1 && !near; a++) for (int b = -1; b <= 1 && !near; b++) for (int c = -1; c <= 1; c++) {
          bool k2; const bool s2 = solid(xp + a, y + b, z + c, k2);
          if (k2 && !s2) { near = true; break; }
        }
        if (near) flag[(size_t)xp * L->xs + (size_t)y * nz + z] = 1;
      }
  if (!C->d_bflag) HC_HIP(hipMalloc((void **)&C->d_bflag, L->npad));
  HC_HIP(hipMemcpy(C->d_bflag, flag.data(), L->npad, hipMemcpyHostToDevice));
  C->brep_const = br_const; C->brep_cutoff = br_cutoff_lu; C->brep_timescale = timescale;
  if (!C->brep_enabled) {
    C->brep_enabled = 1;
    if (C->cap > 0 && !C->rep[0]) for (int d = 0; d < 3; d++) { HC_HIP(hipMalloc((void **)&C->rep[d], C->cap * sizeof(double))); HC_HIP(hipMemset(C->rep[d], 0, C->cap * sizeof(double))); }
  }
  return HC_OK;
}

// cellfields->applyBoundaryRepulsionForce() (core/hemoCell.cpp:310-312 -> core/hemoCellParticleField.cpp:891-918)
int hcp_boundary_repulsion(hc_cells *C) {
  HC_REQUIRE(C, "hcp_boundary_repulsion: null pointer");
  HC_REQUIRE(C->brep_enabled, "hcp_boundary_repulsion: call hcp_set_boundary_repulsion first");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  if (!C->rep[0]) for (int d = 0; d < 3; d++) { HC_HIP(hipMalloc((void **)&C->rep[d], C->cap * sizeof(double))); HC_HIP(hipMemset(C->rep[d], 0, C->cap * sizeof(double))); }
  const LatView v = make_view(C->L);
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    hipLaunchKernelGGL(boundary_rep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, n, (const uint8_t *)C->d_bflag,
                       (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                       C->rep[0] + f, C->rep[1] + f, C->rep[2] + f, C->brep_const, C->brep_cutoff,
                       (const int *)(C->d_vert_cell + f), (const int *)C->d_tag, (const unsigned char *)(C->d_vdead + f));
    HC_HIP(hipGetLastError());
  }
  return HC_OK;
}

// force_repulsion of every vertex, [n][3] in download order
int hcp_download_repulsion(hc_cells *C, double *out) {
  HC_REQUIRE(C && out, "hcp_download_repulsion: null pointer");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_device(C); if (rc != HC_OK) return rc;
  HC_HIP(hipStreamSynchronize(hc::stream()));
  std::vector<double> tmp;
  size_t o = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv;
    tmp.resize((size_t)n);
    for (int d = 0; d < 3; d++) {
      if (n && C->rep[d]) HC_HIP(hipMemcpy(tmp.data(), C->rep[d] + C->first[t], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
      else std::fill(tmp.begin(), tmp.end(), 0.0);
      for (long i = 0; i < n; i++) out[o + 3 * (size_t)i + d] = tmp[(size_t)i];
    }
    o += 3 * (size_t)n;
  }
  return HC_OK;
}


}  // extern "C"
