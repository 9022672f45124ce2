// Membrane mechanics on the GPU: rbcHighOrderModel and pltSimpleModel forces, per-cell information.
//
// Replaces (file:line in the HemoCell tree):
//   core/hemoCellParticleField.cpp:633-675        applyConstitutiveModel
//   mechanics/rbcHighOrderModel.cpp:38-207, mechanics/pltSimpleModel.cpp:44-208
//   helper/cellInfo.cpp                           volume, area, position, bounding box per cell
//
// The membrane models are evaluated in GATHER form: one workgroup per cell,
// vertex positions in LDS, each vertex sums the contributions of its incident
// triangles / ring / edges in exactly the order in which the reference's
// scatter loops would have added them (triangle loop, volume loop, vertex
// loop, edge loop), so no atomics are needed and the result is reproducible
// and (with -ffp-contract=off) bit-identical to the scatter form.
#include "cells.h"

namespace {

// ----------------------------------------------------------------------------
// membrane mechanics
struct MechArgs {
  int model, nv, nt, ne, nie;
  const int *tri, *edge, *ebt, *ebo, *iedge;
  const int *vtri, *vtri_k, *vedge, *vedge_s, *bsrc, *vouter, *vinner, *vinner_s, *ring, *nring;
  const double *tri_area_eq, *edge_len_eq, *edge_angle_eq, *patch_eq, *iedge_len_eq;
  double volume_eq, area_mean_eq, edge_mean_eq;
  double k_volume, k_area, k_link, k_bend, eta_m;
  const double *px, *py, *pz, *vx, *vy, *vz;   // already offset to the type's first vertex
  double *fx, *fy, *fz;
  double *comp;        // optional [6][ncells*nv][3]
  long ncv;            // ncells*nv (stride of comp)
  const int *tag;      // per cell of this type: 0 complete, 1 gone, 2 incomplete
};

#define MaxCellVolumetricChange 0.01   // config/constant_defaults.h:157-173
#define MaxCellSurfaceAreaChange 0.09
#define MaxCellBendingAngle 0.0555
#define MaxPLTBendingAngle 2.467
#define MaxCellPersistenceLength 9.0
#define FORCE_LIMIT_PN 50.0

constexpr int MD = CellTables::MAXD;

__device__ __forceinline__ double norm3(double a, double b, double c) { double r = 0.0; r += a * a; r += b * b; r += c * c; return sqrt(r); }
__device__ __forceinline__ double dot3(double a0, double a1, double a2, double b0, double b1, double b2) { double r = 0.0; r += a0 * b0; r += a1 * b1; r += a2 * b2; return r; }

// area, unit normal and area-force magnitude of triangle t from the LDS copy of the positions (helper/array.h:270-285,
// rbcHighOrderModel.cpp:70-79).  The RBC kernel evaluates this where it is needed instead of keeping five doubles per
// triangle in LDS: same operations, same bits, and the workgroup drops from 92 KB to 41 KB of LDS (three per CU).
struct TriGeom { double area, nx, ny, nz; };
__device__ __forceinline__ TriGeom tri_geom(const double *xs, const double *ys, const double *zs, int i0, int i1, int i2) {
  const double v0x = xs[i0], v0y = ys[i0], v0z = zs[i0];
  const double e1x = xs[i1] - v0x, e1y = ys[i1] - v0y, e1z = zs[i1] - v0z, e2x = xs[i2] - v0x, e2y = ys[i2] - v0y, e2z = zs[i2] - v0z;
  TriGeom g;
  g.nx = e1y * e2z - e1z * e2y; g.ny = e1z * e2x - e1x * e2z; g.nz = e1x * e2y - e1y * e2x;
  const double nn = norm3(g.nx, g.ny, g.nz);
  if (nn != 0.0) { g.area = 0.5 * nn; g.nx /= nn; g.ny /= nn; g.nz /= nn; } else { g.area = 0.0; g.nx = g.ny = g.nz = 0.0; }
  return g;
}
__device__ __forceinline__ double tri_signed_volume(const double *xs, const double *ys, const double *zs, int i0, int i1, int i2) {
  const double v0x = xs[i0], v0y = ys[i0], v0z = zs[i0], v1x = xs[i1], v1y = ys[i1], v1z = zs[i1], v2x = xs[i2], v2y = ys[i2], v2z = zs[i2];
  const double v210 = v2x * v1y * v0z, v120 = v1x * v2y * v0z, v201 = v2x * v0y * v1z;
  const double v021 = v0x * v2y * v1z, v102 = v1x * v0y * v2z, v012 = v0x * v1y * v2z;
  return (-v210 + v120 + v201 - v021 - v102 + v012);
}

// one workgroup = one cell.  LDS: positions and the per-triangle signed-volume terms; RBC: per-vertex bending vector;
// PLT (small mesh): per-triangle {area, unit normal, area-force magnitude} and per-edge {link, visc, bending} vectors.
template <int MODEL, bool SEPARATE>
__global__ __launch_bounds__(256) void mechanics_kernel(MechArgs m) {
  extern __shared__ double lds[];
  constexpr bool PLT = MODEL != HC_MODEL_RBC_HO;
  const int nv = m.nv, nt = m.nt, ne = m.ne;
  double *xs = lds, *ys = xs + nv, *zs = ys + nv;
  double *tV = zs + nv;
  double *tA = tV + nt, *tNx = tA + nt, *tNy = tNx + nt, *tNz = tNy + nt, *tAfm = tNz + nt;   // PLT only
  double *ex = PLT ? tAfm + nt : tV + nt;  // RBC: B[3][nv]; PLT: edge vectors [9][ne]
  __shared__ double s_volume_force;
  const int tid = threadIdx.x, nth = blockDim.x;
  const long base = (long)blockIdx.x * nv;
  const int state = m.tag[blockIdx.x];
  if (state == 1) return;   // the cell is gone
  if (state == 2) {
    // incomplete cell: applyConstitutiveModel zeroes the force of every particle of the type it finds, but only complete
    // cells reach ParticleMechanics (core/hemoCellParticleField.cpp:634-669)
    if (!SEPARATE) for (int i = tid; i < nv; i += nth) { m.fx[base + i] = 0.0; m.fy[base + i] = 0.0; m.fz[base + i] = 0.0; }
    else for (int i = tid; i < nv; i += nth) for (int c = 0; c < 6; c++) for (int d = 0; d < 3; d++) m.comp[((long)c * m.ncv + base + i) * 3 + d] = 0.0;
    return;
  }

  for (int i = tid; i < nv; i += nth) { xs[i] = m.px[base + i]; ys[i] = m.py[base + i]; zs[i] = m.pz[base + i]; }
  __syncthreads();

  // ---- per-triangle quantities (rbcHighOrderModel.cpp:56-98 / pltSimpleModel.cpp:57-99)
  for (int t = tid; t < nt; t += nth) {
    const int i0 = m.tri[3 * t], i1 = m.tri[3 * t + 1], i2 = m.tri[3 * t + 2];
    tV[t] = tri_signed_volume(xs, ys, zs, i0, i1, i2);
    if (PLT) {
      const TriGeom g = tri_geom(xs, ys, zs, i0, i1, i2);
      tA[t] = g.area; tNx[t] = g.nx; tNy[t] = g.ny; tNz[t] = g.nz;
      const double aeq = m.tri_area_eq[t];
      const double areaRatio = (g.area - aeq) / aeq;
      tAfm[t] = m.k_area * (areaRatio + areaRatio / fabs(MaxCellSurfaceAreaChange - areaRatio * areaRatio));
    }
  }
  __syncthreads();
  if (tid == nth - 1) {   // a lane of the last wave, which has the smallest share of the pass below
    // the reference accumulates the signed-volume terms sequentially in triangle order; do the same so
    // that every copy of a cell (other GPUs, the CPU oracle) gets the same bits.  Runs beside the bending / edge pass.
    double volume = 0.0;
#pragma unroll 8
    for (int t = 0; t < nt; t++) volume += tV[t];
    volume *= (1.0 / 6.0);
    const double vf = (volume - m.volume_eq) / m.volume_eq;
    s_volume_force = -m.k_volume * vf / fabs(MaxCellVolumetricChange - vf * vf);
  }

  if (MODEL == HC_MODEL_RBC_HO) {
    // ---- per-vertex bending vector (rbcHighOrderModel.cpp:127-160)
    double *Bx = ex, *By = ex + nv, *Bz = ex + 2 * nv;
    for (int i = tid; i < nv; i += nth) {
      const int nn = m.nring[i];
      const double x = xs[i], y = ys[i], z = zs[i];
      double sx = 0., sy = 0., sz = 0.;
      for (int j = 0; j < nn; j++) { const int r = m.ring[6 * i + j]; sx += xs[r]; sy += ys[r]; sz += zs[r]; }
      const double dvx = sx / nn - x, dvy = sy / nn - y, dvz = sz / nn - z;
      double pnx = 0., pny = 0., pnz = 0.;
      for (int j = 0; j < nn; j++) {
        const int ra = m.ring[6 * i + j], rb = m.ring[6 * i + (j + 1 == nn ? 0 : j + 1)];
        const double ax = xs[ra] - x, ay = ys[ra] - y, az = zs[ra] - z, bx = xs[rb] - x, by = ys[rb] - y, bz = zs[rb] - z;
        double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const double l = norm3(cx, cy, cz);
        cx /= l; cy /= l; cz /= l;
        pnx += cx; pny += cy; pnz += cz;
      }
      const double l = norm3(pnx, pny, pnz);
      pnx /= l; pny /= l; pnz /= l;
      const double ndev = dot3(pnx, pny, pnz, dvx, dvy, dvz);
      const double dDev = (ndev - m.patch_eq[i]) / m.edge_mean_eq;
      const double mag = m.k_bend * (dDev + dDev / fabs(MaxCellBendingAngle - dDev * dDev));
      Bx[i] = mag * pnx; By[i] = mag * pny; Bz[i] = mag * pnz;
    }
  } else {
    // ---- per-edge vectors (pltSimpleModel.cpp:120-183): link, viscosity, dihedral bending
    double *Lx = ex, *Ly = ex + ne, *Lz = ex + 2 * ne, *Vx = ex + 3 * ne, *Vy = ex + 4 * ne, *Vz = ex + 5 * ne,
           *Gx = ex + 6 * ne, *Gy = ex + 7 * ne, *Gz = ex + 8 * ne;
    for (int e = tid; e < ne; e += nth) {
      const int e0 = m.edge[2 * e], e1 = m.edge[2 * e + 1];
      const double evx = xs[e1] - xs[e0], evy = ys[e1] - ys[e0], evz = zs[e1] - zs[e0];
      const double el = sqrt(evx * evx + evy * evy + evz * evz);
      const double ux = evx / el, uy = evy / el, uz = evz / el;
      const double leq = m.edge_len_eq[e];
      const double ef = (el - leq) / leq;
      const double fs = m.k_link * (ef + ef / fabs(MaxCellPersistenceLength - ef * ef));
      Lx[e] = ux * fs; Ly[e] = uy * fs; Lz[e] = uz * fs;
      const double rvx = m.vx[base + e1] - m.vx[base + e0], rvy = m.vy[base + e1] - m.vy[base + e0], rvz = m.vz[base + e1] - m.vz[base + e0];
      const double pr = dot3(rvx, rvy, rvz, ux, uy, uz);
      double wx = m.eta_m * (pr * ux), wy = m.eta_m * (pr * uy), wz = m.eta_m * (pr * uz);
      const double wm = norm3(wx, wy, wz);
      if (wm > FORCE_LIMIT_PN / 4.0) { const double sc = (FORCE_LIMIT_PN / 4.0) / wm; wx *= sc; wy *= sc; wz *= sc; }
      Vx[e] = wx; Vy[e] = wy; Vz[e] = wz;
      const int b0 = m.ebt[2 * e], b1 = m.ebt[2 * e + 1];
      const double a = tNx[b0] + tNx[b1], b = tNy[b0] + tNy[b1], c = tNz[b0] + tNz[b1];
      // getAngleBetweenFaces (helper/geometryUtils.h:49-52)
      const double crx = tNy[b0] * tNz[b1] - tNz[b0] * tNy[b1], cry = tNz[b0] * tNx[b1] - tNx[b0] * tNz[b1], crz = tNx[b0] * tNy[b1] - tNy[b0] * tNx[b1];
      const double angle = atan2(dot3(crx, cry, crz, ux, uy, uz), dot3(tNx[b0], tNy[b0], tNz[b0], tNx[b1], tNy[b1], tNz[b1]));
      const double af = angle - m.edge_angle_eq[e];
      const double fm = m.k_bend * (af + af / fabs(MaxPLTBendingAngle - af * af));
      Gx[e] = (fm * a) * 0.5; Gy[e] = (fm * b) * 0.5; Gz[e] = (fm * c) * 0.5;
    }
  }
  __syncthreads();
  const double volume_force = s_volume_force;

  // ---- per-vertex gather in the reference's accumulation order
  for (int i = tid; i < nv; i += nth) {
    // components: 0 volume, 1 area, 2 bending, 3 link, 4 visc, 5 inner link; unified mode uses slot 0 only
    double acc[SEPARATE ? 6 : 1][3];
#pragma unroll
    for (int c = 0; c < (SEPARATE ? 6 : 1); c++) acc[c][0] = acc[c][1] = acc[c][2] = 0.0;
#define ACC(C) acc[SEPARATE ? (C) : 0]
    const double x = xs[i], y = ys[i], z = zs[i];
    for (int k = 0; k < MD; k++) {  // area force, triangle order
      const int t = m.vtri[MD * i + k];
      if (t < 0) break;
      const int i0 = m.tri[3 * t], i1 = m.tri[3 * t + 1], i2 = m.tri[3 * t + 2];
      const double cx = (xs[i0] + xs[i1] + xs[i2]) / 3.0, cy = (ys[i0] + ys[i1] + ys[i2]) / 3.0, cz = (zs[i0] + zs[i1] + zs[i2]) / 3.0;
      double afm;
      if (PLT) afm = tAfm[t];
      else {
        const TriGeom g = tri_geom(xs, ys, zs, i0, i1, i2);
        const double aeq = m.tri_area_eq[t];
        const double areaRatio = (g.area - aeq) / aeq;
        afm = m.k_area * (areaRatio + areaRatio / fabs(MaxCellSurfaceAreaChange - areaRatio * areaRatio));
      }
      ACC(1)[0] += afm * (cx - x); ACC(1)[1] += afm * (cy - y); ACC(1)[2] += afm * (cz - z);
    }
    for (int k = 0; k < MD; k++) {  // volume force, triangle order (rbcHighOrderModel.cpp:107-113)
      const int t = m.vtri[MD * i + k];
      if (t < 0) break;
      TriGeom g;
      if (PLT) { g.area = tA[t]; g.nx = tNx[t]; g.ny = tNy[t]; g.nz = tNz[t]; }
      else g = tri_geom(xs, ys, zs, m.tri[3 * t], m.tri[3 * t + 1], m.tri[3 * t + 2]);
      const double sc = g.area / m.area_mean_eq;
      ACC(0)[0] += (volume_force * g.nx) * sc; ACC(0)[1] += (volume_force * g.ny) * sc; ACC(0)[2] += (volume_force * g.nz) * sc;
    }
    if (MODEL == HC_MODEL_RBC_HO) {
      const double *Bx = ex, *By = ex + nv, *Bz = ex + 2 * nv;
      for (int k = 0; k < MD; k++) {  // bending: own vector, or -B/n of a ring neighbour, ascending source id
        const int src = m.bsrc[MD * i + k];
        if (src < 0) break;
        if (src == i) { ACC(2)[0] += Bx[i]; ACC(2)[1] += By[i]; ACC(2)[2] += Bz[i]; }
        else { const int nn = m.nring[src]; ACC(2)[0] += -Bx[src] / nn; ACC(2)[1] += -By[src] / nn; ACC(2)[2] += -Bz[src] / nn; }
      }
      for (int k = 0; k < MD; k++) {  // links (rbcHighOrderModel.cpp:169-204)
        const int e = m.vedge[MD * i + k];
        if (e < 0) break;
        const int e0 = m.edge[2 * e], e1 = m.edge[2 * e + 1];
        const double evx = xs[e1] - xs[e0], evy = ys[e1] - ys[e0], evz = zs[e1] - zs[e0];
        const double el = norm3(evx, evy, evz);
        const double ux = evx / el, uy = evy / el, uz = evz / el;
        const double leq = m.edge_len_eq[e];
        const double ef = (el - leq) / leq;
        const double fs = m.k_link * (ef + ef / fabs(MaxCellPersistenceLength - ef * ef));
        const double frx = ux * fs, fry = uy * fs, frz = uz * fs;
        const bool first = m.vedge_s[MD * i + k] > 0;
        if (first) { ACC(3)[0] += frx; ACC(3)[1] += fry; ACC(3)[2] += frz; } else { ACC(3)[0] -= frx; ACC(3)[1] -= fry; ACC(3)[2] -= frz; }
        if (m.eta_m != 0.0) {
          const double rvx = m.vx[base + e1] - m.vx[base + e0], rvy = m.vy[base + e1] - m.vy[base + e0], rvz = m.vz[base + e1] - m.vz[base + e0];
          const double pr = dot3(rvx, rvy, rvz, ux, uy, uz);
          double wx = m.eta_m * (pr * ux), wy = m.eta_m * (pr * uy), wz = m.eta_m * (pr * uz);
          const double wm = norm3(wx, wy, wz);
          if (wm > FORCE_LIMIT_PN / 4.0) { const double sc = (FORCE_LIMIT_PN / 4.0) / wm; wx *= sc; wy *= sc; wz *= sc; }
          if (first) { ACC(4)[0] += wx; ACC(4)[1] += wy; ACC(4)[2] += wz; } else { ACC(4)[0] -= wx; ACC(4)[1] -= wy; ACC(4)[2] -= wz; }
        }
      }
    } else {
      const double *Lx = ex, *Ly = ex + ne, *Lz = ex + 2 * ne, *Vx = ex + 3 * ne, *Vy = ex + 4 * ne, *Vz = ex + 5 * ne,
                   *Gx = ex + 6 * ne, *Gy = ex + 7 * ne, *Gz = ex + 8 * ne;
      // merge of the vertex's own edges and the edges it is an outer point of, ascending edge id
      int ka = 0, kb = 0;
      while (true) {
        const int ea = ka < MD ? m.vedge[MD * i + ka] : -1, eb = kb < MD ? m.vouter[MD * i + kb] : -1;
        if (ea < 0 && eb < 0) break;
        if (eb < 0 || (ea >= 0 && ea < eb)) {
          const bool first = m.vedge_s[MD * i + ka] > 0;
          if (first) { ACC(3)[0] += Lx[ea]; ACC(3)[1] += Ly[ea]; ACC(3)[2] += Lz[ea]; ACC(4)[0] += Vx[ea]; ACC(4)[1] += Vy[ea]; ACC(4)[2] += Vz[ea]; }
          else { ACC(3)[0] -= Lx[ea]; ACC(3)[1] -= Ly[ea]; ACC(3)[2] -= Lz[ea]; ACC(4)[0] -= Vx[ea]; ACC(4)[1] -= Vy[ea]; ACC(4)[2] -= Vz[ea]; }
          ACC(2)[0] += Gx[ea]; ACC(2)[1] += Gy[ea]; ACC(2)[2] += Gz[ea];
          ka++;
        } else {
          ACC(2)[0] -= Gx[eb]; ACC(2)[1] -= Gy[eb]; ACC(2)[2] -= Gz[eb];
          kb++;
        }
      }
      for (int k = 0; k < MD; k++) {  // inner links (pltSimpleModel.cpp:186-205)
        const int e = m.vinner[MD * i + k];
        if (e < 0) break;
        const int e0 = m.iedge[2 * e], e1 = m.iedge[2 * e + 1];
        const double evx = xs[e1] - xs[e0], evy = ys[e1] - ys[e0], evz = zs[e1] - zs[e0];
        const double el = sqrt(evx * evx + evy * evy + evz * evz);
        const double ux = evx / el, uy = evy / el, uz = evz / el;
        const double leq = m.iedge_len_eq[e];
        const double ef = (el - leq) / leq;
        const double fs = m.k_link * 5.0 * ef;
        if (m.vinner_s[MD * i + k] > 0) { ACC(5)[0] += ux * fs; ACC(5)[1] += uy * fs; ACC(5)[2] += uz * fs; }
        else { ACC(5)[0] -= ux * fs; ACC(5)[1] -= uy * fs; ACC(5)[2] -= uz * fs; }
      }
    }
#undef ACC
    if (SEPARATE) {
#pragma unroll
      for (int c = 0; c < 6; c++)
        for (int d = 0; d < 3; d++) m.comp[((long)c * m.ncv + base + i) * 3 + d] = acc[c][d];
    } else {
      m.fx[base + i] = acc[0][0]; m.fy[base + i] = acc[0][1]; m.fz[base + i] = acc[0][2];
    }
  }
}

// per-cell volume / area / bbox / centroid (helper/cellInfo.cpp:39-80,140-180)
__global__ __launch_bounds__(256) void cell_info_kernel(int nv, int nt, const int *tri, const double *px, const double *py, const double *pz,
                                                        double *volume, double *area, double *bbox, double *centroid) {
  __shared__ double red[256][11];
  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * nv;
  double vol = 0, ar = 0, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, cs[3] = {0, 0, 0};
  for (int t = tid; t < nt; t += 256) {
    const long i0 = base + tri[3 * t], i1 = base + tri[3 * t + 1], i2 = base + tri[3 * t + 2];
    const double v0x = px[i0], v0y = py[i0], v0z = pz[i0], v1x = px[i1], v1y = py[i1], v1z = pz[i1], v2x = px[i2], v2y = py[i2], v2z = pz[i2];
    vol += (-v2x * v1y * v0z + v1x * v2y * v0z + v2x * v0y * v1z - v0x * v2y * v1z - v1x * v0y * v2z + v0x * v1y * v2z);
    const double e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z, e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z;
    const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    ar += 0.5 * sqrt(nx * nx + ny * ny + nz * nz);
  }
  for (int i = tid; i < nv; i += 256) {
    const double p[3] = {px[base + i], py[base + i], pz[base + i]};
    for (int d = 0; d < 3; d++) { lo[d] = fmin(lo[d], p[d]); hi[d] = fmax(hi[d], p[d]); cs[d] += p[d]; }
  }
  red[tid][0] = vol; red[tid][1] = ar;
  for (int d = 0; d < 3; d++) { red[tid][2 + d] = lo[d]; red[tid][5 + d] = hi[d]; red[tid][8 + d] = cs[d]; }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[tid][0] += red[tid + s][0]; red[tid][1] += red[tid + s][1];
      for (int d = 0; d < 3; d++) {
        red[tid][2 + d] = fmin(red[tid][2 + d], red[tid + s][2 + d]);
        red[tid][5 + d] = fmax(red[tid][5 + d], red[tid + s][5 + d]);
        red[tid][8 + d] += red[tid + s][8 + d];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const long c = blockIdx.x;
    volume[c] = red[0][0] / 6.0; area[c] = red[0][1];
    // bbox order x0 x1 y0 y1 z0 z1 (helper/cellInfo.cpp:148-160)
    for (int d = 0; d < 3; d++) { bbox[6 * c + 2 * d] = red[0][2 + d]; bbox[6 * c + 2 * d + 1] = red[0][5 + d]; centroid[3 * c + d] = red[0][8 + d] / nv; }
  }
}

}  // namespace

static MechArgs mech_args(const hc_cells *C, int t) {
  const hc_celltype *T = C->types[t];
  MechArgs m;
  m.model = T->host.model; m.nv = T->host.nv; m.nt = T->host.nt; m.ne = T->host.ne; m.nie = T->host.nie;
  m.tri = T->d_tri; m.edge = T->d_edge; m.ebt = T->d_ebt; m.ebo = T->d_ebo; m.iedge = T->d_iedge;
  m.vtri = T->d_vtri; m.vtri_k = T->d_vtri_k; m.vedge = T->d_vedge; m.vedge_s = T->d_vedge_s; m.bsrc = T->d_bsrc;
  m.vouter = T->d_vouter; m.vinner = T->d_vinner; m.vinner_s = T->d_vinner_s; m.ring = T->d_ring; m.nring = T->d_nring;
  m.tri_area_eq = T->d_tri_area_eq; m.edge_len_eq = T->d_edge_len_eq; m.edge_angle_eq = T->d_edge_angle_eq;
  m.patch_eq = T->d_patch_eq; m.iedge_len_eq = T->d_iedge_len_eq;
  m.volume_eq = T->host.volume_eq; m.area_mean_eq = T->host.area_mean_eq; m.edge_mean_eq = T->host.edge_mean_eq;
  m.k_volume = T->host.k_volume; m.k_area = T->host.k_area; m.k_link = T->host.k_link; m.k_bend = T->host.k_bend; m.eta_m = T->host.eta_m;
  const long f = C->first[t];
  m.px = C->pos[0] + f; m.py = C->pos[1] + f; m.pz = C->pos[2] + f;
  m.vx = C->vel[0] + f; m.vy = C->vel[1] + f; m.vz = C->vel[2] + f;
  m.fx = C->frc[0] + f; m.fy = C->frc[1] + f; m.fz = C->frc[2] + f;
  m.comp = nullptr; m.ncv = C->ncells[t] * T->host.nv;
  m.tag = C->d_tag + C->cell0[t];
  return m;
}

static size_t mech_lds_bytes(const CellTables &T) {
  if (T.model == HC_MODEL_RBC_HO) return (6 * (size_t)T.nv + (size_t)T.nt) * sizeof(double);   // positions, bending vectors, signed-volume terms
  return (3 * (size_t)T.nv + 6 * (size_t)T.nt + 9 * (size_t)T.ne) * sizeof(double);
}

static int launch_mechanics(hc_cells *C, int t, double *comp) {
  if (C->ncells[t] == 0) return HC_OK;
  MechArgs m = mech_args(C, t);
  m.comp = comp;
  const CellTables &T = C->types[t]->host;
  const size_t lds = mech_lds_bytes(T);
  HC_REQUIRE(lds <= 160 * 1024 - 64, "mechanics: cell type does not fit the 160 KiB LDS of a CU");
  const int threads = T.nv > 128 ? 256 : 128;
  const dim3 grid((unsigned)C->ncells[t]);
#define LAUNCH(MODEL, SEP)                                                                                        \
  do {                                                                                                            \
    HC_HIP(hipFuncSetAttribute((const void *)mechanics_kernel<MODEL, SEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((mechanics_kernel<MODEL, SEP>), grid, dim3(threads), lds, hc::stream(), m);                \
  } while (0)
  if (T.model == HC_MODEL_RBC_HO) { if (comp) LAUNCH(HC_MODEL_RBC_HO, true); else LAUNCH(HC_MODEL_RBC_HO, false); }
  else { if (comp) LAUNCH(HC_MODEL_PLT_SIMPLE, true); else LAUNCH(HC_MODEL_PLT_SIMPLE, false); }
#undef LAUNCH
  HC_HIP(hipGetLastError());
  return HC_OK;
}

extern "C" {

int hcp_mechanics(hc_cells *C, long iter, int forced) {
  HC_REQUIRE(C, "hcp_mechanics: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  hc::ProfScope prof(hc::PK_MECH);
  for (int t = 0; t < C->ntypes; t++) {
    if (!(iter % C->timescale[t] == 0 || forced)) continue;  // core/hemoCellParticleField.cpp:655
    rc = launch_mechanics(C, t, nullptr);
    if (rc != HC_OK) return rc;
  }
  return HC_OK;
}

// persistent device + pinned host scratch of the information calls (drivers call them at every measurement step):
// no allocation, one asynchronous copy into pinned memory, one wait
static int info_scratch(hc_cells *C, size_t doubles) {
  if (C->info_cap >= doubles) return HC_OK;
  HC_HIP(hipStreamSynchronize(hc::stream()));
  if (C->d_info) HC_HIP(hipFree(C->d_info));
  if (C->h_info) HC_HIP(hipHostFree(C->h_info));
  C->d_info = C->h_info = nullptr; C->info_cap = 0;
  const size_t cap = doubles + doubles / 4 + 1024;
  HC_HIP(hipMalloc((void **)&C->d_info, cap * sizeof(double)));
  HC_HIP(hipHostMalloc((void **)&C->h_info, cap * sizeof(double), hipHostMallocDefault));
  C->info_cap = cap;
  return HC_OK;
}

int hcp_mechanics_components(hc_cells *C, int type, double *comp) {
  HC_REQUIRE(C && comp && type >= 0 && type < C->ntypes, "hcp_mechanics_components: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long n = C->ncells[type] * C->types[type]->host.nv;
  if (n == 0) return HC_OK;
  rc = info_scratch(C, (size_t)(18 * n)); if (rc != HC_OK) return rc;
  rc = launch_mechanics(C, type, C->d_info); if (rc != HC_OK) return rc;
  HC_HIP(hipMemcpyAsync(C->h_info, C->d_info, (size_t)(18 * n) * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  std::memcpy(comp, C->h_info, (size_t)(18 * n) * sizeof(double));
  return HC_OK;
}

int hcp_cell_info(hc_cells *C, int type, double *volume, double *area, double *bbox, double *centroid) {
  HC_REQUIRE(C && volume && area && bbox && centroid && type >= 0 && type < C->ntypes, "hcp_cell_info: bad arguments");
  int rc = settle(C); if (rc != HC_OK) return rc;
  rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long nc = C->ncells[type];
  if (nc == 0) return HC_OK;
  const CellTables &T = C->types[type]->host;
  rc = info_scratch(C, (size_t)(11 * nc)); if (rc != HC_OK) return rc;
  double *d = C->d_info;
  const long f = C->first[type];
  hipLaunchKernelGGL(cell_info_kernel, dim3((unsigned)nc), dim3(256), 0, hc::stream(), T.nv, T.nt, (const int *)C->types[type]->d_tri,
                     (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f), d, d + nc, d + 2 * nc, d + 8 * nc);
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(C->h_info, d, (size_t)(11 * nc) * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  std::memcpy(volume, C->h_info, (size_t)nc * sizeof(double));
  std::memcpy(area, C->h_info + nc, (size_t)nc * sizeof(double));
  std::memcpy(bbox, C->h_info + 2 * nc, (size_t)(6 * nc) * sizeof(double));
  std::memcpy(centroid, C->h_info + 8 * nc, (size_t)(3 * nc) * sizeof(double));
  return HC_OK;
}

}  // extern "C"
