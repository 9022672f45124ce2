// Immersed-boundary coupling on the GPU: phi2 force spreading and velocity interpolation.
//
// Replaces (file:line in the HemoCell tree):
//   core/immersedBoundaryMethod.h:62-138          interpolationCoefficientsPhi2 (cells.h: phi2_stencil)
//   core/hemoCellParticleField.cpp:841-863        spreadParticleForce
//   core/hemoCellParticleField.cpp:819-839        interpolateFluidVelocity
#include "cells.h"
#include <hipcub/hipcub.hpp>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace {

// ----------------------------------------------------------------------------
// spread
__global__ __launch_bounds__(256) void ibm_spread_kernel(LatView v, long n, const double *px, const double *py, const double *pz,
                                                         double *fx, double *fy, double *fz, const double *rx, const double *ry, const double *rz,
                                                         double *F, int limit_on, double f_limit, const int *vert_cell, const int *tag, const unsigned char *dead) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (dead[i] || tag[vert_cell[i]] == 1) return;   // removed particle / cell gone
  double f0 = fx[i], f1 = fy[i], f2 = fz[i];
  if (limit_on) {  // FORCE_LIMIT cap, core/hemoCellParticleField.cpp:848-852 (mutates sv.force)
    const double mag = sqrt((f0 * f0 + f1 * f1) + f2 * f2);
    if (mag > f_limit) {
      const double sc = f_limit / mag;
      f0 *= sc; f1 *= sc; f2 *= sc;
      fx[i] = f0; fy[i] = f1; fz[i] = f2;
    }
  }
  Stencil s;
  phi2_stencil(v, px[i], py[i], pz[i], s);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (s.node[k] < 0) continue;
    v.dirty[s.node[k] >> 4] = v.epoch;
    // external.data[d] += (force_repulsion[d] + force[d]) * weight  (:857-859)
    unsafeAtomicAdd(&F[s.node[k]], ((rx ? rx[i] : 0.0) + f0) * s.w[k]);
    unsafeAtomicAdd(&F[v.npad + s.node[k]], ((ry ? ry[i] : 0.0) + f1) * s.w[k]);
    unsafeAtomicAdd(&F[2 * v.npad + s.node[k]], ((rz ? rz[i] : 0.0) + f2) * s.w[k]);
  }
}

// ----------------------------------------------------------------------------
// reproducible spread (hc_set_reproducible_spread).  The reference adds particle by particle in storage order
// (core/hemoCellParticleField.cpp:841-863), so a node's force is a sum in a fixed order and a run repeats bit for bit; the
// atomic kernels above add in whatever order the hardware takes them.  Here every (particle, admitted stencil node) becomes an
// entry keyed by its node, written in the canonical order (cell type, cell id, vertex id, stencil node); a stable radix
// sort by node groups the entries of a node without disturbing that order, and ONE thread per node sums them in sequence and
// adds the sum to the node.  The order does not depend on the storage slots of the cells, so every slab that holds a node
// forms the bits the single domain forms.
__global__ __launch_bounds__(256) void spread_emit_kernel(LatView v, int nv, long n, const int *order, long ebase, const double *px, const double *py,
                                                          const double *pz, double *fx, double *fy, double *fz, const double *rx, const double *ry,
                                                          const double *rz, int limit_on, double f_limit, const int *tag, const unsigned char *dead,
                                                          unsigned int *keys, int *vals, double *c0, double *c1, double *c2) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;   // (rank of the cell in id order) * nv + vertex
  if (g >= n) return;
  const int cell = order[g / nv];
  const long i = (long)cell * nv + (g % nv);
  const long e0 = (ebase + g) * 8;
  const bool live = tag[cell] != 1 && !dead[i];
  Stencil s;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  if (live) {
    f0 = fx[i]; f1 = fy[i]; f2 = fz[i];
    if (limit_on) {  // FORCE_LIMIT cap, core/hemoCellParticleField.cpp:848-852 (mutates sv.force)
      const double mag = sqrt((f0 * f0 + f1 * f1) + f2 * f2);
      if (mag > f_limit) { const double sc = f_limit / mag; f0 *= sc; f1 *= sc; f2 *= sc; fx[i] = f0; fy[i] = f1; fz[i] = f2; }
    }
    if (rx) { f0 = rx[i] + f0; f1 = ry[i] + f1; f2 = rz[i] + f2; }   // force_repulsion + force, :857-859
    phi2_stencil(v, px[i], py[i], pz[i], s);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const bool adm = live && s.node[k] >= 0;
    keys[e0 + k] = adm ? (unsigned int)s.node[k] : 0xffffffffu;
    vals[e0 + k] = (int)(e0 + k);
    c0[e0 + k] = adm ? f0 * s.w[k] : 0.0; c1[e0 + k] = adm ? f1 * s.w[k] : 0.0; c2[e0 + k] = adm ? f2 * s.w[k] : 0.0;
  }
}

__global__ __launch_bounds__(256) void spread_gather_kernel(LatView v, long n, const unsigned int *keys, const int *vals, const double *c0, const double *c1,
                                                            const double *c2, double *F) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const unsigned int key = keys[s];
  if (key == 0xffffffffu || (s > 0 && keys[s - 1] == key)) return;   // not the first entry of its node
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (long q = s; q < n && keys[q] == key; q++) { const int e = vals[q]; a0 += c0[e]; a1 += c1[e]; a2 += c2[e]; }
  v.dirty[key >> 4] = v.epoch;
  F[key] += a0; F[v.npad + key] += a1; F[2 * v.npad + key] += a2;   // this thread alone touches the node
}

// ----------------------------------------------------------------------------
// interpolate: v = sum_j w_j * (j/rho + F/2)(node_j) on the post-stream state
struct PopView {
  const double *f; const double *F; double bx, by, bz; long qs;   // qs: population stride
  hc::BodyRegions reg;
};

__device__ __forceinline__ void node_velocity(const LatView &v, const PopView &pv, int lx, int ly, int lz, long node, double u[3]) {
  // A node on the OUTER halo plane of a slab would pull from beyond the allocation.  Only particles whose nearest node lies
  // outside the slab have such a node in their stencil, and their interpolated velocity is never used (the owner's record
  // replaces it, "a local particle wins"), so the value does not matter -- the access must not happen.
  if (v.halo_x && (lx <= -HALO || lx >= v.nx + HALO - 1)) { u[0] = u[1] = u[2] = 0.0; return; }
  // first halo plane of a slab: the neighbour that owns the node has evaluated this very expression and sent the result
  // (3 doubles per node instead of the 19 population planes the gather below would need, slab.hip)
  if (v.halo_x && (lx == -1 || lx == v.nx)) {
    const double *hu = v.halo_u[lx < 0 ? 0 : 1];
    if (hu) { const long k = (long)ly * v.nz + lz; u[0] = hu[k]; u[1] = hu[v.ny_nz + k]; u[2] = hu[2L * v.ny_nz + k]; return; }
  }
  // gather S(node,q) = P(node - c_q, q) with the same wrap rules as the collide kernel
  long xm = -(long)v.plane, xp = (long)v.plane;
  if (v.wrap_x) { if (lx == 0) xm = (long)(v.nx - 1) * v.plane; if (lx == v.nx - 1) xp = -(long)(v.nx - 1) * v.plane; }
  int ym = -v.nz, yp = v.nz, zm = -1, zp = 1; bool ymk = true, ypk = true, zmk = true, zpk = true;
  if (ly == 0) { if (v.per_y) ym = (v.ny - 1) * v.nz; else ymk = false; }
  if (ly == v.ny - 1) { if (v.per_y) yp = -(v.ny - 1) * v.nz; else ypk = false; }
  if (lz == 0) { if (v.per_z) zm = v.nz - 1; else zmk = false; }
  if (lz == v.nz - 1) { if (v.per_z) zp = -(v.nz - 1); else zpk = false; }
  double r = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
#define M(Q, CX, CY, CZ)                                                              \
  {                                                                                   \
    long off = 0; bool ok = true;                                                     \
    if (CX == 1) off += xm; else if (CX == -1) off += xp;                             \
    if (CY == 1) { off += ym; ok = ok && ymk; } else if (CY == -1) { off += yp; ok = ok && ypk; } \
    if (CZ == 1) { off += zm; ok = ok && zmk; } else if (CZ == -1) { off += zp; ok = ok && zpk; } \
    const double fq = ok ? pv.f[(long)Q * pv.qs + node + off] : 0.0;                 \
    r += fq;                                                                          \
    if (CX == 1) jx += fq; else if (CX == -1) jx += -fq;                              \
    if (CY == 1) jy += fq; else if (CY == -1) jy += -fq;                              \
    if (CZ == 1) jz += fq; else if (CZ == -1) jz += -fq;                              \
  }
  M(0, 0, 0, 0) M(1, -1, 0, 0) M(2, 0, -1, 0) M(3, 0, 0, -1) M(4, -1, -1, 0) M(5, -1, 1, 0)
  M(6, -1, 0, -1) M(7, -1, 0, 1) M(8, 0, -1, -1) M(9, 0, -1, 1) M(10, 1, 0, 0) M(11, 0, 1, 0)
  M(12, 0, 0, 1) M(13, 1, 1, 0) M(14, 1, -1, 0) M(15, 1, 0, 1) M(16, 1, 0, -1) M(17, 0, 1, 1)
  M(18, 0, 1, -1)
#undef M
  const double invRho = 1.0 / (1.0 + r);
  double bx = pv.bx, by = pv.by, bz = pv.bz;
  if (pv.reg.n) {   // a halo plane of a slab, or the wrapped image, is the global node next door
    int xg = v.x0 + lx;
    if (xg < 0) xg += v.nx_global; else if (xg >= v.nx_global) xg -= v.nx_global;
    region_force(pv.reg, xg, ly, lz, bx, by, bz);
  }
  u[0] = jx * invRho + (bx + pv.F[node]) / 2.0;
  u[1] = jy * invRho + (by + pv.F[v.npad + node]) / 2.0;
  u[2] = jz * invRho + (bz + pv.F[2 * v.npad + node]) / 2.0;
}

// node velocities of one face plane of a slab (lx = 0 or nx - 1), for the neighbour whose first halo plane it is
__global__ __launch_bounds__(256) void face_velocity_kernel(LatView v, PopView pv, int lx0, double *out0, int lx1, double *out1) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= v.ny_nz) return;
  const int lx = blockIdx.y ? lx1 : lx0;           // blockIdx.y: which of the (up to two) planes of the launch
  double *out = blockIdx.y ? out1 : out0;
  const int ly = k / v.nz, lz = k - ly * v.nz;
  const long node = (long)(lx + HALO) * v.plane + (long)ly * v.nz + lz;
  double u[3] = {0.0, 0.0, 0.0};
  if (v.mask[node] == 0) node_velocity(v, pv, lx, ly, lz, node, u);   // stencils admit fluid nodes only
  out[k] = u[0]; out[v.ny_nz + k] = u[1]; out[2L * v.ny_nz + k] = u[2];
}

__global__ __launch_bounds__(256) void ibm_interpolate_kernel(LatView v, PopView pv, long n, const double *px, const double *py,
                                                              const double *pz, double *vx, double *vy, double *vz, const int *vert_cell, const int *tag,
                                                              const unsigned char *dead) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (dead[i] || tag[vert_cell[i]] == 1) return;
  Stencil s;
  phi2_stencil(v, px[i], py[i], pz[i], s);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (s.node[k] < 0) continue;
    double u[3];
    node_velocity(v, pv, s.lx[k], s.ly[k], s.lz[k], s.node[k], u);
    a0 += (u[0] * s.w[k]); a1 += (u[1] * s.w[k]); a2 += (u[2] * s.w[k]);
  }
  vx[i] = a0; vy[i] = a1; vz[i] = a2;
}


// ----------------------------------------------------------------------------
// LDS-tiled IBM kernels: one workgroup per cell.
//
// All 8-node stencils of a cell fall into the cell's bounding box (+1).  Spread: the workgroup
// accumulates one force component at a time on an LDS tile of that box (ds_add_f64), then flushes only the
// touched nodes to HBM with one fp64 atomic each, z-contiguous -- several times fewer, better shaped
// global atomics than one per (vertex, node, component).  Interpolate: the nodes the cell touches are
// compacted, the node velocity (19-population gather + moments) is evaluated once per node into LDS, and
// every vertex then blends its 8 values from LDS.
constexpr int TILE_CAP = 5832;       // nodes per tile: 18 x 18 x 18, any orientation of a 642-vertex RBC; three workgroups per CU fit the 160 KB LDS
constexpr int NODE_CAP = 1536;       // distinct nodes of one cell for the interpolation (an RBC touches ~1300)

// bounding box of a cell's stencil nodes: origin o (global), extent e, origin ow in local wrapped coordinates,
// reciprocals for the index decode
struct Tile { int o[3]; int e[3]; int vol; int ow[3]; float r1, r2; };

// tile index -> (tx, ty, tz) without integer division.  Exact for i < 2^16: (i + 0.5) / e is at least 0.5 / e
// away from an integer while the float error stays below 1e-3 / e.
__device__ __forceinline__ void tile_decode(const Tile &t, int i, int &tx, int &ty, int &tz) {
  const int q = (int)(((float)i + 0.5f) * t.r2);
  tz = i - q * t.e[2];
  tx = (int)(((float)q + 0.5f) * t.r1);
  ty = q - tx * t.e[1];
}

__device__ __forceinline__ int stencil_base(const LatView &v, double px, double py, double pz, int b[3]) {
  const double p[3] = {px, py, pz};
#pragma unroll
  for (int a = 0; a < 3; a++) { const long c = nearest_node(p[a]); b[a] = (int)c + ((p[a] < (double)c) ? -1 : 0); }
  return 0;
}

// global lattice element of tile entry i (only meaningful for entries that were admitted by a stencil)
__device__ __forceinline__ long tile_node(const LatView &v, const Tile &t, int i, int &lx, int &ly, int &lz) {
  int tx, ty, tz;
  tile_decode(t, i, tx, ty, tz);
  lx = t.ow[0] + tx; ly = t.ow[1] + ty; lz = t.ow[2] + tz;
  if (v.wrap_x && lx >= v.nx) lx -= v.nx;     // the tile is no wider than the domain (cell_prologue), one wrap suffices
  if (v.per_y && ly >= v.ny) ly -= v.ny;
  if (v.per_z && lz >= v.nz) lz -= v.nz;
  return (long)(lx + HALO) * v.plane + ly * v.nz + lz;
}

// Workgroups are handed to the 8 XCDs round robin, and every XCD has its own L2.  Cells are stored in placement order,
// neighbours in space mostly next to each other; giving each XCD a contiguous range of cells lets neighbouring cells,
// which share the lattice lines around their common boundary, meet in one L2 (interpolation -3 %, spread -1 % on the
// 256^3 pipe, three runs each way).
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int per = n >> 3;            // cells per XCD in the swizzled part
  const int main_n = per << 3;
  if (b >= main_n) return b;          // the remainder keeps its place
  return (b & 7) * per + (b >> 3);
}

// compact per-vertex stencil kept in registers across the passes of the cell kernels
struct VStencil { double w[8]; int base; unsigned adm; };   // base = tile index of the lowest corner; adm = admitted-node bits

constexpr int NVPT = 3;        // vertices per thread held in registers (642 vertices / 256 threads)
// Eight waves per cell (512 threads, two vertices each) were tried for both kernels: the spread gains 3.5 % at 10 % hematocrit
// and nothing at 25 % (and loses once its loads are reordered), the interpolation loses 9-15 % (184 VGPRs: one cell per CU);
// capping the interpolation at 168 / 128 VGPRs for three cells per CU spills and loses 15 %.  Both kernels wait 80 % of their
// wave cycles (rocprofv3 SQ_WAIT_ANY + SQ_WAIT_INST_ANY) and respond to neither more waves nor shorter workgroups.
constexpr int MAXW = 4;        // waves per workgroup

// -DHC_IBM_PHASE_TIMES (scratch builds only): thread 0 of every workgroup adds the shader cycles between consecutive stamps to
// g_phase[k]; hc_debug_phase_times() reads and clears them
#ifdef HC_IBM_PHASE_TIMES
__device__ unsigned long long g_phase[32];
#define PH_PARAM , long long &ph_last
#define PH_ARG , ph_last
#define PH_INIT long long ph_last = clock64();
#define STAMP(k) do { if (threadIdx.x == 0) { const long long ph_now = clock64(); atomicAdd(&g_phase[k], (unsigned long long)(ph_now - ph_last)); ph_last = ph_now; } } while (0)
#else
#define PH_PARAM
#define PH_ARG
#define PH_INIT
#define STAMP(k) do {} while (0)
#endif

// bounding box of all stencil nodes of the cell: per-thread min/max -> wave shuffles -> LDS -> everyone
__device__ __forceinline__ void block_bbox(int lo[3], int hi[3], int *s_red, Tile &t) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo[a] = min(lo[a], __shfl_xor(lo[a], off)); hi[a] = max(hi[a], __shfl_xor(hi[a], off)); }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { s_red[(tid >> 6) * 6 + a] = lo[a]; s_red[(tid >> 6) * 6 + 3 + a] = hi[a]; }
  }
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int l = s_red[a], h = s_red[3 + a];
    for (int w = 1; w < nw; w++) { l = min(l, s_red[w * 6 + a]); h = max(h, s_red[w * 6 + 3 + a]); }
    t.o[a] = l; t.e[a] = h - l + 1;
  }
  const long vol = (long)t.e[0] * t.e[1] * t.e[2];
  t.vol = vol > 0x7fffffff ? 0x7fffffff : (int)vol;
}

// local (wrapped) origin and decode reciprocals; false when the tile cannot be used
__device__ __forceinline__ bool tile_finish(const LatView &v, Tile &t) {
  if (t.vol > TILE_CAP) return false;
  if ((v.wrap_x && t.e[0] > v.nx) || (v.per_y && t.e[1] > v.ny) || (v.per_z && t.e[2] > v.nz)) return false;
  t.ow[0] = v.wrap_x ? (int)pmod((long)t.o[0] - v.x0, v.nx) : t.o[0] - v.x0;
  t.ow[1] = v.per_y ? (int)pmod(t.o[1], v.ny) : t.o[1];
  t.ow[2] = v.per_z ? (int)pmod(t.o[2], v.nz) : t.o[2];
  t.r1 = 1.0f / (float)t.e[1]; t.r2 = 1.0f / (float)t.e[2];
  return true;
}

// mask class of tile entry i: 0 fluid, 1/2 boundary, 3 not addressable (outside the domain / halo range)
__device__ __forceinline__ unsigned char tile_mask(const LatView &v, const Tile &t, int i) {
  int tx, ty, tz;
  tile_decode(t, i, tx, ty, tz);
  int lx = t.ow[0] + tx, ly = t.ow[1] + ty, lz = t.ow[2] + tz;
  if (v.wrap_x) { if (lx >= v.nx) lx -= v.nx; }
  else if (v.halo_x) { if (lx < -HALO || lx >= v.nx + HALO) return 3; }
  else if (lx < 0 || lx >= v.nx) return 3;
  if (v.per_y) { if (ly >= v.ny) ly -= v.ny; } else if (ly < 0 || ly >= v.ny) return 3;
  if (v.per_z) { if (lz >= v.nz) lz -= v.nz; } else if (lz < 0 || lz >= v.nz) return 3;
  return v.mask[(long)(lx + HALO) * v.plane + ly * v.nz + lz];
}

// interpolationCoefficientsPhi2 against the LDS copy of the mask; same arithmetic and visiting order as phi2_stencil
__device__ __forceinline__ void tile_stencil(const Tile &t, const unsigned char *mt, double px, double py, double pz, const int b[3], VStencil &o) {
  const int sy = t.e[2], sx = t.e[1] * t.e[2];
  o.base = ((b[0] - t.o[0]) * t.e[1] + (b[1] - t.o[1])) * t.e[2] + (b[2] - t.o[2]);
  o.adm = 0;
  double total = 0.0;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int idx = i * 4 + j * 2 + k;
        const double weight = phi2(px - (double)(b[0] + i)) * phi2(py - (double)(b[1] + j)) * phi2(pz - (double)(b[2] + k));
        const bool adm = (weight != 0.0) && (!mt || mt[o.base + i * sx + j * sy + k] == 0);   // mt == nullptr: no wall anywhere near this cell
        if (adm) { total += weight; o.adm |= 1u << idx; }
        o.w[idx] = adm ? weight : 0.0;
      }
  const double coeff = 1.0 / total;
#pragma unroll
  for (int idx = 0; idx < 8; idx++) o.w[idx] *= coeff;
}

// Is every node of the tile an ordinary fluid node inside the domain?  Then interpolationCoefficientsPhi2 admits every node
// with a non-zero weight and the mask need not be looked at (most cells of a vessel are nowhere near its wall).  Answered
// from one byte per 8 x 8 x 8 brick of the padded lattice (hc_lattice::wallbrick: set when the brick holds a non-fluid node or
// touches a face the stencils cannot cross); a tile covers at most 4 x 4 x 4 bricks, one per lane of wave 0.
__device__ __forceinline__ bool tile_is_clear(const LatView &v, const Tile &t, int *s_near) {
  const int n[3] = {v.nx, v.ny, v.nz};
  bool inside = v.wallbrick != nullptr;
  int b0[3], nb[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int lo = a == 0 ? t.ow[0] + HALO : t.ow[a], hi = lo + t.e[a] - 1, lim = a == 0 ? v.nx + 2 * HALO : n[a];   // x in padded planes
    inside = inside && lo >= 0 && hi < lim && (a != 0 || !v.wrap_x || (t.ow[0] >= 0 && t.ow[0] + t.e[0] <= v.nx));
    b0[a] = lo >> 3; nb[a] = (hi >> 3) - b0[a] + 1;
  }
  if (!inside) return false;   // uniform: the tile wraps around a periodic face or leaves the addressable range
  if (nb[0] * nb[1] * nb[2] > 64) return false;   // uniform: a thin, long tile (5832 nodes can span more bricks than one wave looks at) takes the mask path
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    bool near = false;
    if (l < nb[0] * nb[1] * nb[2]) {
      const int bz = l % nb[2], by = (l / nb[2]) % nb[1], bx = l / (nb[2] * nb[1]);
      near = v.wallbrick[((long)(b0[0] + bx) * v.nby + (b0[1] + by)) * v.nbz + (b0[2] + bz)] != 0;
    }
    const unsigned long long any = __ballot(near);
    if (l == 0) *s_near = any != 0ull;
  }
  __syncthreads();
  return *s_near == 0;
}

// shared prologue of the two cell kernels: positions -> registers, tile, mask tile (only near walls), stencils.
// returns false (uniformly) when the cell does not fit the tile and the caller must take the fallback path
// dead: removed particles of an INCOMPLETE cell (null for a complete one); they take no part in anything
__device__ __forceinline__ bool cell_prologue(const LatView &v, int nv, long base, const double *px, const double *py, const double *pz,
                                              const unsigned char *dead, int *s_red, int *s_near, unsigned char *mt, Tile &t, VStencil vs[NVPT] PH_PARAM) {
  const int tid = threadIdx.x, nth = blockDim.x;
  double p[NVPT][3]; int b[NVPT][3]; bool live[NVPT];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
  for (int j = 0; j < NVPT; j++) {
    const int i = tid + j * nth;
    live[j] = i < nv && !(dead && dead[base + i]);
    if (live[j]) {
      p[j][0] = px[base + i]; p[j][1] = py[base + i]; p[j][2] = pz[base + i];
      stencil_base(v, p[j][0], p[j][1], p[j][2], b[j]);
#pragma unroll
      for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], b[j][a]); hi[a] = max(hi[a], b[j][a] + 1); }
    }
  }
  block_bbox(lo, hi, s_red, t);
  STAMP(1);
  if (!tile_finish(v, t) || nv > NVPT * nth) return false;
  const bool clear = tile_is_clear(v, t, s_near);   // uniform
  STAMP(2);
  if (!clear) {
#pragma unroll 4
    for (int i = tid; i < t.vol; i += nth) mt[i] = tile_mask(v, t, i);
    __syncthreads();
  }
  STAMP(3);
#pragma unroll
  for (int j = 0; j < NVPT; j++) {
    vs[j].adm = 0; vs[j].base = 0;
    if (live[j]) tile_stencil(t, clear ? nullptr : mt, p[j][0], p[j][1], p[j][2], b[j], vs[j]);
  }
  STAMP(4);
  return true;
}

__global__ __launch_bounds__(256) void ibm_spread_cell_kernel(LatView v, int nv, const double *px, const double *py, const double *pz,
                                                              double *fx, double *fy, double *fz, const double *rx, const double *ry, const double *rz,
                                                              double *F, int limit_on, double f_limit, int xcd_ranges, const int *tag, const unsigned char *vdead) {
  __shared__ double tile[TILE_CAP];
  __shared__ unsigned char mt[TILE_CAP];
  __shared__ int s_red[6 * MAXW], s_near;
  const int tid = threadIdx.x, nth = blockDim.x;
  PH_INIT
  const int cell = xcd_ranges ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int state = tag[cell];
  if (state == 1) return;                                        // the cell is gone
  STAMP(0);
  const unsigned char *dead = state == 2 ? vdead : nullptr;      // incomplete: skip its removed particles
  const long base = (long)cell * nv;
  const bool in_regs = nv <= NVPT * nth;
  // forces of this thread's vertices: the loads are issued here, together with the position loads of the prologue (nothing
  // is stored in between, so both travel in one round trip); the FORCE_LIMIT cap follows once the tile is known
  double f[NVPT][3];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < NVPT; j++) {
      const int i = tid + j * nth;
      f[j][0] = f[j][1] = f[j][2] = 0.0;
      if (i < nv) { f[j][0] = fx[base + i]; f[j][1] = fy[base + i]; f[j][2] = fz[base + i]; }
    }
  }
  Tile t; VStencil vs[NVPT];
  bool tiled = cell_prologue(v, nv, base, px, py, pz, dead, s_red, &s_near, mt, t, vs PH_ARG);
  // FORCE_LIMIT cap, core/hemoCellParticleField.cpp:848-852 (mutates sv.force)
  if (limit_on) {
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < NVPT; j++) {
        const int i = tid + j * nth;
        if (i < nv) {
          const double mag = sqrt((f[j][0] * f[j][0] + f[j][1] * f[j][1]) + f[j][2] * f[j][2]);
          if (mag > f_limit) { const double sc = f_limit / mag; f[j][0] *= sc; f[j][1] *= sc; f[j][2] *= sc; fx[base + i] = f[j][0]; fy[base + i] = f[j][1]; fz[base + i] = f[j][2]; }
        }
      }
    } else {
      for (int i = tid; i < nv; i += nth) {
        const double f0 = fx[base + i], f1 = fy[base + i], f2 = fz[base + i];
        const double mag = sqrt((f0 * f0 + f1 * f1) + f2 * f2);
        if (mag > f_limit) { const double sc = f_limit / mag; fx[base + i] = f0 * sc; fy[base + i] = f1 * sc; fz[base + i] = f2 * sc; }
      }
    }
  }
  if (!tiled) {
    // cell larger than the tile (or mesh larger than the register budget): direct global atomics
    for (int i = tid; i < nv; i += nth) {
      if (dead && dead[base + i]) continue;
      Stencil s;
      phi2_stencil(v, px[base + i], py[base + i], pz[base + i], s);
      const double f0 = (rx ? rx[base + i] : 0.0) + fx[base + i], f1 = (ry ? ry[base + i] : 0.0) + fy[base + i], f2 = (rz ? rz[base + i] : 0.0) + fz[base + i];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (s.node[k] < 0) continue;
        v.dirty[s.node[k] >> 4] = v.epoch;
        unsafeAtomicAdd(&F[s.node[k]], f0 * s.w[k]);
        unsafeAtomicAdd(&F[v.npad + s.node[k]], f1 * s.w[k]);
        unsafeAtomicAdd(&F[2 * v.npad + s.node[k]], f2 * s.w[k]);
      }
    }
    return;
  }
  // One force component at a time on the LDS tile (ds_add_f64), then one fp64 atomic to HBM per touched node.  (All three
  // components in one pass over compacted per-node accumulators -- the interpolation's node list -- halves a workgroup's
  // own time and LOSES 11-16 % of the kernel's: the atomics then leave in one burst, and it is their rate at the memory side
  // that bounds the kernel)
  const int sy = t.e[2], sx = t.e[1] * t.e[2];
  for (int i = tid; i < t.vol; i += nth) tile[i] = 0.0;   // once: the flush below leaves the tile zeroed for the next component
  __syncthreads();
  STAMP(5);
#pragma unroll
  for (int comp = 0; comp < 3; comp++) {
    const double *rc = comp == 0 ? rx : comp == 1 ? ry : rz;
    double *Fc = F + (long)comp * v.npad;
#pragma unroll
    for (int j = 0; j < NVPT; j++) {
      const int i = tid + j * nth;
      if (i >= nv || !vs[j].adm) continue;
      const double fv = (rc ? rc[base + i] : 0.0) + f[j][comp];   // force_repulsion + force, :857-859
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (vs[j].adm & (1u << k)) atomicAdd(&tile[vs[j].base + (k >> 2) * sx + ((k >> 1) & 1) * sy + (k & 1)], fv * vs[j].w[k]);
    }
    __syncthreads();
    STAMP(6 + 2 * comp);
    for (int i = tid; i < t.vol; i += nth) {
      const double val = tile[i];
      if (val != 0.0) {
        if (comp < 2) tile[i] = 0.0;
        int lx, ly, lz; const long node = tile_node(v, t, i, lx, ly, lz); v.dirty[node >> 4] = v.epoch;
        unsafeAtomicAdd(&Fc[node], val);
      }
    }
    if (comp < 2) __syncthreads();
    STAMP(7 + 2 * comp);
  }
}

__global__ __launch_bounds__(256) void ibm_interpolate_cell_kernel(LatView v, PopView pv, int nv, const double *px, const double *py,
                                                                   const double *pz, double *vx, double *vy, double *vz, const int *slots, int xcd_ranges,
                                                                   const int *tag, const unsigned char *vdead) {
  // 54 KB in all, so that three workgroups share a CU: 16-bit slots, and the node list reuses the mask tile
  // (the mask is only read while the stencils are formed)
  constexpr unsigned short FREE = 0xFFFF, MARK = 0xFFFE;
  __shared__ unsigned short slot[TILE_CAP];
  __shared__ __attribute__((aligned(16))) unsigned char raw[TILE_CAP > 2 * NODE_CAP ? TILE_CAP : 2 * NODE_CAP];
  unsigned char *mt = raw; unsigned short *list = reinterpret_cast<unsigned short *>(raw);
  __shared__ double ux[NODE_CAP], uy[NODE_CAP], uz[NODE_CAP];
  __shared__ int s_red[6 * MAXW], s_count, s_near;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int cell = slots ? slots[blockIdx.x] : (xcd_ranges ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x);   // slots: only the listed cells of the type
  const int state = tag[cell];
  if (state == 1) return;
  const unsigned char *dead = state == 2 ? vdead : nullptr;
  const long base = (long)cell * nv;
  Tile t; VStencil vs[NVPT];
  PH_INIT
  bool tiled = cell_prologue(v, nv, base, px, py, pz, dead, s_red, &s_near, mt, t, vs PH_ARG);
  const int sy = t.e[2], sx = t.e[1] * t.e[2];
  if (tiled) {
    for (int i = tid; i < t.vol; i += nth) slot[i] = FREE;
    if (tid == 0) s_count = 0;
    __syncthreads();   // also: every thread is done reading mt, list may overwrite it
#pragma unroll
    for (int j = 0; j < NVPT; j++)   // mark the admitted nodes
#pragma unroll
      for (int k = 0; k < 8; k++) if (vs[j].adm & (1u << k)) slot[vs[j].base + (k >> 2) * sx + ((k >> 1) & 1) * sy + (k & 1)] = MARK;
    __syncthreads();
    for (int i0 = 0; i0 < t.vol; i0 += nth) {   // compact: one LDS atomic per wave, ranks within the wave from its ballot
      const int i = i0 + tid;
      const bool marked = i < t.vol && slot[i] == MARK;
      const unsigned long long b = __ballot(marked);
      int first = 0;
      if ((tid & 63) == 0 && b) first = atomicAdd(&s_count, (int)__popcll(b));
      first = __shfl(first, 0);
      if (marked) {
        const int n = first + (int)__popcll(b & ((1ull << (tid & 63)) - 1ull));
        if (n < NODE_CAP) { slot[i] = (unsigned short)n; list[n] = (unsigned short)i; }
      }
    }
    __syncthreads();
    STAMP(12);
    if (s_count > NODE_CAP) tiled = false;   // uniform: s_count is shared
  }
  if (tiled) {
    const int n = s_count;
    for (int k = tid; k < n; k += nth) {       // node velocity once per node
      int lx, ly, lz;
      const long node = tile_node(v, t, list[k], lx, ly, lz);
      double u[3];
      node_velocity(v, pv, lx, ly, lz, node, u);
      ux[k] = u[0]; uy[k] = u[1]; uz[k] = u[2];
    }
    __syncthreads();
    STAMP(13);
#pragma unroll
    for (int j = 0; j < NVPT; j++) {
      const int i = tid + j * nth;
      if (i >= nv) continue;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (!(vs[j].adm & (1u << k))) continue;
        const int q = slot[vs[j].base + (k >> 2) * sx + ((k >> 1) & 1) * sy + (k & 1)];
        a0 += (ux[q] * vs[j].w[k]); a1 += (uy[q] * vs[j].w[k]); a2 += (uz[q] * vs[j].w[k]);
      }
      vx[base + i] = a0; vy[base + i] = a1; vz[base + i] = a2;
    }
    STAMP(14);
    return;
  }
  for (int i = tid; i < nv; i += nth) {   // fallback: per-vertex gathers
    if (dead && dead[base + i]) continue;
    Stencil s;
    phi2_stencil(v, px[base + i], py[base + i], pz[base + i], s);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (s.node[k] < 0) continue;
      double u[3];
      node_velocity(v, pv, s.lx[k], s.ly[k], s.lz[k], s.node[k], u);
      a0 += (u[0] * s.w[k]); a1 += (u[1] * s.w[k]); a2 += (u[2] * s.w[k]);
    }
    vx[base + i] = a0; vy[base + i] = a1; vz[base + i] = a2;
  }
}

}  // namespace

static int g_ibm_per_vertex = 0;  // 1: one thread per vertex with direct global atomics (kept for A/B and as reference)
extern "C" int hc_debug_ibm_per_vertex(int on) { g_ibm_per_vertex = on; return HC_OK; }
static int g_reproducible = -1;   // -1: not chosen yet (HEMOCELL_REPRODUCIBLE_SPREAD decides at first use)
static bool reproducible() {
  if (g_reproducible < 0) { const char *e = std::getenv("HEMOCELL_REPRODUCIBLE_SPREAD"); g_reproducible = (e && *e && *e != '0') ? 1 : 0; }
  return g_reproducible == 1;
}
extern "C" int hc_set_reproducible_spread(int on) { g_reproducible = on ? 1 : 0; return HC_OK; }

// the gather form of the spread: entries -> stable sort by node -> one sequential sum per node
static int spread_reproducible(hc_cells *C, int force_limit) {
  const LatView v = make_view(C->L);
  HC_REQUIRE(v.npad * 3 < 0xffffffffL && C->nverts * 8 < 0x7fffffffL, "reproducible spread: lattice or vertex count beyond its 32-bit keys");
  const long n_e = C->nverts * 8;
  if (n_e > C->det_cap) {
    HC_HIP(hipDeviceSynchronize());
    for (int k = 0; k < 2; k++) { if (C->det_keys[k]) HC_HIP(hipFree(C->det_keys[k])); if (C->det_vals[k]) HC_HIP(hipFree(C->det_vals[k])); C->det_keys[k] = nullptr; C->det_vals[k] = nullptr; }
    for (int k = 0; k < 3; k++) { if (C->det_val[k]) HC_HIP(hipFree(C->det_val[k])); C->det_val[k] = nullptr; }
    if (C->det_tmp) HC_HIP(hipFree(C->det_tmp));
    C->det_tmp = nullptr; C->det_tmp_bytes = 0;
    C->det_cap = n_e + n_e / 4 + 4096;
    for (int k = 0; k < 2; k++) { HC_HIP(hipMalloc((void **)&C->det_keys[k], C->det_cap * sizeof(unsigned int))); HC_HIP(hipMalloc((void **)&C->det_vals[k], C->det_cap * sizeof(int))); }
    for (int k = 0; k < 3; k++) HC_HIP(hipMalloc((void **)&C->det_val[k], C->det_cap * sizeof(double)));
    HC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, C->det_tmp_bytes, C->det_keys[0], C->det_keys[1], C->det_vals[0], C->det_vals[1], (int)C->det_cap, 0, 32, hc::stream()));
    HC_HIP(hipMalloc(&C->det_tmp, C->det_tmp_bytes));
  }
  // cell slots in ascending cell id, type by type: the canonical order of the entries
  std::vector<int> order((size_t)0);
  std::vector<long> obase((size_t)C->ntypes, 0);
  for (int t = 0; t < C->ntypes; t++) {
    const size_t nc = (size_t)C->ncells[t], o = order.size();
    obase[(size_t)t] = (long)o;
    order.resize(o + nc);
    std::iota(order.begin() + (long)o, order.end(), 0);
    const std::vector<long> &ids = C->hids[t];
    std::stable_sort(order.begin() + (long)o, order.end(), [&](int a, int b) { return ids[(size_t)a] < ids[(size_t)b]; });
  }
  int *d_order = nullptr;
  int rc = stage_ints(C, 2, &d_order, order.data(), (int)order.size()); if (rc != HC_OK) return rc;
  long ebase = 0;
  for (int t = 0; t < C->ntypes; t++) {
    const int nv = C->types[t]->host.nv;
    const long n = C->ncells[t] * nv, f = C->first[t];
    if (n == 0) continue;
    const bool rep = C->rep_on();
    hipLaunchKernelGGL(spread_emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, nv, n, (const int *)(d_order + obase[(size_t)t]), ebase,
                       (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f), C->frc[0] + f, C->frc[1] + f, C->frc[2] + f,
                       rep ? (const double *)(C->rep[0] + f) : nullptr, rep ? (const double *)(C->rep[1] + f) : nullptr, rep ? (const double *)(C->rep[2] + f) : nullptr,
                       force_limit, C->P.f_limit, (const int *)(C->d_tag + C->cell0[t]), (const unsigned char *)(C->d_vdead + f), C->det_keys[0], C->det_vals[0],
                       C->det_val[0], C->det_val[1], C->det_val[2]);
    HC_HIP(hipGetLastError());
    ebase += n;
  }
  // only the bits a node index needs: 2^bits > npad, so the low bits of an invalid key (all ones) still sort behind every node
  int bits = 1;
  while (bits < 32 && (1L << bits) <= v.npad) bits++;
  size_t tmp = C->det_tmp_bytes;
  HC_HIP(hipcub::DeviceRadixSort::SortPairs(C->det_tmp, tmp, C->det_keys[0], C->det_keys[1], C->det_vals[0], C->det_vals[1], (int)n_e, 0, bits, hc::stream()));
  hipLaunchKernelGGL(spread_gather_kernel, dim3((unsigned)((n_e + 255) / 256)), dim3(256), 0, hc::stream(), v, n_e, (const unsigned int *)C->det_keys[1],
                     (const int *)C->det_vals[1], (const double *)C->det_val[0], (const double *)C->det_val[1], (const double *)C->det_val[2], C->L->force[C->L->fcur]);
  HC_HIP(hipGetLastError());
  return HC_OK;
}
#ifdef HC_IBM_PHASE_TIMES
extern "C" int hc_debug_phase_times(double *out) {
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return HC_ERR_HIP;
  for (int k = 0; k < 32; k++) out[k] = (double)h[k];
  std::memset(h, 0, sizeof(h));
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h)) != hipSuccess) return HC_ERR_HIP;
  return HC_OK;
}
#endif

extern "C" {

int hcp_spread(hc_cells *C, int force_limit) {
  HC_REQUIRE(C, "hcp_spread: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  hc::ProfScope prof(hc::PK_SPREAD);
  if (reproducible()) return spread_reproducible(C, force_limit);
  const LatView v = make_view(C->L);
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    const int nv = C->types[t]->host.nv;
    const double *rp[3] = {C->rep_on() ? C->rep[0] + f : nullptr, C->rep_on() ? C->rep[1] + f : nullptr, C->rep_on() ? C->rep[2] + f : nullptr};
    if (g_ibm_per_vertex)
      hipLaunchKernelGGL(ibm_spread_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, n,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->frc[0] + f, C->frc[1] + f, C->frc[2] + f, rp[0], rp[1], rp[2], C->L->force[C->L->fcur], force_limit, C->P.f_limit,
                         (const int *)(C->d_vert_cell + f), (const int *)C->d_tag, (const unsigned char *)(C->d_vdead + f));
    else {
      hipLaunchKernelGGL(ibm_spread_cell_kernel, dim3((unsigned)C->ncells[t]), dim3(nv > 128 ? 256 : 128), 0, hc::stream(), v, nv,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->frc[0] + f, C->frc[1] + f, C->frc[2] + f, rp[0], rp[1], rp[2], C->L->force[C->L->fcur], force_limit, C->P.f_limit, 1,
                         (const int *)(C->d_tag + C->cell0[t]), (const unsigned char *)(C->d_vdead + f));
    }
    HC_HIP(hipGetLastError());
  }
  return HC_OK;
}

int hcp_interpolate(hc_cells *C) {
  HC_REQUIRE(C, "hcp_interpolate: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  hc::ProfScope prof(hc::PK_INTERP);
  const hc_lattice *L = C->L;
  const LatView v = make_view(L);
  // state after hcl_step_end: f[cur] holds the populations just written, force[(fcur+2)%3] the force they were collided with
  PopView pv{L->f[L->cur], L->force[(L->fcur + 2) % 3], L->body[0], L->body[1], L->body[2], (long)L->qstride, L->regions};
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    const int nv = C->types[t]->host.nv;
    if (g_ibm_per_vertex)
      hipLaunchKernelGGL(ibm_interpolate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, pv, n,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->vel[0] + f, C->vel[1] + f, C->vel[2] + f, (const int *)(C->d_vert_cell + f), (const int *)C->d_tag,
                         (const unsigned char *)(C->d_vdead + f));
    else {
      hipLaunchKernelGGL(ibm_interpolate_cell_kernel, dim3((unsigned)C->ncells[t]), dim3(nv > 128 ? 256 : 128), 0, hc::stream(), v, pv, nv,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->vel[0] + f, C->vel[1] + f, C->vel[2] + f, (const int *)nullptr, 1, (const int *)(C->d_tag + C->cell0[t]),
                         (const unsigned char *)(C->d_vdead + f));
    }
    HC_HIP(hipGetLastError());
  }
  return HC_OK;
}

// the message of a velocity update (slab.hip): u = j/rho + F/2 on this slab's face plane `side`, post-stream state, packed as
// [3][ny*nz] for the neighbour on that side.  The face plane and the plane behind it have been collided and the halo plane
// in front of it holds the neighbour's crossing populations.
int hcl_face_velocity_pack(hc_lattice *L, int side, double *dev_buf) {
  HC_REQUIRE(L && dev_buf && (side == 0 || side == 1), "hcl_face_velocity_pack: bad arguments");
  LatView v = make_view(L);
  v.halo_u[0] = v.halo_u[1] = nullptr;   // own planes only: nothing here reads a halo velocity
  PopView pv{L->f[L->cur], L->force[(L->fcur + 2) % 3], L->body[0], L->body[1], L->body[2], (long)L->qstride, L->regions};
  hipLaunchKernelGGL(face_velocity_kernel, dim3((unsigned)((L->plane + 255) / 256), 1, 1), dim3(256), 0, hc::stream(), v, pv, side == 0 ? 0 : L->nx - 1, dev_buf, 0, (double *)nullptr);
  HC_HIP(hipGetLastError());
  return HC_OK;
}
// both face planes in one launch (either buffer may be null)
int hcl_face_velocity_pack_both(hc_lattice *L, double *dev_lo, double *dev_hi) {
  HC_REQUIRE(L, "hcl_face_velocity_pack_both: null lattice");
  if (!dev_lo && !dev_hi) return HC_OK;
  if (!dev_lo || !dev_hi) return hcl_face_velocity_pack(L, dev_lo ? 0 : 1, dev_lo ? dev_lo : dev_hi);
  LatView v = make_view(L);
  v.halo_u[0] = v.halo_u[1] = nullptr;
  PopView pv{L->f[L->cur], L->force[(L->fcur + 2) % 3], L->body[0], L->body[1], L->body[2], (long)L->qstride, L->regions};
  hipLaunchKernelGGL(face_velocity_kernel, dim3((unsigned)((L->plane + 255) / 256), 2, 1), dim3(256), 0, hc::stream(), v, pv, 0, dev_lo, L->nx - 1, dev_hi);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

// inspection: the same plane of velocities on the host, [3][ny*nz] (tests; the message itself never touches the host)
int hcl_download_face_velocity(hc_lattice *L, int side, double *host_u) {
  HC_REQUIRE(L && host_u && (side == 0 || side == 1), "hcl_download_face_velocity: bad arguments");
  double *d = nullptr;
  HC_HIP(hipMalloc((void **)&d, 3 * L->plane * sizeof(double)));
  int rc = hcl_face_velocity_pack(L, side, d);
  if (rc == HC_OK && hipMemcpyAsync(host_u, d, 3 * L->plane * sizeof(double), hipMemcpyDeviceToHost, hc::stream()) != hipSuccess) { hc::set_error("hcl_download_face_velocity: copy failed"); rc = HC_ERR_HIP; }
  if (rc == HC_OK && hipStreamSynchronize(hc::stream()) != hipSuccess) { hc::set_error("hcl_download_face_velocity: synchronise failed"); rc = HC_ERR_HIP; }
  hipFree(d);
  return rc;
}

int hcp_interpolate_cells(hc_cells *C, int type, const int *slots, int n) { return hcc::interpolate_cells_staged(C, type, slots, n, 1); }

}  // extern "C"

// which: the staging slot the list travels through (a caller that issues several lists in one step gives each its own, so that
// staging the next one does not wait for the copy of the previous one, which may sit behind a whole collide in the stream)
int hcc::interpolate_cells_staged(hc_cells *C, int type, const int *slots, int n, int which) {
  HC_REQUIRE(C && type >= 0 && type < C->ntypes && n >= 0 && which >= 0 && which < 19, "hcp_interpolate_cells: bad arguments");
  if (n == 0) return HC_OK;
  HC_REQUIRE(slots, "hcp_interpolate_cells: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  for (int i = 0; i < n; i++) HC_REQUIRE(slots[i] >= 0 && slots[i] < C->ncells[type], "hcp_interpolate_cells: slot out of range");
  int *d_slots = nullptr;
  rc = stage_ints(C, which, &d_slots, slots, n); if (rc != HC_OK) return rc;
  hc::ProfScope prof(hc::PK_INTERP);
  const hc_lattice *L = C->L;
  const LatView v = make_view(L);
  PopView pv{L->f[L->cur], L->force[(L->fcur + 2) % 3], L->body[0], L->body[1], L->body[2], (long)L->qstride, L->regions};
  const long f = C->first[type];
  const int nv = C->types[type]->host.nv;
  hipLaunchKernelGGL(ibm_interpolate_cell_kernel, dim3((unsigned)n), dim3(nv > 128 ? 256 : 128), 0, hc::stream(), v, pv, nv,
                     (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                     C->vel[0] + f, C->vel[1] + f, C->vel[2] + f, (const int *)d_slots, 0, (const int *)(C->d_tag + C->cell0[type]),
                     (const unsigned char *)(C->d_vdead + f));
  HC_HIP(hipGetLastError());
  return HC_OK;
}
