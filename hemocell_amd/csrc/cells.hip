// Membrane vertices on the GPU: phi2 immersed-boundary spread / interpolate,
// Euler advance, rbcHighOrderModel and pltSimpleModel forces.
//
// Replaces (file:line in the HemoCell tree):
//   core/immersedBoundaryMethod.h:62-138          interpolationCoefficientsPhi2
//   core/hemoCellParticleField.cpp:841-863        spreadParticleForce
//   core/hemoCellParticleField.cpp:819-839        interpolateFluidVelocity
//   core/hemoCellParticle.h:188-203, core/hemoCellParticleField.cpp:566-588   advance
//   core/hemoCellParticleField.cpp:633-675        applyConstitutiveModel
//   mechanics/rbcHighOrderModel.cpp:38-207, mechanics/pltSimpleModel.cpp:44-208
//
// Layout (HBM): vertices are a structure of arrays pos/vel/frc[3][n], cell
// major (a cell's nv vertices are contiguous, cells of one type contiguous), so
// a wavefront touches 64 consecutive doubles per component and a mechanics
// workgroup stages one whole cell in LDS with coalesced loads.
//
// The membrane models are evaluated in GATHER form: one workgroup per cell,
// vertex positions in LDS, each vertex sums the contributions of its incident
// triangles / ring / edges in exactly the order in which the reference's
// scatter loops would have added them (triangle loop, volume loop, vertex
// loop, edge loop), so no atomics are needed and the result is reproducible
// and (with -ffp-contract=off) bit-identical to the scatter form.
#include "common.h"
#include "mesh.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cmath>
#include <cstring>

using namespace hc;

struct hc_celltype {
  CellTables host;
  int *d_tri = nullptr, *d_edge = nullptr, *d_ebt = nullptr, *d_ebo = nullptr, *d_iedge = nullptr;
  int *d_vtri = nullptr, *d_vtri_k = nullptr, *d_vedge = nullptr, *d_vedge_s = nullptr, *d_bsrc = nullptr;
  int *d_vouter = nullptr, *d_vinner = nullptr, *d_vinner_s = nullptr, *d_ring = nullptr, *d_nring = nullptr;
  double *d_tri_area_eq = nullptr, *d_edge_len_eq = nullptr, *d_edge_angle_eq = nullptr, *d_patch_eq = nullptr,
         *d_iedge_len_eq = nullptr;
};

struct hc_cells {
  hc_lattice *L = nullptr;
  hc_params P;
  int ntypes = 0;
  hc_celltype *types[8];
  int timescale[8];
  std::vector<double> hpos[8];   // host staging per type: [ncells*nv][3]
  std::vector<double> hvel[8], hfrc[8];
  std::vector<long> hids[8];
  bool host_dirty = false;       // host staging newer than device
  long nverts = 0, cap = 0;      // live vertices (all types); allocated vertex capacity
  long ncells[8] = {0};
  long capc[8] = {0};            // per-type capacity in cells (device regions are fixed-size per type)
  long first[8] = {0};           // first vertex of each type's region on the device
  long cell0[8] = {0};           // first cell slot of each type's region
  double *pos[3] = {nullptr, nullptr, nullptr}, *vel[3] = {nullptr, nullptr, nullptr}, *frc[3] = {nullptr, nullptr, nullptr};
  int *d_tag = nullptr;          // per-cell deletion tags (all types, slot order)
  long tag_cap = 0;
  int *h_ntag = nullptr;         // pinned host copy of the tag counter
  int *d_ntag = nullptr;         // device counter of tagged cells
  int *d_vert_cell = nullptr;    // [cap] cell slot of every vertex
  // vertex-vertex repulsion (core/hemoCellParticleField.cpp:677-743); arrays exist only once it is enabled
  double *rep[3] = {nullptr, nullptr, nullptr};
  int rep_enabled = 0, rep_timescale = 1; double rep_const = 0, rep_cutoff = 0;
  // boundary particles (core/hemoCellParticleField.cpp:865-918): flag map of the wall nodes that repel vertices
  int brep_enabled = 0, brep_timescale = 1; double brep_const = 0, brep_cutoff = 0; uint8_t *d_bflag = nullptr;
  bool rep_on() const { return rep_enabled || brep_enabled; }
  unsigned int *d_keys[2] = {nullptr, nullptr}; int *d_vals[2] = {nullptr, nullptr}; void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0; long sort_cap = 0;
  int *d_iscratch[2] = {nullptr, nullptr};   // staged slot lists of the envelope exchange (stream ordered, no sync)
  size_t iscratch_cap[2] = {0, 0};
  long n_deleted = 0;
};

namespace {

// ----------------------------------------------------------------------------
// lattice view for the IBM kernels
struct LatView {
  const uint8_t *mask;
  int nx, ny, nz, plane; long npad;
  int x0;                 // global x of local plane 0
  int wrap_x, halo_x;     // single periodic slab: wrap; multi slab: one halo plane is addressable
  int per_y, per_z;
  int nx_global;
  uint8_t *dirty; uint8_t epoch;   // dirty map of the force buffer spread adds to (see common.h)
};

LatView make_view(const hc_lattice *L) {
  LatView v;
  v.mask = L->mask; v.nx = L->nx; v.ny = L->ny; v.nz = L->nz; v.plane = (int)L->plane; v.npad = (long)L->npad;
  v.x0 = L->x0; v.wrap_x = (L->n_slabs == 1 && L->periodic[0]) ? 1 : 0; v.halo_x = L->n_slabs > 1 ? 1 : 0;
  v.per_y = L->periodic[1]; v.per_z = L->periodic[2]; v.nx_global = L->nx_global;
  v.dirty = L->fdirty[L->fcur]; v.epoch = L->fepoch[L->fcur];
  return v;
}

__device__ __forceinline__ long pmod(long a, long n) { long r = a % n; return r < 0 ? r + n : r; }

// phi2 (core/immersedBoundaryMethod.h:37-41)
__device__ __forceinline__ double phi2(double x) { x = fabs(x); x = 1.0 - x; return x > 0.0 ? x : 0.0; }

struct Stencil {
  long node[8];     // padded-lattice element index, -1 when not admitted
  double w[8];      // normalised weights
  int lx[8], ly[8], lz[8];  // local (wrapped) coordinates of the node, for the population gather
};

// interpolationCoefficientsPhi2 (core/immersedBoundaryMethod.h:62-138).  Per axis only the pair
// {centre-1, centre} (x < centre) or {centre, centre+1} can carry a non-zero tent weight, and visiting
// the 2x2x2 pairs in ascending offset order is the reference's 27-node loop with its zero-weight skips.
__device__ __forceinline__ long nearest_node(double x) { return (long)floor(x + 0.5); }

__device__ __forceinline__ void phi2_stencil(const LatView &v, double px, double py, double pz, Stencil &s) {
  // weights are formed in GLOBAL coordinates (identical bits on every slab that holds a copy of the
  // vertex); only the node index is made slab-local.  plint(x+0.5) of the reference (:86) truncates,
  // which equals floor on the block-relative coordinates (>= 0) it is applied to; floor is used so that
  // a periodic image at negative x picks the same nodes as its wrapped position.
  const double p[3] = {px, py, pz};
  long c[3]; int d0[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { c[a] = nearest_node(p[a]); d0[a] = (p[a] < (double)c[a]) ? -1 : 0; }
  double total = 0.0;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int idx = i * 4 + j * 2 + k;
        const long gx = c[0] + d0[0] + i, gy = c[1] + d0[1] + j, gz = c[2] + d0[2] + k;
        long lx = gx - v.x0, ly = gy, lz = gz;
        bool ok = true;
        if (v.wrap_x) lx = pmod(lx, v.nx);
        else if (v.halo_x) ok = ok && (lx >= -HALO && lx < v.nx + HALO);
        else ok = ok && (lx >= 0 && lx < v.nx);
        if (gy < 0 || gy >= v.ny) { if (v.per_y) ly = pmod(gy, v.ny); else ok = false; }
        if (gz < 0 || gz >= v.nz) { if (v.per_z) lz = pmod(gz, v.nz); else ok = false; }
        double weight = 0.0; long node = -1;
        if (ok) {
          weight = phi2(p[0] - (double)gx) * phi2(p[1] - (double)gy) * phi2(p[2] - (double)gz);
          if (weight != 0.0) {
            node = (lx + HALO) * (long)v.plane + ly * v.nz + lz;
            if (v.mask[node] != 0) node = -1;
          }
        }
        if (node >= 0) total += weight;
        s.node[idx] = node; s.w[idx] = weight; s.lx[idx] = (int)lx; s.ly[idx] = (int)ly; s.lz[idx] = (int)lz;
      }
  const double coeff = 1.0 / total;
#pragma unroll
  for (int idx = 0; idx < 8; idx++) s.w[idx] *= coeff;
}

// ----------------------------------------------------------------------------
// spread
__global__ __launch_bounds__(256) void ibm_spread_kernel(LatView v, long n, const double *px, const double *py, const double *pz,
                                                         double *fx, double *fy, double *fz, const double *rx, const double *ry, const double *rz,
                                                         double *F, int limit_on, double f_limit) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
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
// interpolate: v = sum_j w_j * (j/rho + F/2)(node_j) on the post-stream state
struct PopView {
  const double *f; const double *F; double bx, by, bz;
};

__device__ __forceinline__ void node_velocity(const LatView &v, const PopView &pv, int lx, int ly, int lz, long node, double u[3]) {
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
    const double fq = ok ? pv.f[(long)Q * v.npad + node + off] : 0.0;                 \
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
  u[0] = jx * invRho + (pv.bx + pv.F[node]) / 2.0;
  u[1] = jy * invRho + (pv.by + pv.F[v.npad + node]) / 2.0;
  u[2] = jz * invRho + (pv.bz + pv.F[2 * v.npad + node]) / 2.0;
}

__global__ __launch_bounds__(256) void ibm_interpolate_kernel(LatView v, PopView pv, long n, const double *px, const double *py,
                                                              const double *pz, double *vx, double *vy, double *vz) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
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

// compact per-vertex stencil kept in registers across the passes of the cell kernels
struct VStencil { double w[8]; int base; unsigned adm; };   // base = tile index of the lowest corner; adm = admitted-node bits

constexpr int NVPT = 3;        // vertices per thread held in registers (642 vertices / 256 threads)
constexpr int MAXW = 4;        // waves per workgroup

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
        const bool adm = (weight != 0.0) && (mt[o.base + i * sx + j * sy + k] == 0);
        if (adm) { total += weight; o.adm |= 1u << idx; }
        o.w[idx] = adm ? weight : 0.0;
      }
  const double coeff = 1.0 / total;
#pragma unroll
  for (int idx = 0; idx < 8; idx++) o.w[idx] *= coeff;
}

// shared prologue of the two cell kernels: positions -> registers, tile, mask tile, stencils.
// returns false (uniformly) when the cell does not fit the tile and the caller must take the fallback path
__device__ __forceinline__ bool cell_prologue(const LatView &v, int nv, long base, const double *px, const double *py, const double *pz,
                                              int *s_red, unsigned char *mt, Tile &t, VStencil vs[NVPT]) {
  const int tid = threadIdx.x, nth = blockDim.x;
  double p[NVPT][3]; int b[NVPT][3];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
  for (int j = 0; j < NVPT; j++) {
    const int i = tid + j * nth;
    if (i < nv) {
      p[j][0] = px[base + i]; p[j][1] = py[base + i]; p[j][2] = pz[base + i];
      stencil_base(v, p[j][0], p[j][1], p[j][2], b[j]);
#pragma unroll
      for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], b[j][a]); hi[a] = max(hi[a], b[j][a] + 1); }
    }
  }
  block_bbox(lo, hi, s_red, t);
  if (!tile_finish(v, t) || nv > NVPT * nth) return false;
#pragma unroll 4
  for (int i = tid; i < t.vol; i += nth) mt[i] = tile_mask(v, t, i);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NVPT; j++) {
    vs[j].adm = 0;
    if (tid + j * nth < nv) tile_stencil(t, mt, p[j][0], p[j][1], p[j][2], b[j], vs[j]);
  }
  return true;
}

__global__ __launch_bounds__(256) void ibm_spread_cell_kernel(LatView v, int nv, const double *px, const double *py, const double *pz,
                                                              double *fx, double *fy, double *fz, const double *rx, const double *ry, const double *rz,
                                                              double *F, int limit_on, double f_limit) {
  __shared__ double tile[TILE_CAP];
  __shared__ unsigned char mt[TILE_CAP];
  __shared__ int s_red[6 * MAXW];
  const int tid = threadIdx.x, nth = blockDim.x;
  const long base = (long)blockIdx.x * nv;
  // FORCE_LIMIT cap, core/hemoCellParticleField.cpp:848-852 (mutates sv.force)
  if (limit_on) {
    for (int i = tid; i < nv; i += nth) {
      const double f0 = fx[base + i], f1 = fy[base + i], f2 = fz[base + i];
      const double mag = sqrt((f0 * f0 + f1 * f1) + f2 * f2);
      if (mag > f_limit) { const double sc = f_limit / mag; fx[base + i] = f0 * sc; fy[base + i] = f1 * sc; fz[base + i] = f2 * sc; }
    }
  }
  Tile t; VStencil vs[NVPT];
  if (!cell_prologue(v, nv, base, px, py, pz, s_red, mt, t, vs)) {
    // cell larger than the tile (or mesh larger than the register budget): direct global atomics
    for (int i = tid; i < nv; i += nth) {
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
  const int sy = t.e[2], sx = t.e[1] * t.e[2];
  for (int comp = 0; comp < 3; comp++) {
    const double *fc = comp == 0 ? fx : comp == 1 ? fy : fz;
    const double *rc = comp == 0 ? rx : comp == 1 ? ry : rz;
    double *Fc = F + (long)comp * v.npad;
    for (int i = tid; i < t.vol; i += nth) tile[i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NVPT; j++) {
      const int i = tid + j * nth;
      if (i >= nv) continue;
      const double f = (rc ? rc[base + i] : 0.0) + fc[base + i];   // force_repulsion + force, :857-859
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (vs[j].adm & (1u << k)) atomicAdd(&tile[vs[j].base + (k >> 2) * sx + ((k >> 1) & 1) * sy + (k & 1)], f * vs[j].w[k]);
    }
    __syncthreads();
    for (int i = tid; i < t.vol; i += nth) {
      const double val = tile[i];
      if (val != 0.0) { int lx, ly, lz; const long node = tile_node(v, t, i, lx, ly, lz); v.dirty[node >> 4] = v.epoch; unsafeAtomicAdd(&Fc[node], val); }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void ibm_interpolate_cell_kernel(LatView v, PopView pv, int nv, const double *px, const double *py,
                                                                   const double *pz, double *vx, double *vy, double *vz) {
  // 54 KB in all, so that three workgroups share a CU: 16-bit slots, and the node list reuses the mask tile
  // (the mask is only read while the stencils are formed)
  constexpr unsigned short FREE = 0xFFFF, MARK = 0xFFFE;
  __shared__ unsigned short slot[TILE_CAP];
  __shared__ __attribute__((aligned(16))) unsigned char raw[TILE_CAP > 2 * NODE_CAP ? TILE_CAP : 2 * NODE_CAP];
  unsigned char *mt = raw; unsigned short *list = reinterpret_cast<unsigned short *>(raw);
  __shared__ double ux[NODE_CAP], uy[NODE_CAP], uz[NODE_CAP];
  __shared__ int s_red[6 * MAXW], s_count;
  const int tid = threadIdx.x, nth = blockDim.x;
  const long base = (long)blockIdx.x * nv;
  Tile t; VStencil vs[NVPT];
  bool tiled = cell_prologue(v, nv, base, px, py, pz, s_red, mt, t, vs);
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
    for (int i = tid; i < t.vol; i += nth) {   // compact
      if (slot[i] == MARK) { const int n = atomicAdd(&s_count, 1); if (n < NODE_CAP) { slot[i] = (unsigned short)n; list[n] = (unsigned short)i; } }
    }
    __syncthreads();
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
    return;
  }
  for (int i = tid; i < nv; i += nth) {   // fallback: per-vertex gathers
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

// one workgroup = one cell.  LDS: positions, per-triangle {volume term, area, unit normal, area-force
// magnitude}, per-vertex bending vector (RBC) or per-edge {link, visc, bending} vectors (PLT).
template <int MODEL, bool SEPARATE>
__global__ __launch_bounds__(256) void mechanics_kernel(MechArgs m) {
  extern __shared__ double lds[];
  const int nv = m.nv, nt = m.nt, ne = m.ne;
  double *xs = lds, *ys = xs + nv, *zs = ys + nv;
  double *tV = zs + nv, *tA = tV + nt, *tNx = tA + nt, *tNy = tNx + nt, *tNz = tNy + nt, *tAfm = tNz + nt;
  double *ex = tAfm + nt;  // RBC: B[3][nv]; PLT: edge vectors [9][ne]
  __shared__ double s_volume_force;
  const int tid = threadIdx.x, nth = blockDim.x;
  const long base = (long)blockIdx.x * nv;

  for (int i = tid; i < nv; i += nth) { xs[i] = m.px[base + i]; ys[i] = m.py[base + i]; zs[i] = m.pz[base + i]; }
  __syncthreads();

  // ---- per-triangle quantities (rbcHighOrderModel.cpp:56-98 / pltSimpleModel.cpp:57-99)
  for (int t = tid; t < nt; t += nth) {
    const int i0 = m.tri[3 * t], i1 = m.tri[3 * t + 1], i2 = m.tri[3 * t + 2];
    const double v0x = xs[i0], v0y = ys[i0], v0z = zs[i0], v1x = xs[i1], v1y = ys[i1], v1z = zs[i1], v2x = xs[i2], v2y = ys[i2], v2z = zs[i2];
    const double v210 = v2x * v1y * v0z, v120 = v1x * v2y * v0z, v201 = v2x * v0y * v1z;
    const double v021 = v0x * v2y * v1z, v102 = v1x * v0y * v2z, v012 = v0x * v1y * v2z;
    tV[t] = (-v210 + v120 + v201 - v021 - v102 + v012);
    const double e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z, e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z;
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double nn = norm3(nx, ny, nz);
    double area;
    if (nn != 0.0) { area = 0.5 * nn; nx /= nn; ny /= nn; nz /= nn; } else { area = 0.0; nx = ny = nz = 0.0; }
    tA[t] = area; tNx[t] = nx; tNy[t] = ny; tNz[t] = nz;
    const double aeq = m.tri_area_eq[t];
    const double areaRatio = (area - aeq) / aeq;
    tAfm[t] = m.k_area * (areaRatio + areaRatio / fabs(MaxCellSurfaceAreaChange - areaRatio * areaRatio));
  }
  __syncthreads();
  if (tid == 0) {
    // the reference accumulates the signed-volume terms sequentially in triangle order; do the same so
    // that every copy of a cell (other GPUs, the CPU oracle) gets the same bits
    double volume = 0.0;
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
      const double afm = tAfm[t];
      ACC(1)[0] += afm * (cx - x); ACC(1)[1] += afm * (cy - y); ACC(1)[2] += afm * (cz - z);
    }
    for (int k = 0; k < MD; k++) {  // volume force, triangle order (rbcHighOrderModel.cpp:107-113)
      const int t = m.vtri[MD * i + k];
      if (t < 0) break;
      const double sc = tA[t] / m.area_mean_eq;
      ACC(0)[0] += (volume_force * tNx[t]) * sc; ACC(0)[1] += (volume_force * tNy[t]) * sc; ACC(0)[2] += (volume_force * tNz[t]) * sc;
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

__global__ void add_vertex_force_kernel(int n, const long *idx, const double *f, double *fx, double *fy, double *fz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long v = idx[i];
  fx[v] += f[3 * i]; fy[v] += f[3 * i + 1]; fz[v] += f[3 * i + 2];
}

// ---------------------------------------------------------------------------- multi-slab cell exchange
// per-cell [min_x, max_x, number of vertices whose nearest node lies in this slab]
__global__ __launch_bounds__(256) void cell_extent_kernel(int nv, const double *px, double *out, int x0, int nx) {
  __shared__ double lo[256], hi[256];
  __shared__ int own[256];
  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * nv;
  double a = 1e300, b = -1e300; int o = 0;
  for (int i = tid; i < nv; i += 256) {
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
  if (tid == 0) { out[3 * blockIdx.x] = lo[0]; out[3 * blockIdx.x + 1] = hi[0]; out[3 * blockIdx.x + 2] = (double)own[0]; }
}

struct VertArrays { double *p[3], *v[3], *f[3], *r[3]; };   // r: repulsion force arrays or null

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
                                                           int x0, int nx) {
  const long dst = (long)slots[blockIdx.x] * nv, src = (long)blockIdx.x * nv;
  const bool fresh = is_new[blockIdx.x] != 0;
  for (int i = threadIdx.x; i < nv; i += 256) {
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
  for (int i = threadIdx.x; i < nv; i += 256)
    for (int d = 0; d < 3; d++) {
      a.p[d][dst + i] = a.p[d][src + i]; a.v[d][dst + i] = a.v[d][src + i]; a.f[d][dst + i] = a.f[d][src + i];
      if (a.r[d]) a.r[d][dst + i] = a.r[d][src + i];
    }
}

// ParticleInfo statistics (helper/particleInfo.cpp:30-95): magnitude of v (what 1) or of force + force_repulsion (what 2)
// over the vertices this slab owns (findParticles(localDomain))
__global__ __launch_bounds__(256) void vertex_stats_kernel(long n, int what, int all_owned, int x0, int nx, const double *px, const double *a0,
                                                           const double *a1, const double *a2, const double *r0, const double *r1, const double *r2,
                                                           double *partial, int accumulate) {
  StatAcc acc{1e300, -1e300, 0.0, 0};
  if (accumulate) { const double *o = partial + 4 * blockIdx.x; if (threadIdx.x == 0 && o[3] > 0) { acc.mn = o[0]; acc.mx = o[1]; acc.sum = o[2]; acc.n = (long)o[3]; } }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)STAT_BLOCKS * 256) {
    if (!all_owned) { const long gx = nearest_node(px[i]) - x0; if (gx < 0 || gx >= nx) continue; }
    double v0 = a0[i], v1 = a1[i], v2 = a2[i];
    if (what == 2 && r0) { v0 = v0 + r0[i]; v1 = v1 + r1[i]; v2 = v2 + r2[i]; }
    stat_add(acc, sqrt(v0 * v0 + v1 * v1 + v2 * v2));
  }
  __syncthreads();   // every thread has read the previous partial before it is overwritten
  stat_block_store(acc, partial);
}

__global__ __launch_bounds__(256) void owned_count_kernel(long n, const double *px, int x0, int nx, unsigned long long *count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  int mine = 0;
  if (i < n) { const long gx = nearest_node(px[i]) - x0; mine = (gx >= 0 && gx < nx) ? 1 : 0; }
  const unsigned long long b = __ballot(mine);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

// ---------------------------------------------------------------------------- vertex-vertex repulsion
// applyRepulsionForce (core/hemoCellParticleField.cpp:677-743): vertices are binned by their nearest lattice
// node (update_pg, :137-168); two vertices of DIFFERENT cells in the same or in adjacent bins that are closer
// than r_cutoff repel each other with r_const * (r_cutoff / d) along their separation.  The reference visits a
// same-bin pair twice (its inner loop runs over ordered pairs there), so those pairs count double.  Gather form:
// every vertex sums over the 27 bins around its own; the bins come from a radix sort of (bin, vertex).
__global__ void rep_keys_kernel(LatView v, long n, long first, long packed0, const double *px, const double *py, const double *pz, unsigned int *keys, int *vals) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long lx = nearest_node(px[i]) - v.x0, ly = nearest_node(py[i]), lz = nearest_node(pz[i]);
  bool ok = true;
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
    const int lz = key % v.nz, ly = (key / v.nz) % v.ny, lxp = key / v.plane;   // lxp = padded x
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
                                                           double *rx, double *ry, double *rz, double br_const, double br_cutoff) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
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

__global__ void fill_vert_cell_kernel(long n, int nv, int cell0, long first, int *vert_cell) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  vert_cell[first + i] = cell0 + (int)(i / nv);
}

// ----------------------------------------------------------------------------
template <typename T>
int upload_vec(T **dst, const std::vector<T> &src) {
  *dst = nullptr;
  const size_t n = src.size() ? src.size() : 1;
  HC_HIP(hipMalloc((void **)dst, n * sizeof(T)));
  if (src.size()) HC_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return HC_OK;
}
template <size_t N>
std::vector<int> flatten(const std::vector<std::array<long, N>> &v) {
  std::vector<int> o; o.reserve(v.size() * N);
  for (auto &a : v) for (long x : a) o.push_back((int)x);
  return o;
}

}  // namespace

static int g_ibm_per_vertex = 0;  // 1: one thread per vertex with direct global atomics (kept for A/B and as reference)
extern "C" int hc_debug_ibm_per_vertex(int on) { g_ibm_per_vertex = on; return HC_OK; }

static int free_device_arrays(hc_cells *C) {
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
static int sync_to_device(hc_cells *C) {
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
static int sync_to_host(hc_cells *C) {
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
  return m;
}

static size_t mech_lds_bytes(const CellTables &T) {
  const size_t extra = T.model == HC_MODEL_RBC_HO ? 3 * (size_t)T.nv : 9 * (size_t)T.ne;
  return (3 * (size_t)T.nv + 6 * (size_t)T.nt + extra) * sizeof(double);
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
    return mask[(size_t)(lx + HALO) * L->plane + (size_t)ly * L->nz + lz] != 0;
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
  // vertex_index counts vertices in download order (types packed back to back); device regions have gaps
  std::vector<long> dev_idx((size_t)n);
  for (int i = 0; i < n; i++) {
    long v = vertex_index[i], packed0 = 0; bool found = false;
    for (int t = 0; t < C->ntypes && !found; t++) {
      const long nt = C->ncells[t] * C->types[t]->host.nv;
      if (v >= packed0 && v < packed0 + nt) { dev_idx[(size_t)i] = C->first[t] + (v - packed0); found = true; }
      packed0 += nt;
    }
    HC_REQUIRE(found, "hcp_add_vertex_force: vertex index out of range");
  }
  long *d_idx = nullptr; double *d_f = nullptr;
  HC_HIP(hipMalloc((void **)&d_idx, n * sizeof(long)));
  HC_HIP(hipMalloc((void **)&d_f, 3 * n * sizeof(double)));
  HC_HIP(hipMemcpyAsync(d_idx, dev_idx.data(), n * sizeof(long), hipMemcpyHostToDevice, hc::stream()));
  HC_HIP(hipMemcpyAsync(d_f, f, 3 * n * sizeof(double), hipMemcpyHostToDevice, hc::stream()));
  hipLaunchKernelGGL(add_vertex_force_kernel, dim3((n + 255) / 256), dim3(256), 0, hc::stream(), n, (const long *)d_idx, (const double *)d_f, C->frc[0], C->frc[1], C->frc[2]);
  HC_HIP(hipGetLastError());
  HC_HIP(hipStreamSynchronize(hc::stream()));
  hipFree(d_idx); hipFree(d_f);
  return HC_OK;
}

int hcp_spread(hc_cells *C, int force_limit) {
  HC_REQUIRE(C, "hcp_spread: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  if (C->nverts == 0) return HC_OK;
  hc::ProfScope prof(hc::PK_SPREAD);
  const LatView v = make_view(C->L);
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    const int nv = C->types[t]->host.nv;
    const double *rp[3] = {C->rep_on() ? C->rep[0] + f : nullptr, C->rep_on() ? C->rep[1] + f : nullptr, C->rep_on() ? C->rep[2] + f : nullptr};
    if (g_ibm_per_vertex)
      hipLaunchKernelGGL(ibm_spread_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, n,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->frc[0] + f, C->frc[1] + f, C->frc[2] + f, rp[0], rp[1], rp[2], C->L->force[C->L->fcur], force_limit, C->P.f_limit);
    else
      hipLaunchKernelGGL(ibm_spread_cell_kernel, dim3((unsigned)C->ncells[t]), dim3(nv > 128 ? 256 : 128), 0, hc::stream(), v, nv,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->frc[0] + f, C->frc[1] + f, C->frc[2] + f, rp[0], rp[1], rp[2], C->L->force[C->L->fcur], force_limit, C->P.f_limit);
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
  // state after hcl_step_end: f[cur] holds the populations just written, force[1-fcur] the force they were collided with
  PopView pv{L->f[L->cur], L->force[1 - L->fcur], L->body[0], L->body[1], L->body[2]};
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv, f = C->first[t];
    if (n == 0) continue;
    const int nv = C->types[t]->host.nv;
    if (g_ibm_per_vertex)
      hipLaunchKernelGGL(ibm_interpolate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), v, pv, n,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->vel[0] + f, C->vel[1] + f, C->vel[2] + f);
    else
      hipLaunchKernelGGL(ibm_interpolate_cell_kernel, dim3((unsigned)C->ncells[t]), dim3(nv > 128 ? 256 : 128), 0, hc::stream(), v, pv, nv,
                         (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f),
                         C->vel[0] + f, C->vel[1] + f, C->vel[2] + f);
    HC_HIP(hipGetLastError());
  }
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

int hcp_mechanics_components(hc_cells *C, int type, double *comp) {
  HC_REQUIRE(C && comp && type >= 0 && type < C->ntypes, "hcp_mechanics_components: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long n = C->ncells[type] * C->types[type]->host.nv;
  if (n == 0) return HC_OK;
  double *d = nullptr;
  HC_HIP(hipMalloc((void **)&d, (size_t)(18 * n) * sizeof(double)));
  rc = launch_mechanics(C, type, d);
  if (rc == HC_OK) {
    hipError_t e = hipStreamSynchronize(hc::stream());  // the library stream is non-blocking
    if (e == hipSuccess) e = hipMemcpy(comp, d, (size_t)(18 * n) * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hc::hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
  }
  hipFree(d);
  return rc;
}

int hcp_cell_info(hc_cells *C, int type, double *volume, double *area, double *bbox, double *centroid) {
  HC_REQUIRE(C && volume && area && bbox && centroid && type >= 0 && type < C->ntypes, "hcp_cell_info: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long nc = C->ncells[type];
  if (nc == 0) return HC_OK;
  const CellTables &T = C->types[type]->host;
  double *d = nullptr;
  HC_HIP(hipMalloc((void **)&d, (size_t)(11 * nc) * sizeof(double)));
  const long f = C->first[type];
  hipLaunchKernelGGL(cell_info_kernel, dim3((unsigned)nc), dim3(256), 0, hc::stream(), T.nv, T.nt, (const int *)C->types[type]->d_tri,
                     (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f), d, d + nc, d + 2 * nc, d + 8 * nc);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(hc::stream());
  if (e == hipSuccess) e = hipMemcpy(volume, d, nc * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(area, d + nc, nc * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(bbox, d + 2 * nc, 6 * nc * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(centroid, d + 8 * nc, 3 * nc * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return hc::hip_fail(e, "hcp_cell_info", __FILE__, __LINE__);
  return HC_OK;
}

int hc_iterate(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit, int deletion_check_every) {
  HC_REQUIRE(L && C && iter, "hc_iterate: null pointer");
  HC_REQUIRE(C->L == L, "hc_iterate: cells are bound to a different lattice");
  HC_REQUIRE(L->n_slabs == 1, "hc_iterate: single-slab stepping only; multi-slab runs are driven phase by phase with halo exchange");
  HC_REQUIRE(particle_timescale >= 1 && deletion_check_every >= 1, "hc_iterate: timescales must be >= 1");
  int rc;
  for (int s = 0; s < n; s++) {
    const long it = *iter;
    if (C->rep_enabled && it % C->rep_timescale == 0) { if ((rc = hcp_repulsion(C)) != HC_OK) return rc; }   // core/hemoCell.cpp:307-309
    if (C->brep_enabled && it % C->brep_timescale == 0) { if ((rc = hcp_boundary_repulsion(C)) != HC_OK) return rc; }   // :310-312
    if ((rc = hcp_spread(C, force_limit)) != HC_OK) return rc;                  // :313
    if ((rc = hcl_collide_stream_part(L, 0)) != HC_OK) return rc;               // :317
    hcl_step_end(L);
    if (it % particle_timescale == 0) { if ((rc = hcp_interpolate(C)) != HC_OK) return rc; }   // :327-332
    if ((rc = hcp_advance(C, (it % deletion_check_every) == 0)) != HC_OK) return rc;           // :342
    if ((rc = hcp_mechanics(C, it, 0)) != HC_OK) return rc;                     // :345
    *iter = it + 1;                                                             // :374 (force zeroing is fused into the collide kernel)
  }
  return HC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- multi-slab cell exchange (host side)
static VertArrays vert_arrays(hc_cells *C, int t) {
  VertArrays a;
  for (int d = 0; d < 3; d++) {
    a.p[d] = C->pos[d] + C->first[t]; a.v[d] = C->vel[d] + C->first[t]; a.f[d] = C->frc[d] + C->first[t];
    a.r[d] = C->rep[d] ? C->rep[d] + C->first[t] : nullptr;
  }
  return a;
}
// stage a small host int array on the device in a persistent scratch slot; the copy and every later use are
// ordered on the library stream, so no host synchronisation is needed
static int stage_ints(hc_cells *C, int which, int **d, const int *h, int n) {
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

extern "C" {

int hcp_cell_extents(hc_cells *C, int type, double *minmax) {
  HC_REQUIRE(C && minmax && type >= 0 && type < C->ntypes, "hcp_cell_extents: bad arguments");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  const long nc = C->ncells[type];
  if (nc == 0) return HC_OK;
  double *d = nullptr;
  HC_HIP(hipMalloc((void **)&d, (size_t)(3 * nc) * sizeof(double)));
  hipLaunchKernelGGL(cell_extent_kernel, dim3((unsigned)nc), dim3(256), 0, hc::stream(), C->types[type]->host.nv, (const double *)(C->pos[0] + C->first[type]), d, C->L->x0, C->L->nx);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(minmax, d, (size_t)(3 * nc) * sizeof(double), hipMemcpyDeviceToHost, hc::stream());
  if (e == hipSuccess) e = hipStreamSynchronize(hc::stream());
  hipFree(d);
  if (e != hipSuccess) return hc::hip_fail(e, "hcp_cell_extents", __FILE__, __LINE__);
  return HC_OK;
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
    const size_t add = (size_t)n_new * C->types[type]->host.nv * 3;
    C->hpos[type].resize(C->hpos[type].size() + add, 0.0); C->hvel[type].resize(C->hvel[type].size() + add, 0.0); C->hfrc[type].resize(C->hfrc[type].size() + add, 0.0);
    for (int i = 0; i < n; i++) if (is_new[i]) C->hids[type].push_back(cell_ids[i]);
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
                     vert_arrays(C, type), dev_buf, C->L->x0, C->L->nx);
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
  C->ncells[type] = new_nc;
  C->nverts -= (long)n * C->types[type]->host.nv;
  return HC_OK;
}

int hcp_owned_vertices(hc_cells *C, long *n_owned) {
  HC_REQUIRE(C && n_owned, "hcp_owned_vertices: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  unsigned long long *d = nullptr, h = 0;
  HC_HIP(hipMalloc((void **)&d, sizeof(unsigned long long)));
  HC_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), hc::stream()));
  for (int t = 0; t < C->ntypes; t++) {
    const long n = C->ncells[t] * C->types[t]->host.nv;
    if (n == 0) continue;
    hipLaunchKernelGGL(owned_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hc::stream(), n, (const double *)(C->pos[0] + C->first[t]), C->L->x0, C->L->nx, d);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, hc::stream());
  if (e == hipSuccess) e = hipStreamSynchronize(hc::stream());
  hipFree(d);
  if (e != hipSuccess) return hc::hip_fail(e, "hcp_owned_vertices", __FILE__, __LINE__);
  *n_owned = (long)h;
  return HC_OK;
}

}  // extern "C"

extern "C" {

// ParticleInfo::calculate{Velocity,Force}Statistics (helper/particleInfo.cpp:30-140) as a device reduction
int hcp_vertex_stats(hc_cells *C, int what, double out[3], long *n) {
  HC_REQUIRE(C && out && n && (what == 1 || what == 2), "hcp_vertex_stats: bad arguments (what: 1 velocity, 2 force)");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
  double *d_partial = nullptr;
  HC_HIP(hipMalloc((void **)&d_partial, (size_t)STAT_BLOCKS * 4 * sizeof(double)));
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
                       C->rep[2] ? (const double *)(C->rep[2] + f) : nullptr, d_partial, launched);
    launched = 1;
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) rc = hc::stat_finish(d_partial, out, n);
  hipFree(d_partial);
  if (e != hipSuccess) return hc::hip_fail(e, "hcp_vertex_stats", __FILE__, __LINE__);
  return rc;
}

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
                       (const double *)(C->pos[0] + f), (const double *)(C->pos[1] + f), (const double *)(C->pos[2] + f), C->d_keys[0], C->d_vals[0]);
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
    return L->hmask[((size_t)xp * ny + y) * nz + z] != 0;
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
        if (near) flag[((size_t)xp * ny + y) * nz + z] = 1;
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
                       C->rep[0] + f, C->rep[1] + f, C->rep[2] + f, C->brep_const, C->brep_cutoff);
    HC_HIP(hipGetLastError());
  }
  return HC_OK;
}

// force_repulsion of every vertex, [n][3] in download order
int hcp_download_repulsion(hc_cells *C, double *out) {
  HC_REQUIRE(C && out, "hcp_download_repulsion: null pointer");
  int rc = sync_to_device(C); if (rc != HC_OK) return rc;
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
