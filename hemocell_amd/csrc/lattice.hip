// D3Q19 Guo-forced BGK collide-stream for gfx950.
//
// Replaces Palabos MultiBlockLattice3D::collideAndStream() as called from
// HemoCell::iterate (core/hemoCell.cpp:317) with GuoExternalForceBGKdynamics
// (examples/pipeflow/pipeflow.cpp:71) and BounceBack walls (:73), plus the two
// setExternalVector sweeps of core/hemoCell.cpp:369-371 and
// examples/pipeflow/pipeflow.cpp:144-146 (fused: the body force is a kernel
// argument, the per-node IBM force buffer of the *other* parity is zeroed here).
//
// Layout (HBM): structure of arrays f[q][x+HALO][y][z], z fastest, so that a
// wavefront reads/writes 64 consecutive doubles of one population and an
// x-plane of one population is contiguous (halo pack = plain copies).  Two
// population buffers (A/B "pull" scheme): the stored state is the
// POST-COLLISION field P_t; a step gathers S_t(x,i) = P_{t-1}(x - c_i, i)
// (misaligned reads, aligned writes), collides and writes P_t.  The reference's
// visible state after collideAndStream (post-stream S) is recovered by
// hcl_download_populations with the same gather.
//
// Arithmetic follows oracle/hemo_oracle.c operation for operation (the library
// is built with -ffp-contract=off) so that fluid-only runs are bit-identical.
#include "common.h"
#include "comm.h"
#include <algorithm>

using namespace hc;

namespace {

struct LatArgs {
  const double *fin;
  double *fout;
  const double *Fin;   // IBM force to read   [3][npad]
  double *Fzero;       // IBM force to zero   [3][npad]
  const uint8_t *mask;
  int nx, ny, nz;
  int plane;             // nodes of one x-plane
  long xs;               // elements from x-plane to x-plane (plane + padding)
  long npad;
  long qs;               // population stride (npad + padding)
  int x_begin;
  int x_split, x_jump;   // one launch over two ranges of planes (the planes next to the two faces of a slab): blockIdx.y >= x_split -> x += x_jump
  int wrap_x, per_y, per_z;
  double omega;
  double bx, by, bz;
  const int *row_z0, *row_cum, *blk_row;
  int nblk;
  int ibm;               // 0: no membrane cells are bound to this lattice -> the IBM force buffers are not touched
  const uint8_t *dirty_in, *dirty_zero; uint8_t epoch_in, epoch_zero;
  double wall_u[4][3];   // moving-wall classes 3..6
  int x0, nx_global;     // global x of plane 0 (body-force regions are given in global coordinates)
  BodyRegions reg;
};

// uniform body force, or that of the last region holding the node
__device__ __forceinline__ void body_at(const LatArgs &a, int x, int y, int z, double &bx, double &by, double &bz) {
  bx = a.bx; by = a.by; bz = a.bz;
  if (a.reg.n) region_force(a.reg, a.x0 + x, y, z, bx, by, bz);
}

struct Nbr {  // element offsets to the -1 / +1 neighbour along each axis, and validity
  long xm, xp;
  int ym, yp, zm, zp;
  bool ym_ok, yp_ok, zm_ok, zp_ok;
};

__device__ __forceinline__ Nbr neighbours(const LatArgs &a, int x, int y, int z) {
  Nbr n;
  n.xm = -a.xs; n.xp = a.xs;
  if (a.wrap_x) {
    if (x == 0) n.xm = (long)(a.nx - 1) * a.xs;
    if (x == a.nx - 1) n.xp = -(long)(a.nx - 1) * a.xs;
  }
  n.ym = -a.nz; n.yp = a.nz; n.ym_ok = n.yp_ok = true;
  if (y == 0) { if (a.per_y) n.ym = (a.ny - 1) * a.nz; else n.ym_ok = false; }
  if (y == a.ny - 1) { if (a.per_y) n.yp = -(a.ny - 1) * a.nz; else n.yp_ok = false; }
  n.zm = -1; n.zp = 1; n.zm_ok = n.zp_ok = true;
  if (z == 0) { if (a.per_z) n.zm = a.nz - 1; else n.zm_ok = false; }
  if (z == a.nz - 1) { if (a.per_z) n.zp = -(a.nz - 1); else n.zp_ok = false; }
  return n;
}

// offset from a node to (node - c_q)   [c = +1 -> the -1 neighbour]
template <int CX, int CY, int CZ>
__device__ __forceinline__ long src_off(const Nbr &n, bool &ok) {
  long off = 0; ok = true;
  if (CX == 1) off += n.xm; else if (CX == -1) off += n.xp;
  if (CY == 1) { off += n.ym; ok = ok && n.ym_ok; } else if (CY == -1) { off += n.yp; ok = ok && n.yp_ok; }
  if (CZ == 1) { off += n.zm; ok = ok && n.zm_ok; } else if (CZ == -1) { off += n.zp; ok = ok && n.zp_ok; }
  return off;
}
// offset from a node to (node + c_q)
template <int CX, int CY, int CZ>
__device__ __forceinline__ long dst_off(const Nbr &n, bool &ok) {
  long off = 0; ok = true;
  if (CX == 1) off += n.xp; else if (CX == -1) off += n.xm;
  if (CY == 1) { off += n.yp; ok = ok && n.yp_ok; } else if (CY == -1) { off += n.ym; ok = ok && n.ym_ok; }
  if (CZ == 1) { off += n.zp; ok = ok && n.zp_ok; } else if (CZ == -1) { off += n.zm; ok = ok && n.zm_ok; }
  return off;
}

#define FOR_Q(M)                                                                                     \
  M(0, 0, 0, 0) M(1, -1, 0, 0) M(2, 0, -1, 0) M(3, 0, 0, -1) M(4, -1, -1, 0) M(5, -1, 1, 0)          \
  M(6, -1, 0, -1) M(7, -1, 0, 1) M(8, 0, -1, -1) M(9, 0, -1, 1) M(10, 1, 0, 0) M(11, 0, 1, 0)        \
  M(12, 0, 0, 1) M(13, 1, 1, 0) M(14, 1, -1, 0) M(15, 1, 0, 1) M(16, 1, 0, -1) M(17, 0, 1, 1)        \
  M(18, 0, 1, -1)

__device__ __forceinline__ constexpr double tq(int q) { return q == 0 ? 1. / 3. : ((q >= 1 && q <= 3) || (q >= 10 && q <= 12)) ? 1. / 18. : 1. / 36.; }

// gather the post-stream populations S(node, q) = P(node - c_q, q)
__device__ __forceinline__ void pull(const double *__restrict__ fin, long npad, long node, const Nbr &n, double f[HC_Q]) {
#define M(Q, CX, CY, CZ)                                   \
  {                                                        \
    bool ok; long off = src_off<CX, CY, CZ>(n, ok);        \
    f[Q] = ok ? fin[(long)Q * npad + node + off] : 0.0;    \
  }
  FOR_Q(M)
#undef M
}

// moments in the oracle's order: ascending q, zero-velocity components skipped
__device__ __forceinline__ void moments(const double f[HC_Q], double &rhoBar, double &jx, double &jy, double &jz) {
  double r = 0.0, x = 0.0, y = 0.0, z = 0.0;
#define M(Q, CX, CY, CZ)              \
  r += f[Q];                          \
  if (CX == 1) x += f[Q]; else if (CX == -1) x += -f[Q]; \
  if (CY == 1) y += f[Q]; else if (CY == -1) y += -f[Q]; \
  if (CZ == 1) z += f[Q]; else if (CZ == -1) z += -f[Q];
  FOR_Q(M)
#undef M
  rhoBar = r; jx = x; jy = y; jz = z;
}

template <int CX, int CY, int CZ>
__device__ __forceinline__ double cdot(double a0, double a1, double a2) {
  // ((cx*a0 + cy*a1) + cz*a2) with the zero terms dropped (exact)
  double s = 0.0; bool first = true;
  if (CX != 0) { s = (CX == 1 ? a0 : -a0); first = false; }
  if (CY != 0) { double t = (CY == 1 ? a1 : -a1); s = first ? t : s + t; first = false; }
  if (CZ != 0) { double t = (CZ == 1 ? a2 : -a2); s = first ? t : s + t; first = false; }
  return s;
}

// GuoExternalForceBGKdynamics::collide, operation order of oracle/hemo_oracle.c collide_guo_bgk
__device__ __forceinline__ void collide_guo(double f[HC_Q], double Fx, double Fy, double Fz, double omega) {
  double rhoBar, j0, j1, j2;
  moments(f, rhoBar, j0, j1, j2);
  const double invRho = 1.0 / (1.0 + rhoBar);
  const double rho = 1.0 + rhoBar;
  const double u0 = j0 * invRho + Fx / 2.0, u1 = j1 * invRho + Fy / 2.0, u2 = j2 * invRho + Fz / 2.0;
  j0 = rho * u0; j1 = rho * u1; j2 = rho * u2;
  const double jSqr = j0 * j0 + j1 * j1 + j2 * j2;
  const double one_m_omega = 1.0 - omega;
  const double guo = 1.0 - omega / 2.0;
#define M(Q, CX, CY, CZ)                                                                     \
  {                                                                                          \
    const double c_j = cdot<CX, CY, CZ>(j0, j1, j2);                                         \
    const double feq = tq(Q) * (rhoBar + 3.0 * c_j + invRho * (4.5 * c_j * c_j - 1.5 * jSqr)); \
    f[Q] *= one_m_omega;                                                                     \
    f[Q] += omega * feq;                                                                     \
  }
  FOR_Q(M)
#undef M
#define M(Q, CX, CY, CZ)                                                                     \
  {                                                                                          \
    double c_u = cdot<CX, CY, CZ>(u0, u1, u2);                                               \
    c_u *= 9.0;                                                                              \
    double ft = (((double)CX - u0) * 3.0 + c_u * (double)CX) * Fx;                           \
    ft += (((double)CY - u1) * 3.0 + c_u * (double)CY) * Fy;                                 \
    ft += (((double)CZ - u2) * 3.0 + c_u * (double)CZ) * Fz;                                 \
    ft *= tq(Q);                                                                             \
    ft *= guo;                                                                               \
    f[Q] += ft;                                                                              \
  }
  FOR_Q(M)
#undef M
}

template <bool REGIONS>
__global__ __launch_bounds__(256) void collide_stream_kernel(LatArgs a) {
  // thread -> (y,z) through the active-span map of this plane: consecutive threads walk the spans of
  // consecutive rows, so every lane of every wave (except the last of a plane) has a live node
  const int x = a.x_begin + (int)blockIdx.y + ((int)blockIdx.y >= a.x_split ? a.x_jump : 0);
  const int xp = x + HALO;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int *cum = a.row_cum + (long)xp * (a.ny + 1);
  if (t >= cum[a.ny]) return;
  int lo = a.blk_row[(long)xp * (a.nblk + 1) + blockIdx.x], hi = a.blk_row[(long)xp * (a.nblk + 1) + blockIdx.x + 1];
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cum[mid] <= t) lo = mid; else hi = mid - 1; }
  const int y = lo, z = a.row_z0[(long)xp * a.ny + y] + (t - cum[y]);
  const int p = y * a.nz + z;
  const long node = (long)(x + HALO) * a.xs + p;
  const uint8_t m = a.mask[node];
  if (m == 2) return;   // solid node with no fluid neighbour: inert under full-way bounce-back
  const Nbr n = neighbours(a, x, y, z);
  double f[HC_Q];
  pull(a.fin, a.qs, node, n, f);
  const bool wall = m != 0;
  if (wall) {
    // BounceBack::collide: swap opposite pairs (full-way bounce-back)
#pragma unroll
    for (int i = 1; i <= 9; i++) { double t = f[i]; f[i] = f[i + 9]; f[i + 9] = t; }
    if (m >= 3) {
      // moving no-slip wall: Ladd's momentum term at rho = 1, f_opp(i) = f_i - 6 t_i (c_i.u_w); after the
      // swap slot o holds f_i, so the term of direction i = opp(o) is subtracted from slot o
      const double w0 = a.wall_u[m - 3][0], w1 = a.wall_u[m - 3][1], w2 = a.wall_u[m - 3][2];
#define M(Q, CX, CY, CZ)                                                                  \
      if (Q != 0) {                                                                       \
        constexpr int O = Q <= 9 ? Q + 9 : Q - 9;                                         \
        const double c_u = (double)CX * w0 + (double)CY * w1 + (double)CZ * w2;           \
        f[O] = f[O] - 6.0 * tq(Q) * c_u;                                                  \
      }
      FOR_Q(M)
#undef M
    }
  } else {
    double bx = a.bx, by = a.by, bz = a.bz;
    if (REGIONS) region_force(a.reg, a.x0 + x, y, z, bx, by, bz);
    double Fx = bx, Fy = by, Fz = bz;
    if (a.ibm && a.dirty_in[node >> 4] == a.epoch_in) {   // x + 0.0 == x, so skipping untouched groups changes no bits
      Fx = bx + a.Fin[node]; Fy = by + a.Fin[a.npad + node]; Fz = bz + a.Fin[2 * a.npad + node];
    }
    collide_guo(f, Fx, Fy, Fz, a.omega);
  }
#pragma unroll
  // streamed once and read again only after 5 GB of other traffic: non-temporal stores keep the lines out of the way
  // of the loads (measured: -3 % kernel time on the pipe and on the all-fluid box; non-temporal loads cost 4 %)
  for (int q = 0; q < HC_Q; q++) __builtin_nontemporal_store(f[q], &a.fout[(long)q * a.qs + node]);
  if (a.ibm && a.dirty_zero[node >> 4] == a.epoch_zero) { a.Fzero[node] = 0.0; a.Fzero[a.npad + node] = 0.0; a.Fzero[2 * a.npad + node] = 0.0; }
}

// P(y,i) = mask[y+c_i] ? 0 : feq_i(rho,u): initializeAtEquilibrium in the shifted representation
__global__ void init_eq_kernel(LatArgs a, double rhoBar, double j0, double j1, double j2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const int y = p / a.nz, z = p - y * a.nz;
  const long node = (long)(x + HALO) * a.xs + p;
  const Nbr n = neighbours(a, x, y, z);
  const double invRho = 1.0 / (1.0 + rhoBar);
  const double jSqr = j0 * j0 + j1 * j1 + j2 * j2;
#define M(Q, CX, CY, CZ)                                                                        \
  {                                                                                             \
    bool ok; long off = dst_off<CX, CY, CZ>(n, ok);                                             \
    double v = 0.0;                                                                             \
    if (ok && a.mask[node + off] == 0) {                                                        \
      const double c_j = cdot<CX, CY, CZ>(j0, j1, j2);                                          \
      v = tq(Q) * (rhoBar + 3.0 * c_j + invRho * (4.5 * c_j * c_j - 1.5 * jSqr));               \
    }                                                                                           \
    a.fout[(long)Q * a.qs + node] = v;                                                          \
  }
  FOR_Q(M)
#undef M
}

// AoS [node][19] of the post-stream state, bulk nodes only
__global__ void download_kernel(LatArgs a, double *aos) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const int y = p / a.nz, z = p - y * a.nz;
  const long node = (long)(x + HALO) * a.xs + p;
  const Nbr n = neighbours(a, x, y, z);
  double f[HC_Q];
  pull(a.fin, a.qs, node, n, f);
  const long o = ((long)x * a.plane + p) * HC_Q;
#pragma unroll
  for (int q = 0; q < HC_Q; q++) aos[o + q] = f[q];
}

// inverse of download: P(y,i) = S(y+c_i, i) (0 if the target lies outside)
// lo_ok / hi_ok: the post-stream state of the plane below / above the slab is in aos as well (planes -1 and nx of the host
// numbering, fetched from the neighbouring ranks)
__global__ void upload_kernel(LatArgs a, const double *aos, int lo_ok, int hi_ok) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const int y = p / a.nz, z = p - y * a.nz;
  const long node = (long)(x + HALO) * a.xs + p;
  const long bulk = (long)x * a.plane + p;
  Nbr nl = neighbours(a, x, y, z);   // offsets in the unpadded [x][y][z] numbering of the host array
  nl.xm = nl.xm / a.xs * a.plane; nl.xp = nl.xp / a.xs * a.plane;
#define M(Q, CX, CY, CZ)                                                                  \
  {                                                                                       \
    bool ok; long off = dst_off<CX, CY, CZ>(nl, ok);                                      \
    /* without x wrap the +-x neighbour of a face plane belongs to the neighbouring rank, or lies outside the domain */ \
    if (!a.wrap_x && ((CX == 1 && x == a.nx - 1 && !hi_ok) || (CX == -1 && x == 0 && !lo_ok))) ok = false; \
    a.fout[(long)Q * a.qs + node] = ok ? aos[(bulk + off) * HC_Q + Q] : 0.0;            \
  }
  FOR_Q(M)
#undef M
}

__global__ void rho_u_kernel(LatArgs a, double *rho, double *u) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const int y = p / a.nz, z = p - y * a.nz;
  const long node = (long)(x + HALO) * a.xs + p;
  const Nbr n = neighbours(a, x, y, z);
  double f[HC_Q];
  pull(a.fin, a.qs, node, n, f);
  double rhoBar, j0, j1, j2;
  moments(f, rhoBar, j0, j1, j2);
  const double invRho = 1.0 / (1.0 + rhoBar);
  const long o = (long)x * a.plane + p;
  rho[o] = 1.0 + rhoBar;
  double bx, by, bz;
  body_at(a, x, y, z, bx, by, bz);
  u[3 * o] = j0 * invRho + (bx + a.Fin[node]) / 2.0;
  u[3 * o + 1] = j1 * invRho + (by + a.Fin[a.npad + node]) / 2.0;
  u[3 * o + 2] = j2 * invRho + (bz + a.Fin[2 * a.npad + node]) / 2.0;
}

// Off-equilibrium part of the momentum-flux tensor, as Palabos' momentTemplates::compute_rhoBar_j_PiNeq forms it from the
// stored populations f_i - t_i: Pi_ab = sum_i c_ia c_ib fbar_i - j_a j_b / rho - cs2 rhoBar delta_ab, components in the
// order xx, xy, xz, yy, yz, zz.  Output fields only (shear stress, strain rate); nothing on the step path reads it.
__global__ void pi_neq_kernel(LatArgs a, double *pi) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const int y = p / a.nz, z = p - y * a.nz;
  const long node = (long)(x + HALO) * a.xs + p;
  const Nbr n = neighbours(a, x, y, z);
  double f[HC_Q];
  pull(a.fin, a.qs, node, n, f);
  double rhoBar, j0, j1, j2;
  moments(f, rhoBar, j0, j1, j2);
  const double invRho = 1.0 / (1.0 + rhoBar);
  double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
#define M(Q, CX, CY, CZ)                       \
  if (CX * CX) xx += f[Q];                     \
  if (CX * CY == 1) xy += f[Q]; else if (CX * CY == -1) xy += -f[Q]; \
  if (CX * CZ == 1) xz += f[Q]; else if (CX * CZ == -1) xz += -f[Q]; \
  if (CY * CY) yy += f[Q];                     \
  if (CY * CZ == 1) yz += f[Q]; else if (CY * CZ == -1) yz += -f[Q]; \
  if (CZ * CZ) zz += f[Q];
  FOR_Q(M)
#undef M
  const double cs2 = 1.0 / 3.0;
  const long o = 6 * ((long)x * a.plane + p);
  pi[o] = xx - invRho * j0 * j0 - cs2 * rhoBar;
  pi[o + 1] = xy - invRho * j0 * j1;
  pi[o + 2] = xz - invRho * j0 * j2;
  pi[o + 3] = yy - invRho * j1 * j1 - cs2 * rhoBar;
  pi[o + 4] = yz - invRho * j1 * j2;
  pi[o + 5] = zz - invRho * j2 * j2 - cs2 * rhoBar;
}

__global__ void force_aos_kernel(LatArgs a, double *F) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.plane) return;
  const int x = a.x_begin + blockIdx.y;
  const long node = (long)(x + HALO) * a.xs + p;
  const long o = (long)x * a.plane + p;
  for (int d = 0; d < 3; d++) F[3 * o + d] = a.Fin[d * a.npad + node];
}

// FluidInfo statistics (helper/fluidInfo.cpp:33-96): magnitude of Cell::computeVelocity (what 0) or of the external
// force (what 1) over the non-boundary bulk nodes; what 2: rhoBar = sum of the 19 stored populations, all bulk nodes
__global__ __launch_bounds__(256) void fluid_stats_kernel(LatArgs a, int what, double *partial) {
  StatAcc acc{1e300, -1e300, 0.0, 0};
  const long nbulk = (long)a.nx * a.plane;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nbulk; k += (long)STAT_BLOCKS * 256) {
    const int x = (int)(k / a.plane), p = (int)(k - (long)x * a.plane);
    const long node = (long)(x + HALO) * a.xs + p;
    if (what == 2) {   // mass: sum of the stored populations of EVERY bulk node (walls park what bounces back)
      double r = 0.0;
#pragma unroll
      for (int q = 0; q < HC_Q; q++) r += a.fin[(long)q * a.qs + node];
      stat_add(acc, r);
      continue;
    }
    if (a.mask[node] != 0) continue;
    const int y = p / a.nz, z = p - y * a.nz;
    double bx, by, bz;
    body_at(a, x, y, z, bx, by, bz);
    double Fx = bx, Fy = by, Fz = bz;
    if (a.ibm) { Fx = bx + a.Fin[node]; Fy = by + a.Fin[a.npad + node]; Fz = bz + a.Fin[2 * a.npad + node]; }
    double v0 = Fx, v1 = Fy, v2 = Fz;
    if (what == 0) {
      const Nbr n = neighbours(a, x, y, z);
      double f[HC_Q];
      pull(a.fin, a.qs, node, n, f);
      double rhoBar, j0, j1, j2;
      moments(f, rhoBar, j0, j1, j2);
      const double invRho = 1.0 / (1.0 + rhoBar);
      v0 = j0 * invRho + Fx / 2.0; v1 = j1 * invRho + Fy / 2.0; v2 = j2 * invRho + Fz / 2.0;
    }
    stat_add(acc, sqrt(v0 * v0 + v1 * v1 + v2 * v2));
  }
  stat_block_store(acc, partial);
}

struct HaloArgs {
  double *f;          // population buffer
  double *buf;        // contiguous staging
  double *buf2; int n_first;   // entries e >= n_first belong to the second staging block (both faces in one launch)
  long npad; long xs; int plane;   // npad: population stride, xs: x-plane stride
  int n;              // (population, plane) entries
  int pop[HC_Q + 5];  // population of entry e
  int xp[HC_Q + 5];   // padded x index of its plane
  int to_buf;
};
__global__ void halo_copy_kernel(HaloArgs h) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= h.plane) return;
  const int e = blockIdx.y;
  const long li = (long)h.pop[e] * h.npad + (long)h.xp[e] * h.xs + p;
  double *b = e < h.n_first ? h.buf + (long)e * h.plane + p : h.buf2 + (long)(e - h.n_first) * h.plane + p;
  if (h.to_buf) *b = h.f[li]; else h.f[li] = *b;
}

// clears the 2*HALO halo planes of the three IBM force components
__global__ void zero_force_halo_kernel(double *F, long npad, long xs, int plane, int nx) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= plane) return;
  const int comp = blockIdx.y / (2 * HALO), w = blockIdx.y % (2 * HALO);
  const int xp = w < HALO ? w : nx + w;   // padded plane index: 0..HALO-1 and nx+HALO..nx+2*HALO-1
  F[(long)comp * npad + (long)xp * xs + p] = 0.0;
}

LatArgs make_args(const hc_lattice *L) {
  LatArgs a;
  a.fin = L->f[L->cur]; a.fout = L->f[1 - L->cur];
  const int fprev = (L->fcur + 2) % 3;
  a.Fin = L->force[L->fcur]; a.Fzero = L->force[fprev];
  a.x_split = 0x7fffffff; a.x_jump = 0;
  a.mask = L->mask;
  a.nx = L->nx; a.ny = L->ny; a.nz = L->nz; a.plane = (int)L->plane; a.xs = (long)L->xs; a.npad = (long)L->npad; a.qs = (long)L->qstride;
  a.x_begin = 0;
  a.wrap_x = (L->n_slabs == 1 && L->periodic[0]) ? 1 : 0;
  a.per_y = L->periodic[1]; a.per_z = L->periodic[2];
  a.omega = L->omega; a.bx = L->body[0]; a.by = L->body[1]; a.bz = L->body[2];
  a.x0 = L->x0; a.nx_global = L->nx_global; a.reg = L->regions;
  a.row_z0 = L->row_z0; a.row_cum = L->row_cum; a.blk_row = L->blk_row; a.nblk = L->nblk;
  a.ibm = L->ibm;
  a.dirty_in = L->fdirty[L->fcur]; a.dirty_zero = L->fdirty[fprev]; a.epoch_in = L->fepoch[L->fcur]; a.epoch_zero = L->fepoch[fprev];
  for (int c = 0; c < 4; c++) for (int d = 0; d < 3; d++) a.wall_u[c][d] = L->wall_u[c][d];
  return a;
}

dim3 plane_grid(const hc_lattice *L, int nplanes) { return dim3((unsigned)((L->plane + 255) / 256), (unsigned)nplanes, 1); }

int ensure_scratch(hc_lattice *L, size_t doubles) {
  if (L->scratch_doubles >= doubles) return HC_OK;
  if (L->scratch) HC_HIP(hipFree(L->scratch));
  L->scratch = nullptr; L->scratch_doubles = 0;
  HC_HIP(hipMalloc((void **)&L->scratch, doubles * sizeof(double)));
  L->scratch_doubles = doubles;
  return HC_OK;
}

// (re)build the active-span map from the host mask classes
int rebuild_active_map(hc_lattice *L) {
  const int NX = L->nx + 2 * HALO, ny = L->ny, nz = L->nz;
  std::vector<int> z0((size_t)NX * ny), cum((size_t)NX * (ny + 1));
  int max_active = 0;
  for (int x = 0; x < NX; x++) {
    int c = 0;
    for (int y = 0; y < ny; y++) {
      const uint8_t *row = L->hmask.data() + (size_t)x * L->xs + (size_t)y * nz;
      int a = 0, b = nz;
      while (a < nz && row[a] == 2) a++;
      while (b > a && row[b - 1] == 2) b--;
      z0[(size_t)x * ny + y] = a;
      cum[(size_t)x * (ny + 1) + y] = c;
      c += b - a;
    }
    cum[(size_t)x * (ny + 1) + ny] = c;
    max_active = std::max(max_active, c);
  }
  if (max_active == 0) max_active = 1;
  const int nblk = (max_active + 255) / 256;
  std::vector<int> blk((size_t)NX * (nblk + 1));
  for (int x = 0; x < NX; x++) {
    const int *cx = cum.data() + (size_t)x * (ny + 1);
    int y = 0;
    for (int b = 0; b <= nblk; b++) {
      const long t = (long)b * 256;
      while (y + 1 < ny && cx[y + 1] <= t) y++;
      blk[(size_t)x * (nblk + 1) + b] = y;
    }
  }
  for (int **p : {&L->row_z0, &L->row_cum, &L->blk_row}) if (*p) { HC_HIP(hipFree(*p)); *p = nullptr; }
  HC_HIP(hipMalloc((void **)&L->row_z0, z0.size() * sizeof(int)));
  HC_HIP(hipMalloc((void **)&L->row_cum, cum.size() * sizeof(int)));
  HC_HIP(hipMalloc((void **)&L->blk_row, blk.size() * sizeof(int)));
  HC_HIP(hipMemcpy(L->row_z0, z0.data(), z0.size() * sizeof(int), hipMemcpyHostToDevice));
  HC_HIP(hipMemcpy(L->row_cum, cum.data(), cum.size() * sizeof(int), hipMemcpyHostToDevice));
  HC_HIP(hipMemcpy(L->blk_row, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
  L->nblk = nblk; L->max_active = max_active;
  return HC_OK;
}

// one byte per 8 x 8 x 8 brick (see hc_lattice::wallbrick)
int rebuild_wall_bricks(hc_lattice *L) {
  const int NX = L->nx + 2 * HALO, ny = L->ny, nz = L->nz;
  std::vector<uint8_t> flag((size_t)L->nbx * L->nby * L->nbz, 0);
  for (int bx = 0; bx < L->nbx; bx++)
    for (int by = 0; by < L->nby; by++)
      for (int bz = 0; bz < L->nbz; bz++) {
        bool near = false;
        // faces no stencil crosses: the ends of a non-periodic axis; the halo planes of a slab (a single slab never addresses them)
        if (!L->periodic[1] && (by == 0 || by == L->nby - 1)) near = true;
        if (!L->periodic[2] && (bz == 0 || bz == L->nbz - 1)) near = true;
        if (bx * 8 < HALO + 1 || bx * 8 + 7 >= L->nx + HALO - 1) near = true;
        for (int x = bx * 8; x < std::min(NX, bx * 8 + 8) && !near; x++)
          for (int y = by * 8; y < std::min(ny, by * 8 + 8) && !near; y++) {
            const uint8_t *row = L->hmask.data() + (size_t)x * L->xs + (size_t)y * nz;
            for (int z = bz * 8; z < std::min(nz, bz * 8 + 8); z++) if (row[z] != 0) { near = true; break; }
          }
        flag[((size_t)bx * L->nby + by) * L->nbz + bz] = near ? 1 : 0;
      }
  if (!L->wallbrick) HC_HIP(hipMalloc((void **)&L->wallbrick, flag.size()));
  HC_HIP(hipMemcpy(L->wallbrick, flag.data(), flag.size(), hipMemcpyHostToDevice));
  return HC_OK;
}

// planes [x_begin, x_begin + nplanes) and, in the same launch, [x2, x2 + n2) (n2 = 0: one range)
int launch_collide(hc_lattice *L, int x_begin, int nplanes, int x2 = 0, int n2 = 0) {
  if (nplanes + n2 <= 0) return HC_OK;
  LatArgs a = make_args(L);
  a.x_begin = x_begin;
  if (n2 > 0) { a.x_split = nplanes; a.x_jump = x2 - (x_begin + nplanes); }
  const unsigned ny = (unsigned)(nplanes + n2);
  if (L->regions.n) hipLaunchKernelGGL(collide_stream_kernel<true>, dim3((unsigned)((L->max_active + 255) / 256), ny, 1), dim3(256), 0, hc::stream(), a);
  else hipLaunchKernelGGL(collide_stream_kernel<false>, dim3((unsigned)((L->max_active + 255) / 256), ny, 1), dim3(256), 0, hc::stream(), a);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

}  // namespace

static int g_force_plane_padding = 0;   // tests / A-B runs: 1 = pad the planes of every lattice, -1 = of none, 0 = by size

extern "C" {

int hc_debug_force_plane_padding(int on) { g_force_plane_padding = on > 0 ? 1 : (on < 0 ? -1 : 0); return HC_OK; }

int hcl_create(hc_lattice **out, int nx, int ny, int nz, const int periodic[3], double omega,
               int x0, int nx_global, int n_slabs) {
  HC_REQUIRE(out && periodic, "hcl_create: null pointer");
  HC_REQUIRE(nx >= 2 && ny >= 2 && nz >= 2, "hcl_create: every dimension must be >= 2");
  HC_REQUIRE(n_slabs >= 1 && nx_global >= nx && x0 >= 0 && x0 + nx <= nx_global, "hcl_create: inconsistent slab decomposition");
  HC_REQUIRE((long)ny * nz < (1L << 30) && (long)(nx + 2 * HALO) * (((long)ny + 8) * nz + 32) < (1L << 31), "hcl_create: slab too large for 32-bit plane indexing");
  HC_REQUIRE(omega > 0.0 && omega < 2.0, "hcl_create: omega must be in (0,2)");
  if (hc::stream() == nullptr) { hc::set_error("hcl_create: hc_init() has not been called"); return HC_ERR_STATE; }
  hc_lattice *L = new hc_lattice();
  L->nx = nx; L->ny = ny; L->nz = nz;
  for (int d = 0; d < 3; d++) L->periodic[d] = periodic[d] ? 1 : 0;
  L->x0 = x0; L->nx_global = nx_global; L->n_slabs = n_slabs;
  L->omega = omega;
  L->plane = (size_t)ny * nz;
  // x-plane stride.  (1) A multiple of 16 doubles, so that the x-1 / x+1 neighbours of a 128-byte line are lines too: with
  // an odd plane (the reference's voxelised tubes are 2N+3 x N+3 x N+3 nodes) every read of a population that moves along
  // x straddles one line more (511 x 257 x 257 pipe: -8 % against 512 x 256 x 257).  (2) x-planes whose size is a multiple
  // of 1 MiB (512 x 512 doubles) put the x-1 / x / x+1 planes a kernel streams at the same time at the same offset in the
  // HBM channel interleave; 8 rows of padding between planes take them apart (all-fluid 512^3 box: about +5 %).
  L->xs = (L->plane + 15) / 16 * 16;
  if (g_force_plane_padding > 0 || (g_force_plane_padding == 0 && ((L->plane * sizeof(double)) % (1u << 20)) == 0)) L->xs += (size_t)(8 * nz + 15) / 16 * 16;
  if (g_force_plane_padding < 0) L->xs = L->plane;
  L->npad = (size_t)(nx + 2 * HALO) * L->xs;
  // The 19 population arrays are streamed side by side.  With power-of-two planes (512 x 512 doubles = 2 MiB) and
  // npad a multiple of the plane, all 38 read / write streams of a node sit at the same offset modulo 2 MiB and
  // camp on the same HBM channels (all-fluid 512^3 box: 5.27 TB/s algorithmic against 6.0 for 256^3).  An odd
  // number of 128-byte lines between consecutive populations spreads them over the channels.
  L->qstride = L->npad + 16 * 129;
  L->cur = 0; L->fcur = 0; L->ibm = 0;
  L->body[0] = L->body[1] = L->body[2] = 0.0;
  L->regions.n = 0;
  for (int c = 0; c < 4; c++) for (int d = 0; d < 3; d++) L->wall_u[c][d] = 0.0;
  L->scratch = nullptr; L->scratch_doubles = 0;
  L->f[0] = L->f[1] = nullptr; L->mask = nullptr;
  for (int k = 0; k < 3; k++) { L->force[k] = nullptr; L->fdirty[k] = nullptr; }
  for (int k = 0; k < 2; k++) {
    HC_HIP(hipMalloc((void **)&L->f[k], L->qstride * HC_Q * sizeof(double)));
    HC_HIP(hipMemsetAsync(L->f[k], 0, L->qstride * HC_Q * sizeof(double), hc::stream()));
  }
  for (int k = 0; k < 3; k++) {
    HC_HIP(hipMalloc((void **)&L->force[k], L->npad * 3 * sizeof(double)));
    HC_HIP(hipMemsetAsync(L->force[k], 0, L->npad * 3 * sizeof(double), hc::stream()));
  }
  HC_HIP(hipMalloc((void **)&L->mask, L->npad));
  HC_HIP(hipMemsetAsync(L->mask, 0, L->npad, hc::stream()));
  for (int k = 0; k < 3; k++) {
    HC_HIP(hipMalloc((void **)&L->fdirty[k], L->npad / 16 + 1));
    HC_HIP(hipMemsetAsync(L->fdirty[k], 0, L->npad / 16 + 1, hc::stream()));
    L->fepoch[k] = 1;
  }
  L->hmask.assign(L->npad, 0);   // device numbering; hcl_set_mask marks the padding
  L->row_z0 = L->row_cum = L->blk_row = nullptr;
  L->wallbrick = nullptr; L->nbx = (nx + 2 * HALO + 7) / 8; L->nby = (ny + 7) / 8; L->nbz = (nz + 7) / 8;
  { int rc = rebuild_active_map(L); if (rc != HC_OK) return rc; }
  { int rc = rebuild_wall_bricks(L); if (rc != HC_OK) return rc; }
  HC_HIP(hipStreamSynchronize(hc::stream()));
  *out = L;
  return HC_OK;
}

int hcl_destroy(hc_lattice *L) {
  if (!L) return HC_OK;
  hcs::lattice_destroyed(L);
  hipStreamSynchronize(hc::stream());
  for (int k = 0; k < 2; k++) if (L->f[k]) hipFree(L->f[k]);
  for (int k = 0; k < 3; k++) { if (L->force[k]) hipFree(L->force[k]); if (L->fdirty[k]) hipFree(L->fdirty[k]); }
  if (L->mask) hipFree(L->mask);
  if (L->scratch) hipFree(L->scratch);
  if (L->row_z0) hipFree(L->row_z0);
  if (L->row_cum) hipFree(L->row_cum);
  if (L->blk_row) hipFree(L->blk_row);
  if (L->wallbrick) hipFree(L->wallbrick);
  delete L;
  return HC_OK;
}

int hcl_dims(const hc_lattice *L, int dims[3]) {
  HC_REQUIRE(L && dims, "hcl_dims: null pointer");
  dims[0] = L->nx; dims[1] = L->ny; dims[2] = L->nz;
  return HC_OK;
}

int hcl_node_counts(const hc_lattice *L, long counts[3]) {
  HC_REQUIRE(L && counts, "hcl_node_counts: null pointer");
  long fluid = 0, active = 0;
  for (int x = 0; x < L->nx; x++)
    for (size_t p = 0; p < L->plane; p++) {
      const uint8_t m = L->hmask[(size_t)(x + HALO) * L->xs + p];
      fluid += m == 0; active += m != 2;
    }
  counts[0] = (long)L->nx * (long)L->plane; counts[1] = fluid; counts[2] = active;
  return HC_OK;
}

int hcl_set_mask(hc_lattice *L, const uint8_t *mask_with_halo) {
  HC_REQUIRE(L && mask_with_halo, "hcl_set_mask: null pointer");
  // host copy in the device numbering (x-planes xs apart); the padding between planes is inert solid
  L->hmask.assign(L->npad, 2);
  for (int x = 0; x < L->nx + 2 * HALO; x++)
    for (size_t p = 0; p < L->plane; p++) {
      const uint8_t m = mask_with_halo[(size_t)x * L->plane + p];
      L->hmask[(size_t)x * L->xs + p] = (m >= 3 && m <= 6) ? m : (m ? 1 : 0);   // 1 = bounce-back, 3..6 = moving-wall classes
    }
  // class 2 = solid node without any fluid neighbour.  Full-way bounce-back returns every population to
  // where it came from, so such a node never exchanges anything with the fluid: the collide kernel skips
  // it (no loads, no stores).  Results on fluid nodes are unchanged, bit for bit.
  {
    static const int cx[HC_Q] = HC_CX, cy[HC_Q] = HC_CY, cz[HC_Q] = HC_CZ;
    const int NX = L->nx + 2 * HALO, ny = L->ny, nz = L->nz;
    std::vector<uint8_t> cls(L->hmask);
    for (int x = 0; x < NX; x++)
      for (int y = 0; y < ny; y++)
        for (int z = 0; z < nz; z++) {
          const size_t k = (size_t)x * L->xs + (size_t)y * nz + z;
          if (!L->hmask[k]) continue;
          bool fluid_near = false;
          for (int q = 1; q < HC_Q && !fluid_near; q++) {
            int xx = x + cx[q], yy = y + cy[q], zz = z + cz[q];
            if (xx < 0 || xx >= NX) { fluid_near = true; break; }   // beyond the halo: unknown, keep the node active
            if (yy < 0 || yy >= ny) { if (L->periodic[1]) yy = (yy + ny) % ny; else continue; }
            if (zz < 0 || zz >= nz) { if (L->periodic[2]) zz = (zz + nz) % nz; else continue; }
            if (!L->hmask[(size_t)xx * L->xs + (size_t)yy * nz + zz]) fluid_near = true;
          }
          if (!fluid_near) cls[k] = 2;
        }
    // a 128-byte line (16 doubles) of a population array must be written completely or not at all:
    // partially written lines at the ends of the live spans cost a read-modify-write in the memory system
    // (measured: the pipe ran no faster than a full box although it moves 18 % fewer bytes).  Inert nodes
    // that share a line with a live node are therefore kept as ordinary bounce-back nodes.
    const size_t n = cls.size();
    for (size_t g = 0; g < n; g += 16) {
      const size_t e = std::min(n, g + 16);
      bool live = false;
      for (size_t k = g; k < e; k++) if (cls[k] != 2) { live = true; break; }
      if (live) for (size_t k = g; k < e; k++) if (cls[k] == 2) cls[k] = 1;
    }
    L->hmask.swap(cls);
  }
  { int rc = rebuild_active_map(L); if (rc != HC_OK) return rc; }
  { int rc = rebuild_wall_bricks(L); if (rc != HC_OK) return rc; }
  HC_HIP(hipMemcpyAsync(L->mask, L->hmask.data(), L->npad, hipMemcpyHostToDevice, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_init_equilibrium(hc_lattice *L, double rho, const double u[3]) {
  HC_REQUIRE(L && u, "hcl_init_equilibrium: null pointer");
  LatArgs a = make_args(L);
  a.fout = L->f[L->cur];
  HC_HIP(hipMemsetAsync(L->f[0], 0, L->qstride * HC_Q * sizeof(double), hc::stream()));
  HC_HIP(hipMemsetAsync(L->f[1], 0, L->qstride * HC_Q * sizeof(double), hc::stream()));
  hipLaunchKernelGGL(init_eq_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, rho - 1.0, rho * u[0], rho * u[1], rho * u[2]);
  HC_HIP(hipGetLastError());
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_set_wall_velocity(hc_lattice *L, int wall_class, const double u[3]) {
  HC_REQUIRE(L && u && wall_class >= 3 && wall_class <= 6, "hcl_set_wall_velocity: class must be 3..6");
  for (int d = 0; d < 3; d++) L->wall_u[wall_class - 3][d] = u[d];
  return HC_OK;
}

int hcl_set_body_force(hc_lattice *L, const double F[3]) {
  HC_REQUIRE(L && F, "hcl_set_body_force: null pointer");
  for (int d = 0; d < 3; d++) L->body[d] = F[d];
  return HC_OK;
}

int hcl_set_body_force_regions(hc_lattice *L, int n, const int *boxes, const double *forces) {
  HC_REQUIRE(L && n >= 0 && (n == 0 || (boxes && forces)), "hcl_set_body_force_regions: bad arguments");
  HC_REQUIRE(n <= HC_MAX_FORCE_REGIONS, "hcl_set_body_force_regions: more than HC_MAX_FORCE_REGIONS boxes");
  for (int k = 0; k < n; k++) {
    const int *b = boxes + 6 * k;
    HC_REQUIRE(b[0] <= b[1] && b[2] <= b[3] && b[4] <= b[5], "hcl_set_body_force_regions: empty box");
    for (int i = 0; i < 6; i++) L->regions.box[k][i] = b[i];
    for (int d = 0; d < 3; d++) L->regions.f[k][d] = forces[3 * k + d];
  }
  L->regions.n = n;
  return HC_OK;
}

int hcl_collide_stream_part(hc_lattice *L, int part) {
  HC_REQUIRE(L, "hcl_collide_stream_part: null lattice");
  HC_REQUIRE(part >= 0 && part <= 6, "hcl_collide_stream_part: part must be 0..6");
  HC_REQUIRE(part < 3 || L->nx >= 4, "hcl_collide_stream_part: parts 3, 4 and 6 need a slab of at least 4 planes");
  hc::ProfScope prof(hc::forked() ? hc::PK_COLLIDE_BESIDE : hc::PK_COLLIDE);
  int rc = HC_OK;
  if (part == 0) rc = launch_collide(L, 0, L->nx);
  else if (part == 1) rc = launch_collide(L, 1, L->nx - 2);
  else if (part == 2 || part == 5) rc = launch_collide(L, 0, 1, L->nx - 1, 1);       // both face planes in one launch
  else if (part == 3) rc = launch_collide(L, 2, L->nx - 4);
  else rc = launch_collide(L, 0, 2, L->nx - 2, 2);                                  // the two planes next to each face in one launch
  if (rc == HC_OK && (part == 0 || part == 2 || part == 4)) rc = hcl_zero_force_halos(L);
  return rc;
}

// The collide kernel zeroes the other-parity IBM force on the bulk planes; envelope copies of cells also spread onto the halo
// planes, which have to be cleared as well (one small launch; before hcl_step_end of the same step).  Parts 0, 2 and 4 of
// hcl_collide_stream_part do it themselves; after parts 5 / 6 (the slab schedule) the caller does, once its face message is away.
int hcl_zero_force_halos(hc_lattice *L) {
  HC_REQUIRE(L, "hcl_zero_force_halos: null lattice");
  if (L->n_slabs <= 1 || !L->ibm) return HC_OK;
  hipLaunchKernelGGL(zero_force_halo_kernel, dim3((unsigned)((L->plane + 255) / 256), (unsigned)(2 * HALO * 3), 1), dim3(256), 0, hc::stream(),
                     L->force[(L->fcur + 2) % 3], (long)L->npad, (long)L->xs, (int)L->plane, L->nx);
  HC_HIP(hipGetLastError());
  return HC_OK;
}

int hcl_step_end(hc_lattice *L) {
  HC_REQUIRE(L, "hcl_step_end: null lattice");
  L->cur ^= 1; L->fcur = (L->fcur + 1) % 3;
  L->halo_u_valid = false;   // velocities of a neighbour's face plane belong to the state that has just been replaced
  L->fepoch[L->fcur] = (uint8_t)(L->fepoch[L->fcur] % 255 + 1);   // the buffer spread will add to next gets a fresh epoch (1..255)
  return HC_OK;
}

int hcl_collide_stream(hc_lattice *L, int nsteps) {
  HC_REQUIRE(L, "hcl_collide_stream: null lattice");
  if (L->n_slabs > 1) return hcs::collide_stream_slab(L, nsteps);   // faces exchanged inside, beside the interior collide
  for (int s = 0; s < nsteps; s++) {
    int rc = hcl_collide_stream_part(L, 0);
    if (rc != HC_OK) return rc;
    hcl_step_end(L);
  }
  return HC_OK;
}

int hcl_download_populations(hc_lattice *L, double *f_aos) {
  HC_REQUIRE(L && f_aos, "hcl_download_populations: null pointer");
  if (L->n_slabs > 1) { const int rc = hcl_slab_refresh_halos(L, 2); if (rc != HC_OK) return rc; }   // the post-stream view pulls from the halo planes
  const size_t nd = (size_t)L->nx * L->plane * HC_Q;
  int rc = ensure_scratch(L, nd); if (rc != HC_OK) return rc;
  LatArgs a = make_args(L);
  hipLaunchKernelGGL(download_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, L->scratch);
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(f_aos, L->scratch, nd * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_upload_populations(hc_lattice *L, const double *f_aos) {
  HC_REQUIRE(L && f_aos, "hcl_upload_populations: null pointer");
  const size_t np = (size_t)L->plane * HC_Q, nd = (size_t)L->nx * np;
  int rc = ensure_scratch(L, nd + 2 * np); if (rc != HC_OK) return rc;
  double *bulk = L->scratch + np;   // scratch: plane -1, the nx planes of the slab, plane nx
  HC_HIP(hipMemcpyAsync(bulk, f_aos, nd * sizeof(double), hipMemcpyHostToDevice, hc::stream()));
  int lo_ok = 0, hi_ok = 0;
  if (L->n_slabs > 1) {
    // The stored value of a face node in a direction that crosses the face is the post-stream value of the NEIGHBOUR's first
    // plane: every rank hands its face planes to its neighbours (collective: all ranks of the run upload together, as
    // HemoCell::loadCheckPoint does)
    HC_REQUIRE(hcm::active(), "hcl_upload_populations: a slab of a multi-rank run needs the ranks connected (hc_comm_init) -- the call is collective");
    int lo, hi; hcm::neighbours(L->periodic[0] != 0, lo, hi);
    lo_ok = lo >= 0; hi_ok = hi >= 0;
    rc = hcm::exchange(hc::stream(), L->periodic[0] != 0, bulk, np * sizeof(double), bulk + (size_t)(L->nx - 1) * np, np * sizeof(double), L->scratch, np * sizeof(double),
                       bulk + nd, np * sizeof(double));
    if (rc != HC_OK) return rc;
    hcs::halos_stale(L);
  }
  LatArgs a = make_args(L);
  a.fout = L->f[L->cur];
  hipLaunchKernelGGL(upload_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, (const double *)bulk, lo_ok, hi_ok);
  HC_HIP(hipGetLastError());
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_download_rho_u(hc_lattice *L, double *rho, double *u) {
  HC_REQUIRE(L && rho && u, "hcl_download_rho_u: null pointer");
  if (L->n_slabs > 1) { const int rc = hcl_slab_refresh_halos(L, 2); if (rc != HC_OK) return rc; }
  const size_t n = (size_t)L->nx * L->plane;
  int rc = ensure_scratch(L, n * 4); if (rc != HC_OK) return rc;
  LatArgs a = make_args(L);
  hipLaunchKernelGGL(rho_u_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, L->scratch, L->scratch + n);
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(rho, L->scratch, n * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipMemcpyAsync(u, L->scratch + n, 3 * n * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_download_pi_neq(hc_lattice *L, double *pi) {
  HC_REQUIRE(L && pi, "hcl_download_pi_neq: null pointer");
  if (L->n_slabs > 1) { const int rc = hcl_slab_refresh_halos(L, 2); if (rc != HC_OK) return rc; }
  const size_t n = (size_t)L->nx * L->plane;
  int rc = ensure_scratch(L, n * 6); if (rc != HC_OK) return rc;
  LatArgs a = make_args(L);
  hipLaunchKernelGGL(pi_neq_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, L->scratch);
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(pi, L->scratch, 6 * n * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_download_ibm_force(hc_lattice *L, double *F) {
  HC_REQUIRE(L && F, "hcl_download_ibm_force: null pointer");
  const size_t n = (size_t)L->nx * L->plane;
  int rc = ensure_scratch(L, n * 3); if (rc != HC_OK) return rc;
  LatArgs a = make_args(L);
  hipLaunchKernelGGL(force_aos_kernel, plane_grid(L, L->nx), dim3(256), 0, hc::stream(), a, L->scratch);
  HC_HIP(hipGetLastError());
  HC_HIP(hipMemcpyAsync(F, L->scratch, 3 * n * sizeof(double), hipMemcpyDeviceToHost, hc::stream()));
  HC_HIP(hipStreamSynchronize(hc::stream()));
  return HC_OK;
}

int hcl_zero_ibm_force(hc_lattice *L) {
  HC_REQUIRE(L, "hcl_zero_ibm_force: null lattice");
  HC_HIP(hipMemsetAsync(L->force[L->fcur], 0, L->npad * 3 * sizeof(double), hc::stream()));
  return HC_OK;
}

int hcl_fluid_stats(hc_lattice *L, int what, double out[3], long *n_nodes) {
  HC_REQUIRE(L && out && n_nodes && what >= 0 && what <= 2, "hcl_fluid_stats: bad arguments");
  if (L->n_slabs > 1 && what == 0) { const int rc0 = hcl_slab_refresh_halos(L, 1); if (rc0 != HC_OK) return rc0; }   // velocities on the face planes pull from the halos
  int rc = ensure_scratch(L, (size_t)STAT_BLOCKS * 4); if (rc != HC_OK) return rc;
  LatArgs a = make_args(L);
  hipLaunchKernelGGL(fluid_stats_kernel, dim3(STAT_BLOCKS), dim3(256), 0, hc::stream(), a, what, L->scratch);
  HC_HIP(hipGetLastError());
  return hc::stat_finish(L->scratch, out, n_nodes);
}

size_t hcl_halo_doubles(const hc_lattice *L, int width) {
  if (!L) return 0;
  return (size_t)(width == 1 ? 5 : 14 + 5) * L->plane;
}

// width 1 (every step): the 5 populations that cross the face.  width 2 (before interpolation): everything the
// neighbour needs to evaluate node velocities on its first halo plane.  S(x, i) = P(x - c_i, i) there pulls the 9 populations
// with c_x = 0 from the face plane itself, the 5 moving towards the neighbour from the plane behind it, and the 5 moving
// away from the neighbour out of the neighbour's own first plane; the neighbour's next collide pulls the 5 moving towards
// it from the face plane.  So 14 populations of the face plane and 5 of the plane behind travel: 19 planes, not 24.
static int halo_copy(hc_lattice *L, int side, int width, double *buf, int to_buf, int next = 0) {
  HC_REQUIRE(L && buf, "hcl_halo: null pointer");
  HC_REQUIRE((side == 0 || side == 1) && (width == 1 || width == 2), "hcl_halo: side must be 0/1 and width 1/2");
  HC_REQUIRE(L->nx >= 2 * width, "hcl_halo: slab thinner than the halo");
  static const int cxm[5] = {1, 4, 5, 6, 7};        // c_x = -1
  static const int cxp[5] = {10, 13, 14, 15, 16};   // c_x = +1
  HaloArgs h;
  h.f = L->f[next ? 1 - L->cur : L->cur]; h.buf = buf; h.buf2 = nullptr; h.n_first = 0x7fffffff; h.npad = (long)L->qstride; h.xs = (long)L->xs; h.plane = (int)L->plane; h.to_buf = to_buf; h.n = 0;
  // the populations that travel towards -x (cxm) leave through the low face and arrive in the low neighbour's high
  // halo; those towards +x (cxp) the other way round
  const int *moving = to_buf ? (side == 0 ? cxm : cxp) : (side == 0 ? cxp : cxm);
  // plane next to the face (bulk side when packing, halo side when unpacking) and the one behind it
  const int near = to_buf ? (side == 0 ? HALO : HALO + L->nx - 1) : (side == 0 ? HALO - 1 : HALO + L->nx);
  const int far = to_buf ? (side == 0 ? HALO + 1 : HALO + L->nx - 2) : (side == 0 ? HALO - 2 : HALO + L->nx + 1);
  if (width == 1) {
    for (int k = 0; k < 5; k++) { h.pop[h.n] = moving[k]; h.xp[h.n] = near; h.n++; }
  } else {
    static const int cx[HC_Q] = HC_CX;
    for (int q = 0; q < HC_Q; q++) {
      bool towards = false;
      for (int k = 0; k < 5; k++) towards = towards || moving[k] == q;
      if (cx[q] == 0 || towards) { h.pop[h.n] = q; h.xp[h.n] = near; h.n++; }
    }
    for (int k = 0; k < 5; k++) { h.pop[h.n] = moving[k]; h.xp[h.n] = far; h.n++; }
  }
  hipLaunchKernelGGL(halo_copy_kernel, dim3((unsigned)((L->plane + 255) / 256), (unsigned)h.n, 1), dim3(256), 0, hc::stream(), h);
  HC_HIP(hipGetLastError());
  return HC_OK;
}
// the width-1 message of both faces in one launch (either buffer may be null: a slab at a non-periodic end of the domain)
static int halo_copy_both(hc_lattice *L, double *buf_lo, double *buf_hi, int to_buf, int next) {
  HC_REQUIRE(L, "hcl_halo: null pointer");
  HC_REQUIRE(L->nx >= 2, "hcl_halo: slab thinner than the halo");
  static const int cxm[5] = {1, 4, 5, 6, 7};        // c_x = -1
  static const int cxp[5] = {10, 13, 14, 15, 16};   // c_x = +1
  HaloArgs h;
  h.f = L->f[next ? 1 - L->cur : L->cur]; h.npad = (long)L->qstride; h.xs = (long)L->xs; h.plane = (int)L->plane; h.to_buf = to_buf; h.n = 0;
  h.buf = buf_lo; h.buf2 = buf_hi; h.n_first = buf_lo ? 5 : 0;
  for (int side = 0; side < 2; side++) {
    if (!(side == 0 ? buf_lo : buf_hi)) continue;
    const int *moving = to_buf ? (side == 0 ? cxm : cxp) : (side == 0 ? cxp : cxm);
    const int near = to_buf ? (side == 0 ? HALO : HALO + L->nx - 1) : (side == 0 ? HALO - 1 : HALO + L->nx);
    for (int k = 0; k < 5; k++) { h.pop[h.n] = moving[k]; h.xp[h.n] = near; h.n++; }
  }
  if (h.n == 0) return HC_OK;
  hipLaunchKernelGGL(halo_copy_kernel, dim3((unsigned)((L->plane + 255) / 256), (unsigned)h.n, 1), dim3(256), 0, hc::stream(), h);
  HC_HIP(hipGetLastError());
  return HC_OK;
}
int hcl_halo_pack_both(hc_lattice *L, double *dev_lo, double *dev_hi, int next) { return halo_copy_both(L, dev_lo, dev_hi, 1, next); }
int hcl_halo_unpack_both(hc_lattice *L, const double *dev_lo, const double *dev_hi) { return halo_copy_both(L, (double *)dev_lo, (double *)dev_hi, 0, 0); }
int hcl_halo_pack(hc_lattice *L, int side, int width, double *dev_buf) { return halo_copy(L, side, width, dev_buf, 1); }
int hcl_halo_pack_next(hc_lattice *L, int side, int width, double *dev_buf) { return halo_copy(L, side, width, dev_buf, 1, 1); }
int hcl_halo_unpack(hc_lattice *L, int side, int width, const double *dev_buf) { return halo_copy(L, side, width, (double *)dev_buf, 0); }

double hcl_mlups_bytes_per_node(const hc_lattice *L) {
  // 19 reads + 19 writes of fp64 populations and 1 mask byte; with membrane cells bound to the lattice also
  // 3 reads of the IBM force and 3 zeroing writes (304 B fluid only, 353 B coupled)
  return 19 * 8 * 2 + 1 + ((L && L->ibm) ? 3 * 8 * 2 : 0);
}

}  // extern "C"
