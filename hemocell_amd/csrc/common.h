// Internal declarations shared by the HIP translation units of libhemocell_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/hemocell_amd.h"

namespace hc {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
hipStream_t stream();
hipStream_t comm_stream();   // third stream: the transfers of slab runs
// fork-join onto the library's side stream: fork() makes the side stream wait for everything enqueued so far,
// route(1) sends the following launches there, route(0) back, join() makes the main stream wait for them
int fork();
void route(int side);
int join();
bool forked();   // between fork() and join()

#define HC_HIP(call)                                                         \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) return hc::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define HC_REQUIRE(cond, msg)                 \
  do {                                        \
    if (!(cond)) {                            \
      hc::set_error(std::string(msg));        \
      return HC_ERR_ARG;                      \
    }                                         \
  } while (0)

// per-kernel hipEvent timing (hc_profile_*)
enum ProfKernel { PK_COLLIDE = 0, PK_SPREAD, PK_INTERP, PK_ADVANCE, PK_MECH, PK_COLLIDE_BESIDE, PK_COUNT };   // _BESIDE: collide launches with side-stream work next to them
struct ProfScope {
  int k; bool on;
  hipEvent_t a, b;
  explicit ProfScope(int kernel);
  ~ProfScope();
};

// D3Q19, Palabos ordering: opposite of i (1..9) is i+9.
// (patch/palabos.patch:491-498)
#define HC_CX {0, -1, 0, 0, -1, -1, -1, -1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0}
#define HC_CY {0, 0, -1, 0, -1, 1, 0, 0, -1, -1, 0, 1, 0, 1, -1, 0, 0, 1, 1}
#define HC_CZ {0, 0, 0, -1, 0, 0, -1, 1, -1, 1, 0, 0, 1, 0, 0, 1, -1, 1, -1}

constexpr int HALO = 2;  // x-halo planes on each side of a slab

// Deterministic min / max / sum / count reduction used by the statistics entry points: a fixed grid of STAT_BLOCKS
// workgroups, each thread strides over the items, waves combine by shuffles, workgroups write one partial each and
// the host folds the partials in index order.
constexpr int STAT_BLOCKS = 512;
struct StatAcc { double mn, mx, sum; long n; };
__device__ __forceinline__ void stat_add(StatAcc &a, double v) { a.mn = v < a.mn ? v : a.mn; a.mx = v > a.mx ? v : a.mx; a.sum += v; a.n++; }
__device__ __forceinline__ void stat_block_store(StatAcc a, double *partial /*[STAT_BLOCKS][4]*/) {
  __shared__ double s_red[4][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double mn = __shfl_xor(a.mn, off), mx = __shfl_xor(a.mx, off), sm = __shfl_xor(a.sum, off);
    const long n = __shfl_xor(a.n, off);
    a.mn = mn < a.mn ? mn : a.mn; a.mx = mx > a.mx ? mx : a.mx; a.sum += sm; a.n += n;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[w][0] = a.mn; s_red[w][1] = a.mx; s_red[w][2] = a.sum; s_red[w][3] = (double)a.n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mn = s_red[0][0], mx = s_red[0][1], sm = s_red[0][2], n = s_red[0][3];
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) { mn = s_red[k][0] < mn ? s_red[k][0] : mn; mx = s_red[k][1] > mx ? s_red[k][1] : mx; sm += s_red[k][2]; n += s_red[k][3]; }
    double *o = partial + 4 * blockIdx.x;
    o[0] = mn; o[1] = mx; o[2] = sm; o[3] = n;
  }
}
// folds the partials (device pointer) on the host; out = {min, max, sum}
int stat_finish(const double *d_partial, double out[3], long *n, double *h_pinned = nullptr);   // h_pinned: [STAT_BLOCKS][4] pinned landing block, or a stack copy

}  // namespace hc

namespace hc {
// setExternalVector on part of the domain (cases/kolmogorovFlow/kolmogorovFlow.cpp:136-140): up to HC_MAX_FORCE_REGIONS boxes
// (inclusive global node ranges) whose nodes carry f instead of the uniform body force; a later box overrides an earlier one
struct BodyRegions { int n; int box[HC_MAX_FORCE_REGIONS][6]; double f[HC_MAX_FORCE_REGIONS][3]; };
#ifdef __HIPCC__
__device__ __forceinline__ void region_force(const BodyRegions &r, int xg, int y, int z, double &bx, double &by, double &bz) {
  for (int k = 0; k < r.n; k++) {
    const int *b = r.box[k];
    if (xg >= b[0] && xg <= b[1] && y >= b[2] && y <= b[3] && z >= b[4] && z <= b[5]) { bx = r.f[k][0]; by = r.f[k][1]; bz = r.f[k][2]; }
  }
}
#endif
}  // namespace hc

struct hc_lattice {
  int nx, ny, nz;        // local bulk dims
  int periodic[3];       // global periodicity
  int x0, nx_global, n_slabs;
  double omega;
  size_t plane;          // ny*nz nodes of one x-plane
  size_t xs;             // elements from x-plane to x-plane: plane, or plane + 8 rows of padding (see hcl_create)
  size_t npad;           // (nx+2*HALO)*xs
  size_t qstride;        // doubles from population q to q+1 of the same node: npad + padding (see hcl_create)
  double *f[2];          // [19][npad] post-collision populations (fBar), ping-pong
  int cur;               // f[cur] is read by the next collide
  double *force[3];      // [3][npad] IBM force accumulators, rotated: fcur -> (fcur+1)%3 every step
  int fcur;              // force[fcur] is the one spread adds to / collide reads; force[(fcur+2)%3] is the previous
                         // step's (what the interpolation after a collide reads, and what the NEXT collide zeroes);
                         // force[(fcur+1)%3] is already clean, so the spread of the next step may run beside this collide
  int ibm;               // set once membrane cells are bound (hcp_create): collide then reads/zeroes the force buffers
  // dirty map of the IBM force buffers: one byte per group of 16 consecutive nodes (one 128-byte line of a
  // force component) holding the epoch in which spread last touched the group.  The collide kernel reads /
  // zeroes a group only when its byte equals the buffer's current epoch, so untouched lines cost no traffic.
  // Epochs are never cleared (no races); an aliased stale epoch only causes a harmless extra read / zeroing.
  uint8_t *fdirty[3];
  uint8_t fepoch[3];
  // one byte per 8 x 8 x 8 brick of the padded lattice: 1 = the brick holds a non-fluid node or touches a face that stencils
  // cannot cross (the IBM kernels skip the mask look-ups for cells whose tile meets no such brick)
  uint8_t *wallbrick; int nbx, nby, nbz;
  uint8_t *mask;         // [npad]
  std::vector<uint8_t> hmask;  // host copy (cell placement tests against it)
  double body[3];
  hc::BodyRegions regions;   // boxes with their own body force (hcl_set_body_force_regions), global node coordinates
  double wall_u[4][3];   // velocities of the moving-wall mask classes 3..6
  // active-node map of the collide kernel: per padded plane and row, the z-span that holds every node
  // which is not an inert solid, flattened so that a launch only creates threads for those spans
  int *row_z0, *row_cum, *blk_row;   // [NX*ny], [NX*(ny+1)], [NX*(nblk+1)]
  int nblk, max_active;
  double *scratch;       // download staging
  size_t scratch_doubles;
  // slab runs: node velocities u = j/rho + F/2 of the two neighbours' face planes, evaluated there by their owner after the
  // last collide (slab.hip); [side][3][plane].  The interpolation reads them for stencil nodes on the first halo plane
  // instead of gathering 19 populations there.  Valid from the exchange until the next hcl_step_end.
  double *halo_u[2] = {nullptr, nullptr};
  bool halo_u_valid = false;
};

// slab.hip: HemoCell::iterate / collideAndStream on one x-slab of a multi-GPU run (halo and envelope exchange inside)
namespace hcs {
int iterate_slab(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit);
int collide_stream_slab(hc_lattice *L, int nsteps);
void lattice_destroyed(hc_lattice *L);
void halos_stale(hc_lattice *L);   // the populations were replaced from outside: the halo planes have to be fetched again
void set_overlap(int on);
}
