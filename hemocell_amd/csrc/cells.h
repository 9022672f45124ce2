// Internal declarations shared by the translation units that work on membrane vertices
// (cells.hip: storage and stepping, ibm.hip: spread / interpolate, mechanics.hip: membrane models,
// exchange.hip: slab envelopes and statistics, repulsion.hip: vertex-vertex and boundary repulsion).
//
// Layout (HBM): vertices are a structure of arrays pos/vel/frc[3][n], cell major (a cell's nv vertices are
// contiguous, cells of one type contiguous in a fixed-capacity region), so a wavefront touches 64 consecutive
// doubles per component and a mechanics workgroup stages one whole cell in LDS with coalesced loads.
#pragma once
#include "common.h"
#include "mesh.h"
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
  std::vector<double> hrep[8];   // force_repulsion staging (filled only while a repulsion is enabled)
  std::vector<long> hids[8];
  std::vector<int> htag[8];              // per cell: 0 complete, 2 incomplete (a gone cell never reaches the host staging)
  std::vector<unsigned char> hdead[8];   // per vertex: 1 = this particle was removed (reference deletion mode)
  bool host_dirty = false;       // host staging newer than device
  long nverts = 0, cap = 0;      // live vertices (all types); allocated vertex capacity
  long ncells[8] = {0};
  long capc[8] = {0};            // per-type capacity in cells (device regions are fixed-size per type)
  long first[8] = {0};           // first vertex of each type's region on the device
  long cell0[8] = {0};           // first cell slot of each type's region
  double *pos[3] = {nullptr, nullptr, nullptr}, *vel[3] = {nullptr, nullptr, nullptr}, *frc[3] = {nullptr, nullptr, nullptr};
  // Deletion lives on the device (core/hemoCellParticleField.cpp:566-588, :304-321 without a host round trip):
  // d_tag per cell slot: 0 complete, 1 gone (every kernel skips the cell; the slot is reclaimed at the next
  // compaction), 2 incomplete (the reference's state after removeParticles(1) took single particles out: no
  // mechanics, forces zeroed at the next material step, the remaining particles are still spread / interpolated /
  // advanced); d_vdead per vertex: 1 = removed particle.
  int *d_tag = nullptr;          // [tag_cap] all types, slot order
  long tag_cap = 0;
  unsigned char *d_vdead = nullptr;   // [cap]
  int *h_ntag_dev = nullptr;     // device view of h_ntag
  int *h_ntag = nullptr;         // pinned host copy of the counters {cells gone and not yet compacted, cells made incomplete}
  int *d_ntag = nullptr;         // device counters [2]
  hipEvent_t ntag_ev = nullptr;  // completion of an asynchronous counter read (hc_iterate polls it, never waits)
  bool ntag_pending = false, maybe_tagged = false;
  int del_mode = HC_DELETE_PARTICLE;
  int *d_vert_cell = nullptr;    // [cap] cell slot of every vertex
  // vertex-vertex repulsion (core/hemoCellParticleField.cpp:677-743); arrays exist only once it is enabled
  double *rep[3] = {nullptr, nullptr, nullptr};
  int rep_enabled = 0, rep_timescale = 1; double rep_const = 0, rep_cutoff = 0;
  // boundary particles (core/hemoCellParticleField.cpp:865-918): flag map of the wall nodes that repel vertices
  int brep_enabled = 0, brep_timescale = 1; double brep_const = 0, brep_cutoff = 0; uint8_t *d_bflag = nullptr;
  bool rep_on() const { return rep_enabled || brep_enabled; }
  unsigned int *d_keys[2] = {nullptr, nullptr}; int *d_vals[2] = {nullptr, nullptr}; void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0; long sort_cap = 0;
  // staged slot lists (envelope exchange, interpolate_cells, remove): pinned host block -> device block, stream ordered;
  // the event guards the pinned block against being rewritten while its copy is still in flight
  // staged slot lists: 0, 1 envelope exchange; 2 reproducible spread; 3 + 2 * type + half: the two halves of a slab's velocity update
  int *d_iscratch[19] = {nullptr}, *h_iscratch[19] = {nullptr}; hipEvent_t iscratch_ev[19] = {nullptr};
  // asynchronous cell extents (hcp_cell_extents_begin / _end): device block, pinned host block and event per type
  double *d_ext[8] = {nullptr}, *h_ext[8] = {nullptr}; long ext_cap[8] = {0}, ext_n[8] = {0}; hipEvent_t ext_done[8] = {nullptr}; bool ext_pending[8] = {false};
  // staging of hcp_add_vertex_force (called every iteration by the stretch drivers): pinned host block + device block
  // [n indices | 3n force components], grown on demand; the event guards the pinned block against reuse in flight
  char *h_vf = nullptr, *d_vf = nullptr; size_t vf_cap = 0; hipEvent_t vf_done = nullptr;
  size_t iscratch_cap[19] = {0};
  // reproducible spread (hc_set_reproducible_spread): (node, entry) pairs of every (particle, stencil node), both sort buffers,
  // and the three force components of every entry
  unsigned int *det_keys[2] = {nullptr, nullptr}; int *det_vals[2] = {nullptr, nullptr}; double *det_val[3] = {nullptr, nullptr, nullptr};
  void *det_tmp = nullptr; size_t det_tmp_bytes = 0; long det_cap = 0;
  double *d_stat = nullptr, *h_stat = nullptr;   // [STAT_BLOCKS][4] partials of the statistics reductions, device and pinned host
  double *d_info = nullptr, *h_info = nullptr; size_t info_cap = 0;   // scratch of the information calls (hcp_cell_info, hcp_mechanics_components, statistics)
  // slab runs: ids of cells this rank compacted away on its own (a host query between two envelope synchronisations:
  // counts, output, deleteIncompleteCells) -- the next synchronisation tells the other holders, or their copy would come
  // back as a fresh complete cell
  std::vector<long> slab_gone[8];
  // slab runs: a cell that reaches within e_share lattice units of a face is replicated on the neighbour (the cell-granular
  // form of the reference's particle envelope, core/hemoCell.cpp:139; hcp_set_envelope).  env_viol counts the copies that
  // arrived too late -- a particle already sat on the receiving slab's side when its cell first got there (mapped pinned
  // memory, written by unpack_cells_kernel)
  double e_share = 4.0;
  int *h_env_viol = nullptr, *d_env_viol = nullptr;
  std::vector<long> slab_rejected;   // (type, cell id) pairs hcp_add_cell rejected at a wall on this slab, until hcp_slab_sync_placement
  long n_deleted = 0;              // cells removed entirely
  long n_particles_deleted = 0;    // single particles removed (reference mode), including those of cells removed later
};

namespace hcc {

constexpr double E_SHARE_DEFAULT = 4.0, E_SHARE_MIN = 2.0;   // slab runs, hc_cells::e_share: default, and the least that still covers the IBM stencil

// ----------------------------------------------------------------------------
// lattice view for the IBM kernels
struct LatView {
  const uint8_t *mask;
  int nx, ny, nz; long plane; long npad;   // plane: elements from x-plane to x-plane (hc_lattice::xs)
  int x0;                 // global x of local plane 0
  int wrap_x, halo_x;     // single periodic slab: wrap; multi slab: one halo plane is addressable
  int per_y, per_z;
  int nx_global;
  uint8_t *dirty; uint8_t epoch;   // dirty map of the force buffer spread adds to (see common.h)
  const uint8_t *wallbrick; int nby, nbz;   // wall proximity per 8^3 brick (see common.h)
  const double *halo_u[2];                  // slab runs: owner-evaluated node velocities of the first halo planes (null: gather them here)
  int ny_nz;                                // nodes of one x-plane (stride of the three components of halo_u)
};

inline LatView make_view(const hc_lattice *L) {
  LatView v;
  v.mask = L->mask; v.nx = L->nx; v.ny = L->ny; v.nz = L->nz; v.plane = (long)L->xs; v.npad = (long)L->npad;
  v.x0 = L->x0; v.wrap_x = (L->n_slabs == 1 && L->periodic[0]) ? 1 : 0; v.halo_x = L->n_slabs > 1 ? 1 : 0;
  v.per_y = L->periodic[1]; v.per_z = L->periodic[2]; v.nx_global = L->nx_global;
  v.dirty = L->fdirty[L->fcur]; v.epoch = L->fepoch[L->fcur];
  v.wallbrick = L->wallbrick; v.nby = L->nby; v.nbz = L->nbz;
  v.halo_u[0] = L->halo_u_valid ? L->halo_u[0] : nullptr; v.halo_u[1] = L->halo_u_valid ? L->halo_u[1] : nullptr; v.ny_nz = (int)L->plane;
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

struct VertArrays { double *p[3], *v[3], *f[3], *r[3]; unsigned char *dead; int *tag; };   // r: repulsion force arrays or null; dead / tag offset to the type

// ----------------------------------------------------------------------------
template <typename T>
inline int upload_vec(T **dst, const std::vector<T> &src) {
  *dst = nullptr;
  const size_t n = src.size() ? src.size() : 1;
  HC_HIP(hipMalloc((void **)dst, n * sizeof(T)));
  if (src.size()) HC_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return HC_OK;
}
template <size_t N>
inline std::vector<int> flatten(const std::vector<std::array<long, N>> &v) {
  std::vector<int> o; o.reserve(v.size() * N);
  for (auto &a : v) for (long x : a) o.push_back((int)x);
  return o;
}

// storage management (cells.hip)
int free_device_arrays(hc_cells *C);
int sync_to_device(hc_cells *C);   // host staging -> device arrays when the host copy is newer
int sync_to_host(hc_cells *C);     // device arrays -> host staging before host-side edits
// make the host's view of the cell set current: reads the deletion counters (one small blocking copy, and only when an
// advance ran since the last time) and compacts gone cells away.  Every entry point that reports or edits cells calls it.
int settle(hc_cells *C);
VertArrays vert_arrays(hc_cells *C, int t);
// stage a small host int array on the device in a persistent scratch slot (through a pinned block; stream ordered)
int stage_ints(hc_cells *C, int which, int **d, const int *h, int n);
void host_append_state(hc_cells *C, int type, long cell_id);
int interpolate_cells_staged(hc_cells *C, int type, const int *slots, int n, int which);   // ibm.hip: hcp_interpolate_cells through staging slot `which`

}  // namespace hcc
using namespace hcc;
