/*
 * hemocell_amd.h -- C ABI of the MI355X-native IB-LBM hot path.
 *
 * Drop-in boundary for the one data-parallel path of HemoCell:
 * the D3Q19 Guo-BGK collide-stream, the phi2 immersed-boundary spread /
 * interpolate, the Euler vertex advance and the rbcHighOrderModel /
 * pltSimpleModel membrane forces.  Every entry point names the reference
 * interface it replaces (file:line relative to the HemoCell tree).  Plain
 * pointers and sizes only; no C++/torch types cross this boundary.  All
 * functions return 0 on success and a non-zero code on failure, with the
 * message available from hc_last_error() (the reference logs and exit(1)s,
 * e.g. core/hemoCell.cpp:75-78; a library must not, so the host layer above
 * this ABI does that).  The library is HIP-only: there is no CPU fallback and
 * every call fails loudly when no gfx950 device is usable.
 *
 * Threading / process model: one process per GPU, one hc_* context per process
 * (core/hemoCell.cpp:75-79 has the same rule).
 */
#ifndef HEMOCELL_AMD_H
#define HEMOCELL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HC_OK 0
#define HC_ERR_HIP 1
#define HC_ERR_ARG 2
#define HC_ERR_STATE 3

#define HC_Q 19

typedef struct hc_lattice hc_lattice;   /* one x-slab of the D3Q19 lattice on this GPU          */
typedef struct hc_celltype hc_celltype; /* CommonCellConstants + moduli of one cell type         */
typedef struct hc_cells hc_cells;       /* all membrane vertices held by this GPU (SoA)          */

/* ------------------------------------------------------------------ runtime */
const char *hc_last_error(void);
/* selects the HIP device, checks it is gfx950; replaces plb::plbInit (core/hemoCell.cpp:80-86).  A multi-rank run passes
 * its LOCAL rank modulo hc_device_count (one process per GPU); hc_comm_init_env does that by itself. */
int hc_init(int device);
int hc_device_count(int *count);
/* run every kernel of this library on an existing hipStream_t (e.g. the host framework's current stream); NULL = library stream */
int hc_set_stream(void *hip_stream);
int hc_synchronize(void);
/* fork-join onto a second stream owned by the library, for phase-by-phase callers (slab runs) that want what
 * hc_iterate does on its own: hc_fork() makes the side stream wait for everything enqueued so far, hc_route(1) sends the
 * launches of the following calls there (hc_route(0): back to the main stream), hc_join() makes the main stream wait
 * for them.  Between two velocity updates advance, mechanics and the next spread run beside the collide this way. */
int hc_fork(void);
int hc_route(int side);
int hc_join(void);
int hc_side_stream(void **hip_stream);  /* the side stream's hipStream_t, for a host framework that has to enqueue its own work (transfers) there */
/* hc_iterate overlaps those phases by itself (default); 0 = strictly one stream (A/B measurements) */
int hc_set_overlap(int on);
/* per-launch hipEvent timing of the dominant kernel (helper/profiler.h:46-77 "collideAndStream" timer) */
int hc_profile_enable(int on);
int hc_profile_read(const char *kernel, double *total_ms, long *launches); /* "collide_stream" (every launch) = "collide_stream_alone" + "collide_stream_beside" (launches with advance / spread on the side stream next to them), "ibm_spread", "ibm_interpolate", "advance", "mechanics" */
int hc_profile_reset(void);
/* identifies the build of the dominant kernel: first 16 hex digits of the SHA-256 of csrc/lattice.hip (a committed PMC
 * traffic figure is only quoted next to a timing when it was measured on the same kernel) */
const char *hc_build_tag(void);
/* device-to-device copy of `bytes` on the library stream, `repeats` times: read + written GB/s of the GPU in hand */
int hc_measure_copy_bandwidth(size_t bytes, int repeats, double *gbytes_per_s);
/* Reproducible force spreading.  The reference adds the particles' forces to the lattice particle by particle, in storage
 * order (core/hemoCellParticleField.cpp:841-863): a run repeats bit for bit.  The default kernels here add with fp64 atomics in
 * whatever order the hardware takes them, so two runs of one input differ in the last bits.  on = 1 selects the gather form:
 * every (particle, stencil node) contribution is keyed by its node, sorted stably in (cell type, cell id, vertex id) order and
 * summed by one thread per node -- identical bits from run to run and from one slab decomposition to another, at several times
 * the cost of the atomic kernel.  Also selected by HEMOCELL_REPRODUCIBLE_SPREAD=1 in the environment. */
int hc_set_reproducible_spread(int on);
/* A/B switch: 1 = per-vertex IBM kernels with direct global atomics instead of the LDS-tiled per-cell kernels */
int hc_debug_ibm_per_vertex(int on);
int hc_debug_force_plane_padding(int on); /* tests / A-B runs: lattices created afterwards get the padded x-plane stride whatever their size (1), never (-1), by size (0, default) */

/* ---------------------------------------------------------------- ranks (one process per GPU of one node)
 * Replaces what plb::plbInit / MPI give the reference (core/hemoCell.cpp:80-86, the MPI calls of SURVEY.md section 2.3):
 * a rank, a world size and neighbour exchange.  Two layers:
 *   control plane  a TCP mesh between the ranks (loopback / MASTER_ADDR): bootstrap, barrier, reductions of a few
 *                  scalars at output cadence, agreement on the placed cells.  Never on the stepping path.
 *   data plane     HC_TRANSPORT_RCCL: ncclSend / ncclRecv between x-neighbours, grouped, on the library's side stream
 *                  (RCCL over xGMI; /opt/rocm/lib/librccl.so.1 is loaded on demand); HC_TRANSPORT_TCP: the same messages
 *                  staged through pinned host memory and the mesh -- for ranks that SHARE a GPU (RCCL refuses two ranks
 *                  on one device), i.e. rehearsals and tests on a one-GPU box.
 * hc_comm_init_env reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torch.distributed.run's names; also
 * OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK and PMI_RANK / PMI_SIZE of an mpirun), HEMOCELL_PORT (default MASTER_PORT + 1017)
 * and HEMOCELL_TRANSPORT = rccl | tcp (unset: HC_TRANSPORT_AUTO), selects the device LOCAL_RANK modulo the device count (hc_init) and
 * connects.  Without those variables it is a one-rank world and does nothing. */
#define HC_TRANSPORT_NONE 0
#define HC_TRANSPORT_RCCL 1
#define HC_TRANSPORT_TCP 2
#define HC_TRANSPORT_AUTO 3   /* RCCL when every rank can set it up and a ring self-test completes; otherwise TCP, announced on stderr */
int hc_comm_init_env(void);
/* explicit form; init_device != 0: also hc_init(local_rank % device count).  transport HC_TRANSPORT_NONE builds the control
 * plane only (host-side use without a GPU: the CPU tests of the mesh). */
int hc_comm_init(int rank, int world, int local_rank, const char *master_addr, int port, int transport, int init_device);
int hc_comm_finalize(void);
int hc_comm_info(int *rank, int *world, int *transport);
int hc_comm_barrier(void);
/* v[n] <- reduction over all ranks, folded in rank order on rank 0 (deterministic); op: 0 sum, 1 min, 2 max */
int hc_comm_allreduce(double *v, int n, int op);
int hc_comm_bcast(void *buf, size_t bytes, int root);
/* all[r * bytes ..] <- rank r's block, on every rank (the reference's HemoCellGatheringFunctional, core/hemoCellFunctional.h:101-112) */
int hc_comm_allgather(const void *mine, size_t bytes, void *all);
/* neighbour exchange of HOST buffers over the control plane with the same routing rule as the data plane (my low-face
 * message is the low neighbour's high-halo message; the ring closes when periodic != 0): what the CPU tests check */
int hc_comm_exchange_host(int periodic, const void *send_lo, size_t n_lo, const void *send_hi, size_t n_hi,
                          void *recv_lo, size_t m_lo, void *recv_hi, size_t m_hi);
/* counters of the slab schedule since the last reset (reset != 0 clears them): out[0..7] = envelope records sent, new
 * copies created, copies dropped, cells deleted at a wall, iterations, host seconds spent enqueueing them, host seconds
 * waiting for the id headers, velocity-update steps */
int hc_slab_stats(hc_lattice *L, double out[8], int reset);

/* ------------------------------------------------------------------ lattice */
/* MultiBlockLattice3D<T,DESCRIPTOR>(nx,ny,nz, new GuoExternalForceBGKdynamics(omega))
 * (examples/pipeflow/pipeflow.cpp:66-71).  nx is the LOCAL slab thickness;
 * x-halos of width 2 are allocated on both sides.  periodic[0] with
 * n_slabs==1 wraps in-kernel; with n_slabs>1 the caller exchanges halos
 * (hcl_halo_pack / hcl_halo_unpack).  x0 = global x of local plane 0. */
int hcl_create(hc_lattice **out, int nx, int ny, int nz, const int periodic[3], double omega,
               int x0, int nx_global, int n_slabs);
int hcl_destroy(hc_lattice *L);
/* defineDynamics(lattice, flagMatrix, bbox, new BounceBack(1.), 0) (examples/pipeflow/pipeflow.cpp:73):
 * mask[node]=1 -> full-way bounce-back / isBoundary (3..6: moving wall classes, see hcl_set_wall_velocity); node = z + nz*(y + ny*x_local), x_local in [-2, nx+2)
 * i.e. the array holds (nx+4)*ny*nz bytes including the halo planes. */
int hcl_set_mask(hc_lattice *L, const uint8_t *mask_with_halo);
/* HemoCell::latticeEquilibrium(rho,u) + lattice->initialize() (core/hemoCell.cpp:129-133) */
int hcl_init_equilibrium(hc_lattice *L, double rho, const double u[3]);
/* setExternalVector(lattice, bbox, forceBeginsAt, F) (core/hemoCell.cpp:369-371, examples/pipeflow/pipeflow.cpp:144-146):
 * the uniform driving force; the per-node IBM force is kept separately and zeroed by the collide kernel */
int hcl_set_body_force(hc_lattice *L, const double F[3]);
/* setExternalVector on PART of the domain (cases/kolmogorovFlow/kolmogorovFlow.cpp:136-140: +F on one half of the box, -F on
 * the other): n <= HC_MAX_FORCE_REGIONS boxes, boxes[k] = {x0, x1, y0, y1, z0, z1} inclusive GLOBAL node ranges, whose nodes
 * carry forces[k] instead of the uniform body force; where boxes overlap the later one holds, as with consecutive
 * setExternalVector calls.  n = 0 removes them.  Evaluated inside the kernels from these few numbers: no force field in HBM. */
#define HC_MAX_FORCE_REGIONS 4
int hcl_set_body_force_regions(hc_lattice *L, int n, const int *boxes, const double *forces);
/* setBoundaryVelocity(lattice, box, u) on nodes flagged by setVelocityConditionOnBlockBoundaries
 * (helper/hemocellInit.hh:71-86): mask classes 3..6 are moving no-slip walls with velocity u (full-way
 * bounce-back + Ladd momentum term; stand-in for Palabos' regularised boundary, which is not available) */
int hcl_set_wall_velocity(hc_lattice *L, int wall_class, const double u[3]);
/* lattice->collideAndStream() (core/hemoCell.cpp:317), n times (fluid-only stepping).  On a slab of a multi-rank run
 * (n_slabs > 1, hc_comm_init* done) the faces are exchanged inside, interior planes colliding meanwhile. */
int hcl_collide_stream(hc_lattice *L, int nsteps);
/* bring the x-halo planes of a slab up to date (width 1 or 2, see hcl_halo_doubles) through the data plane; the
 * download / statistics entry points do it by themselves */
int hcl_slab_refresh_halos(hc_lattice *L, int width);
/* one collide-stream of this slab; halos must be current. part: 0 = all planes, 1 = interior planes
 * (those that do not read halo data), 2 = the two face planes; or 3 = planes 2..nx-3, 4 = the two planes next to each
 * face (what the node velocities of a face plane are evaluated from, hcl_face_velocity_pack, so that the messages of a velocity
 * update can travel while part 3 runs).
 * hcl_step_end() flips the buffers. */
int hcl_collide_stream_part(hc_lattice *L, int part);
/* parts 5 and 6: as 2 and 4, but the halo planes of the IBM force buffer being retired are left for the caller to clear with
 * hcl_zero_force_halos (before hcl_step_end) -- the slab schedule sends its face message first */
int hcl_zero_force_halos(hc_lattice *L);
int hcl_step_end(hc_lattice *L);
/* populations in the reference's own layout: AoS [node][19], node = z + nz*(y + ny*x), values f_i - t_i
 * as Palabos stores them (post-stream state, i.e. what Cell::operator[] returns after collideAndStream) */
int hcl_download_populations(hc_lattice *L, double *f_aos);
int hcl_upload_populations(hc_lattice *L, const double *f_aos);   /* on a slab of a multi-rank run: collective (the ranks exchange their face planes) */
/* rho[n] and u[n][3] = Cell::computeVelocity (j/rho + F/2), local bulk nodes */
int hcl_download_rho_u(hc_lattice *L, double *rho, double *u);
/* pi[n][6] = the off-equilibrium momentum flux of the post-stream populations (xx, xy, xz, yy, yz, zz), the quantity
 * Cell::computeShearStress and computeStrainRateFromStress scale (io/FluidHdf5IO.hh:419-443, :497-552), local bulk nodes */
int hcl_download_pi_neq(hc_lattice *L, double *pi);
/* IBM force field currently accumulated (without the body force), [node][3] */
/* FluidInfo::calculate{Velocity,Force}Statistics (helper/fluidInfo.cpp:33-118) as a device reduction: out = {min, max,
 * sum} of the magnitude over the non-boundary bulk nodes of this slab, *n_nodes their number.  what: 0 =
 * Cell::computeVelocity, 1 = external force (body force + the IBM field spread for the coming step), 2 = rhoBar (sum of
 * the 19 stored populations) over ALL bulk nodes, walls included: its sum is the conserved mass.  Deterministic. */
int hcl_fluid_stats(hc_lattice *L, int what, double out[3], long *n_nodes);
int hcl_download_ibm_force(hc_lattice *L, double *F);
int hcl_zero_ibm_force(hc_lattice *L);
/* halo exchange (Palabos duplicateOverlaps(staticVariables), core/hemoCell.cpp:142).  width = 1 (5
 * populations per face, enough for one collide-stream) or 2 (needed before an IBM interpolation: the 14
 * populations of the face plane that do not move away from the neighbour and, from the plane behind it, the 5
 * that stream onto the neighbour's first halo plane).  side 0 = low-x face, 1 = high-x face.  Buffers are DEVICE pointers of
 * hcl_halo_doubles(L,width) doubles each. */
size_t hcl_halo_doubles(const hc_lattice *L, int width);
int hcl_halo_pack(hc_lattice *L, int side, int width, double *dev_buf);
int hcl_halo_unpack(hc_lattice *L, int side, int width, const double *dev_buf);
/* as hcl_halo_pack, but from the buffer the collide-stream in progress is writing (between
 * hcl_collide_stream_part(L, 4) and hcl_step_end): lets the message leave before the interior planes are done */
int hcl_halo_pack_next(hc_lattice *L, int side, int width, double *dev_buf);
/* the width-1 message of BOTH faces in one launch (what hc_iterate uses every step; a null buffer = no neighbour on that side);
 * next != 0: from the buffer the collide in progress is writing */
int hcl_halo_pack_both(hc_lattice *L, double *dev_lo, double *dev_hi, int next);
int hcl_halo_unpack_both(hc_lattice *L, const double *dev_lo, const double *dev_hi);
/* the message of a velocity update between slabs: node velocities u = j/rho + F/2 (Cell::computeVelocity for
 * ExternalForceDynamics, what core/hemoCellParticleField.cpp:833 blends) of this slab's face plane `side`, evaluated by
 * their owner on the post-stream state and packed as [3][ny*nz] doubles (device pointer) for the neighbour whose first halo
 * plane it is -- 3 planes instead of the 19 population planes of a width-2 halo.  hc_iterate exchanges them itself. */
int hcl_face_velocity_pack(hc_lattice *L, int side, double *dev_buf);
int hcl_face_velocity_pack_both(hc_lattice *L, double *dev_lo, double *dev_hi);   /* both face planes in one launch; a null buffer = no neighbour there */
int hcl_download_face_velocity(hc_lattice *L, int side, double *host_u /*[3][ny*nz]*/);   /* the same plane on the host, for inspection */
int hcl_dims(const hc_lattice *L, int dims[3]);
/* counts[0] = bulk nodes of this slab, [1] = fluid nodes (GuoExternalForceBGKdynamics), [2] = nodes the collide kernel
 * loads and stores (everything but solid nodes without a fluid neighbour, which full-way bounce-back leaves inert) */
int hcl_node_counts(const hc_lattice *L, long counts[3]);
double hcl_mlups_bytes_per_node(const hc_lattice *L); /* algorithmic bytes per node update of the collide kernel */

/* --------------------------------------------------------------- cell types */
#define HC_MODEL_RBC_HO 0     /* mechanics/rbcHighOrderModel.cpp */
#define HC_MODEL_PLT_SIMPLE 1 /* mechanics/pltSimpleModel.cpp    */
#define HC_SHAPE_RBC_FROM_SPHERE 1       /* config/constant_defaults.h:80 */
#define HC_SHAPE_ELLIPSOID_FROM_SPHERE 6 /* config/constant_defaults.h:81 */

typedef struct hc_params { /* Parameters::lbm_base_parameters, mechanics/constantConversion.cpp:36-59 */
  double dx, dt, nu_p, rho_p, kBT_p;
  double tau, nu_lbm, dm, df, f_limit, kBT_lbm;
} hc_params;
int hc_params_base(hc_params *P, double dx, double dt, double nu_p, double rho_p, double kBT_p);

typedef struct hc_material { /* <MaterialModel> of RBC.xml / PLT.xml */
  double kLink, kArea, kVolume, kBend, eta_m;
  double radius;       /* [m] */
  int min_triangles;
  double aspect_ratio; /* ellipsoid only */
  const long *inner_edges; /* [n_inner][2] (PLT.xml:14-38) or NULL */
  int n_inner;
} hc_material;

/* hemocell.addCellType<Model>(name, constructType) (hemocell.h:122-128): builds the mesh
 * (helper/meshGeneratingFunctions.hh), CommonCellConstants (mechanics/commonCellConstants.cpp:70-409) and
 * the moduli (mechanics/cellMechanics.h:50-78), and uploads the tables. */
int hcp_celltype_create(hc_celltype **out, int model, int shape, const hc_params *P, const hc_material *M);
int hcp_celltype_destroy(hc_celltype *T);
/* table sizes: out[0..3] = vertices, triangles, edges, inner edges */
int hcp_celltype_sizes(const hc_celltype *T, int out[4]);
/* host copies of the tables for inspection / output writers; any pointer may be NULL */
int hcp_celltype_tables(const hc_celltype *T, double *vertices /*[nv][3]*/, long *triangles /*[nt][3]*/,
                        long *edges /*[ne][2]*/, double *edge_length_eq, double *edge_angle_eq,
                        double *triangle_area_eq, long *vertex_vertexes /*[nv][6]*/, double *patch_dist_eq,
                        double scalars[9] /* volume_eq, area_mean_eq, edge_mean_eq, angle_mean_eq, k_volume, k_area, k_link, k_bend, eta_m */);

/* the remaining CommonCellConstants tables (mechanics/commonCellConstants.h:64-85); any pointer may be NULL */
int hcp_celltype_tables2(const hc_celltype *T, long *edge_bending_triangles /*[ne][2]*/, long *edge_bending_outer_points /*[ne][2]*/,
                         long *inner_edges /*[nie][2]*/, double *inner_edge_length_eq /*[nie]*/, int *vertex_n_vertexes /*[nv]*/);

/* ---------------------------------------------------------------- cells */
/* HemoCellFields / HemoCellParticleField (core/hemoCellFields.h:103-158, core/hemoCellParticleField.h:39-207) */
int hcp_create(hc_cells **out, hc_lattice *L, const hc_params *P);
int hcp_destroy(hc_cells *C);
int hcp_add_type(hc_cells *C, hc_celltype *T, int material_timescale /* setMaterialTimeScaleSeparation */, int *type_index);
/* loadParticles() (io/readPositionsBloodCells.cpp:290-361) for one cell: centre in lattice units (GLOBAL
 * coordinates), angles in radians, already negated as :228-229 does.  placed=0 when a vertex falls on /
 * within min_dist_um of a boundary node (:139-164) and the cell is dropped.
 * On a slab (n_slabs > 1) every rank offers every cell: the call keeps it when one of its particles has its nearest node in
 * this slab or it reaches within the envelope of a face (periodic images are tried shifted by +-nx_global,
 * core/hemoCellParticleDataTransfer.cpp:33-65), placed=0 otherwise; a rank sees the wall only next to its slab, so
 * hcp_slab_sync_placement must follow the last cell: it drops on every rank the cells any rank rejected and reports how
 * many distinct cells of each type the whole domain now holds. */
int hcp_add_cell(hc_cells *C, int type, long cell_id, const double centre_lu[3], const double angles[3],
                 double min_dist_um, int *placed);
/* slot for a cell whose state is restored afterwards with hcp_upload (checkpoint resume,
 * core/hemoCellFields.cpp:240-275): undeformed mesh at centre_lu, no wall test */
int hcp_add_cell_unchecked(hc_cells *C, int type, long cell_id, const double centre_lu[3], const double angles[3]);
/* n_vertices counts the listed vertices (cells x vertices per cell; an incomplete cell keeps its slots), n_deleted the
 * cells removed entirely so far */
int hcp_slab_sync_placement(hc_cells *C, long *global_cells_per_type /*[n types]*/);
/* The particle envelope of a slab run: <particleEnvelope> of the case configuration in lattice units
 * (examples/pipeflow/config.xml:34), read at core/hemoCell.cpp:139 and handed to HemoCellFields (core/hemoCellFields.cpp:39-43).
 * The reference replicates single particles within that distance of a block face on the neighbour; here whole cells are
 * replicated, so an envelope E acts as share = E - (diameter of the largest cell type): how far a cell may still travel
 * towards the face, between two velocity updates, before a complete copy must exist on the other side (default share: 4 lu).
 * Call after the cell types are added and before the first cell is placed.  A request the slab width cannot support
 * (nx >= 2 * diameter + 2 * share) is clamped and the share in use returned; a slab too thin for the minimum (2 lu) is an
 * error.  Every copy that arrives late -- a particle already within reach of the receiving slab's nodes when its cell first
 * gets there -- is counted on the device (hcp_envelope) and fails the hc_iterate call that notices it. */
int hcp_set_envelope(hc_cells *C, double particle_envelope_lu, double *share_in_use);
int hcp_envelope(const hc_cells *C, double *share_in_use, long *late_copies);
int hcp_counts(hc_cells *C, long *n_vertices, long *n_cells, long *n_deleted);
int hcp_type_range(hc_cells *C, int type, long *first_vertex, long *n_cells);
/* What happens to a particle whose nearest node is a boundary after advance (core/hemoCellParticleField.cpp:566-588):
 *   HC_DELETE_PARTICLE (default, the reference): removeParticles(1) takes that particle out (:304-321); its cell is
 *     incomplete from then on -- no mechanics (:634-652), forces of the remaining particles zeroed at the next material
 *     step (:660-667), still spread / interpolated / advanced -- until hcp_delete_incomplete_cells removes the rest
 *     (deleteIncompleteCells, :512-553: the reference calls it at every writeOutput, core/hemoCell.cpp:248-252, and at
 *     velocity-update steps when verbose.cellsDeletedInfo is set, :361-363);
 *   HC_DELETE_CELL: the whole cell goes at once.
 * Either way the decision is taken on the device inside the advance kernel and costs no host round trip; slab runs
 * (n_slabs > 1) remove an incomplete cell on all its holders at the next velocity update, i.e. behave as the reference
 * with cellsDeletedInfo. */
#define HC_DELETE_PARTICLE 0
#define HC_DELETE_CELL 1
int hcp_set_deletion_mode(hc_cells *C, int mode);
int hcp_delete_incomplete_cells(hc_cells *C, long *n_cells_removed);
/* cells removed entirely / single particles removed so far; incomplete cells and missing particles currently listed */
int hcp_deletion_counts(hc_cells *C, long *cells_removed, long *particles_removed, long *incomplete_cells, long *missing_particles);
/* alive[i] = 0 for a removed particle of an incomplete cell, in hcp_download order */
int hcp_download_alive(hc_cells *C, unsigned char *alive);
/* serializeValues_t fields as [n][3] arrays in cell-major order; what: 0 position 1 velocity 2 force */
int hcp_download(hc_cells *C, int what, double *out);
int hcp_upload(hc_cells *C, int what, const double *in);
int hcp_download_cell_ids(hc_cells *C, long *ids);
/* The same state in the reference's own particle record, HemoCellParticle::serializeValues_t (core/hemoCellParticle.h:45-63,
 * 120 bytes: v @0, position @24, force @48, force_repulsion @72, plint cellId @96, uint16 vertexId @104, uint restime @108,
 * uchar celltype @112), so that a binding can hand over the contents of libhemocell's std::vector<HemoCellParticle>:
 * download writes one record per vertex (n_records = hcp_counts vertices; types, cells, vertices in order);
 * upload replaces the whole population by the given records, in any order, every cell complete. */
int hcp_download_records(hc_cells *C, void *records, long n_records);
int hcp_upload_records(hc_cells *C, const void *records, long n_records);
/* HemoCellStretch::ForceForcedLsps (helper/hemoCellStretch.cpp:63-78): sv.force += f on listed vertices */
int hcp_add_vertex_force(hc_cells *C, const long *vertex_index, int n, const double *f /*[n][3]*/);
/* hemocell.setRepulsion(k, cutoff) + setRepulsionTimeScaleSeperation (core/hemoCell.cpp:394-397,420-426); the
 * cutoff is given in lattice units (the facade converts from micrometres).  hc_iterate then evaluates
 * cellfields->applyRepulsionForce() (core/hemoCell.cpp:307-309 -> core/hemoCellParticleField.cpp:677-743) every
 * `timescale` iterations; spread adds force_repulsion + force as the reference does. */
/* ParticleInfo::calculate{Velocity,Force}Statistics (helper/particleInfo.cpp:30-140) as a device reduction over the
 * vertices this slab owns: out = {min, max, sum} of |v| (what 1) or |force + force_repulsion| (what 2). */
int hcp_vertex_stats(hc_cells *C, int what, double out[3], long *n);
int hcp_set_repulsion(hc_cells *C, double r_const, double r_cutoff_lu, int timescale);
int hcp_repulsion(hc_cells *C);
int hcp_download_repulsion(hc_cells *C, double *out /*[n][3]*/);
/* hemocell.enableBoundaryParticles(k, cutoff, timestep) (core/hemoCell.cpp:428-436): wall nodes with a non-wall node
 * among their 26 neighbours (populateBoundaryParticles, core/hemoCellParticleField.cpp:865-890) push the vertices
 * binned around them; hcp_boundary_repulsion is cellfields->applyBoundaryRepulsionForce() (core/hemoCell.cpp:310-312 ->
 * core/hemoCellParticleField.cpp:891-918) and ADDS to force_repulsion, which only hcp_repulsion resets (:703).
 * hc_iterate applies it every `timescale` iterations.  Call after hcl_set_mask. */
int hcp_set_boundary_repulsion(hc_cells *C, double br_const, double br_cutoff_lu, int timescale);
int hcp_boundary_repulsion(hc_cells *C);
/* cellfields->spreadParticleForce() (core/hemoCell.cpp:313 -> core/hemoCellParticleField.cpp:841-863) */
int hcp_spread(hc_cells *C, int force_limit);
/* cellfields->interpolateFluidVelocity() (core/hemoCell.cpp:329 -> core/hemoCellParticleField.cpp:819-839) */
int hcp_interpolate(hc_cells *C);
/* the same for the listed cells of one type only (slab runs interpolate the cells that cross a face first, so that
 * their records travel while the rest is interpolated) */
int hcp_interpolate_cells(hc_cells *C, int type, const int *slots, int n);
/* cellfields->advanceParticles() (core/hemoCell.cpp:342 -> core/hemoCellParticleField.cpp:566-588) */
int hcp_advance(hc_cells *C, int check_deletions);
/* cellfields->applyConstitutiveModel(forced) (core/hemoCell.cpp:345 -> core/hemoCellParticleField.cpp:633-675) */
int hcp_mechanics(hc_cells *C, long iter, int forced);
/* separate_force_vectors output mode (core/hemoCellParticleField.cpp:590-614): comp = [6][n][3]
 * (volume, area, bending, link, visc, inner link) for the vertices of one type */
int hcp_mechanics_components(hc_cells *C, int type, double *comp);
/* HemoCell::iterate() (core/hemoCell.cpp:299-376) n times, followed each time by the driver's body-force
 * re-application (examples/pipeflow/pipeflow.cpp:144-146).  particle_timescale =
 * setParticleVelocityUpdateTimeScaleSeparation. iter is read and advanced.  Particles that reach a wall are deleted on
 * the device at EVERY iteration (hcp_set_deletion_mode); deletion_check_every is only how often the host looks, without
 * waiting, whether storage can be compacted.  On a slab of a multi-rank run (n_slabs > 1) the same call runs the slab
 * schedule: faces of the 5 crossing populations every step and the particle envelopes at every velocity update
 * (syncEnvelopes, core/hemoCellFields.cpp:377-499) travel over the data plane beside the interior collide. */
int hc_iterate(hc_lattice *L, hc_cells *C, long *iter, int n, int particle_timescale, int force_limit,
               int deletion_check_every);
/* ---- multi-slab particle envelopes: HemoCellFields::syncEnvelopes (core/hemoCellFields.cpp:377-499) and
 * HemoCellParticleDataTransfer::send/receive (core/hemoCellParticleDataTransfer.cpp:33-180).  The host
 * decides WHICH cells cross a slab face (from hcp_cell_extents); the records move device-to-device.
 * A record is 9 doubles per vertex: position, velocity, force (the mutable part of serializeValues_t,
 * core/hemoCellParticle.h:45-63); x_shift is the periodic offset (+-nx_global) of :33-65. */
int hcp_cell_extents(hc_cells *C, int type, double *ext /*[n_cells][3] = min x, max x, #vertices whose nearest node is in this slab*/);
/* asynchronous form: _begin enqueues the kernel and the copy into pinned staging and returns; _end waits for that copy
 * only (an event), not for work enqueued afterwards, so the extents can be started early in a step and picked up
 * without draining the stream.  The cell set must not change in between. */
int hcp_cell_extents_begin(hc_cells *C, int type);
int hcp_cell_extents_end(hc_cells *C, int type, double *ext);
size_t hcp_record_doubles(const hc_cells *C, int type); /* doubles per cell record */
int hcp_pack_cells(hc_cells *C, int type, const int *slots, int n, double x_shift, double *dev_buf);
/* merge rule of HemoCellParticleField::addParticle (core/hemoCellParticleField.cpp:173-235): a local vertex
 * (nearest node in this slab) is kept, any other is overwritten; is_new cells are appended (slots must be
 * n_cells, n_cells+1, ... in order) */
int hcp_unpack_cells(hc_cells *C, int type, const int *slots, const long *cell_ids, const int *is_new, int n,
                     const double *dev_buf);
/* deleteNonLocalParticles (core/hemoCellFields.cpp:676-688) at cell granularity; holes are filled from the tail */
int hcp_remove_cells(hc_cells *C, int type, const int *slots, int n);
int hcp_owned_vertices(hc_cells *C, long *n_owned); /* vertices whose nearest node lies in this slab */

/* CellInformationFunctionals (helper/cellInfo.cpp:39-80,140-180): per cell volume, area, bbox[6], centroid[3] */
int hcp_cell_info(hc_cells *C, int type, double *volume, double *area, double *bbox, double *centroid);

#ifdef __cplusplus
}
#endif
#endif
