/*
 * hemo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64) of the HemoCell IB-LBM hot path, used as the
 * parity oracle for the HIP kernels in hemocell_amd/csrc and as bench.py's
 * `cpu_baseline` ("kind": "port").  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The product path
 * (libhemocell_amd.so) never links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the read-only reference tree).  The fluid half (collide-stream) lives in
 * Palabos v2.3.0, which is NOT in the reference tree (setup.sh:9-23 downloads
 * it): its arithmetic is restated from the published algorithm (Guo-forced BGK,
 * D3Q19, full-way bounce-back) and from the call sites / the patch
 * (patch/palabos.patch:244-249, 459-466, 491-498) -- bit-level fluid parity
 * with Palabos is therefore UNPINNED; see DESIGN.md "Oracle pinning".
 */
#ifndef HEMO_ORACLE_H
#define HEMO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_Q 19

/* ------------------------------------------------------------------ lattice */
typedef struct orc_lattice {
  int nx, ny, nz;          /* node index = z + nz*(y + ny*x)  (patch/palabos.patch:244-249) */
  int periodic[3];
  double omega;
  double *f;               /* [n][19] AoS, stored as f_i - t_i (Palabos fBar convention)   */
  double *ftmp;            /* second buffer for collide -> stream                          */
  double *force;           /* [n][3]  Cell::external.data[0..2]                            */
  unsigned char *mask;     /* 0 = GuoExternalForceBGKdynamics, 1 = BounceBack (isBoundary), 3..6 = moving wall class */
  int nthreads;            /* OpenMP threads used by orc_collide_stream (cpu_baseline)     */
  double wall_u[4][3];     /* velocity of the moving-wall classes 3..6                     */
  int fused;               /* cpu_baseline only: orc_collide_stream runs as ONE pass (orc_collide_stream_fused) */
} orc_lattice;

orc_lattice *orc_lattice_create(int nx, int ny, int nz, const int periodic[3], double omega);
void orc_lattice_destroy(orc_lattice *L);
void orc_lattice_set_mask(orc_lattice *L, const unsigned char *mask);
void orc_lattice_init_equilibrium(orc_lattice *L, double rho, const double u[3]);
void orc_lattice_set_force_uniform(orc_lattice *L, const double F[3]);
void orc_lattice_set_force_box(orc_lattice *L, const int box[6], const double F[3]);
void orc_collide_stream(orc_lattice *L);
/* the same step as one pass over the lattice (collide in registers, push to the neighbours in the second buffer):
 * same arithmetic, same bits, a third of the memory traffic -- the faster of the two CPU baselines of bench.py */
void orc_collide_stream_fused(orc_lattice *L);
/* rho and u = j/rho + F/2 of the current (post-stream) populations */
void orc_node_rho_u(const orc_lattice *L, long node, double *rho, double u[3]);
void orc_lattice_set_threads(orc_lattice *L, int n);
void orc_lattice_set_wall_velocity(orc_lattice *L, int cls, const double u[3]);

/* D3Q19 tables (Palabos ordering) */
extern const int    orc_c[ORC_Q][3];
extern const double orc_t[ORC_Q];

/* --------------------------------------------------------------- parameters */
typedef struct orc_params {
  double dx, dt, nu_p, rho_p, kBT_p;
  double tau, nu_lbm, dm, df, f_limit, kBT_lbm;
} orc_params;
void orc_params_base(orc_params *P, double dx, double dt, double nu_p, double rho_p, double kBT_p);

/* --------------------------------------------------------------- cell types */
#define ORC_MODEL_RBC_HO 0
#define ORC_MODEL_PLT_SIMPLE 1

typedef struct orc_celltype {
  int model;
  int nv, nt, ne, nie;
  double *vertices;        /* [nv][3] reference (undeformed) mesh, lattice units */
  long *triangles;         /* [nt][3] */
  long *edges;             /* [ne][2] */
  double *edge_length_eq;  /* [ne] */
  double *edge_angle_eq;   /* [ne] */
  long *edge_bending_triangles; /* [ne][2] */
  long *edge_bending_outer;     /* [ne][2] */
  double *triangle_area_eq;     /* [nt] */
  long *vertex_vertexes;        /* [nv][6] ring-ordered, -1 padded */
  int *vertex_n_vertexes;       /* [nv] */
  double *patch_dist_eq;        /* [nv] surface_patch_center_dist_eq_list */
  long *inner_edges;            /* [nie][2] */
  double *inner_edge_length_eq; /* [nie] */
  double volume_eq, area_mean_eq, edge_mean_eq, angle_mean_eq;
  double k_volume, k_area, k_link, k_bend, eta_m;
  int timescale;                /* stepMaterialEvery */
} orc_celltype;

/* shape: 1 = RBC_FROM_SPHERE (icosahedron), 6 = ELLIPSOID_FROM_SPHERE (octahedron) */
orc_celltype *orc_celltype_create(int model, int shape, double radius_lu, int min_triangles,
                                  double aspect_ratio, const long *inner_edges, int n_inner);
void orc_celltype_destroy(orc_celltype *T);
void orc_celltype_set_moduli(orc_celltype *T, const orc_params *P, double kLink, double kArea,
                             double kVolume, double kBend, double eta_m_si);
double orc_mesh_surface(const orc_celltype *T);

/* membrane forces for ONE complete cell.  pos/vel: [nv][3].  force: [nv][3]
 * accumulated INTO (caller zeroes).  If comp != NULL it is [6][nv][3]
 * (volume, area, bending, link, visc, inner_link) written separately, like the
 * reference's separate_force_vectors mode. flags: bit0 area, bit1 volume,
 * bit2 bending, bit3 link(+visc), bit4 inner links; 0x1f = all.            */
void orc_cell_forces(const orc_celltype *T, const double *pos, const double *vel, double *force,
                     double *comp, int flags);

/* ---------------------------------------------------------------------- IBM */
/* phi2 stencil (core/immersedBoundaryMethod.h:62-138). returns count (<=8) */
int orc_phi2_stencil(const orc_lattice *L, const double pos[3], long nodes[8], double weights[8]);

/* ---------------------------------------------------------------- particles */
typedef struct orc_particle {     /* = HemoCellParticle::serializeValues_t, 120 B */
  double v[3];
  double position[3];
  double force[3];
  double force_repulsion[3];
  long cellId;
  unsigned short vertexId;
  unsigned int restime;
  unsigned char celltype;
} orc_particle;

typedef struct orc_sim {
  orc_lattice *L;
  orc_params P;
  int ntypes;
  orc_celltype *types[8];
  long ncells[8];           /* live complete cells per type                   */
  orc_particle *particles;  /* cell-major: type0 cells, then type1 ...        */
  long np;
  /* cached stencils from the last spread (reference caches kernelLocations) */
  long *st_nodes; double *st_w; int *st_n;
  int particle_velocity_timescale;   /* stepParticleEvery */
  int force_limit_enabled;
  long iter;
  long cells_deleted;
  double body_force[3];     /* driver's setExternalVector after each iterate */
  int rep_enabled, rep_timescale; double rep_const, rep_cutoff;   /* setRepulsion / setRepulsionTimeScaleSeperation */
  int brep_enabled, brep_timescale; double brep_const, brep_cutoff;   /* enableBoundaryParticles (core/hemoCell.cpp:428-436) */
  /* What advanceParticles does with a particle whose nearest node is a boundary (core/hemoCellParticleField.cpp:566-588):
   * 0 (default) = the reference: removeParticles(1) takes that particle out (:304-321), the cell stays behind incomplete --
   * no mechanics (:634-652), its forces zeroed at the next material step (:660-667), still spread / interpolated / advanced --
   * until deleteIncompleteCells (:512-553; the reference calls it at writeOutput, core/hemoCell.cpp:248-252).
   * 1 = the whole cell is removed at once (the product's HC_DELETE_CELL mode). */
  int deletion_mode;
  unsigned char *dead;      /* [np] 1 = this particle was removed (its record stays in place so that cell-major indexing holds) */
  long particles_deleted;
  /* setExternalVector on sub-boxes, re-applied by the driver around every iterate like the uniform force
   * (cases/kolmogorovFlow/kolmogorovFlow.cpp:136-140): inclusive node ranges {x0,x1,y0,y1,z0,z1}, applied in order */
  int n_regions; int region_box[4][6]; double region_force[4][3];
} orc_sim;

orc_sim *orc_sim_create(orc_lattice *L, const orc_params *P);
void orc_sim_destroy(orc_sim *S);   /* does not destroy L or the cell types */
int orc_sim_add_type(orc_sim *S, orc_celltype *T);
/* place one cell: centre (lattice units), angles in radians already negated as
 * io/readPositionsBloodCells.cpp:228-229 does. returns 1 if placed, 0 if rejected */
int orc_sim_add_cell(orc_sim *S, int type, const double centre_lu[3], const double angles[3],
                     double min_dist_from_solid_um);
void orc_sim_spread(orc_sim *S);
void orc_sim_interpolate(orc_sim *S);
void orc_sim_advance(orc_sim *S);
void orc_sim_mechanics(orc_sim *S, int forced);
void orc_sim_iterate(orc_sim *S);
void orc_sim_repulsion(orc_sim *S, double r_const, double r_cutoff_lu);
void orc_sim_boundary_repulsion(orc_sim *S, double br_const, double br_cutoff_lu);
long orc_sim_type_offset(const orc_sim *S, int type);
/* what: 0 position, 1 velocity, 2 force, 3 force_repulsion; arrays [np][3] */
void orc_sim_get(const orc_sim *S, int what, double *out);
void orc_sim_set(orc_sim *S, int what, const double *in);
void orc_sim_add_vertex_force(orc_sim *S, long particle, const double f[3]);
/* HemoCellParticleField::deleteIncompleteCells (core/hemoCellParticleField.cpp:512-553); returns the cells removed */
long orc_sim_delete_incomplete_cells(orc_sim *S);
void orc_sim_get_alive(const orc_sim *S, unsigned char *alive);

#ifdef __cplusplus
}
#endif
#endif
