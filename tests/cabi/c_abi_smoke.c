/* Plain C99 client of the C ABI: what a non-C++ host (cgo, JNI, ctypes, Fortran ...) sees.  One RBC in a small periodic
 * pipe, 50 x HemoCell::iterate, then a few sanity numbers on stdout.  Compiled by tests/test_capi_symbols.py (CPU: link
 * check) and run by tests/test_gpu_parity.py (GPU). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hemocell_amd.h"

#define CHECK(call) do { if ((call) != HC_OK) { fprintf(stderr, "%s failed: %s\n", #call, hc_last_error()); return 1; } } while (0)

int main(void) {
  const int nx = 32, ny = 26, nz = 26;
  int periodic[3] = {1, 0, 0};
  hc_params P;
  hc_lattice *L = NULL;
  hc_cells *C = NULL;
  hc_celltype *T = NULL;
  CHECK(hc_init(0));
  CHECK(hc_params_base(&P, 5e-7, 1e-7, 1.1e-6, 1025.0, 4.100531391e-21));
  CHECK(hcl_create(&L, nx, ny, nz, periodic, 1.0 / P.tau, 0, nx, 1));
  {
    /* analytic cylinder along x, radius (ny-2)/2, mask with the two halo planes on each side */
    unsigned char *mask = (unsigned char *)calloc((size_t)(nx + 4) * ny * nz, 1);
    const double R = (ny - 2) / 2.0, cy = (ny - 1) / 2.0, cz = (nz - 1) / 2.0;
    for (int x = 0; x < nx + 4; x++) for (int y = 0; y < ny; y++) for (int z = 0; z < nz; z++)
      mask[((size_t)x * ny + y) * nz + z] = (sqrt((y - cy) * (y - cy) + (z - cz) * (z - cz)) > R) ? 1 : 0;
    CHECK(hcl_set_mask(L, mask));
    free(mask);
  }
  { double u0[3] = {0, 0, 0}, F[3] = {2e-6, 0, 0}; CHECK(hcl_init_equilibrium(L, 1.0, u0)); CHECK(hcl_set_body_force(L, F)); }
  CHECK(hcp_create(&C, L, &P));
  {
    hc_material M;
    memset(&M, 0, sizeof(M));
    M.radius = 3.91e-6; M.min_triangles = 600; M.kLink = 15.0; M.kArea = 5.0; M.kVolume = 20.0; M.kBend = 80.0; M.eta_m = 0.0;
    CHECK(hcp_celltype_create(&T, HC_MODEL_RBC_HO, HC_SHAPE_RBC_FROM_SPHERE, &P, &M));
    CHECK(hcp_add_type(C, T, 1, NULL));
  }
  {
    double centre[3] = {16.0, 12.5, 12.5}, angles[3] = {-1.5707963267948966, 0, 0};
    int placed = 0;
    CHECK(hcp_add_cell(C, 0, 0, centre, angles, 0.0, &placed));
    if (!placed) { fprintf(stderr, "cell rejected\n"); return 1; }
  }
  CHECK(hcp_mechanics(C, 0, 1));
  { long it = 0; CHECK(hc_iterate(L, C, &it, 50, 1, 1, 1)); printf("ITER %ld\n", it); }
  {
    long nv = 0, nc = 0, nd = 0; double vs[3], fs[3]; long n1 = 0, n2 = 0;
    double vol, area, bbox[6], cen[3];
    CHECK(hcp_counts(C, &nv, &nc, &nd));
    CHECK(hcp_vertex_stats(C, 1, vs, &n1));
    CHECK(hcl_fluid_stats(L, 0, fs, &n2));
    CHECK(hcp_cell_info(C, 0, &vol, &area, bbox, cen));
    printf("CELLS %ld VERTICES %ld DELETED %ld\n", nc, nv, nd);
    printf("VMAX %.6e UMAX %.6e VOLUME %.6f CENTRE %.6f %.6f %.6f\n", vs[1], fs[1], vol, cen[0], cen[1], cen[2]);
  }
  CHECK(hcp_destroy(C)); CHECK(hcp_celltype_destroy(T)); CHECK(hcl_destroy(L));
  return 0;
}
