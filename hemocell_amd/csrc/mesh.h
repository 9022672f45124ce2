// Host-side construction of a cell type: surface mesh, CommonCellConstants
// tables in gather form, moduli.  Product code (not the oracle).
#pragma once
#include <array>
#include <vector>
#include "../../include/hemocell_amd.h"

namespace hc {

using Vec3 = std::array<double, 3>;

struct CellTables {
  int model = 0;
  int nv = 0, nt = 0, ne = 0, nie = 0;
  std::vector<Vec3> vertices;                  // undeformed mesh, lattice units
  std::vector<std::array<long, 3>> triangles;  // hemoCellField.cpp:78-83
  std::vector<std::array<long, 2>> edges;      // commonCellConstants.cpp:81-93
  std::vector<double> edge_length_eq, edge_angle_eq, triangle_area_eq, patch_dist_eq;
  std::vector<std::array<long, 2>> edge_bending_triangles, edge_bending_outer;
  std::vector<std::array<long, 6>> vertex_vertexes;  // ring ordered, -1 padded
  std::vector<int> vertex_n_vertexes;
  std::vector<std::array<long, 2>> inner_edges;
  std::vector<double> inner_edge_length_eq;
  double volume_eq = 0, area_mean_eq = 0, edge_mean_eq = 0, angle_mean_eq = 0;
  double diameter = 0;                         // of the undeformed mesh, lattice units
  double k_volume = 0, k_area = 0, k_link = 0, k_bend = 0, eta_m = 0;

  // ---- gather form used by the kernels (all int32, -1 padded) ----
  static constexpr int MAXD = 8;                    // max incident elements kept per vertex
  std::vector<int> vtri;      // [nv][MAXD] incident triangles, ascending id
  std::vector<int> vtri_k;    // [nv][MAXD] corner index (0,1,2) of the vertex in that triangle
  std::vector<int> vedge;     // [nv][MAXD] incident edges, ascending id
  std::vector<int> vedge_s;   // [nv][MAXD] +1 if vertex is edge[0], -1 if edge[1]
  std::vector<int> bsrc;      // [nv][MAXD] RBC bending sources: {self} U ring, ascending vertex id
  // PLT: per vertex, edges that touch it as an outer point of the dihedral pair
  std::vector<int> vouter;    // [nv][MAXD] ascending edge id
  std::vector<int> vinner;    // [nv][MAXD] inner edges, ascending
  std::vector<int> vinner_s;  // [nv][MAXD]
};

// builds everything from the material description; returns non-empty error on failure
std::string build_cell_tables(CellTables &T, int model, int shape, const hc_params &P, const hc_material &M);

// rotateTriangularMeshXYZ of io/readPositionsBloodCells.cpp:40-111 as a 3x3 matrix
void rotation_matrix_xyz(double alpha, double beta, double gamma, double R[3][3]);

}  // namespace hc
