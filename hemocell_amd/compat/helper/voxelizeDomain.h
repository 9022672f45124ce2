// helper/voxelizeDomain.h: STL -> flag matrix (1 inside, 0 outside), tube ends opened.
// Replacement for Palabos' voxelizer (helper/voxelizeDomain.cpp:76-152): the mesh is scaled so that its
// extent along refDir spans refDirLength lattice cells and shifted by one margin cell; a node is inside when a
// +z ray from it crosses the closed surface an odd number of times.  Domain sizing follows the description in
// DESIGN.md; Palabos' exact rounding is not available (UNPINNED).
#pragma once
#include "../hemocell.h"

namespace hemo {

inline bool read_stl(const std::string &fn, std::vector<std::array<double, 9>> &tris) {
  std::ifstream f(fn.c_str(), std::ios::binary);
  if (!f.is_open()) return false;
  std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (all.compare(0, 5, "solid") == 0 && all.find("facet") != std::string::npos) {
    std::istringstream s(all); std::string w; std::array<double, 9> t; int k = 0;
    while (s >> w) if (w == "vertex") { s >> t[k] >> t[k + 1] >> t[k + 2]; k += 3; if (k == 9) { tris.push_back(t); k = 0; } }
    return !tris.empty();
  }
  if (all.size() < 84) return false;
  uint32_t n; std::memcpy(&n, all.data() + 80, 4);
  for (uint32_t i = 0; i < n && 84 + 50 * (size_t)(i + 1) <= all.size(); i++) {
    float v[12]; std::memcpy(v, all.data() + 84 + 50 * (size_t)i, 48);
    std::array<double, 9> t; for (int k = 0; k < 9; k++) t[k] = v[3 + k];
    tris.push_back(t);
  }
  return !tris.empty();
}

inline void getFlagMatrixFromSTL(std::string meshFileName, plint /*extendedEnvelopeWidth*/, plint refDirLength, plint refDir,
                                 VoxelizedDomain3D<T> *&voxelizedDomain, MultiScalarField3D<int> *&flagMatrix, plint /*blockSize*/, int /*particleEnvelope*/) {
  std::vector<std::array<double, 9>> tris;
  if (!read_stl(meshFileName, tris)) { hlog << "(Voxelizer) Error: " << meshFileName << " is not an existing stl file." << endl; std::exit(1); }
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (auto &t : tris) for (int k = 0; k < 9; k++) { lo[k % 3] = std::min(lo[k % 3], t[k]); hi[k % 3] = std::max(hi[k % 3], t[k]); }
  const plint margin = 1;
  const double dxs = (hi[refDir] - lo[refDir]) / refDirLength;
  plint n[3];
  for (int d = 0; d < 3; d++) n[d] = (plint)std::llround((hi[d] - lo[d]) / dxs) + 1 + 2 * margin;
  for (auto &t : tris) for (int k = 0; k < 9; k++) t[k] = (t[k] - lo[k % 3]) / dxs + margin;   // lattice units
  MultiBlockManagement3D m; m.nx = n[0]; m.ny = n[1]; m.nz = n[2];
  voxelizedDomain = new VoxelizedDomain3D<T>(m);
  flagMatrix = new MultiScalarField3D<int>(n[0], n[1], n[2], 0);
  // parity of +z ray crossings per (x,y) column; a tiny irrational offset keeps rays off edges and vertices
  const double ex = 1.234567e-7, ey = 2.345678e-7;
  for (plint x = 0; x < n[0]; x++)
    for (plint y = 0; y < n[1]; y++) {
      std::vector<double> hits;
      const double px = x + ex, py = y + ey;
      for (auto &t : tris) {
        const double ax = t[0] - px, ay = t[1] - py, bx = t[3] - px, by = t[4] - py, cx = t[6] - px, cy = t[7] - py;
        const double d0 = ax * by - ay * bx, d1 = bx * cy - by * cx, d2 = cx * ay - cy * ax;
        if ((d0 > 0 && d1 > 0 && d2 > 0) || (d0 < 0 && d1 < 0 && d2 < 0)) { const double s = d0 + d1 + d2; hits.push_back((d1 * t[2] + d2 * t[5] + d0 * t[8]) / s); }
      }
      std::sort(hits.begin(), hits.end());
      for (plint z = 0; z < n[2]; z++) {
        size_t above = 0; for (double h : hits) if (h > z) above++;
        flagMatrix->get(x, y, z) = (above % 2) ? 1 : 0;
      }
    }
  // open the two x ends by copying the neighbouring slice (helper/voxelizeDomain.cpp:141-150)
  for (plint y = 0; y < n[1]; y++) for (plint z = 0; z < n[2]; z++) {
    flagMatrix->get(1, y, z) = flagMatrix->get(2, y, z); flagMatrix->get(0, y, z) = flagMatrix->get(1, y, z);
    flagMatrix->get(n[0] - 2, y, z) = flagMatrix->get(n[0] - 3, y, z); flagMatrix->get(n[0] - 1, y, z) = flagMatrix->get(n[0] - 2, y, z);
  }
  hlog << "(main) Voxelisation is done. Resulting domain parameters are: " << n[0] << "-by-" << n[1] << "-by-" << n[2] << endl;
}

#if __cplusplus < 201703L
inline void getFlagMatrixFromSTL(std::string meshFileName, plint extendedEnvelopeWidth, plint refDirLength, plint refDir,
                                 std::auto_ptr<VoxelizedDomain3D<T>> &voxelizedDomain, std::auto_ptr<MultiScalarField3D<int>> &flagMatrix,
                                 plint blockSize, int particleEnvelope) {
  VoxelizedDomain3D<T> *v = nullptr; MultiScalarField3D<int> *f = nullptr;
  getFlagMatrixFromSTL(meshFileName, extendedEnvelopeWidth, refDirLength, refDir, v, f, blockSize, particleEnvelope);
  voxelizedDomain = std::auto_ptr<VoxelizedDomain3D<T>>(v);
  flagMatrix = std::auto_ptr<MultiScalarField3D<int>>(f);
}
#endif

}  // namespace hemo
