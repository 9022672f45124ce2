// Cell-type construction on the host: mesh generation, equilibrium tables,
// moduli.  Mirrors (does not copy) the reference's setup path:
//   helper/meshGeneratingFunctions.hh:32-271   sphere -> RBC / ellipsoid surface
//   core/hemoCellField.cpp:38-118              triangle list from the mesh
//   mechanics/commonCellConstants.cpp:70-409   equilibrium tables
//   mechanics/cellMechanics.h:50-78            moduli in lattice units
// Vertex numbering: first occurrence over the triangle list (Palabos
// TriangleSet -> DEFscaledMesh); confirmed by PLT.xml's InnerEdges, whose 21
// hard-coded vertex pairs are exact antipodes / mirror pairs under this
// numbering (tests/test_oracle_pins.py).
#include "mesh.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <string>
#include <tuple>

namespace hc {

namespace {

constexpr double kPi = 3.14159265358979323846;  // config/constant_defaults.h:121-123

struct Tri { Vec3 p[3]; };

inline Vec3 sub(const Vec3 &a, const Vec3 &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
  return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
// helper/array.h:228-244 accumulate from zero, component order
inline double dot(const Vec3 &a, const Vec3 &b) { double r = 0; for (int i = 0; i < 3; i++) r += a[i] * b[i]; return r; }
inline double norm(const Vec3 &a) { double r = 0; for (int i = 0; i < 3; i++) r += a[i] * a[i]; return std::sqrt(r); }

Vec3 unit_normal(const Vec3 &v0, const Vec3 &v1, const Vec3 &v2, double *area = nullptr) {
  Vec3 n = cross(sub(v1, v0), sub(v2, v0));
  const double nn = norm(n);
  if (nn != 0.0) { if (area) *area = 0.5 * nn; n[0] /= nn; n[1] /= nn; n[2] /= nn; }
  else { if (area) *area = 0; n = {0, 0, 0}; }
  return n;
}

// z-x-z Euler rotation of a triangle soup (Palabos TriangleSet::rotate)
void rotate_zxz(std::vector<Tri> &t, double phi, double theta, double psi) {
  const double a[3][3] = {{1, 0, 0}, {0, std::cos(theta), -std::sin(theta)}, {0, std::sin(theta), std::cos(theta)}};
  const double b[3][3] = {{std::cos(phi), -std::sin(phi), 0}, {std::sin(phi), std::cos(phi), 0}, {0, 0, 1}};
  const double b2[3][3] = {{std::cos(psi), -std::sin(psi), 0}, {std::sin(psi), std::cos(psi), 0}, {0, 0, 1}};
  double c[3][3], m[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { c[i][j] = 0; for (int k = 0; k < 3; k++) c[i][j] += a[i][k] * b[k][j]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { m[i][j] = 0; for (int k = 0; k < 3; k++) m[i][j] += b2[i][k] * c[k][j]; }
  for (auto &tr : t) for (auto &p : tr.p) {
    const Vec3 x = p;
    for (int i = 0; i < 3; i++) { double s = 0; for (int j = 0; j < 3; j++) s += m[i][j] * x[j]; p[i] = s; }
  }
}

// 1 -> 4 subdivision with midpoints projected on the unit sphere
// (helper/meshGeneratingFunctions.hh:109-146: centre triangle replaces the
// parent in place, the three corner triangles are appended)
void subdivide_until(std::vector<Tri> &t, long min_triangles) {
  while ((long)t.size() < min_triangles) {
    const size_t n = t.size();
    t.reserve(4 * n);
    for (size_t i = 0; i < n; i++) {
      const Vec3 va = t[i].p[0], vb = t[i].p[1], vc = t[i].p[2];
      Vec3 vd, ve, vf;
      for (int d = 0; d < 3; d++) { vd[d] = 0.5 * (va[d] + vb[d]); ve[d] = 0.5 * (vb[d] + vc[d]); vf[d] = 0.5 * (vc[d] + va[d]); }
      const double nd = norm(vd), ne = norm(ve), nf = norm(vf);
      for (int d = 0; d < 3; d++) { vd[d] /= nd; ve[d] /= ne; vf[d] /= nf; }
      t[i] = Tri{{vd, ve, vf}};
      t.push_back(Tri{{va, vd, vf}});
      t.push_back(Tri{{vd, vb, ve}});
      t.push_back(Tri{{vf, ve, vc}});
    }
  }
}

std::vector<Tri> icosphere(long min_triangles) {
  // helper/meshGeneratingFunctions.hh:44-106
  const double tau = -0.8506508084, one = -0.5257311121;
  const Vec3 v[13] = {{0, 0, 0},
                      {tau, one, 0}, {-tau, one, 0}, {-tau, -one, 0}, {tau, -one, 0},
                      {one, 0, tau}, {one, 0, -tau}, {-one, 0, -tau}, {-one, 0, tau},
                      {0, tau, one}, {0, -tau, one}, {0, -tau, -one}, {0, tau, -one}};
  static const int faces[20][3] = {{5, 8, 9}, {5, 10, 8}, {6, 12, 7}, {6, 7, 11}, {1, 4, 5}, {1, 6, 4}, {3, 2, 8},
                                   {3, 7, 2}, {9, 12, 1}, {9, 2, 12}, {10, 4, 11}, {10, 11, 3}, {9, 1, 5},
                                   {12, 6, 1}, {5, 4, 10}, {6, 11, 4}, {8, 2, 9}, {7, 12, 2}, {8, 10, 3}, {7, 3, 11}};
  std::vector<Tri> t;
  for (auto &f : faces) t.push_back(Tri{{v[f[0]], v[f[1]], v[f[2]]}});
  subdivide_until(t, min_triangles);
  return t;
}

std::vector<Tri> octasphere(long min_triangles) {
  // Palabos constructSphere (initialSphereShape 0, helper/meshGeneratingFunctions.h:55)
  const Vec3 a{1, 0, 0}, b{0, 1, 0}, c{-1, 0, 0}, d{0, -1, 0}, e{0, 0, 1}, f{0, 0, -1};
  std::vector<Tri> t = {Tri{{e, a, b}}, Tri{{e, b, c}}, Tri{{e, c, d}}, Tri{{e, d, a}},
                        Tri{{f, b, a}}, Tri{{f, c, b}}, Tri{{f, d, c}}, Tri{{f, a, d}}};
  subdivide_until(t, min_triangles);
  return t;
}

void build_mesh(CellTables &T, int shape, double radius, long min_triangles, double aspect) {
  std::vector<Tri> t;
  if (shape == HC_SHAPE_RBC_FROM_SPHERE) {
    // constructRBCFromSphere, helper/meshGeneratingFunctions.hh:217-243
    t = icosphere(min_triangles);
    rotate_zxz(t, kPi / 2.0, kPi / 2.0, 0.);
    for (auto &tr : t) for (auto &p : tr.p) {
      // spherePointToRBCPoint (R = 1), :155-171
      double r2 = p[0] * p[0] + p[1] * p[1];
      const double val = p[2];
      const int sign = (0.0 < val) - (val < 0.0);
      if (1 - r2 < 0) r2 = 1;
      const double C0 = 0.054322, C2 = 1.001279, C4 = -0.561381;
      p[2] = sign * 1.0 * std::sqrt(1 - r2) * (C0 + C2 * r2 + C4 * r2 * r2);
    }
    for (auto &tr : t) for (auto &p : tr.p) for (int d = 0; d < 3; d++) p[d] *= radius;
    rotate_zxz(t, kPi / 2.0, kPi / 2.0, 0.);
  } else {
    // constructEllipsoidFromSphere, :246-271
    t = octasphere(min_triangles);
    rotate_zxz(t, kPi / 2.0, kPi / 2.0, 0.);
    for (auto &tr : t) for (auto &p : tr.p) {
      double r2 = p[0] * p[0] + p[1] * p[1];
      const double val = p[2];
      const int sign = (0.0 < val) - (val < 0.0);
      if (1 - r2 < 0) r2 = 1;
      p[0] *= radius; p[1] *= radius;
      p[2] = sign * aspect * radius * std::sqrt(1 - r2);
    }
    rotate_zxz(t, kPi / 2.0, kPi / 2.0, 0.);
  }
  // weld: first-occurrence numbering
  using Key = std::tuple<int64_t, int64_t, int64_t>;
  std::map<Key, long> seen;
  T.vertices.clear(); T.triangles.clear();
  for (const auto &tr : t) {
    std::array<long, 3> ids;
    for (int k = 0; k < 3; k++) {
      const Vec3 &p = tr.p[k];
      Key key{std::llround(p[0] * 1e8), std::llround(p[1] * 1e8), std::llround(p[2] * 1e8)};
      auto it = seen.find(key);
      if (it == seen.end()) { it = seen.emplace(key, (long)T.vertices.size()).first; T.vertices.push_back(p); }
      ids[k] = it->second;
    }
    T.triangles.push_back(ids);
  }
  T.nv = (int)T.vertices.size(); T.nt = (int)T.triangles.size();
  // Cells.getMesh().inflate() (helper/meshGeneratingFunctions.h:92): 1e-3 lu along the
  // vertex normal; amount chosen as documented in DESIGN.md ("Oracle pinning")
  std::vector<Vec3> vn(T.nv, Vec3{0, 0, 0});
  for (const auto &tr : T.triangles) {
    const Vec3 n = unit_normal(T.vertices[tr[0]], T.vertices[tr[1]], T.vertices[tr[2]]);
    for (int k = 0; k < 3; k++) for (int d = 0; d < 3; d++) vn[tr[k]][d] += n[d];
  }
  for (int i = 0; i < T.nv; i++) {
    const double l = norm(vn[i]);
    for (int d = 0; d < 3; d++) T.vertices[i][d] += 1.e-3 * (vn[i][d] / l);
  }
}

}  // namespace

void rotation_matrix_xyz(double alpha, double beta, double gamma, double a[3][3]) {
  // io/readPositionsBloodCells.cpp:47-98, same index pattern ("column-first")
  double b[3][3], c[3][3];
  a[0][0] = 1; a[0][1] = 0; a[0][2] = 0;
  a[1][0] = 0; a[1][1] = std::cos(alpha); a[1][2] = std::sin(alpha);
  a[2][0] = 0; a[2][1] = -std::sin(alpha); a[2][2] = std::cos(alpha);
  b[0][0] = std::cos(beta); b[0][1] = 0; b[0][2] = -std::sin(beta);
  b[1][0] = 0; b[1][1] = 1; b[1][2] = 0;
  b[2][0] = std::sin(beta); b[2][1] = 0; b[2][2] = std::cos(beta);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { c[i][j] = 0; for (int k = 0; k < 3; k++) c[i][j] += a[k][j] * b[i][k]; }
  b[0][0] = std::cos(gamma); b[0][1] = std::sin(gamma); b[0][2] = 0;
  b[1][0] = -std::sin(gamma); b[1][1] = std::cos(gamma); b[1][2] = 0;
  b[2][0] = 0; b[2][1] = 0; b[2][2] = 1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = 0; for (int k = 0; k < 3; k++) a[i][j] += c[k][j] * b[i][k]; }
}

std::string build_cell_tables(CellTables &T, int model, int shape, const hc_params &P, const hc_material &M) {
  if (model != HC_MODEL_RBC_HO && model != HC_MODEL_PLT_SIMPLE) return "unknown mechanics model";
  if (shape != HC_SHAPE_RBC_FROM_SPHERE && shape != HC_SHAPE_ELLIPSOID_FROM_SPHERE) return "unsupported construct type";
  if (!(M.radius > 0) || M.min_triangles < 8) return "material: radius must be > 0 and min_triangles >= 8";
  T.model = model;
  build_mesh(T, shape, M.radius / P.dx, M.min_triangles, M.aspect_ratio);
  const int nv = T.nv, nt = T.nt;
  const auto &V = T.vertices;

  // directed half-edge -> triangle
  std::map<std::pair<long, long>, long> half;
  for (long t = 0; t < nt; t++) for (int k = 0; k < 3; k++) half[{T.triangles[t][k], T.triangles[t][(k + 1) % 3]}] = t;
  if ((long)half.size() != 3L * nt) return "mesh is not an oriented manifold";

  // edges (a < b) in triangle order
  T.edges.clear();
  for (long t = 0; t < nt; t++) {
    const auto &tr = T.triangles[t];
    if (tr[0] < tr[1]) T.edges.push_back({tr[0], tr[1]});
    if (tr[1] < tr[2]) T.edges.push_back({tr[1], tr[2]});
    if (tr[2] < tr[0]) T.edges.push_back({tr[2], tr[0]});
  }
  T.ne = (int)T.edges.size();
  const int ne = T.ne;
  T.edge_length_eq.resize(ne); T.edge_angle_eq.resize(ne);
  T.edge_bending_triangles.resize(ne); T.edge_bending_outer.resize(ne);
  for (int e = 0; e < ne; e++) {
    const long e0 = T.edges[e][0], e1 = T.edges[e][1];
    Vec3 ev = sub(V[e1], V[e0]);
    const double el = norm(ev);
    T.edge_length_eq[e] = el;
    // adjacent triangles: [0] holds the directed edge e1->e0, [1] holds e0->e1 (restoring
    // orientation of the dihedral law; Palabos getAdjacentTriangleIds is not available)
    auto i0 = half.find({e1, e0}), i1 = half.find({e0, e1});
    if (i0 == half.end() || i1 == half.end()) return "open mesh: edge with a single adjacent triangle";
    const long ta = i0->second, tb = i1->second;
    T.edge_bending_triangles[e] = {ta, tb};
    const Vec3 n1 = unit_normal(V[T.triangles[ta][0]], V[T.triangles[ta][1]], V[T.triangles[ta][2]]);
    const Vec3 n2 = unit_normal(V[T.triangles[tb][0]], V[T.triangles[tb][1]], V[T.triangles[tb][2]]);
    ev[0] /= el; ev[1] /= el; ev[2] /= el;
    T.edge_angle_eq[e] = std::atan2(dot(cross(n1, n2), ev), dot(n1, n2));  // helper/geometryUtils.h:49-52
    for (int i = 0; i < 3; i++) {
      if (T.triangles[ta][i] != e0 && T.triangles[ta][i] != e1) T.edge_bending_outer[e][0] = T.triangles[ta][i];
      if (T.triangles[tb][i] != e0 && T.triangles[tb][i] != e1) T.edge_bending_outer[e][1] = T.triangles[tb][i];
    }
  }
  // inner edges
  T.inner_edges.clear(); T.inner_edge_length_eq.clear();
  for (int e = 0; e < M.n_inner; e++) {
    const long a = M.inner_edges[2 * e], b = M.inner_edges[2 * e + 1];
    if (a < 0 || b < 0 || a >= nv || b >= nv) return "inner edge vertex id out of range";
    T.inner_edges.push_back({a, b});
    T.inner_edge_length_eq.push_back(norm(sub(V[b], V[a])));
  }
  T.nie = (int)T.inner_edges.size();
  // areas
  T.triangle_area_eq.resize(nt);
  for (long t = 0; t < nt; t++) unit_normal(V[T.triangles[t][0]], V[T.triangles[t][1]], V[T.triangles[t][2]], &T.triangle_area_eq[t]);

  // gather lists (ascending ids by construction)
  const int D = CellTables::MAXD;
  auto fill = [&](std::vector<int> &v) { v.assign((size_t)nv * D, -1); };
  fill(T.vtri); fill(T.vtri_k); fill(T.vedge); fill(T.vedge_s); fill(T.bsrc); fill(T.vouter); fill(T.vinner); fill(T.vinner_s);
  std::vector<int> cnt(nv, 0);
  for (long t = 0; t < nt; t++) for (int k = 0; k < 3; k++) {
    const long v = T.triangles[t][k];
    if (cnt[v] >= D) return "vertex valence exceeds MAXD";
    T.vtri[v * D + cnt[v]] = (int)t; T.vtri_k[v * D + cnt[v]] = k; cnt[v]++;
  }
  std::fill(cnt.begin(), cnt.end(), 0);
  for (int e = 0; e < ne; e++) for (int s = 0; s < 2; s++) {
    const long v = T.edges[e][s];
    if (cnt[v] >= D) return "vertex valence exceeds MAXD";
    T.vedge[v * D + cnt[v]] = e; T.vedge_s[v * D + cnt[v]] = s == 0 ? 1 : -1; cnt[v]++;
  }
  std::fill(cnt.begin(), cnt.end(), 0);
  for (int e = 0; e < ne; e++) for (int s = 0; s < 2; s++) {
    const long v = T.edge_bending_outer[e][s];
    if (cnt[v] >= D) return "vertex valence exceeds MAXD";
    T.vouter[v * D + cnt[v]++] = e;
  }
  std::fill(cnt.begin(), cnt.end(), 0);
  for (int e = 0; e < T.nie; e++) for (int s = 0; s < 2; s++) {
    const long v = T.inner_edges[e][s];
    if (cnt[v] >= D) return "too many inner edges on one vertex";
    T.vinner[v * D + cnt[v]] = e; T.vinner_s[v * D + cnt[v]] = s == 0 ? 1 : -1; cnt[v]++;
  }

  // volume_eq: MeshMetrics::getVolume, helper/meshMetrics.h:167-177
  double vol = 0.0;
  for (int iv = 0; iv < nv; iv++)
    for (int k = 0; k < D; k++) {
      const int t = T.vtri[(size_t)iv * D + k];
      if (t < 0) break;
      const auto &tr = T.triangles[t];
      vol += dot(V[tr[0]], cross(V[tr[1]], V[tr[2]])) / 6.0 / 3.0;
    }
  T.volume_eq = vol;
  {   // diameter of the undeformed mesh (twice the largest distance from its bounding-box centre): sizes the particle envelope of slab runs
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, r2 = 0.0;
    for (const Vec3 &v : T.vertices) for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], v[d]); hi[d] = std::max(hi[d], v[d]); }
    for (const Vec3 &v : T.vertices) { double q = 0; for (int d = 0; d < 3; d++) { const double c = v[d] - 0.5 * (lo[d] + hi[d]); q += c * c; } r2 = std::max(r2, q); }
    T.diameter = 2.0 * std::sqrt(r2);
  }
  double s = 0; for (double a : T.triangle_area_eq) s += a; T.area_mean_eq = s / nt;
  s = 0; for (double l : T.edge_length_eq) s += l; T.edge_mean_eq = s / ne;
  s = 0; for (double a : T.edge_angle_eq) s += a; T.angle_mean_eq = s / ne;

  // neighbour rings: first neighbour in edge-list order, then walk the fan
  // (mechanics/commonCellConstants.cpp:201-271)
  T.vertex_vertexes.assign(nv, {-1, -1, -1, -1, -1, -1});
  T.vertex_n_vertexes.assign(nv, 0);
  for (int e = 0; e < ne; e++) {
    const long a = T.edges[e][0], b = T.edges[e][1];
    if (T.vertex_n_vertexes[a] >= 6 || T.vertex_n_vertexes[b] >= 6) return "vertex with more than 6 neighbours";
    T.vertex_vertexes[a][T.vertex_n_vertexes[a]++] = b;
    T.vertex_vertexes[b][T.vertex_n_vertexes[b]++] = a;
  }
  for (long v = 0; v < nv; v++) {
    long cur = T.vertex_vertexes[v][0];
    for (int n = 1; n < T.vertex_n_vertexes[v]; n++) {
      auto it = half.find({v, cur});
      if (it == half.end()) return "ring walk failed";
      const auto &tr = T.triangles[it->second];
      long next = -1;
      for (int k = 0; k < 3; k++) if (tr[k] == v) next = tr[(k + 2) % 3];
      cur = next;
      T.vertex_vertexes[v][n] = cur;
    }
  }
  // equilibrium patch-centre distance (:274-305)
  T.patch_dist_eq.resize(nv);
  for (long i = 0; i < nv; i++) {
    const int nn = T.vertex_n_vertexes[i];
    Vec3 sum{0, 0, 0};
    for (int j = 0; j < nn; j++) for (int d = 0; d < 3; d++) sum[d] += V[T.vertex_vertexes[i][j]][d];
    const Vec3 mid{sum[0] / nn, sum[1] / nn, sum[2] / nn};
    const Vec3 dev = sub(mid, V[i]);
    Vec3 pn{0, 0, 0};
    for (int j = 0; j < nn; j++) {
      Vec3 tn = cross(sub(V[T.vertex_vertexes[i][j]], V[i]), sub(V[T.vertex_vertexes[i][(j + 1) % nn]], V[i]));
      const double l = norm(tn);
      for (int d = 0; d < 3; d++) { tn[d] /= l; pn[d] += tn[d]; }
    }
    const double l = norm(pn);
    pn[0] /= l; pn[1] /= l; pn[2] /= l;
    T.patch_dist_eq[i] = dot(pn, dev);
  }
  // bending sources for the gather form: {self} U ring, ascending
  for (long i = 0; i < nv; i++) {
    std::vector<long> src(T.vertex_vertexes[i].begin(), T.vertex_vertexes[i].begin() + T.vertex_n_vertexes[i]);
    src.push_back(i);
    std::sort(src.begin(), src.end());
    for (size_t k = 0; k < src.size(); k++) T.bsrc[i * D + k] = (int)src[k];
  }

  // moduli, mechanics/cellMechanics.h:50-78
  const double plc = 7.5e-9 / P.dx;
  T.k_link = M.kLink * P.kBT_lbm / plc;
  const double eqLength = 5e-7 / P.dx;
  T.k_bend = M.kBend * P.kBT_lbm / eqLength;
  const double NfacesScaling = 1280.0 / nt;
  T.k_volume = M.kVolume * NfacesScaling * P.kBT_lbm / eqLength;
  T.k_area = M.kArea * NfacesScaling * P.kBT_lbm / (eqLength);
  T.eta_m = M.eta_m * P.dx / P.dt / P.df;
  return "";
}

}  // namespace hc

extern "C" int hc_params_base(hc_params *P, double dx, double dt, double nu_p, double rho_p, double kBT_p) {
  if (!P) return HC_ERR_ARG;
  // Parameters::lbm_base_parameters, mechanics/constantConversion.cpp:36-59
  P->dx = dx; P->dt = dt; P->nu_p = nu_p; P->rho_p = rho_p; P->kBT_p = kBT_p;
  if (dt < 0.0) {
    P->tau = 1.0;
    P->nu_lbm = 1.0 / 3.0 * (P->tau - 0.5);
    P->dt = P->nu_lbm / nu_p * (dx * dx);
  } else {
    P->nu_lbm = nu_p * dt / (dx * dx);
    P->tau = 3.0 * P->nu_lbm + 0.5;
  }
  P->dm = rho_p * (dx * dx * dx);
  P->df = P->dm * dx / (P->dt * P->dt);
  P->f_limit = 50.0 / 1.0e12 / P->df;  // FORCE_LIMIT = 50 pN, config/constant_defaults.h:73-75
  P->kBT_lbm = kBT_p / (P->df * dx);
  return HC_OK;
}
