// HDF5 output with the reference's file layout (row f, rank 2 of SURVEY.md §8):
//   <out>/hdf5/<iter %012d>/<Name>.<iter %012d>.p.<blockId>.h5        io/ParticleHdf5IO.cpp:60, io/FluidHdf5IO.hh:89
// Cell files: root attributes dx, dt, iteration, processorId, numberOfProcessors, numberOfParticles,
// numberOfTriangles; float32 [n][3|1] datasets named as io/hemoCellParticleFieldOutputFunctions.cpp:53-426,
// int32 "Triangles" / "InnerLinks" with per-cell vertex offsets; chunks of <= 1000 rows, deflate 7
// (io/ParticleHdf5IO.cpp:85-151).  Fluid file: attributes numberOfCells, subdomainSize {Nz,Ny,Nx},
// relativePosition {z,y,x} - 1.5, dxdydz; float32 [Nz+2][Ny+2][Nx+2][C] datasets incl. a one-node envelope,
// SI scaling when outputInSiUnits (io/FluidHdf5IO.hh:92-287).  Compiled only when HEMOCELL_WITH_HDF5 is defined.
#pragma once
#ifdef HEMOCELL_WITH_HDF5
#include <hdf5.h>
#include <hdf5_hl.h>
#include "hemocell.h"

namespace hemo {

inline string zeroPadNumber(unsigned int n) { char b[32]; std::snprintf(b, sizeof(b), "%012u", n); return b; }

inline void h5_write_2d(hid_t file, const string &name, const vector<float> &data, hsize_t rows, hsize_t cols) {
  hsize_t dim[2] = {rows, cols};
  hsize_t chunk[2] = {std::max<hsize_t>(1, std::min<hsize_t>(1000, rows)), std::max<hsize_t>(1, cols)};
  hid_t sid = H5Screate_simple(2, dim, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
  H5Pset_chunk(pl, 2, chunk); H5Pset_deflate(pl, 7);
  hid_t did = H5Dcreate2(file, name.c_str(), H5T_NATIVE_FLOAT, sid, H5P_DEFAULT, pl, H5P_DEFAULT);
  H5Dwrite(did, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data());
  H5Dclose(did); H5Pclose(pl); H5Sclose(sid);
}
inline void h5_write_2d_int(hid_t file, const string &name, const vector<int> &data, hsize_t rows, hsize_t cols) {
  hsize_t dim[2] = {rows, cols};
  hsize_t chunk[2] = {std::max<hsize_t>(1, std::min<hsize_t>(1000, rows)), std::max<hsize_t>(1, cols)};
  hid_t sid = H5Screate_simple(2, dim, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
  H5Pset_chunk(pl, 2, chunk); H5Pset_deflate(pl, 7);
  hid_t did = H5Dcreate2(file, name.c_str(), H5T_NATIVE_INT, sid, H5P_DEFAULT, pl, H5P_DEFAULT);
  H5Dwrite(did, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data());
  H5Dclose(did); H5Pclose(pl); H5Sclose(sid);
}

// writeCellField3D_HDF5 (io/ParticleHdf5IO.cpp:36-176)
inline void writeCellField3D_HDF5(HemoCell &h, HemoCellField &field, const string &dir) {
  if (field.desiredOutputVariables.empty()) return;
  hc_cells *c = h.cellfields->device();
  long fv = 0, nc = 0; hcp_type_range(c, (int)field.ctype, &fv, &nc);
  long nvt = 0, nct = 0; hcp_counts(c, &nvt, &nct, nullptr);
  // the cells this rank reports: all of them on one GPU, those whose centre lies in its slab otherwise (the other holder of
  // an envelope copy writes it; the reference writes each particle from the block that owns it)
  vector<long> sel;
  {
    vector<double> V((size_t)nc), A((size_t)nc), B(6 * (size_t)nc), P(3 * (size_t)nc);
    if (nc && global.world > 1) hc_check(hcp_cell_info(c, (int)field.ctype, V.data(), A.data(), B.data(), P.data()), "hcp_cell_info");
    for (long k = 0; k < nc; k++) if (global.world == 1 || CellInformationFunctionals::centre_local(&h, &P[3 * (size_t)k])) sel.push_back(k);
  }
  const long full_nc = nc; (void)full_nc;
  nc = (long)sel.size();
  const int nvc = field.numVertex;
  auto srcv = [&](long i) { return fv + sel[(size_t)(i / nvc)] * nvc + i % nvc; };   // output row -> vertex in download order
  const long n = nc * field.numVertex;
  vector<double> pos(3 * (size_t)nvt), vel(3 * (size_t)nvt), frc(3 * (size_t)nvt), comp;
  if (nvt) { hcp_download(c, 0, pos.data()); hcp_download(c, 1, vel.data()); hcp_download(c, 2, frc.data()); }
  vector<double> rep(3 * (size_t)nvt, 0.0);
  if (nvt) hc_check(hcp_download_repulsion(c, rep.data()), "hcp_download_repulsion");   // zeros while no repulsion is enabled
  // the reference writes each particle where its block holds it, i.e. inside the domain; positions here are not
  // re-wrapped when a cell crosses a periodic face, so they are wrapped vertex by vertex for the output
  {
    const plint dims[3] = {h.lattice->getNx(), h.lattice->getNy(), h.lattice->getNz()};
    for (int d = 0; d < 3; d++) {
      if (!h.lattice->per.p[d]) continue;
      for (long i = 0; i < nvt; i++) { double &x = pos[(size_t)(3 * i + d)]; x -= (double)dims[d] * std::floor((x + 0.5) / (double)dims[d]); }
    }
  }
  vector<long> ids((size_t)nct); if (nct) hcp_download_cell_ids(c, ids.data());
  long first_cell = 0; for (unsigned int t = 0; t < field.ctype; t++) { long f2, n2; hcp_type_range(c, (int)t, &f2, &n2); first_cell += n2; }
  const string fileName = dir + "/" + field.name + "." + zeroPadNumber(h.iter) + ".p." + std::to_string(global.rank) + ".h5";   // one file per block (io/ParticleHdf5IO.cpp:60)
  hid_t file = H5Fcreate(fileName.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  double dx = Parameters::dx, dt = Parameters::dt; long it = h.iter, np_ = global.world; int id = global.rank;
  H5LTset_attribute_double(file, "/", "dx", &dx, 1); H5LTset_attribute_double(file, "/", "dt", &dt, 1);
  H5LTset_attribute_long(file, "/", "iteration", &it, 1); H5LTset_attribute_int(file, "/", "processorId", &id, 1);
  H5LTset_attribute_long(file, "/", "numberOfProcessors", &np_, 1);
  const bool si = h.outputInSiUnits;
  auto vec3 = [&](const vector<double> &src, double scale) { vector<float> o(3 * (size_t)n); for (long i = 0; i < n; i++) for (int d = 0; d < 3; d++) o[(size_t)(3 * i + d)] = (float)(src[(size_t)(3 * srcv(i) + d)] * scale); return o; };
  bool triangles = false, innerlinks = false;
  for (int var : field.desiredOutputVariables) {
    switch (var) {
      case OUTPUT_POSITION: { h5_write_2d(file, "Position", vec3(pos, si ? Parameters::dx : 1.0), n, 3); long nP = n; H5LTset_attribute_long(file, "/", "numberOfParticles", &nP, 1); break; }
      case OUTPUT_VELOCITY: h5_write_2d(file, "Velocity", vec3(vel, si ? Parameters::dx / Parameters::dt : 1.0), n, 3); break;
      case OUTPUT_FORCE: {   // force_total = force + force_repulsion (core/hemoCellParticleField.cpp:596)
        vector<double> tot(frc); for (size_t i = 0; i < tot.size(); i++) tot[i] += rep[i];
        // With separate force vectors requested the reference first re-evaluates the model where this iteration is a material
        // step (separateForceVectors -> applyConstitutiveModel(), :590-598), so its total is the sum of the vectors it writes.
        // Same here for the OUTPUT (the forces the simulation holds are left alone; the reference overwrites them as a side effect).
        bool separate = false;
        for (int v2 : field.desiredOutputVariables) separate = separate || (v2 >= OUTPUT_FORCE_VOLUME && v2 <= OUTPUT_FORCE_INNER_LINK);
        const long nfull = full_nc * field.numVertex;
        if (separate && nfull && h.iter % field.timescale == 0) {
          if (comp.empty()) { comp.resize(18 * (size_t)nfull); hc_check(hcp_mechanics_components(c, (int)field.ctype, comp.data()), "hcp_mechanics_components"); }
          for (long i = 0; i < nfull; i++) for (int d = 0; d < 3; d++) {
            double sum = 0; for (int slot = 0; slot < 6; slot++) sum += comp[(size_t)(slot * 3 * nfull + 3 * i + d)];
            tot[(size_t)(3 * (fv + i) + d)] = sum + rep[(size_t)(3 * (fv + i) + d)];
          }
        }
        h5_write_2d(file, "Total force", vec3(tot, si ? Parameters::df : 1.0), n, 3); break; }
      case OUTPUT_FORCE_VOLUME: case OUTPUT_FORCE_AREA: case OUTPUT_FORCE_BENDING: case OUTPUT_FORCE_LINK: case OUTPUT_FORCE_VISC: case OUTPUT_FORCE_INNER_LINK: {
        const long nfull = full_nc * field.numVertex;
        if (comp.empty() && nfull) { comp.resize(18 * (size_t)nfull); hc_check(hcp_mechanics_components(c, (int)field.ctype, comp.data()), "hcp_mechanics_components"); }
        const int slot = var == OUTPUT_FORCE_VOLUME ? 0 : var == OUTPUT_FORCE_AREA ? 1 : var == OUTPUT_FORCE_BENDING ? 2 : var == OUTPUT_FORCE_LINK ? 3 : var == OUTPUT_FORCE_VISC ? 4 : 5;
        static const char *names[6] = {"Volume force", "Area force", "Bending force", "Link force", "Viscous force", "Inner link force"};
        vector<float> o(3 * (size_t)n); for (long i = 0; i < n; i++) for (int d = 0; d < 3; d++) o[(size_t)(3 * i + d)] = (float)(comp[(size_t)(slot * 3 * nfull + 3 * (srcv(i) - fv) + d)] * (si ? Parameters::df : 1.0));
        h5_write_2d(file, names[slot], o, n, 3); break; }
      case OUTPUT_FORCE_REPULSION: h5_write_2d(file, "Repulsion force", vec3(rep, si ? Parameters::df : 1.0), n, 3); break;
      case OUTPUT_VERTEX_ID: { vector<float> o((size_t)n); for (long i = 0; i < n; i++) o[(size_t)i] = (float)(i % field.numVertex); h5_write_2d(file, "Vertex Id", o, n, 1); break; }
      case OUTPUT_CELL_ID: { vector<float> o((size_t)n); for (long i = 0; i < n; i++) o[(size_t)i] = (float)ids[(size_t)(first_cell + sel[(size_t)(i / field.numVertex)])]; h5_write_2d(file, "Cell Id", o, n, 1); break; }
      case OUTPUT_RES_TIME: h5_write_2d(file, "Res Time", vector<float>((size_t)n, 0.f), n, 1); break;
      case OUTPUT_TRIANGLES: triangles = true; break;
      case OUTPUT_INNER_LINKS: innerlinks = true; break;
      default: break;
    }
  }
  if (triangles) {   // vertex indices offset by numVertex per cell (io/hemoCellParticleFieldOutputFunctions.cpp:345-364)
    vector<int> tri(3 * (size_t)(nc * field.numTriangles));
    for (long cc = 0; cc < nc; cc++) for (int t = 0; t < field.numTriangles; t++) for (int k = 0; k < 3; k++)
      tri[(size_t)((cc * field.numTriangles + t) * 3 + k)] = (int)(field.triangles[3 * (size_t)t + k] + cc * field.numVertex);
    h5_write_2d_int(file, "Triangles", tri, (hsize_t)(nc * field.numTriangles), 3);
    long nT = nc * field.numTriangles; H5LTset_attribute_long(file, "/", "numberOfTriangles", &nT, 1);
  }
  if (innerlinks && !field.innerEdges.empty()) {
    const long ni = (long)field.innerEdges.size() / 2;
    vector<int> li(2 * (size_t)(nc * ni));
    for (long cc = 0; cc < nc; cc++) for (long e = 0; e < ni; e++) for (int k = 0; k < 2; k++) li[(size_t)((cc * ni + e) * 2 + k)] = (int)(field.innerEdges[2 * (size_t)e + k] + cc * field.numVertex);
    h5_write_2d_int(file, "InnerLinks", li, (hsize_t)(nc * ni), 2);
    long nL = nc * ni; H5LTset_attribute_long(file, "/", "numberOfInnerLinks", &nL, 1);
  }
  H5Fclose(file);
}

// writeFluidField_HDF5 (io/FluidHdf5IO.hh:60-210)
inline void writeFluidField_HDF5(HemoCell &h, const string &dir) {
  if (h.fluidOutputs.empty()) return;
  auto *L = h.lattice; hc_lattice *d = L->device();
  const plint nx = L->nxl, ny = L->ny, nz = L->nz, x0 = L->x0;   // this rank's block
  const size_t plane = (size_t)ny * nz, nn = (size_t)nx * plane, ne = (size_t)(nx + 2) * plane;
  // Every field lives on the block plus one x-plane on either side (index (x + 1) * plane + y * nz + z, x = -1 .. nx): the
  // reference writes a one-node envelope around each block, filled from the neighbouring block (io/FluidHdf5IO.hh:215-287 loop
  // over odomain +- 1).  The planes come from the periodic image on one rank, from the neighbour rank's face otherwise, and
  // repeat the face where the domain ends.  y and z envelopes wrap or repeat locally.
  auto gx = [&](plint x) -> plint {   // global plane of local plane x, -1 where the domain ends
    plint g = x0 + x;
    if (g < 0 || g >= L->nx) { if (!L->per.p[0]) return -1; g = ((g % L->nx) + L->nx) % L->nx; }
    return g;
  };
  auto fill_ends = [&](vector<double> &a, int C) {   // a: [(nx + 2) * plane][C], interior already in place
    const size_t pc = plane * (size_t)C;
    if (global.world > 1) {
      vector<double> rlo(pc), rhi(pc);
      hc_check(hc_comm_exchange_host(L->per.p[0] ? 1 : 0, a.data() + pc, pc * sizeof(double), a.data() + (size_t)nx * pc, pc * sizeof(double), rlo.data(), pc * sizeof(double),
                                     rhi.data(), pc * sizeof(double)), "hc_comm_exchange_host");
      if (gx(-1) >= 0) std::copy(rlo.begin(), rlo.end(), a.begin());
      if (gx(nx) >= 0) std::copy(rhi.begin(), rhi.end(), a.begin() + (size_t)(nx + 1) * pc);
    } else if (L->per.p[0]) {
      std::copy(a.begin() + (size_t)nx * pc, a.begin() + (size_t)(nx + 1) * pc, a.begin());
      std::copy(a.begin() + pc, a.begin() + 2 * pc, a.begin() + (size_t)(nx + 1) * pc);
    }
    if (gx(-1) < 0) std::copy(a.begin() + pc, a.begin() + 2 * pc, a.begin());
    if (gx(nx) < 0) std::copy(a.begin() + (size_t)nx * pc, a.begin() + (size_t)(nx + 1) * pc, a.begin() + (size_t)(nx + 1) * pc);
  };
  // node classes of the extended block from the global flags
  vector<uint8_t> maskx(ne), bbx(ne, 0);
  for (plint x = -1; x <= nx; x++) {
    const plint g = gx(x) >= 0 ? gx(x) : gx(x < 0 ? 0 : nx - 1);
    std::copy(L->mask.begin() + (size_t)g * plane, L->mask.begin() + (size_t)(g + 1) * plane, maskx.begin() + (size_t)(x + 1) * plane);
    if (!L->bounce_back.empty()) std::copy(L->bounce_back.begin() + (size_t)g * plane, L->bounce_back.begin() + (size_t)(g + 1) * plane, bbx.begin() + (size_t)(x + 1) * plane);
  }
  vector<double> rho(ne), u(3 * ne);
  {
    vector<double> r0(nn), u0(3 * nn);
    hc_check(hcl_download_rho_u(d, r0.data(), u0.data()), "hcl_download_rho_u");
    std::copy(r0.begin(), r0.end(), rho.begin() + plane); std::copy(u0.begin(), u0.end(), u.begin() + 3 * plane);
  }
  fill_ends(rho, 1); fill_ends(u, 3);
  // cell.computeVelocity / computeDensity go through the node's dynamics (io/FluidHdf5IO.hh:223,273): a BounceBack node answers
  // zero velocity and the density it was constructed with, whatever its populations hold; a velocity-condition node the
  // velocity it imposes (setBoundaryVelocity; zero until one is given)
  for (size_t k = 0; k < ne; k++) {
    const uint8_t m = maskx[k];
    if (m == 0) continue;
    if (bbx[k]) { rho[k] = L->bb_rho; u[3 * k] = u[3 * k + 1] = u[3 * k + 2] = 0; continue; }
    for (int c = 0; c < 3; c++) u[3 * k + c] = m >= 3 ? (double)L->wall_u[(size_t)(m - 3)][(size_t)c] : 0.0;
  }
  const string fileName = dir + "/Fluid." + zeroPadNumber(h.iter) + ".p." + std::to_string(global.rank) + ".h5";
  hid_t file = H5Fcreate(fileName.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  double dx = Parameters::dx, dt = Parameters::dt; long it = h.iter; int id = global.rank;
  H5LTset_attribute_double(file, "/", "dx", &dx, 1); H5LTset_attribute_double(file, "/", "dt", &dt, 1);
  H5LTset_attribute_long(file, "/", "iteration", &it, 1); H5LTset_attribute_int(file, "/", "processorId", &id, 1);
  const hsize_t Nx = nx + 2, Ny = ny + 2, Nz = nz + 2;   // one-node envelope on each side for paraview
  int ncells = (int)(Nx * Ny * Nz); int sub[3] = {(int)Nz, (int)Ny, (int)Nx};
  float dxdydz[3] = {1.f, 1.f, 1.f}, rel[3] = {-1.5f, -1.5f, (float)x0 - 1.5f};   // {z, y, x} of the block's first node - 1.5 (io/FluidHdf5IO.hh:112-118)
  const bool si = h.outputInSiUnits;
  if (si) for (int k = 0; k < 3; k++) { rel[k] *= (float)Parameters::dx; dxdydz[k] = (float)Parameters::dx; }
  H5LTset_attribute_int(file, "/", "numberOfCells", &ncells, 1); H5LTset_attribute_int(file, "/", "subdomainSize", sub, 3);
  H5LTset_attribute_float(file, "/", "relativePosition", rel, 3); H5LTset_attribute_float(file, "/", "dxdydz", dxdydz, 3);
  auto src = [&](plint v, plint n, bool per) { if (v < 0 || v >= n) return per ? ((v % n) + n) % n : std::min<plint>(std::max<plint>(v, 0), n - 1); return v; };
  // extended index of output node (x, y, z), x in -1 .. nx (beyond that: the outermost plane again), y and z in -1 .. n
  auto node_of = [&](plint x, plint y, plint z) {
    const plint xe = std::min<plint>(std::max<plint>(x, -1), nx) + 1;
    return ((size_t)xe * ny + (size_t)src(y, ny, L->per.p[1])) * nz + (size_t)src(z, nz, L->per.p[2]);
  };
  auto write_raw = [&](const string &name, int C, const vector<float> &out) {
    hsize_t dim[4] = {Nz, Ny, Nx, (hsize_t)C}, chunk[4] = {std::min<hsize_t>(1000, Nz), std::min<hsize_t>(1000, Ny), std::min<hsize_t>(1000, Nx), (hsize_t)C};
    hid_t sid = H5Screate_simple(4, dim, NULL), pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 4, chunk); H5Pset_deflate(pl, 7);
    hid_t did = H5Dcreate2(file, name.c_str(), H5T_NATIVE_FLOAT, sid, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(did, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.data());
    H5Dclose(did); H5Pclose(pl); H5Sclose(sid);
  };
  auto write4 = [&](const string &name, int C, const std::function<float(size_t, int)> &val) {   // val(extended index, component)
    vector<float> out((size_t)(Nx * Ny * Nz) * C); size_t o = 0;
    for (plint z = -1; z <= nz; z++) for (plint y = -1; y <= ny; y++) for (plint x = -1; x <= nx; x++) {
      const size_t k = node_of(x, y, z);
      for (int cidx = 0; cidx < C; cidx++) out[o++] = val(k, cidx);
    }
    write_raw(name, C, out);
  };
  vector<double> pi;   // off-equilibrium momentum flux, fetched once if a stress or strain-rate field is asked for
  auto need_pi = [&]() {
    if (!pi.empty()) return;
    vector<double> p0(6 * nn);
    hc_check(hcl_download_pi_neq(d, p0.data()), "hcl_download_pi_neq");
    pi.assign(6 * ne, 0.0); std::copy(p0.begin(), p0.end(), pi.begin() + 6 * plane);
    fill_ends(pi, 6);
  };
  for (int var : h.fluidOutputs) {
    if (var == OUTPUT_VELOCITY) write4("Velocity", 3, [&](size_t k, int cidx) { return (float)(u[3 * k + cidx] * (si ? Parameters::dx / Parameters::dt : 1.0)); });
    else if (var == OUTPUT_FORCE) {   // the external field as the driver left it (io/FluidHdf5IO.hh:240-262)
      const auto ext = L->external_now();
      write4("Force", 3, [&](size_t k, int cidx) {
        const plint xe = (plint)(k / plane) - 1, g = gx(xe) >= 0 ? gx(xe) : gx(xe < 0 ? 0 : nx - 1);
        return (float)(ext.at(g, (plint)(k / (size_t)nz % (size_t)ny), (plint)(k % (size_t)nz))[cidx] * (si ? Parameters::df : 1.0));
      });
    }
    else if (var == OUTPUT_DENSITY) write4("Density", 1, [&](size_t k, int) { return (float)(rho[k] * (si ? Parameters::df / (Parameters::dx * Parameters::dx) : 1.0)); });
    else if (var == OUTPUT_BOUNDARY) write4("Boundary", 1, [&](size_t k, int) { return maskx[k] ? 1.f : 0.f; });
    else if (var == OUTPUT_OMEGA)   // getDynamics().getOmega(), scaled like a stress in SI as the reference does (io/FluidHdf5IO.hh:352-372); a BounceBack node has none
      write4("Omega", 1, [&](size_t k, int) { return (float)((bbx[k] ? 0.0 : (double)L->omega) * (si ? Parameters::df / (Parameters::dx * Parameters::dx) : 1.0)); });
    else if (var == OUTPUT_SHEAR_STRESS) {   // Cell::computeShearStress (io/FluidHdf5IO.hh:404-432): (omega/2 - 1) PiNeq in the bulk, zero on BounceBack nodes
      need_pi();
      const double pre = 0.5 * (double)L->omega - 1.0;
      write4("ShearStress", 6, [&](size_t k, int cidx) { return (float)(bbx[k] ? 0.0 : pre * pi[6 * k + cidx] * (si ? Parameters::df / (Parameters::dx * Parameters::dx) : 1.0)); });
    } else if (var == OUTPUT_STRAIN_RATE) {   // computeStrainRateFromStress (io/FluidHdf5IO.hh:497-552): -omega / (2 cs2 rho) PiNeq
      need_pi();
      write4("StrainRate", 6, [&](size_t k, int cidx) { return (float)(bbx[k] ? 0.0 : -1.5 * (double)L->omega / rho[k] * pi[6 * k + cidx] * (si ? 1.0 / Parameters::dt : 1.0)); });
    } else if (var == OUTPUT_SHEAR_RATE) {   // central differences of computeVelocity (io/FluidHdf5IO.hh:434-495); row a*3+b = d u_a / d x_b
      vector<float> out((size_t)(Nx * Ny * Nz) * 9); size_t o = 0;
      for (plint z = -1; z <= nz; z++) for (plint y = -1; y <= ny; y++) for (plint x = -1; x <= nx; x++) {
        const size_t p[3] = {node_of(x + 1, y, z), node_of(x, y + 1, z), node_of(x, y, z + 1)}, m[3] = {node_of(x - 1, y, z), node_of(x, y - 1, z), node_of(x, y, z - 1)};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) out[o++] = (float)((u[3 * p[b] + a] - u[3 * m[b] + a]) / 2 * (si ? 1.0 / Parameters::dt : 1.0));
      }
      write_raw("ShearRate", 9, out);
    } else if (var == OUTPUT_CELL_DENSITY) {   // vertices per node, one dataset per cell type (io/FluidHdf5IO.hh:374-402)
      hc_cells *c = h.cellfields->device();
      long nvt = 0; hcp_counts(c, &nvt, nullptr, nullptr);
      vector<double> pos(3 * (size_t)nvt); if (nvt) hcp_download(c, 0, pos.data());
      vector<unsigned char> alive((size_t)nvt, 1); if (nvt) hc_check(hcp_download_alive(c, alive.data()), "hcp_download_alive");
      for (unsigned int t = 0; t < h.cellfields->size(); t++) {
        long fv = 0, nc = 0; hcp_type_range(c, (int)t, &fv, &nc);
        const HemoCellField &cf = *(*h.cellfields)[t];
        vector<float> out((size_t)(Nx * Ny * Nz), 0.f);
        for (long i = fv; i < fv + nc * cf.numVertex; i++) {
          if (!alive[(size_t)i]) continue;
          // the reference indexes the (Nx+2)-wide array with the node coordinate itself, i.e. one node towards the origin of
          // where the other datasets put that node (:388-393); kept, so that files compare
          plint q[3]; const plint dims[3] = {L->nx, ny, nz};
          for (int a = 0; a < 3; a++) { q[a] = (plint)std::floor(pos[(size_t)(3 * i + a)] + 0.5); if (L->per.p[a]) q[a] = ((q[a] % dims[a]) + dims[a]) % dims[a]; }
          q[0] -= x0;
          if (q[0] < 0 || q[0] >= nx || q[1] < 0 || q[1] >= ny || q[2] < 0 || q[2] >= nz) continue;   // findParticles(localDomain): this block's vertices
          out[(size_t)q[0] + (size_t)q[1] * Nx + (size_t)q[2] * Nx * Ny] += 1.f;
        }
        if (si) for (float &v : out) v *= (float)cf.volumeFractionOfLspPerNode;
        write_raw("CellDensity_" + cf.name, 1, out);
      }
    } else if (var == OUTPUT_BINDING_SITES || var == OUTPUT_INTERIOR_POINTS) {   // neither feature exists here: what the reference writes without them (:306-350)
      pcout << (var == OUTPUT_BINDING_SITES ? "(FluidHdf5) (Error) OUTPUT_BINDING_SITES requested, but binding sites not used, outputting a zero field"
                                            : "(FluidHdf5) (Error) OUTPUT_INTERIOR_POINTS requested, but interior viscosity not used, outputting a zero field") << endl;
      write_raw(var == OUTPUT_BINDING_SITES ? "BindingSites" : "InteriorPoints", 1, vector<float>((size_t)(Nx * Ny * Nz), 0.f));
    }
  }
  H5Fclose(file);
}

}  // namespace hemo
#endif
