// Exercises the plugin surface of SURVEY.md section 8(b) that the four reference drivers do not touch:
//   plugin_driver surface  <config.xml>   HemoCellParticleField (particles, get_particles_per_cell, get_lpc, localDomain),
//                                         HemoCellField::kernelMethod, CellMechanics::cellConstants, phase methods of HemoCellFields
//   plugin_driver usermodel <config.xml>  addCellType<user model> -> refused with log + exit(1)
//   plugin_driver userkernel <config.xml> a user IBM kernelMethod -> refused with log + exit(1)
//   plugin_driver forceonce <config.xml>  setExternalVector ONCE, then ten iterations: the reference zeroes the field at the end of
//                                         every iterate() (core/hemoCell.cpp:369-371), so only the first one is driven
#include "hemocell.h"
#include "fluidInfo.h"
#include "rbcHighOrderModel.h"
#include "user_model.h"
#include <cstring>

using namespace hemo;

static void myKernel(plb::BlockLattice3D<T, DESCRIPTOR> &, HemoCellParticle &) {}

int main(int argc, char *argv[]) {
  if (argc < 3) { std::cout << "usage: plugin_driver surface|usermodel|userkernel config.xml" << std::endl; return 2; }
  const std::string mode = argv[1];
  HemoCell hemocell(argv[2], argc, argv);
  Config *cfg = hemocell.cfg;
  param::lbm_base_parameters(*cfg);
  const plint nx = 40, ny = 30, nz = 30;
  hemocell.lattice = new MultiBlockLattice3D<T, DESCRIPTOR>(defaultMultiBlockPolicy3D().getMultiBlockManagement(nx, ny, nz, 1),
                                                            defaultMultiBlockPolicy3D().getBlockCommunicator(), defaultMultiBlockPolicy3D().getCombinedStatistics(),
                                                            defaultMultiBlockPolicy3D().getMultiCellAccess<T, DESCRIPTOR>(),
                                                            new GuoExternalForceBGKdynamics<T, DESCRIPTOR>(1.0 / param::tau));
  hemocell.lattice->toggleInternalStatistics(false);
  hemocell.lattice->periodicity().toggleAll(true);
  hemocell.latticeEquilibrium(1., hemo::Array<T, 3>({0., 0., 0.}));
  hemocell.lattice->initialize();
  hemocell.initializeCellfield();
  if (mode == "usermodel") {
    hemocell.addCellType<SpringToCentroidModel>("RBC", RBC_FROM_SPHERE);
    std::cout << "NOT REFUSED" << std::endl;
    return 0;
  }
  hemocell.addCellType<RbcHighOrderModel>("RBC", RBC_FROM_SPHERE);
  hemocell.setMaterialTimeScaleSeparation("RBC", 1);
  hemocell.setParticleVelocityUpdateTimeScaleSeparation(1);
  HemoCellField &field = *(*hemocell.cellfields)["RBC"];
  if (field.kernelMethod != interpolationCoefficientsPhi2) { std::cout << "kernelMethod default wrong" << std::endl; return 1; }
  if (mode == "userkernel") field.kernelMethod = myKernel;
  hemocell.loadParticles();
  if (mode == "userkernel") { std::cout << "NOT REFUSED" << std::endl; return 0; }
  if (mode == "forceonce") {
    const T F = 1e-5;
    setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt, plb::Array<T, 3>(F, 0., 0.));
    for (int it = 0; it < 10; it++) hemocell.iterate();
    FluidStatistics once = FluidInfo::calculateVelocityStatistics(&hemocell);
    for (int it = 0; it < 10; it++) {   // the pattern of the shipped drivers: written again after every iteration
      hemocell.iterate();
      setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt, plb::Array<T, 3>(F, 0., 0.));
    }
    FluidStatistics again = FluidInfo::calculateVelocityStatistics(&hemocell);
    std::printf("forceonce F %.6e after_ten_with_one_write %.6e after_ten_more_written_every_time %.6e\n", F, once.avg, again.avg);
    return 0;
  }

  // mechanics/cellMechanics.h:39: the constants a model reads
  const CommonCellConstants &cc = field.mechanics->cellConstants;
  std::cout << "constants " << cc.triangle_list.size() << " " << cc.edge_list.size() << " " << cc.vertex_vertexes.size() << " " << cc.edge_bending_triangles_list.size()
            << " " << cc.vertex_edges.size() << " " << cc.volume_eq << std::endl;
  for (int it = 0; it < 5; it++) hemocell.iterate();
  // core/hemoCellFields.h:161 + core/hemoCellParticleField.h:138,175-178
  HemoCellParticleField &pf = hemocell.cellfields->immersedParticles->getComponent(0);
  const std::map<int, std::vector<int>> &ppc = pf.get_particles_per_cell();
  const std::map<int, bool> &lpc = pf.get_lpc();
  size_t complete = 0;
  for (auto &kv : ppc) { bool all = true; for (int i : kv.second) all = all && i >= 0; complete += all; }
  double fsum[3] = {0, 0, 0}, vmax = 0;
  for (HemoCellParticle &p : pf.particles) { for (int d = 0; d < 3; d++) fsum[d] += p.sv.force[d]; vmax = std::max(vmax, std::fabs(p.sv.v[0])); }
  std::cout << "particles " << pf.particles.size() << " cells " << ppc.size() << " complete " << complete << " lpc " << lpc.size() << " localDomain "
            << pf.localDomain.x0 << " " << pf.localDomain.x1 << " getsize " << pf.getsize() << std::endl;
  std::cout << "force_sum " << fsum[0] << " " << fsum[1] << " " << fsum[2] << " vmax " << vmax << std::endl;
  // an edit through the view goes back to the device: shift the cell by one node, then iterate on
  for (HemoCellParticle &p : pf.particles) p.sv.position[1] += 1.0;
  pf.upload();
  hemocell.cellfields->applyConstitutiveModel(true);
  hemocell.cellfields->spreadParticleForce();
  hemocell.iterate();
  CellInformationFunctionals::calculateCellPosition(&hemocell);
  for (auto &kv : CellInformationFunctionals::info_per_cell) std::cout << "cell " << kv.first << " y " << kv.second.position[1] << std::endl;
  // phase methods a driver may call one by one (core/hemoCellFields.h:103-158)
  hemocell.setRepulsion(2e-22, 0.7);
  hemocell.cellfields->applyRepulsionForce();
  hemocell.cellfields->separate_force_vectors(); hemocell.cellfields->unify_force_vectors();
  hemocell.cellfields->syncEnvelopes(); hemocell.cellfields->deleteNonLocalParticles(3);
  hemocell.doLoadBalance();
  if (hemocell.calculateFractionalLoadImbalance() != 0) { std::cout << "load imbalance of a single slab is not 0" << std::endl; return 1; }
  hemocell.iterate();
  std::cout << "SURFACE OK" << std::endl;
  return 0;
}
