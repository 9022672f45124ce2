// a checkpoint written while a cell is incomplete (single particles removed at a wall,
// core/hemoCellParticleField.cpp:304-321) must load again -- the reference's load path runs deleteIncompleteCells afterwards
// (core/hemoCellFields.cpp:272-274) -- and a truncated dump must be refused, not resumed from.
//   checkpoint_incomplete config.xml save   push cell 1 into the pipe wall, saveCheckPoint, loadCheckPoint
//   checkpoint_incomplete config.xml load   loadCheckPoint only (tests/test_gpu_compat_driver.py truncates the dump first)
#ifndef HEMOCELL_COMPAT_MAIN
#define HEMOCELL_COMPAT_MAIN
#endif
#include "hemocell.h"
#include "rbcHighOrderModel.h"

using namespace hemo;

int main(int argc, char *argv[]) {
  if (argc < 3) { cout << "Usage: " << argv[0] << " <configuration.xml> save|load" << endl; return -1; }
  const string mode = argv[2];
  HemoCell hemocell(argv[1], argc, argv);
  Config *cfg = hemocell.cfg;
  param::lbm_base_parameters(*cfg);
  const plint nx = 96, ny = 34, nz = 34;
  std::unique_ptr<MultiScalarField3D<int>> flagMatrix;
  std::unique_ptr<VoxelizedDomain3D<T>> voxelizedDomain;
  getFlagMatrixCylinder(nx, ny, nz, voxelizedDomain, flagMatrix);
  hemocell.initializeLattice(voxelizedDomain->getMultiBlockManagement());
  defineDynamics(*hemocell.lattice, *flagMatrix, hemocell.lattice->getBoundingBox(), new BounceBack<T, DESCRIPTOR>(1.), 0);
  hemocell.lattice->toggleInternalStatistics(false);
  hemocell.lattice->periodicity().toggleAll(false);
  hemocell.latticeEquilibrium(1., plb::Array<T, 3>(0., 0., 0.));
  hemocell.lattice->initialize();
  hemocell.initializeCellfield();
  hemocell.addCellType<RbcHighOrderModel>("RBC", RBC_FROM_SPHERE);
  hemocell.setMaterialTimeScaleSeparation("RBC", 4);
  hemocell.setParticleVelocityUpdateTimeScaleSeparation(60);
  hemocell.setOutputs("RBC", {OUTPUT_POSITION});
  hemocell.setFluidOutputs({OUTPUT_VELOCITY});
  hemocell.setSystemPeriodicity(0, true);
  hemocell.loadParticles();
  hc_cells *c = hemocell.cellfields->device();
  long nv = 0, nc = 0, inc = 0, miss = 0;
  if (mode == "save") {
    hemocell.iterate();                                       // iteration 0 interpolates; velocities are then held for 60 iterations
    c = hemocell.cellfields->device();
    hcp_counts(c, &nv, &nc, nullptr);
    vector<double> vel((size_t)(3 * nv));
    hc_check(hcp_download(c, 1, vel.data()), "hcp_download");
    for (long i = nv / 2; i < nv; i++) { vel[3 * i] = 0.0; vel[3 * i + 1] = 0.0; vel[3 * i + 2] = 0.1; }   // cell 1 towards the wall
    hc_check(hcp_upload(c, 1, vel.data()), "hcp_upload");
    for (int i = 0; i < 55; i++) hemocell.iterate();
    c = hemocell.cellfields->device();
    hcp_counts(c, &nv, &nc, nullptr); hcp_deletion_counts(c, nullptr, nullptr, &inc, &miss);
    std::printf("BEFORE cells %ld incomplete %ld missing %ld\n", nc, inc, miss);
    hemocell.saveCheckPoint();
  }
  hemocell.loadCheckPoint();
  c = hemocell.cellfields->device();
  hcp_counts(c, &nv, &nc, nullptr); hcp_deletion_counts(c, nullptr, nullptr, &inc, &miss);
  std::printf("RESUMED iteration %u cells %ld incomplete %ld missing %ld\n", hemocell.iter, nc, inc, miss);
  for (int i = 0; i < 8; i++) hemocell.iterate();             // and the run goes on (a material step among them)
  c = hemocell.cellfields->device();
  hcp_counts(c, &nv, &nc, nullptr);
  std::printf("CONTINUED iteration %u cells %ld\n", hemocell.iter, nc);
  return 0;
}
