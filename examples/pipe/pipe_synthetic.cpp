// Synthetic pipe flow driven through the source-level facade: the structure of a HemoCell pipeflow case
// (warm-up, iterate + re-applied driving force, periodic statistics) on an analytic cylinder instead of a
// voxelised STL.  Prints "STAT iter cells viscosity mean_force_pN" lines that the caller checks against the
// reference's validation bounds (tests/validation/pipeflow/test_pipeflow.cpp:87-106).
#define HEMOCELL_COMPAT_MAIN
#include "hemocell.h"
#include "cellInfo.h"
#include "fluidInfo.h"
#include "particleInfo.h"
#include "pltSimpleModel.h"
#include "rbcHighOrderModel.h"
#include "writeCellInfoCSV.h"

using namespace hemo;

int main(int argc, char *argv[]) {
  if (argc < 2) { cout << "Usage: " << argv[0] << " <configuration.xml>" << endl; return -1; }
  HemoCell hemocell(argv[1], argc, argv);
  Config *cfg = hemocell.cfg;

  const plint ny = (*cfg)["domain"]["refDirN"].read<int>() + 2, nz = ny, nx = (*cfg)["domain"]["lengthN"].read<int>();
  std::unique_ptr<MultiScalarField3D<int>> flagMatrix;
  std::unique_ptr<VoxelizedDomain3D<T>> voxelizedDomain;
  getFlagMatrixCylinder(nx, ny, nz, voxelizedDomain, flagMatrix);
  param::lbm_pipe_parameters(*cfg, flagMatrix.get());
  param::printParameters();

  hemocell.lattice = new MultiBlockLattice3D<T, DESCRIPTOR>(
      voxelizedDomain->getMultiBlockManagement(), defaultMultiBlockPolicy3D().getBlockCommunicator(),
      defaultMultiBlockPolicy3D().getCombinedStatistics(), defaultMultiBlockPolicy3D().getMultiCellAccess<T, DESCRIPTOR>(),
      new GuoExternalForceBGKdynamics<T, DESCRIPTOR>(1.0 / param::tau));
  defineDynamics(*hemocell.lattice, *flagMatrix, hemocell.lattice->getBoundingBox(), new BounceBack<T, DESCRIPTOR>(1.), 0);
  hemocell.lattice->toggleInternalStatistics(false);
  hemocell.lattice->periodicity().toggleAll(false);
  hemocell.latticeEquilibrium(1., plb::Array<T, 3>(0., 0., 0.));
  const T drivingForce = 8 * param::nu_lbm * (param::u_lbm_max * 0.5) / param::pipe_radius / param::pipe_radius;
  hemocell.lattice->initialize();

  hemocell.initializeCellfield();
  hemocell.addCellType<RbcHighOrderModel>("RBC", RBC_FROM_SPHERE);
  hemocell.setMaterialTimeScaleSeparation("RBC", (*cfg)["ibm"]["stepMaterialEvery"].read<int>());
  hemocell.setInitialMinimumDistanceFromSolid("RBC", 0.5);
  hemocell.addCellType<PltSimpleModel>("PLT", ELLIPSOID_FROM_SPHERE);
  hemocell.setMaterialTimeScaleSeparation("PLT", (*cfg)["ibm"]["stepMaterialEvery"].read<int>());
  hemocell.setParticleVelocityUpdateTimeScaleSeparation((*cfg)["ibm"]["stepParticleEvery"].read<int>());
  vector<int> outputs = {OUTPUT_POSITION, OUTPUT_TRIANGLES, OUTPUT_FORCE, OUTPUT_FORCE_VOLUME, OUTPUT_FORCE_BENDING, OUTPUT_FORCE_LINK, OUTPUT_FORCE_AREA, OUTPUT_FORCE_VISC, OUTPUT_CELL_ID, OUTPUT_VERTEX_ID};
  hemocell.setOutputs("RBC", outputs);
  outputs.push_back(OUTPUT_INNER_LINKS); outputs.push_back(OUTPUT_FORCE_INNER_LINK);
  hemocell.setOutputs("PLT", outputs);
  hemocell.setFluidOutputs({OUTPUT_VELOCITY, OUTPUT_DENSITY, OUTPUT_FORCE, OUTPUT_BOUNDARY});
  hemocell.setSystemPeriodicity(0, true);
  if (not cfg->checkpointed) hemocell.loadParticles();
  else hemocell.loadCheckPoint();       // started with <out>/checkpoint/checkpoint.xml as configuration

  setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt,
                    plb::Array<T, DESCRIPTOR<T>::d>(drivingForce, 0.0, 0.0));
  if (hemocell.iter == 0)
    for (plint i = 0; i < (*cfg)["parameters"]["warmup"].read<plint>(); ++i) hemocell.lattice->collideAndStream();

  if (std::getenv("HEMOCELL_PRINT_KERNEL_TIMES")) { hemocell.flush(); hemo::global.statistics.start(); }
  const unsigned tcheckpoint = (*cfg)["sim"]["tcheckpoint"].read<unsigned int>();
  const unsigned tmax = (*cfg)["sim"]["tmax"].read<unsigned int>(), tmeas = (*cfg)["sim"]["tmeas"].read<unsigned int>();
  while (hemocell.iter < tmax) {
    hemocell.iterate();
    setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt,
                      plb::Array<T, DESCRIPTOR<T>::d>(drivingForce, 0.0, 0.0));
    if (hemocell.iter % tmeas == 0) {
      FluidStatistics finfo = FluidInfo::calculateVelocityStatistics(&hemocell);
      ParticleStatistics pinfo = ParticleInfo::calculateForceStatistics(&hemocell);
      std::printf("STAT %u %lu %lu %lu %.8f %.8f\n", hemocell.iter, CellInformationFunctionals::getTotalNumberOfCells(&hemocell),
                  CellInformationFunctionals::getNumberOfCellsFromType(&hemocell, "RBC"), CellInformationFunctionals::getNumberOfCellsFromType(&hemocell, "PLT"),
                  (param::u_lbm_max * 0.5) / finfo.avg, pinfo.avg * param::df * 1.0e12);
    }
    if (hemocell.iter % tcheckpoint == 0) hemocell.saveCheckPoint();
  }
  if (std::getenv("HEMOCELL_PRINT_KERNEL_TIMES")) {   // the library's hipEvent timers (what bench.py reports as kernel_ms), for a like-for-like comparison
    const char *names[] = {"collide_stream_alone", "collide_stream_beside", "ibm_spread", "ibm_interpolate", "advance", "mechanics"};
    for (const char *k : names) { double ms = 0; long n = 0; hc_profile_read(k, &ms, &n); std::printf("KERNEL %s %.4f ms x %ld\n", k, n ? ms / n : 0.0, n); }
  }
  hemocell.writeOutput();   // <out>/hdf5/<iter>/{RBC,PLT,Fluid}.<iter>.p.0.h5 + <out>/csv
  return 0;
}
