// Every fluid output variable of io/FluidHdf5IO.hh:139-199 on a small pipe with one RBC, written once after the flow has
// settled: tests/test_gpu_compat_driver.py checks the datasets against each other (shear rate = central differences of the
// velocity dataset, strain rate = its symmetric part, stress = 2 mu strain rate, ...).
#ifndef HEMOCELL_COMPAT_MAIN
#define HEMOCELL_COMPAT_MAIN
#endif
#include "hemocell.h"
#include "rbcHighOrderModel.h"

using namespace hemo;

int main(int argc, char *argv[]) {
  if (argc < 2) { cout << "Usage: " << argv[0] << " <configuration.xml>" << endl; return -1; }
  HemoCell hemocell(argv[1], argc, argv);
  Config *cfg = hemocell.cfg;
  param::lbm_base_parameters(*cfg);
  const plint nx = 96, ny = 34, nz = 34;   // two slabs of 48 planes when started as two ranks
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
  hemocell.setMaterialTimeScaleSeparation("RBC", 1);
  hemocell.setParticleVelocityUpdateTimeScaleSeparation(1);
  hemocell.setOutputs("RBC", {OUTPUT_POSITION});
  hemocell.setFluidOutputs({OUTPUT_VELOCITY, OUTPUT_DENSITY, OUTPUT_FORCE, OUTPUT_BOUNDARY, OUTPUT_OMEGA, OUTPUT_SHEAR_STRESS, OUTPUT_SHEAR_RATE,
                            OUTPUT_STRAIN_RATE, OUTPUT_CELL_DENSITY, OUTPUT_BINDING_SITES, OUTPUT_INTERIOR_POINTS});
  hemocell.setSystemPeriodicity(0, true);
  hemocell.loadParticles();
  const T R = (ny - 2) / 2.0, umax = 0.02;
  const T drivingForce = 4 * param::nu_lbm * umax / (R * R);
  setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt,
                    plb::Array<T, DESCRIPTOR<T>::d>(drivingForce, 0.0, 0.0));
  for (int i = 0; i < 6000; i++) hemocell.lattice->collideAndStream();
  for (int i = 0; i < 10; i++) {
    hemocell.iterate();
    setExternalVector(*hemocell.lattice, hemocell.lattice->getBoundingBox(), DESCRIPTOR<T>::ExternalField::forceBeginsAt,
                      plb::Array<T, DESCRIPTOR<T>::d>(drivingForce, 0.0, 0.0));
  }
  hemocell.writeOutput();
  std::printf("PARAMS tau %.10f dx %.6e dt %.6e df %.6e fraction %.8f\n", param::tau, param::dx, param::dt, param::df, (*hemocell.cellfields)["RBC"]->volumeFractionOfLspPerNode);
  return 0;
}
