// helper/hemocellInit.hh:71-93 iniLatticeSquareCouette: shear between a top and a bottom wall moving in +-x
#pragma once
#include "../hemocell.h"

template <typename U, template <class V> class Descriptor>
void iniLatticeSquareCouette(plb::MultiBlockLattice3D<U, Descriptor> &lattice, plint nx, plint ny, plint nz,
                             plb::OnLatticeBoundaryCondition3D<U, Descriptor> &boundaryCondition, U shearRate) {
  const plb::Box3D top(0, nx - 1, 0, ny - 1, nz - 1, nz - 1), bottom(0, nx - 1, 0, ny - 1, 0, 0);
  lattice.periodicity().toggle(0, true);
  lattice.periodicity().toggle(1, true);
  lattice.periodicity().toggle(2, false);
  boundaryCondition.setVelocityConditionOnBlockBoundaries(lattice, top);
  boundaryCondition.setVelocityConditionOnBlockBoundaries(lattice, bottom);
  // The reference's velocity nodes ARE the wall (z = 0 and z = nz-1, +-vHalf there, helper/hemocellInit.hh:82-84).  This
  // back end's moving wall (bounce-back + momentum term) acts half a node inside the wall node, so it is given the
  // velocity the reference's linear profile has at that position: every fluid node then sees the reference's
  // u(z) = vHalf (1 - 2 z / (nz-1)), i.e. the same shear rate.
  const U vHalf = (nz - 1) * shearRate * 0.5;
  const U vWall = vHalf * (U)(nz - 2) / (U)(nz - 1);
  plb::setBoundaryVelocity(lattice, top, plb::Array<U, 3>(-vWall, 0.0, 0.0));
  plb::setBoundaryVelocity(lattice, bottom, plb::Array<U, 3>(vWall, 0.0, 0.0));
  plb::setExternalVector(lattice, lattice.getBoundingBox(), Descriptor<U>::ExternalField::forceBeginsAt, plb::Array<U, 3>(0.0, 0.0, 0.0));
  lattice.initialize();
}
