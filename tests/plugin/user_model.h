// A mechanics model as a HemoCell user would write it against the reference's plugin interface
// (mechanics/cellMechanics.h:36-47, the shape of mechanics/rbcHighOrderModel.h): constructed as
// new Model(Config&, HemoCellField&) by HemoCell::addCellType<Model> (hemocell.h:122-128), overriding
// ParticleMechanics(map<int, vector<HemoCellParticle*>>&, const map<int,bool>&, pluint) and statistics().
// Nothing in here knows about the GPU back end: it must COMPILE unchanged against hemocell_amd/compat and be
// refused at addCellType with the reference's log + exit(1), because its force law is host code.
#ifndef TEST_USER_MODEL_H
#define TEST_USER_MODEL_H
#include "cellMechanics.h"
#include "hemoCellField.h"
#include "config.h"

namespace hemo {

class SpringToCentroidModel : public CellMechanics {
 public:
  HemoCellField &cellField;
  const T k_volume, k_area, k_link, k_bend, eta_m;

  SpringToCentroidModel(Config &modelCfg_, HemoCellField &cellField_)
      : CellMechanics(cellField_, modelCfg_), cellField(cellField_),
        k_volume(calculate_kVolume(modelCfg_, *cellField_.meshmetric)), k_area(calculate_kArea(modelCfg_, *cellField_.meshmetric)),
        k_link(calculate_kLink(modelCfg_, *cellField_.meshmetric)), k_bend(calculate_kBend(modelCfg_, *cellField_.meshmetric)),
        eta_m(calculate_etaM(modelCfg_)) {}

  void ParticleMechanics(std::map<int, std::vector<HemoCellParticle *>> &particles_per_cell, const std::map<int, bool> &lpc, pluint ctype) {
    for (const auto &pair : lpc) {
      const int &cid = pair.first;
      std::vector<HemoCellParticle *> &cell = particles_per_cell[cid];
      if (cell.size() == 0) continue;
      if (cell[0]->sv.celltype != ctype) continue;
      hemo::Array<T, 3> centre({0., 0., 0.});
      for (HemoCellParticle *p : cell) centre = centre + p->sv.position / T(cell.size());
      for (const hemo::Array<plint, 3> &triangle : cellConstants.triangle_list)
        for (int k = 0; k < 3; k++) *cell[triangle[k]]->force_area = *cell[triangle[k]]->force_area + (centre - cell[triangle[k]]->sv.position) * k_area;
      for (unsigned int e = 0; e < cellConstants.edge_list.size(); e++) {
        const hemo::Array<plint, 2> &edge = cellConstants.edge_list[e];
        const hemo::Array<T, 3> d = cell[edge[1]]->sv.position - cell[edge[0]]->sv.position;
        *cell[edge[0]]->force_link = *cell[edge[0]]->force_link + d * (k_link / cellConstants.edge_length_eq_list[e]);
        *cell[edge[1]]->force_link = *cell[edge[1]]->force_link - d * (k_link / cellConstants.edge_length_eq_list[e]);
      }
    }
  }
  void statistics() {
    hlog << "(Cell-mechanics model) SpringToCentroidModel for " << cellField.name << ": k_area " << k_area << " k_link " << k_link
         << " volume_eq " << cellConstants.volume_eq << " edges " << cellConstants.edge_list.size() << std::endl;
  }
};

}  // namespace hemo
#endif
