#!/usr/bin/env python3
"""Generate tests/golden/material_patch.json from the reference's own numpy
material tester (tools/materialTester/getModuli/{mesh,rbcHO}.py).

Run in the build container only (needs /root/reference); the JSON it writes is
the committed fixture.  The fixture is data: seeded node positions of the
7-node hexagonal patch (SI units), the reference's k_link / k_area / l_eq /
area_eq, and the node forces its Cell.calcConstitutiveForces() returns
(link law k(ef+ef/(9-ef^2)) + area law k(r+r/(0.09-r^2)), velocities zero so
the membrane-viscosity term vanishes).
"""
import json, os, sys
import numpy as np

REF = os.environ.get("HEMOCELL_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "tools", "materialTester"))
from getModuli.rbcHO import Cell  # noqa: E402

def main():
    rng = np.random.default_rng(20260404)
    cases = []
    for l_eq in (0.5e-6, 0.35e-6):
        for amp in (0.0, 0.01, 0.03, 0.06):
            model = Cell(l_eq)
            nodes = model.mesh.nodes
            for n in nodes:
                n.position = n.position + amp * l_eq * rng.standard_normal(3)
                n.force.fill(0.0)
            pos = np.array([n.position for n in nodes])
            model.calcConstitutiveForces()
            frc = np.array([n.force for n in nodes])
            idx = {id(n): i for i, n in enumerate(nodes)}
            edges = [[idx[id(e[0])], idx[id(e[1])]] for e in model.mesh.edges]
            faces = [[idx[id(f[0])], idx[id(f[1])], idx[id(f[2])]] for f in model.mesh.faces]
            cases.append(dict(l_eq=l_eq, amp=amp, area_eq=float(model.area_eq),
                              k_link=float(model.k_link), k_area=float(model.k_area),
                              edges=edges, faces=faces,
                              positions=pos.tolist(), forces=frc.tolist()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "material_patch.json")
    with open(out, "w") as f:
        json.dump(dict(source="tools/materialTester/getModuli/rbcHO.py:Cell.calcConstitutiveForces",
                       cases=cases), f, indent=1)
    print("wrote", out, len(cases), "cases")

if __name__ == "__main__":
    main()
