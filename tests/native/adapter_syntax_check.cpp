// Typo guard (g++ -fsyntax-only, never linked): the adapter header against the stub declarations in adapter_stubs/, used the way
// the reference's src/main.cpp uses the class (:52, :88-100, :290-292).  See adapter_stubs/README.md -- not parity evidence.
#include "adapter_geometrycentral.h"

#include <memory>

static std::unique_ptr<SignedHeatGridSolver> gridSolver;

double use_like_main_cpp(VertexPositionGeometry& geometry, pointcloud::PointPositionNormalGeometry& pointGeom, bool isCloud) {
    gridSolver = std::unique_ptr<SignedHeatGridSolver>(new SignedHeatGridSolver());
    gridSolver->VERBOSE = false;
    SignedHeat3DOptions SHM_OPTIONS;
    SHM_OPTIONS.hCoef = 1.0;
    SHM_OPTIONS.rebuild = false;
    SHM_OPTIONS.fastIntegration = true;
    Vector<double> phi = isCloud ? gridSolver->computeDistance(pointGeom, SHM_OPTIONS) : gridSolver->computeDistance(geometry, SHM_OPTIONS);
    return phi.size() ? phi[0] : 0.;
}
