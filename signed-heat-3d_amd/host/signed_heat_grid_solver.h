// SignedHeatGridSolver -- same class surface as the reference's include/signed_heat_grid_solver.h:11-22
// (constructor, two computeDistance overloads, public VERBOSE), backed by the gfx950 library through the
// C ABI of include/shm_grid.h.  The reference's private Eigen state (laplaceMat, poissonSolver, faceAreas, ...)
// has no counterpart: the Laplacian is matrix-free on the device and the dead Cholesky factorisation
// (signed_heat_grid_solver.cpp:30, never solved with) is not reproduced.
#pragma once
#include <memory>
#include <vector>

#include "../../include/shm_grid.h"
#include "signed_heat_3d.h"

namespace shm_host {

// Allocator that default-initialises: VectorXd(n) allocates without zero-filling, like the Eigen::VectorXd it stands for (the
// reference's `Vector<double> phi` is written by the solve, never read before); a value-initialising std::vector would page in and
// zero 1 GB at 512^3 before the device copy overwrites it.
template <typename T> struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... Args> void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
};
using VectorXd = std::vector<double, DefaultInitAllocator<double>>;  // stands where geometrycentral::Vector<double> (Eigen::VectorXd) stands in the demo

class SignedHeatGridSolver {
  public:
    SignedHeatGridSolver();
    explicit SignedHeatGridSolver(const GridBackendOptions& backend);
    ~SignedHeatGridSolver();

    // signed_heat_grid_solver.cpp:5-114
    VectorXd computeDistance(VertexPositionGeometry& geometry, const SignedHeat3DOptions& options = SignedHeat3DOptions());
    // signed_heat_grid_solver.cpp:116-222
    VectorXd computeDistance(PointPositionNormalGeometry& pointGeom, const SignedHeat3DOptions& options = SignedHeat3DOptions());

    bool VERBOSE = true;

    // Headless stand-in for the demo's contour()/export path (src/main.cpp:116-128,167-191, done there by Polyscope's marching
    // cubes): isosurface of the phi of the LAST computeDistance() call, extracted on the device (marching cubes: shm_grid_isosurface).
    void isosurface(double isoval, std::vector<Vector3>& vertices, std::vector<std::array<size_t, 3>>& faces);

    // Read-only views of the grid block the reference keeps private (used by the CLI / tests / the Polyscope
    // side effect `registerVolumeGrid("domain", {nx,ny,nz}, bboxMin, bboxMax)`, :35 / :143).
    size_t gridSize() const { return nx; }
    Vector3 gridMin() const { return bboxMin; }
    Vector3 gridMax() const { return bboxMax; }
    double gridCell() const { return cellSize; }
    const shm_stats& lastStats() const { return stats; }
    // Sources as handed to the device (pos, wnormal, area, lambda) -- exposed for parity tests.
    const std::vector<double>& lastSourcePositions() const { return srcPos; }
    const std::vector<double>& lastSourceWeightedNormals() const { return srcWn; }
    const std::vector<double>& lastSourceAreas() const { return srcArea; }
    double lastLambda() const { return lambda; }

  private:
    GridBackendOptions backend;
    shm_solver* handle = nullptr;
    bool gridBuilt = false;  // plays the role of `poissonSolver != nullptr` (:8); never set by the point overload (:119)
    size_t nx = 0, ny = 0, nz = 0;
    Vector3 bboxMin, bboxMax;
    double shortTime = 0., cellSize = 0., lambda = 0.;
    std::vector<double> srcPos, srcWn, srcArea;
    shm_stats stats{};

    void ensureHandle();
    void buildGrid(const Vector3& c, double r, const SignedHeat3DOptions& options);
    VectorXd solveOnDevice(bool scrub, const SignedHeat3DOptions& options);
};

}  // namespace shm_host
