// shm_grid_cli -- headless re-creation of the demo's solve() for the grid solver (src/main.cpp:68-114, 227-262).
// Flags follow the reference (`--g/--grid`, `--f/--fast`, `--V/--verbose`, README.md:65-71) plus the `--h` the README
// documents but the reference never parsed, `--t`, and backend/output knobs.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "mesh_io.h"
#include "signed_heat_grid_solver.h"

using namespace shm_host;

static void usage() {
    std::cout << "  shm_grid_cli {mesh} {OPTIONS}\n\n    Solve for generalized signed distance on a background grid (MI355X).\n\n"
                 "  OPTIONS\n      --help            Display this help menu\n      mesh              A mesh (.obj) or point cloud (.pc) file\n"
                 "      --g, --grid       Solve on a background grid (always on: the tet path is not part of this build)\n"
                 "      --f, --fast       Less accurate, faster integration (BFS)\n      --V, --verbose    Verbose output\n"
                 "      --h <hCoef>       Grid resolution: n = 2*2^(hCoef+3) nodes per side (default 0 -> 16^3)\n"
                 "      --t <tCoef>       Diffusion time coefficient (default 1)\n      --fp32            Compute in fp32 (default fp64)\n"
                 "      --exact-step1     fp64 only: every (node, source) pair of Step 1 in fp64 like the reference (default: error-budgeted tiers)\n"
                 "      --tol <x>         Projected-CG relative residual tolerance\n      --device <i>      HIP device ordinal\n"
                 "      --out <file>      Write phi as raw little-endian float64 (n^3 values, x fastest)\n"
                 "      --iso <value>     Contour phi at this value (default 0) and export the isosurface\n"
                 "      --export <file>   OBJ file of the isosurface (the demo writes ../export/isosurface.obj)\n";
}

int main(int argc, char** argv) {
    std::string path, out, exportPath;
    double isoval = 0.;
    SignedHeat3DOptions opts;
    GridBackendOptions backend;
    bool verbose = false;
    for (int a = 1; a < argc; a++) {
        const std::string s = argv[a];
        auto need = [&](const char* what) -> const char* {
            if (a + 1 >= argc) {
                std::cerr << "missing value for " << what << std::endl;
                exit(1);
            }
            return argv[++a];
        };
        if (s == "--help") { usage(); return 0; }
        else if (s == "--g" || s == "--grid") {}
        else if (s == "--f" || s == "--fast") opts.fastIntegration = true;
        else if (s == "--V" || s == "--verbose") verbose = true;
        else if (s == "--h") opts.hCoef = atof(need("--h"));
        else if (s == "--t") opts.tCoef = atof(need("--t"));
        else if (s == "--fp32") backend.precision = 32;
        else if (s == "--exact-step1") backend.exactStep1 = true;
        else if (s == "--tol") backend.tol = atof(need("--tol"));
        else if (s == "--device") backend.device = atoi(need("--device"));
        else if (s == "--out") out = need("--out");
        else if (s == "--iso") isoval = atof(need("--iso"));
        else if (s == "--export") exportPath = need("--export");
        else if (!s.empty() && s[0] == '-') { std::cerr << "Flag could not be matched: " << s << std::endl; usage(); return 1; }
        else path = s;
    }
    if (path.empty()) {
        std::cerr << "Please specify a mesh file as argument." << std::endl;
        return EXIT_FAILURE;
    }
    try {
        SignedHeatGridSolver solver(backend);
        solver.VERBOSE = verbose;
        const std::string ext = path.substr(path.find_last_of(".") + 1);
        VectorXd phi;
        const auto t1 = std::chrono::high_resolution_clock::now();
        if (ext != "pc") {
            VertexPositionGeometry geometry = readSurfaceMesh(path);
            phi = solver.computeDistance(geometry, opts);
        } else {
            PointPositionNormalGeometry pointGeom = readPointCloud(path);
            phi = solver.computeDistance(pointGeom, opts);
        }
        const auto t2 = std::chrono::high_resolution_clock::now();
        if (verbose) std::cerr << "Solve time (s): " << std::chrono::duration<double>(t2 - t1).count() << std::endl;
        const auto mm = std::minmax_element(phi.begin(), phi.end());
        std::cerr << "min: " << *mm.first << "\tmax: " << *mm.second << std::endl;  // src/main.cpp:101
        if (!out.empty()) {
            std::ofstream f(out, std::ios::binary);
            f.write((const char*)phi.data(), (std::streamsize)(phi.size() * sizeof(double)));
            std::cerr << "phi (" << solver.gridSize() << "^3 float64) written to " << out << std::endl;
        }
        if (!exportPath.empty()) {
            std::vector<Vector3> iv;
            std::vector<std::array<size_t, 3>> jf;
            solver.isosurface(isoval, iv, jf);
            writeSurfaceMesh(iv, jf, exportPath);
            std::cerr << "Isosurface written to " << exportPath << " (" << iv.size() << " vertices, " << jf.size() << " triangles)" << std::endl;
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    return EXIT_SUCCESS;
}
