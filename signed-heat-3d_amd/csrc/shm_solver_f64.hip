// Solver<double> and every kernel instantiation it launches: one of the two large translation units of libshm_grid.so (the other precision is its twin; they compile in parallel).
#include "shm_solver.hip.h"

namespace shm {
SolverBase* make_solver_f64(const shm_config& cfg) { return new Solver<double>(cfg); }
}  // namespace shm
