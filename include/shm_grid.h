/* shm_grid.h -- C ABI of the MI355X (gfx950) regular-grid Signed Heat Method solver.
 *
 * Drop-in boundary for the hot path of nzfeng/signed-heat-3d:
 *     SignedHeatGridSolver::computeDistance()      src/signed_heat_grid_solver.cpp:5-114  (mesh source)
 *                                                  src/signed_heat_grid_solver.cpp:116-222 (point source)
 * The reference has no FFI; the C++ adapter (signed-heat-3d_amd/host/signed_heat_grid_solver.h, same
 * class surface as include/signed_heat_grid_solver.h:11-22 of the reference) computes the cheap
 * geometry-dependent pre-processing on the host (centroid/radius/meanEdgeLength/setFaceVectorAreas/
 * barycenter, signed_heat_3d.cpp:3-89, signed_heat_grid_solver.cpp:498-503) and hands flat arrays to
 * this library.  Everything from the Step-1 summation to the final shift runs on the GPU.
 *
 * Conventions: plain pointers and sizes only; the caller owns every host buffer; the library owns
 * all device memory; no exception crosses this boundary; every function returns an shm_status and
 * shm_grid_last_error() gives the message; one solve at a time per handle; a handle pins one HIP
 * device and its own streams.  Node flattening: idx = i + j*n + k*n*n (x fastest),
 * signed_heat_grid_solver.cpp:505-508.  There is NO CPU fallback: without a HIP device
 * shm_grid_create fails with SHM_ERR_HIP.
 */
#ifndef SHM_GRID_H
#define SHM_GRID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHM_GRID_ABI_VERSION 5 /* 5: shm_opts grew by dual_form and step1_budget (round 5); 2: shm_stats grew by cg_form (round 2); 3: by pairs_fp64 / pairs_fp32 / conv_launches (round 3); 4: shm_opts grew by step1_arith, shm_stats by pairs_redone,
                                * shm_grid_run_conv_arith and shm_grid_get_field_planes added (round 4); callers allocate shm_opts / shm_stats by this header */

typedef struct shm_solver shm_solver; /* opaque */

typedef enum {
    SHM_OK = 0,
    SHM_ERR_INVALID = 1,   /* bad argument / inconsistent sizes */
    SHM_ERR_HIP = 2,       /* HIP runtime error (incl. "no device") */
    SHM_ERR_NOMEM = 3,     /* allocation failed */
    SHM_ERR_BREAKDOWN = 4, /* CG breakdown (p.Kp <= 0 or non-finite residual) */
    SHM_ERR_NOCONV = 5,    /* max_iters reached before tol (phi is still returned) */
    SHM_ERR_RCCL = 6,      /* RCCL error */
    SHM_ERR_STATE = 7,     /* call order violated (e.g. solve before set_problem) */
    SHM_ERR_SINGULAR = 8   /* A A^T not positive definite */
} shm_status;

enum { SHM_F64 = 64, SHM_F32 = 32 };

/* Construction-time configuration (replaces `new SignedHeatGridSolver()`, src/main.cpp:290). */
typedef struct {
    int32_t device;      /* HIP device ordinal */
    int32_t precision;   /* SHM_F64 (reference arithmetic) or SHM_F32 */
    int32_t local_slabs; /* z-slabs owned by THIS process (>=1).  1 in production; >1 runs the same
                            slab/halo/partial-reduction code path on one GPU (loop-back transport). */
    int32_t rank;        /* rank of this process among `world` processes (0 when world==1) */
    int32_t world;       /* number of processes = GPUs; z-slabs are split rank-major */
    int32_t verbose;     /* mirrors SignedHeatGridSolver::VERBOSE (signed_heat_grid_solver.h:22) */
    const void* rccl_unique_id; /* 128-byte ncclUniqueId from shm_comm_unique_id() of rank 0; NULL iff world==1 */
    int32_t slab_plan;   /* SHM_SLAB_PLAN_EQUAL (0): n / slabs planes each (shm_plan_slab).  SHM_SLAB_PLAN_STEP1: planes weighted by the Step-1 work the
                          * culling / precision tiers leave in them (shm_step1_plane_weights + shm_plan_slab_weighted; every rank derives the same plan from
                          * the sources).  The weighted plan serves the default multi-rank solve (Steps 1-2 on slabs, D^T Y gathered, whole-grid dual solve)
                          * and the plain stencil CG; the slab-distributed transforms (DUAL_SLABS, PRIMAL + DCT) need equal slabs and are refused with it:
                          * their two all-to-alls turn P z-slabs into P y-pencils by exchanging P x P congruent blocks of (n/P) x (n/P) x n elements, addressed
                          * in the sweeps by shift / mask (segment = n/P rows, a power of two).  Unequal slabs would need unequal pencils, per-pair counts in the
                          * exchange and division-based addressing in the y sweeps -- for a solve phase that is 3-10 % of a multi-rank solve, while the weighted
                          * plan exists to balance Step 1 (80-97 %).  A caller that wants both runs the default (gathered) solve, which has both. */
} shm_config;
enum { SHM_SLAB_PLAN_EQUAL = 0, SHM_SLAB_PLAN_STEP1 = 1 };

/* Source geometry, already reduced to what the hot loops read
 * (signed_heat_grid_solver.cpp:53-57 mesh / :162-166 points). */
typedef struct {
    int64_t S;             /* number of sources: faces (mesh) or points (cloud) */
    const double* pos;     /* [3S] face barycenters (:498-503) or point positions */
    const double* wnormal; /* [3S] N_f*A_f (:57) or n_p*A_p (:166) */
    const double* area;    /* [S]  A_f (shoelace, signed_heat_3d.cpp:74-88) or tufted dual area */
    double lambda;         /* 1/sqrt(tCoef*h^2) (:42-44 / :151-153) */
} shm_sources;

/* Grid block of signed_heat_grid_solver.cpp:13-26. */
typedef struct {
    int32_t n;          /* nodes per side: nx=ny=nz = (size_t)(2*2^(hCoef+3)) */
    double bbox_min[3]; /* centroid - radius*scale */
    double cell;        /* 2*s/(n-1) */
} shm_grid;

typedef struct {
    int32_t fast_integration; /* SignedHeat3DOptions::fastIntegration (signed_heat_3d.h:27) */
    int32_t scrub_nonfinite;  /* 1 = mesh overload's divYt scrub (:72-74); 0 = point overload (:180) */
    double tol;               /* relative residual at which the iteration stops; <=0 -> default 1e-8 (fp64) / 1e-5 (fp32).
                               * PRIMAL: ||P r|| <= tol * ||P b||, r the grid residual projected on null(A).
                               * DUAL / DUAL_SLABS: ||r_mu|| <= tol * ||r_mu,0||, r_mu = Pm(A K^+ b - S mu) the residual of the
                               * m-dimensional multiplier system (what that solver iterates on).  Both bound the error of phi; they
                               * are not the same number (measured at 256^3: tol 1e-8 -> L_inf(phi) 4e-10..3e-9 for DUAL, 4e-8 PRIMAL).
                               * DUAL with the explicit S^-1 (stats.cg_form == 2) is a direct solve plus iterative refinement: at most min(max_iters, 6)
                               * passes; it lands at ~eps * cond(S) (1e-14 ... 1e-11) in the first.  A tol below that floor does not end in
                               * SHM_ERR_NOCONV: once a pass stops reducing the residual and rel_residual < 1e-9 the solve is accepted as converged and
                               * stats.rel_residual reports what was reached. */
    int32_t max_iters;        /* <=0 -> default 20*n */
    int32_t check_every;      /* residual is inspected on the host every this many iterations; <=0 -> 32 (4 with the preconditioner) */
    int32_t preconditioner;   /* SHM_PRECOND_AUTO | _NONE | _DCT  (primal solver only) */
    int32_t solver;           /* SHM_SOLVER_AUTO | _PRIMAL | _DUAL | _DUAL_SLABS */
    int32_t step1_arith;      /* SHM_STEP1_AUTO | SHM_STEP1_EXACT_F64: arithmetic of the Step-1 summation in an SHM_F64 handle (ignored by SHM_F32 handles) */
    int32_t dual_form;        /* SHM_DUAL_AUTO | _DIRECT | _EXPLICIT_S_CG | _THROUGH_GRID: which form of the dual solver runs (ABI 5; see below) */
    double step1_budget;      /* error budget of STEP1_AUTO's precision tiers on the normalised field Y; 0 -> 1e-8; accepted range [1e-12, 1e-3] (ABI 5) */
} shm_opts;

/* Form of the dual solver (SHM_SOLVER_DUAL / AUTO on one process; the same operator in every form, so the same phi up to the tolerance).
 * AUTO:          chosen per problem: the direct solve where the inverse of S hides behind Step 1, else CG on the explicit S or through the grid.
 * DIRECT:        S = A K^+ A^T assembled from the image-sum Green's table and INVERTED beside Step 1; the solve is two dense mat-vecs + refinement passes.
 *                Applies for m <= 16384 rows and n <= 512, on one z-slab or -- round 6 -- on a power-of-two number of equal z-slabs with n a power of two (S and S^-1 are
 *                then replicated on every rank and K^+ runs on the slabs: SHM_SOLVER_DUAL_SLABS, and what AUTO picks on four or more ranks for 256 <= n <= 512 with S <= 16384 sources);
 *                elsewhere the request falls back to AUTO's choice.
 * EXPLICIT_S_CG: S assembled, CG on it (one dense mat-vec per iteration), preconditioned by (A A^T)^-1 (A K A^T) (A A^T)^-1.
 * THROUGH_GRID:  CG with S applied through the grid (scatter, five transform sweeps, gather per iteration); no m x m matrix is formed.
 * stats.cg_form reports what ran (2 / 3 / 0).
 * shm_opts is 48 bytes in ABI 5 (40 in ABI 4): shm_grid_solve reads sizeof(shm_opts) bytes from the caller, so a caller MUST check shm_grid_abi_version() == 5 before
 * passing one -- a struct of an older ABI would be over-read and its trailing fields (dual_form, step1_budget) taken from whatever follows it. */
enum { SHM_DUAL_AUTO = 0, SHM_DUAL_DIRECT = 1, SHM_DUAL_EXPLICIT_S_CG = 2, SHM_DUAL_THROUGH_GRID = 3 };

/* Arithmetic of Step 1 (the N*S direct summation, signed_heat_grid_solver.cpp:48-65 / :157-174; yukawaPotential, signed_heat_3d.cpp:45-49) in an SHM_F64 handle.
 * AUTO:      error-budgeted precision tiers (csrc/shm_conv_tiered.hip.h): per block of 8 x 8 x 4 nodes, sources whose terms are below e^-8 of the block's
 *            dominant terms are summed in packed fp32, sources whose terms all together stay below 2e-9 of it are dropped (round 6: by the accumulated sum of their
 *            bounds, not S times the worst case); everything else in fp64.  The kernel checks the packed-fp32 sums AND the dropped sources' bound against |X| at
 *            every node and re-evaluates a block's far and dropped sources in fp64 where they could move Y by more than the budget (cancellation regions;
 *            shm_stats.pairs_redone).  max|Y - Y_exact| < 1e-8 (asserted against the C oracle at the full sizes of
 *            BASELINE.json), phi inherits < 1e-9.
 * EXACT_F64: every (node, source) pair in fp64 like the reference (~1.4x the Step-1 time); Y agrees with the serial loops to 1e-11.  (Round 5: the tiered kernel with nothing far
 * and nothing dropped where every pair of the grid stays inside a block's exponent span -- lambda * grid diagonal below ~600 --, the all-fp64 kernel of rounds 1-4 otherwise.)
 * step1_budget (ABI 5) moves AUTO's three thresholds together: a budget b puts the far threshold at e^-(8 - ln(b / 1e-8)), the a-posteriori test at b / eps_far (1e-6, growing with the far terms' exponent beyond 24) and the
 * drop threshold at b / 5.  (With SHM_DEBUG_KNOBS=1 in the environment, SHM_CONV_EXACT=1 forces EXACT_F64 whatever the caller asks: A/B runs and tests.) */
enum { SHM_STEP1_AUTO = 0, SHM_STEP1_EXACT_F64 = 1 };

/* How the KKT system of signed_heat_grid_solver.cpp:101-107 is solved.
 * PRIMAL: projected (optionally DCT-preconditioned) CG on the N grid unknowns -- the matrix-free 7-point-stencil PCG.
 * DUAL:   CG on the m x m Schur complement S = A K^+ A^T (K^+ = DCT fast Poisson solve) for the multipliers, preconditioned by
 *         (A A^T)^-1 (A K A^T) (A A^T)^-1; needs the DCT (see SHM_PRECOND_DCT).  Same solution, ~2x fewer and ~2x cheaper iterations.
 * AUTO:   DUAL when the DCT is available, else PRIMAL.
 * With several processes (world > 1) Steps 1-2 and the divergence always run on the rank's z-slab.  DUAL / AUTO then gather the
 * right-hand side D^T Y (one N-vector, grouped ncclSend/ncclRecv) and every rank runs the single-GPU dual solve on the whole grid
 * (its iteration is m-dimensional and latency-bound: slicing it buys nothing, exchanging its transposes costs more than it saves);
 * DUAL_SLABS keeps the solve distributed as well (z-slab DCT with two all-to-alls per application), PRIMAL is the z-slab stencil
 * CG with halo exchange.  phi comes back per rank for its own planes in every case. */
enum { SHM_SOLVER_AUTO = 0, SHM_SOLVER_PRIMAL = 1, SHM_SOLVER_DUAL = 2, SHM_SOLVER_DUAL_SLABS = 3 };

/* Preconditioner of the projected CG.  DCT = exact fast Poisson solve (3-D DCT-II diagonalises the reference's
 * Neumann Laplacian, signed_heat_grid_solver.cpp:278-334) sandwiched between constraint projections.  n = 2^k in [16,1024]:
 * O(n log n) line transforms, also on a power-of-two number of equal z-slabs dividing n.  Any other n in [4,1024] (the reference's
 * nx = (size_t)(2 * 2^(hCoef+3)) with a fractional hCoef, :24): dense products with the DCT matrix on the fp64 matrix cores, single z-slab.
 * AUTO picks DCT when available, otherwise NONE (plain projected CG). */
enum { SHM_PRECOND_AUTO = 0, SHM_PRECOND_NONE = 1, SHM_PRECOND_DCT = 2 };

typedef struct {
    int32_t n, m;             /* grid side; constraint rows (distinct source cells) */
    int64_t S;
    int32_t iters;            /* projected-CG iterations executed */
    double rel_residual;      /* the quantity `tol` bounds, at exit (PRIMAL: ||P r|| / ||P b||; DUAL: ||r_mu|| / ||r_mu,0||) */
    double shift;             /* area-weighted mean of phi over the sources that was subtracted */
    /* device-side timings (hipEvent, ms) */
    double ms_conv;           /* Steps 1+2 */
    double ms_div;            /* D^T Y (+ scrub) */
    double ms_setup;          /* constraint rows, A A^T and its inverse (dense blocked Gauss-Jordan, or two-level for large m): host wall time from the
                               * start of the set-up until the inverse is ready; runs on a 2nd stream beside ms_conv */
    double ms_wait_setup;     /* time the main stream actually waited for the set-up after the divergence */
    double ms_pcg;            /* projected CG loop */
    double ms_shift;          /* shift + phi write-out */
    double ms_total;          /* whole shm_grid_solve */
    /* per-kernel averages inside the CG loop, hipEvents on the solver's stream around sampled launches */
    double ms_stencil_avg;    /* q = K p + partial p.q                       2NT algorithmic bytes */
    double ms_update_xr_avg;  /* x += a p, r += a q + partial ||r||^2        6NT */
    double ms_project_avg;    /* gather A r, (A A^T)^-1 matvec, scatter A^T u (m-sized, not N-sized) */
    double ms_update_p_avg;   /* p = -z + b p (z = r without preconditioner)  3NT */
    double ms_precond_avg;    /* z = M^-1 r: five DCT sweeps (0 without preconditioner)   10NT(+1NT for r.z) */
    int32_t kernel_samples;   /* how many iterations were sampled for the averages above */
    int32_t preconditioner;   /* SHM_PRECOND_NONE or SHM_PRECOND_DCT: what actually ran */
    int32_t solver;           /* SHM_SOLVER_PRIMAL, SHM_SOLVER_DUAL or SHM_SOLVER_DUAL_SLABS: what actually ran */
    double bytes_per_iter;    /* algorithmic HBM bytes per CG iteration of the decomposition launched */
    int32_t cg_form;          /* DUAL: 0: S = A K^+ A^T applied through the grid (five sweeps per iteration); 3: CG on the explicit S (one dense mat-vec,
                               *    timed in ms_precond_avg); 2: direct solve with the explicit S^-1 (iters = passes, ms_precond_avg = S^-1 mat-vec).
                               * PRIMAL: 0: four N-sized kernels per iteration (11NT; the per-kernel fields above mean what they say).
                               * 1: fused sweeps (8NT): ms_stencil_avg = DIR sweep (p' = -z + beta p, partial p'.Kp'; 3NT),
                               *    ms_update_xr_avg = RES sweep (r += alpha K p', partial ||r||^2; 3NT),
                               *    ms_update_p_avg = x += a0 p0 + a1 p1 (4NT per launch, launched every other iteration)
                               * 4: as 1, but x is updated on half the grid in every iteration (2NT per launch), beside the projection (one GPU, no preconditioner; round 5) */
    double pairs_fp64;        /* (node, source) pairs Step 1 actually evaluated on this rank in the last solve, in fp64 arithmetic ... */
    double pairs_fp32;        /* ... and in (packed) fp32: the tiers of shm_conv_tiered.hip.h; culled / dropped pairs are in neither.
                               * Nominal work is N*S; the Step-1 roofline fraction is computed from these, not from N*S. */
    int32_t conv_launches;    /* kernel launches Step 1 took on this rank in the last solve (its duration ms_conv spans all of them) */
    double pairs_redone;      /* tiered Step 1: pairs first summed in packed fp32 and then evaluated AGAIN in fp64 because the a-posteriori test of their node
                               * block failed (the packed-fp32 sums exceeded 3.3e-3 of |X| at a node: cancellation regions); counted in pairs_fp64 and pairs_fp32 too */
} shm_stats;

/* --- life cycle -------------------------------------------------------------------------------- */
shm_status shm_grid_create(const shm_config* cfg, shm_solver** out);
void shm_grid_destroy(shm_solver* s);
const char* shm_grid_last_error(const shm_solver* s); /* s may be NULL: error of the failed create */
int32_t shm_grid_abi_version(void);

/* --- the hot path ------------------------------------------------------------------------------ */
/* Upload sources + (re)build the grid state.  Equivalent of the `options.rebuild` block
 * (signed_heat_grid_solver.cpp:8-36) plus making the inputs resident in HBM. */
shm_status shm_grid_set_problem(shm_solver* s, const shm_sources* src, const shm_grid* grid);
/* Steps 1-3 + shift on the device; phi stays resident.  (signed_heat_grid_solver.cpp:38-113) */
shm_status shm_grid_solve(shm_solver* s, const shm_opts* opts, shm_stats* stats);
/* Copy phi of the z-planes owned by this process to the host: k in [*k_begin,*k_end), x fastest,
 * (k_end-k_begin)*n*n doubles.  With world==1 that is the whole grid (N = n^3 values). */
shm_status shm_grid_get_phi(shm_solver* s, double* phi_out, int32_t* k_begin, int32_t* k_end);
/* The z-planes [*k_begin, *k_end) this process owns under the slab plan in force (after shm_grid_set_problem): what sizes the buffers of
 * shm_grid_get_phi / shm_grid_get_field.  With SHM_SLAB_PLAN_EQUAL it equals shm_plan_slab(n, world * local_slabs, ...); with the weighted plan ask here. */
shm_status shm_grid_owned_planes(shm_solver* s, int32_t* k_begin, int32_t* k_end);
/* One-shot convenience = set_problem + solve + get_phi (world==1 only): the call a C++ adapter's
 * computeDistance() makes. */
shm_status shm_grid_compute_distance(shm_solver* s, const shm_sources* src, const shm_grid* grid,
                                     const shm_opts* opts, double* phi_out, shm_stats* stats);

/* --- stage access (parity tests call these; same kernels as shm_grid_solve) --------------------- */
typedef enum {
    SHM_FIELD_Y0 = 0, /* normalised X, x component  (Y[3*idx+0], :60-62) */
    SHM_FIELD_Y1 = 1,
    SHM_FIELD_Y2 = 2,
    SHM_FIELD_DIV = 3, /* divYt (:71-74) */
    SHM_FIELD_PHI = 4
} shm_field;
shm_status shm_grid_run_conv(shm_solver* s);                       /* Steps 1+2 only (step1_arith = SHM_STEP1_AUTO) */
shm_status shm_grid_run_conv_arith(shm_solver* s, int32_t step1_arith); /* the same with the arithmetic of shm_opts.step1_arith */
shm_status shm_grid_run_divergence(shm_solver* s, int32_t scrub);  /* needs run_conv */
shm_status shm_grid_get_field(shm_solver* s, shm_field f, double* out /* owned planes */);
/* The same for the z-planes [k_begin, k_end) only (a sub-range of the owned planes): (k_end - k_begin) * n * n doubles.  What the full-size parity tests
 * read: a few planes of Y at 512^3 / 1024^3 instead of three N-vectors. */
shm_status shm_grid_get_field_planes(shm_solver* s, shm_field f, int32_t k_begin, int32_t k_end, double* out);
/* out = L*u with the reference's Laplacian (signed_heat_grid_solver.cpp:278-334); u,out: n^3 doubles
 * on the host (world==1). */
shm_status shm_grid_apply_laplacian(shm_solver* s, const double* u, double* out);
/* Constraint rows (signed_heat_grid_solver.cpp:80-98): nodes/coeffs hold 8*S entries; *m rows written. */
shm_status shm_grid_get_constraints(shm_solver* s, int64_t* nodes, double* coeffs, int32_t* m);
/* The explicit Schur complement S = A K^+ A^T the dual solver iterates on when the problem qualifies (one slab; n = 2^k <= 512 and m <= 4096 (direct solve), or m <= 16384 in the fp64 solve with shm_opts.step1_arith = SHM_STEP1_AUTO where the library estimates the dense mat-vec cheaper than its sparse sweeps through the grid; any other n: m <= 16384): m x m
 * doubles, row-major; *m rows.  SHM_ERR_STATE when the solver applies S through the grid instead.  Test entry point (world==1). */
shm_status shm_grid_get_schur(shm_solver* s, double* out, int32_t* m);
/* v <- v - A^T (A A^T)^-1 A v on the device (the projector inside the CG); v: n^3 doubles, world==1. */
shm_status shm_grid_apply_projector(shm_solver* s, double* v);

/* out = M^-1 v with the DCT preconditioner alone (no projection); v,out: n^3 doubles on the host (world==1). */
shm_status shm_grid_apply_preconditioner(shm_solver* s, const double* v, double* out);

/* --- isosurface of the resident phi (headless stand-in for the demo's contour, src/main.cpp:116-128,167-191: Polyscope's marching cubes on the node
 * scalar quantity, setIsosurfaceLevel + registerIsosurfaceAsMesh).  inside = phi < isovalue, normals towards increasing phi, one vertex per cut grid edge
 * (linear interpolation), welded across cells.  Covers the cells whose lower z-plane this process owns.  Call shm_grid_isosurface, size the buffers, then _get_.
 *   SHM_ISO_MARCHING_CUBES (what shm_grid_isosurface runs, round 5): 256-case table, ambiguous faces resolved per face so that neighbouring cells agree
 *                          (watertight; csrc/shm_mc_table.h, generated by tools/gen_mc_table.py)
 *   SHM_ISO_MARCHING_TETS  (rounds 1-4): Kuhn split of every cell into six tetrahedra; also watertight, about 2.3x the triangles, extra vertices on the
 *                          face and body diagonals */
enum { SHM_ISO_MARCHING_CUBES = 0, SHM_ISO_MARCHING_TETS = 1 };
shm_status shm_grid_isosurface(shm_solver* s, double isovalue, int64_t* n_vertices, int64_t* n_triangles);
shm_status shm_grid_isosurface_ex(shm_solver* s, double isovalue, int32_t method, int64_t* n_vertices, int64_t* n_triangles);
shm_status shm_grid_get_isosurface(shm_solver* s, double* vertices /* [3*nv] */, int64_t* triangles /* [3*nt] */);

/* --- multi-GPU bootstrap ------------------------------------------------------------------------ */
/* Fill 128 bytes with a fresh ncclUniqueId (rank 0 calls this, the launcher broadcasts the bytes). */
shm_status shm_comm_unique_id(void* out128);
/* z-plane range [k0,k1) owned by slab `slab` of `nslabs` for a grid of n planes (pure host logic). */
void shm_plan_slab(int32_t n, int32_t nslabs, int32_t slab, int32_t* k0, int32_t* k1);
/* Relative Step-1 cost of every z-plane (weights[n], pure host logic): the kernels' culling / tier rules evaluated per source on a 12 x 12 sample of
 * node blocks per block layer.  fp64: near pairs 1, packed-fp32 pairs 0.43, dropped 0; fp32: kept pairs 1, skipped 0. */
shm_status shm_step1_plane_weights(const shm_sources* src, const shm_grid* grid, int32_t precision, double* weights);
/* The source-aware variant of shm_plan_slab: contiguous ranges of about equal weight, boundaries at multiples of `granule` planes (4 for the fp64
 * Step 1, 8 for fp32: the kernels' node blocks), at least one granule per slab. */
void shm_plan_slab_weighted(int32_t n, int32_t nslabs, int32_t slab, const double* weights, int32_t granule, int32_t* k0, int32_t* k1);

#ifdef __cplusplus
}
#endif
#endif /* SHM_GRID_H */
