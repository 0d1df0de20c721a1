// libshm_grid.so -- C ABI (include/shm_grid.h) + host orchestration of the gfx950 kernels.
//
// Replaces, for the regular-grid solver of nzfeng/signed-heat-3d:
//   * the serial source-integration loops        signed_heat_grid_solver.cpp:48-65 / :157-174
//   * Eigen assembly of D and the product D^T Y   :70-74 / :179-180
//   * Eigen assembly of L, A, the KKT matrix and its sparse LU (solveSquare)  :80-108 / :186-214
//     -> matrix-free projected CG on null(A) with a dense (A A^T)^-1
//   * the shift                                   :110-111 / :216-217
// One process drives one GPU; the grid is cut into z-slabs (rank-major).  A process may own several
// slabs (loop-back transport, used to exercise the slab logic on one GPU); across processes the halo
// planes and the reduction vectors travel over RCCL (xGMI) on the solver's own stream.
#pragma once
#include "shm_host.hip.h"
#include "shm_kernels.hip.h"
#include "shm_conv_tiered.hip.h"
#include "shm_cg_fused.hip.h"
#include "shm_twolevel.hip.h"
#include "shm_schur.hip.h"
#include "shm_dct.hip.h"
#include "shm_dct_gemm.hip.h"
#include "shm_green_fft.hip.h"

namespace shm {

template <typename T> constexpr int vec_width() { return sizeof(T) == 8 ? 2 : 4; }

template <typename T>
struct Slab {
    int k0 = 0, k1 = 0, nzl = 0;  // owned global planes [k0,k1)
    size_t plane = 0, nown = 0, ntot = 0;
    DevArray<T> Y0, Y1, Y2, r /* also divYt */, x, p, q /* also phi */, z /* preconditioned residual (DCT path only) */;
    DevArray<double> partials, pq_partials /* block partials of p'.Kp' when the RES sweep sums them itself */, red /* [1+m] */, pq /* [1] */, u /* [m] */, sc;
    DevArray<unsigned> proj_ticket;   // arrival counter of the projection's u.w reduction (scatter_nodes_kernel)
    // constraint pieces restricted to owned nodes
    DevArray<int> row_ptr, ent_row, node_ptr;
    DevArray<uint32_t> ent_node, node_id;
    DevArray<double> ent_coef, nent_coef;
    DevArray<ShiftItem> shift_items;
    DevArray<double> dv;  // dual solver m-vectors: mu, r, p, z, t1, t2, g (7 x mp)
    DevArray<int> layer_order;   // tiered Step 1: z-layers of blocks in queue order (layer_order_for)
    DevArray<int> sample_order;               // ... when the far rule is decided from a sample (far_rule_plan)
    DevArray<unsigned char> sample_rule;      // rule per layer of that order: 1 sample, 0 box, 2 decided
    DevArray<unsigned long long> sample_ctr;  // the sample's counters (see ConvParams)
    long long sample_key = -1;
    int sample_blocks = 0;
    long long layer_order_key = -1;
    DevArray<double> div_partials;   // per-workgroup sums of b written by the divergence kernel (div_sum_blocks of them; 0: not available, sum b with sum_kernel)
    int div_sum_blocks = 0;
    DevArray<T> touched_save;   // the touched nodes' entries of b while b - A^T mu stands in their place (final stage of the dual solve)
    DevArray<T> W1, W2;  // DCT work arrays (precision TP == T); W2 only with several slabs (packed transposes)
    DevArray<T> S1, S2, S4;  // sparse-sweep buffers of the dual solver's per-iteration solve (single slab)
    DevArray<int> act_x, act_y;  // active tiles of the x sweeps / y sweeps
    DevArray<unsigned> act_z;    // bit k: z-plane k holds touched nodes
    DevArray<int> act_planes;    // the same as an ascending list (zsolve_sparse_kernel)
    int n_act_planes = 0;
    int n_act_x = 0, n_act_y = 0;
    int n_touched = 0, n_shift = 0;
    GridParams gp{};
};

template <typename T>
struct Solver final : SolverBase {
    shm_config cfg;
    hipStream_t stream = nullptr;   // conv, divergence, CG
    hipStream_t stream2 = nullptr;  // constraint set-up ((A A^T)^-1), overlapped with the Step-1 kernel
    hipStream_t stream3 = nullptr;  // explicit Schur complement of the dual solver: beside the inversion of G (stream2) and Step 1 (created on first use)
    std::unique_ptr<Event> e_sch_in, e_sch_done, e_gs_done;
    bool gs_early = false;   // this solve's Green's table was queued before the host built the rows (enqueue_green_table)
    hipStream_t stream_h = nullptr;  // halo exchange of the fused primal CG, overlapped with the interior z chunks of its DIR sweep (created on first use)
    int n = 0, alloc_n = -1;
    size_t N = 0;
    double cell = 0., lambda = 0.;
    double bbox_min[3] = {0, 0, 0};
    double conv_ctr[3] = {0, 0, 0};   // the grid's centre: origin of the coordinates Step 1 computes in
    double conv_wscale = 1.;          // power of two that brings the largest source weight into (0.5, 1]
    int64_t S = 0;
    std::vector<double> h_pos, h_wn, h_area;
    double area_sum = 0., conv_far_gap = 0., conv_skip_base = 3.0e38, last_host_setup_ms = 0., last_setup_wall_ms = 0.;
    double conv_tier_log = 0., conv_tier_skip_base = 3.0e38;   // tiered fp64 Step 1: far threshold G (nats) and the drop threshold of rounds 3-5, ln(S / eps)
    double conv_drop_K = 64., conv_drop_eps64 = 2e-9, conv_drop_eps32 = 6.0e-8;   // tiered kernels, round 6: the accumulated drop rule (drop_rule_plan)
    double conv_w_span = 0.;                                   // ln(largest / smallest non-zero source weight)
    // fp32 handles: the tiered kernel's view of the sources (fp64 records in clusters of 64, the reference's own weights) beside the fp32 kernel's
    DevArray<double> d_src_t;
    DevArray<float> d_clusters_t;
    int n_clusters_t = 0;
    bool conv_tiered = false;                                  // fp64 only; SHM_CONV_EXACT=1 selects the all-fp64 kernel
    bool conv_tier_exact = false;                              // ... with every pair in its fp64 body (SHM_STEP1_EXACT_F64 where the exponent span allows)
    bool conv_tiered32 = false;                                // fp32 handles: Step 1 through the tiered kernel's packed-fp32 body
    bool fold_pq = false;                                      // fused stencil CG on one GPU: the RES sweep sums the DIR sweep's partials of p'.Kp' itself
    int fold_pq_np = 0;
    int dual_form_req = SHM_DUAL_AUTO;                         // shm_opts.dual_form of the solve in progress
    double step1_budget = 0.;                                  // shm_opts.step1_budget of the solve in progress (<= 0: kTierBudget)
    DevArray<unsigned long long> d_pair_counters;              // [0] fp64 pairs, [1] fp32 pairs evaluated by the last Step 1, [2] pairs evaluated again in fp64 (a-posteriori test)
    DevArray<unsigned> d_unit_counters;                        // tiered Step 1: eight work-queue heads (one per XCD) per launch (zeroed at the start of every Step 1)
    static constexpr int kMaxConvLaunches = 256;
    int conv_launch_index = 0;
    int conv_launches_last = 0;   // launches of the last Step 1 (all slabs of this rank)
    int n_clusters = 0;
    int conv_grid_cap = 1 << 30;
    bool o_fast_hint = false;  // the running solve is a fast-integration one: no constraint set-up beside Step 1
    int num_cus = 256, dct_grid_x16 = 16;
    DevArray<T> d_src;          // [Spad][6] Morton-sorted, padded to whole clusters
    DevArray<float> d_src32;    // same in fp32 (far clusters of the fp64 path)
    DevArray<float> d_clusters; // [n_clusters][4] bounding spheres
    DevArray<double> d_exptab;  // 2^(j/2048), j < 2048 (Step 1 fp64 exponential)
    std::vector<Slab<T>> slabs;
    int total_slabs = 1, first_slab = 0;
    std::vector<int32_t> slab_bounds;   // [total_slabs + 1] plane boundaries of all slabs (every rank holds the whole plan)
    bool slabs_equal = true;            // the plan is the equal-plane one (what the slab-distributed transforms need)
    void slab_range(int slab, int32_t* k0, int32_t* k1) const {
        *k0 = slab_bounds[(size_t)slab];
        *k1 = slab_bounds[(size_t)slab + 1];
    }
    bool have_problem = false, have_conv = false, have_div = false, have_phi = false, have_constraints = false;
    // constraints (replicated)
    std::vector<Row> rows;
    std::vector<std::vector<ShiftItem>> shift_host;
    int m = 0, mp = 0;
    DevArray<double> Ginv, gjP, gjR, gjC;
    DevArray<float> Ginv32;  // single-precision copy for the dual solver's preconditioner
    // two-level inverse of G for large m (shm_twolevel.hip.h); Ginv / Ginv32 then hold the inverse of the |Sigma| x |Sigma| Schur complement
    struct TwoLevel {
        bool on = false;
        int box = 16, P = 0, nI = 0, nS = 0, nSp = 0, ysz = 0;
        DevArray<int> ptrI, ptrS, rowsI, colsS, sepRow, adj_ptr, adj_idx, colour_list, rowBox, chunkBox, chunkCol, tBox, tRow, sBox, sRow;
        DevArray<size_t> offW;
        DevArray<double> gjP, gjR, gjC;   // pivot blocks and panels of the batched Gauss-Jordan over the boxes
        int nChunks = 0, nTChunks = 0, nbMax = 0;
        int schur_ptr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        DevArray<size_t> offD, offE;
        DevArray<double> D, E, Tm, tbuf, ybuf, vS, uS;
        DevArray<float> D32, E32, T32;
        int colour_ptr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        size_t szD = 0, szE = 0;
        TlBoxes view() const { return TlBoxes{ptrI.p, ptrS.p, offD.p, offE.p, rowsI.p, colsS.p}; }
    } tl;
    int ginv_rows = 0, ginv_ld = 0;  // logical size / leading dimension of Ginv (m, mp for the dense inverse; nS, nSp for the two-level one)
    DevArray<int> gjFlag;
    DevArray<double*> d_redptrs;
    // dual solver: B = A K A^T (CSR, replicated) and the m-vectors of its CG (per slab, replicated values)
    DevArray<int> Bptr, Bcol;
    DevArray<double> Bval;
    bool have_B = false;
    // DCT preconditioner (single slab, n = 2^k)
    using TP = T;  // precision of the preconditioner sweeps
    DevArray<Cplx<TP>> d_tw, d_om;
    DevArray<TP> d_lam;
    DevArray<double> d_lam64;  // the same eigenvalues in double (zsolve_sparse_kernel computes in double whatever T)
    DevArray<double> gd_Cm, gd_Ct, gd_W1, gd_W2;   // n not a power of two (shm_dct_gemm.hip.h): the orthonormal DCT-II matrix, its transpose, two n^3 work arrays
    // explicit Schur complement S = A K^+ A^T of the dual solver (shm_schur.hip.h): image-sum Green's table T and its work arrays, the per-row cells /
    // trilinear parameters, S itself (mp x mp, zero-padded)
    DevArray<double> gs_lam, gs_ctab, gs_Cm, gs_Ct, gs_W0, gs_W1, gs_T, Sdense, d_rowT;
    DevArray<int> d_rowX;
    std::vector<double> h_gs_lam, h_gs_ctab, h_rowT;   // host staging outlives the asynchronous uploads
    std::vector<Cplx<double>> h_gs_tw;                 // twiddles of the Green's table's FFT passes (shm_green_fft.hip.h)
    DevArray<Cplx<double>> gs_tw;
    std::vector<int> h_rowX;
    bool have_S = false;
    bool dual_direct_requested = false, dual_direct = false;   // direct dual solve: S^-1 (Sinv) instead of G^-1; requested by solve(), decided in build_constraints()
    DevArray<double> Sinv, Sinv_ones /* S^-1 1: the border of the direct solve, formed once per set-up behind the inversion */;
    // s_setprio(3) in the set-up kernels that share the SIMDs with the tiered Step 1: they are short and on the critical path when Step 1 is (64^3 ... 256^3, thin
    // slabs of a multi-GPU run); where Step 1 outlasts the set-up several times over the raised priority only costs Step 1 issue slots at the wrong moments
    // (512^3: 198.8 -> 196-197 ms, bunny.pc 512^3: 106.0 -> 104-105 ms without it; 128^3: 6.2 -> 7.7 ms, hence the switch).  Decided per solve in build_constraints().
    int setup_prio = 1;
    double conv_est_total_ms = 1e30;   // estimate of this rank's last Step-1 launch (1e30: none was launched -- stand-alone set-up, test entry points)
    int gs_n = 0;   // grid the Green's table in gs_T was built for (0: none)
    double gs_cell = 0.;
    int log2n = 0;
    bool precond_ready = false;
    double* h_pinned = nullptr;
    Rccl::comm_t comm = nullptr;
    int vec = 1;  // vector width usable for this n
    // world > 1: the whole-grid solver every rank runs after the right-hand side has been gathered (see solve_gathered)
    std::unique_ptr<Solver<T>> full;
    bool solve_only = false;  // this instance is such a whole-grid solver: it never runs Steps 1-2, so it owns no Y arrays
    bool full_problem_set = false;   // `full` holds the current problem (ensure_full)
    bool pool_held = false;   // this solver holds a reference on its device's pinned staging chunks (PinnedPool)

    explicit Solver(const shm_config& c, bool solve_only_ = false) : cfg(c), solve_only(solve_only_) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            throw Error(SHM_ERR_HIP, fmt("no HIP device available (%s); this library has no CPU fallback",
                                          e == hipSuccess ? "device count 0" : hipGetErrorString(e)));
        if (cfg.device < 0 || cfg.device >= ndev) throw Error(SHM_ERR_INVALID, fmt("device %d out of range [0,%d)", cfg.device, ndev));
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, cfg.device));
            const char* e = knob("SHM_CONV_SLOTS_PER_CU_X16");  // tuning knob: conv workgroups per CU, in 1/16ths (default 64 = 4, the LDS-limited residency: one persistent wave of workgroups)
            const int x16 = e ? atoi(e) : 64;
            conv_grid_cap = std::max(1, prop.multiProcessorCount * x16 / 16);
            num_cus = prop.multiProcessorCount;
            const char* d = knob("SHM_DCT_GRID_X16");  // tuning knob: DCT workgroups per resident slot, in 1/16ths (16 = one persistent wave of workgroups)
            dct_grid_x16 = d ? std::max(1, atoi(d)) : (1 << 20);
        }
        {   // the set-up stream outranks the main stream so that its short kernels are not starved by the Step-1 kernel
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, greatest));
        }
        HIPCHK(hipHostMalloc((void**)&h_pinned, 64 * sizeof(double)));
        if (cfg.world > 1 && !cfg.rccl_unique_id) throw Error(SHM_ERR_INVALID, "world>1 needs rccl_unique_id");
        if (cfg.rccl_unique_id) {  // also with world == 1: a one-rank communicator exercises the RCCL path
            Rccl& R = Rccl::get();
            Rccl::unique_id id;
            memcpy(&id, cfg.rccl_unique_id, sizeof id);
            R.chk(R.CommInitRank(&comm, cfg.world, id, cfg.rank), "ncclCommInitRank");
        }
        PinnedPool::get().acquire(cfg.device);   // last: a constructor that throws runs no destructor
        pool_held = true;
    }
    ~Solver() override {
        (void)hipSetDevice(cfg.device);
        if (comm) (void)Rccl::get().CommDestroy(comm);
        if (h_pinned) (void)hipHostFree(h_pinned);
        slabs.clear();
        if (stream) (void)hipStreamDestroy(stream);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream3) (void)hipStreamDestroy(stream3);
        if (stream_h) (void)hipStreamDestroy(stream_h);
        if (pool_held) PinnedPool::get().release(cfg.device);
    }

    void log(const char* f, ...) {
        if (!cfg.verbose) return;
        va_list ap;
        va_start(ap, f);
        vfprintf(stderr, f, ap);
        va_end(ap);
        fputc('\n', stderr);
    }

    static int grid_for(size_t work_items, int cap = 2048) {
        size_t b = (work_items + kBlock - 1) / kBlock;
        if (b < 1) b = 1;
        if (b > (size_t)cap) b = cap;
        return (int)b;
    }

    // ------------------------------------------------------------------------------------------
    void set_problem(const shm_sources& src, const shm_grid& g) override {
        HIPCHK(hipSetDevice(cfg.device));
        if (g.n < 2) throw Error(SHM_ERR_INVALID, "grid.n must be >= 2");
        if (!(g.cell > 0.) || !std::isfinite(g.cell)) throw Error(SHM_ERR_INVALID, "grid.cell must be positive and finite");
        if (src.S <= 0 || !src.pos || !src.wnormal || !src.area) throw Error(SHM_ERR_INVALID, "sources: S>0 and non-null arrays required");
        if (!(src.lambda > 0.) || !std::isfinite(src.lambda)) throw Error(SHM_ERR_INVALID, "sources.lambda must be positive and finite");
        if (src.S > (int64_t)1 << 28) throw Error(SHM_ERR_INVALID, "too many sources");
        total_slabs = cfg.world * cfg.local_slabs;
        first_slab = cfg.rank * cfg.local_slabs;
        if (g.n < total_slabs) throw Error(SHM_ERR_INVALID, fmt("grid.n=%d smaller than the number of z-slabs %d", g.n, total_slabs));
        n = g.n;
        N = (size_t)n * n * n;
        cell = g.cell;
        lambda = src.lambda;
        for (int a = 0; a < 3; a++) bbox_min[a] = g.bbox_min[a];
        S = src.S;
        h_pos.assign(src.pos, src.pos + 3 * S);
        h_wn.assign(src.wnormal, src.wnormal + 3 * S);
        h_area.assign(src.area, src.area + S);
        // every source must lie strictly inside the grid (the reference indexes cell+1 unchecked, :443-449)
        for (int64_t s = 0; s < S; s++)
            for (int a = 0; a < 3; a++) {
                const double d = (h_pos[3 * s + a] - bbox_min[a]) / cell;
                if (!(d >= 0.) || !(std::floor(d) + 1. <= (double)(n - 1)))
                    throw Error(SHM_ERR_INVALID, fmt("source %lld lies outside the grid cells", (long long)s));
            }
        area_sum = 0.;
        for (int64_t s = 0; s < S; s++) area_sum += h_area[s];  // sequential like `normalization += A` (:477)
        // Step-1 layout: sources sorted along a Morton curve and cut into clusters of kConvCluster with bounding spheres, so
        // that a node tile can classify whole clusters as "far" (contribution < e^-25 of the tile's dominant term) and
        // evaluate them in fp32; near clusters keep the reference's fp64 arithmetic.  Sum order differs from the
        // reference's (a few ulp); constraint rows and the shift keep the original source order.
        {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int64_t s = 0; s < S; s++)
                for (int a = 0; a < 3; a++) {
                    lo[a] = std::min(lo[a], h_pos[3 * s + a]);
                    hi[a] = std::max(hi[a], h_pos[3 * s + a]);
                }
            const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
            auto spread = [](uint64_t v) {  // 21 bits -> every third bit
                v &= 0x1fffff;
                v = (v | v << 32) & 0x1f00000000ffffULL;
                v = (v | v << 16) & 0x1f0000ff0000ffULL;
                v = (v | v << 8) & 0x100f00f00f00f00fULL;
                v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
                v = (v | v << 2) & 0x1249249249249249ULL;
                return v;
            };
            std::vector<std::pair<uint64_t, int64_t>> order((size_t)S);
            for (int64_t s = 0; s < S; s++) {
                uint64_t code = 0;
                for (int a = 0; a < 3; a++) {
                    const uint64_t q = (uint64_t)std::min(1048575.0, std::max(0.0, (h_pos[3 * s + a] - lo[a]) / ext * 1048575.0));
                    code |= spread(q) << a;
                }
                order[(size_t)s] = {code, s};
            }
            std::sort(order.begin(), order.end());
            // Step 1 only ever needs DIFFERENCES node - source.  Its device copies of the sources and its node coordinates are therefore taken relative to the
            // grid's centre: the fp32 arithmetic (fp32 solve; packed-fp32 tier of the fp64 solve) then rounds coordinates of magnitude <= half the grid's
            // diagonal instead of |centroid| + that -- a mesh that sits far from the origin loses nothing (tests: test_step1_is_translation_invariant).
            for (int a = 0; a < 3; a++) conv_ctr[a] = bbox_min[a] + 0.5 * (double)(n - 1) * cell;
            constexpr int kConvCluster = conv_cluster<T>();
            n_clusters = (int)((S + kConvCluster - 1) / kConvCluster);
            const int64_t Spad = (int64_t)n_clusters * kConvCluster;
            std::vector<T> packed((size_t)Spad * 6, (T)0);
            std::vector<float> packed32((size_t)Spad * 6, 0.f), cl((size_t)n_clusters * kConvClusterRec, 0.f);
            double amin = 1e300, amax = 0.;
            for (int64_t s = 0; s < S; s++) {
                const double w = std::sqrt(h_wn[3 * s] * h_wn[3 * s] + h_wn[3 * s + 1] * h_wn[3 * s + 1] + h_wn[3 * s + 2] * h_wn[3 * s + 2]);
                if (w > 0.) amin = std::min(amin, w);
                amax = std::max(amax, w);
            }
            // Y = X / |X| does not see a common factor of the weights: fp32 arithmetic gets them relative to the largest one, scaled by a POWER OF TWO (exact), so
            // that it sees weights in (0, 1] whatever the mesh's units -- what the exponent-range test of the packed-fp32 tier (shm_conv_tiered.hip.h, `in_range`)
            // and the fp32 solve's sums rely on.  The fp64 copies keep the reference's own weights (its arithmetic incl. its gradual underflow far from the
            // sources); the tiered kernel applies the factor when it converts a far source to fp32.
            conv_wscale = amax > 0. && std::isfinite(amax) ? std::ldexp(1.0, -std::ilogb(amax) - 1) : 1.0;
            conv_w_span = amax > 0. && amin < 1e300 ? std::log(amax / amin) : 0.;
            const double wscale = sizeof(T) == 4 ? conv_wscale : 1.0;
            for (int64_t t = 0; t < Spad; t++) {
                const int64_t s = order[(size_t)std::min<int64_t>(t, S - 1)].second;  // padding repeats the last source with zero weight
                for (int a = 0; a < 3; a++) {
                    packed[6 * t + a] = (T)(h_pos[3 * s + a] - conv_ctr[a]);
                    packed32[6 * t + a] = (float)(h_pos[3 * s + a] - conv_ctr[a]);
                    if (t < S) {
                        packed[6 * t + 3 + a] = (T)(h_wn[3 * s + a] * wscale);
                        packed32[6 * t + 3 + a] = (float)(h_wn[3 * s + a] * wscale);
                    }
                }
            }
            for (int c = 0; c < n_clusters; c++) {
                double cc[3] = {0, 0, 0};
                for (int e = 0; e < kConvCluster; e++)
                    for (int a = 0; a < 3; a++) cc[a] += (double)packed[6 * ((size_t)c * kConvCluster + e) + a];
                for (int a = 0; a < 3; a++) cc[a] /= kConvCluster;
                double rad = 0.;
                for (int e = 0; e < kConvCluster; e++) {
                    double d2 = 0.;
                    for (int a = 0; a < 3; a++) {
                        const double d = (double)packed[6 * ((size_t)c * kConvCluster + e) + a] - cc[a];
                        d2 += d * d;
                    }
                    rad = std::max(rad, std::sqrt(d2));
                }
                for (int a = 0; a < 3; a++) cl[kConvClusterRec * (size_t)c + a] = (float)cc[a];
                cl[kConvClusterRec * (size_t)c + 3] = (float)(rad * 1.00001 + 1e-30);
                double wmax2 = 0., wsum = 0.;  // largest source weight |A N| of the cluster and the sum of its weights (drop rule: ln of them, rounded up)
                for (int e = 0; e < kConvCluster; e++) {
                    double w2 = 0.;
                    for (int a = 0; a < 3; a++) {
                        const double w = (double)packed[6 * ((size_t)c * kConvCluster + e) + 3 + a];
                        w2 += w * w;
                    }
                    wmax2 = std::max(wmax2, w2);
                    wsum += std::sqrt(w2);
                }
                cl[kConvClusterRec * (size_t)c + 4] = wmax2 > 0. ? (float)(0.5 * std::log(wmax2) + 1e-5) : -1.0e30f;
                cl[kConvClusterRec * (size_t)c + 5] = wsum > 0. ? (float)(std::log(wsum) + 1e-5) : -1.0e30f;
            }
            // far when lambda * (d_lo - r_hi) > 25 + ln(Amax/Amin): the cluster's terms are below e^-25 ~ 1.4e-11 of the
            // tile's dominant term, so their fp32 rounding (~1e-5 incl. the exponent) stays below 2e-16 of it
            {
                const char* e = knob("SHM_CONV_FAR_LOG");  // experiment knob: -ln of the relative size below which a cluster goes to fp32
                const double far_log = e ? atof(e) : 25.0;
                conv_far_gap = (far_log + std::log(std::max(1.0, amax / std::max(amin, 1e-300)))) / lambda;
                // skipped clusters: a source further than r_hi + gap from the tile contributes less than (A_s / A_near) e^{-lambda gap} of
                // the tile's dominant term (its nearest source: weight A_near, at most r_hi away).  A cluster with largest weight A_c is
                // skipped when lambda gap > ln(S / eps) + ln(A_c / A_near): all skipped sources together (at most S) then stay below eps
                // of the dominant term (eps = 2^-24 / 2^-53: the arithmetic's own rounding unit).  Exact to rounding; bites when the kernel
                // decays over a small part of the grid (SprayBottle.pc at 1024^3: two thirds of the clusters).
                const double eps = sizeof(T) == 8 ? 1.1e-16 : 6.0e-8;
                const char* sk = knob("SHM_CONV_NO_SKIP");
                // sources whose terms, all S of them together, stay below one rounding unit of a tile's dominant term are dropped (round 3: the bound is
                // S e^-skip, loose by orders of magnitude -- rocker 512^3 fp32 reads the same L_inf against fp64, 5.6e-6, with a budget of 1e-6; until round 3
                // a further safety factor of 64 sat in it: 470 -> 440 ms on that workload).  SHM_CONV_DROP_BUDGET32: A/B knob for the fp32 solve.
                const char* db32 = knob("SHM_CONV_DROP_BUDGET32");
                conv_skip_base = sk ? 3.0e38 : std::log((double)S / (db32 && sizeof(T) == 4 ? atof(db32) : eps));
                // Tiered fp64 Step 1 (shm_conv_tiered.hip.h; default for SHM_F64): per (wave sub-tile, source), terms below e^-G of the sub-tile's dominant
                // terms go through packed fp32.  G from the error budget on Y (1e-8, the stage test's bound and a decade inside the
                // 1e-7 gate on phi): the packed-fp32 tier is measured at <= 5.6e-9 over every data file and grid size; sources whose terms all together stay
                // below 2e-9 of the dominant term are dropped (SHM_CONV_DROP_BUDGET; max|dY| does not move between 1.6e-13 and 2e-9: tools/tier_robustness.py).
                // SHM_CONV_EXACT=1: every pair in the reference's fp64 arithmetic (conv_normalize_kernel<double>; Y to 1e-11 of the C oracle).
                const char* tl = knob("SHM_CONV_TIER_LOG");
                conv_tier_log = tl ? atof(tl) : 8.0;
                const char* db = knob("SHM_CONV_DROP_BUDGET");
                conv_tier_skip_base = sk ? 3.0e38 : std::log((double)S / (db ? atof(db) : 2e-9));
                // Round 6 -- the tiered kernels drop by ACCUMULATED bound (shm_conv_tiered.hip.h): budgets eps (same values as above: 2e-9 of the dominant term in the fp64
                // solve, one fp32 rounding unit in the fp32 solve; 0: nothing is dropped), and K, the number of sources a block is expected to find near the threshold -- the
                // candidate threshold is eps_soft / K, the running sum keeps the rule sound whatever K is.  Sources near the threshold lie in a ring of radius D ~ 25 / lambda
                // around the block on a surface with one source per mean source area a: ~ 2 pi D / (a lambda) of them within one e-fold, i.e. 157 / (a lambda^2); three times
                // that leaves the sum room (tools/r06_drop_sim.py: rocker 512^3 keeps 0.52-0.53 of its pairs for K = 512 ... 1024, 0.62 for K = 256, where the sum binds).
                conv_drop_eps64 = sk ? 0. : (db ? atof(db) : 2e-9);
                conv_drop_eps32 = sk ? 0. : (db32 ? atof(db32) : 6.0e-8);
                conv_drop_K = knob("SHM_CONV_DROP_K") ? atof(knob("SHM_CONV_DROP_K")) : drop_rule_K(S, h_wn.data(), lambda);   // (the formula; refined on a sample of blocks below)
                select_step1_arith(SHM_STEP1_AUTO);
            }
            d_src.upload(packed, stream);
            d_src32.upload(packed32, stream);
            {   // one more level: bounding sphere / largest weight of every LDS fill's worth of clusters (kernel: kChunk = 4 in fp64, 16 in fp32), appended
                // to the cluster records -- a node tile tests the whole fill first and only then its clusters
                const int per = conv_chunk<T>();
                const int nchunks = (n_clusters + per - 1) / per;
                for (int g = 0; g < nchunks; g++) {
                    const int a = g * per, b = std::min(n_clusters, a + per);
                    double cc[3] = {0, 0, 0};
                    for (int c = a; c < b; c++)
                        for (int t = 0; t < 3; t++) cc[t] += cl[kConvClusterRec * (size_t)c + t];
                    for (int t = 0; t < 3; t++) cc[t] /= (b - a);
                    double rad = 0., lnw = -1.0e30;
                    for (int c = a; c < b; c++) {
                        double d2 = 0.;
                        for (int t = 0; t < 3; t++) {
                            const double d = (double)cl[kConvClusterRec * (size_t)c + t] - cc[t];
                            d2 += d * d;
                        }
                        rad = std::max(rad, std::sqrt(d2) + (double)cl[kConvClusterRec * (size_t)c + 3]);
                        lnw = std::max(lnw, (double)cl[kConvClusterRec * (size_t)c + 4]);
                    }
                    for (int t = 0; t < 3; t++) cl.push_back((float)cc[t]);
                    cl.push_back((float)(rad * 1.00001 + 1e-30));
                    cl.push_back((float)lnw);
                    cl.push_back(0.f);   // (record stride; the fill-level records carry no weight sum)
                }
            }
            d_clusters.upload(cl, stream);
            if constexpr (sizeof(T) == 8) {   // (fp64 solve: the tiered kernel reads these very records -- clusters of 64, unscaled weights)
                if (!solve_only && knob("SHM_CONV_DROP_K") == nullptr && knob("SHM_CONV_DROP_K_FORMULA") == nullptr)
                    conv_drop_K = choose_drop_K(reinterpret_cast<const double*>(packed.data()), n_clusters, cl.data(), kConvClusterRec, lambda, n, cell, conv_drop_eps64, S, conv_drop_K);
            }
            if constexpr (sizeof(T) == 4) {
                // fp32 solve through the tiered kernel (conv_tiered_kernel<NPT, float, false>): it stages fp64 source records in clusters of 64 and scales the weights itself
                n_clusters_t = (int)((S + kTierCluster - 1) / kTierCluster);
                const int64_t Sp = (int64_t)n_clusters_t * kTierCluster;
                std::vector<double> pk((size_t)Sp * 6, 0.);
                for (int64_t t = 0; t < Sp; t++) {
                    const int64_t sidx = order[(size_t)std::min<int64_t>(t, S - 1)].second;  // padding repeats the last source with zero weight
                    for (int a = 0; a < 3; a++) {
                        pk[6 * t + a] = h_pos[3 * sidx + a] - conv_ctr[a];
                        if (t < S) pk[6 * t + 3 + a] = h_wn[3 * sidx + a];
                    }
                }
                std::vector<float> clt((size_t)n_clusters_t * kConvClusterRec, 0.f);
                for (int c = 0; c < n_clusters_t; c++) {
                    const double* q = pk.data() + 6 * (size_t)c * kTierCluster;
                    double cc[3] = {0, 0, 0}, rad = 0., wmax2 = 0., wsum = 0.;
                    for (int e = 0; e < kTierCluster; e++)
                        for (int a = 0; a < 3; a++) cc[a] += q[6 * e + a] / kTierCluster;
                    for (int e = 0; e < kTierCluster; e++) {
                        double d2 = 0., w2 = 0.;
                        for (int a = 0; a < 3; a++) {
                            d2 += (q[6 * e + a] - cc[a]) * (q[6 * e + a] - cc[a]);
                            w2 += q[6 * e + 3 + a] * q[6 * e + 3 + a];
                        }
                        rad = std::max(rad, std::sqrt(d2));
                        wmax2 = std::max(wmax2, w2);
                        wsum += std::sqrt(w2);
                    }
                    for (int a = 0; a < 3; a++) clt[kConvClusterRec * (size_t)c + a] = (float)cc[a];
                    clt[kConvClusterRec * (size_t)c + 3] = (float)(rad * 1.00001 + 1e-30);
                    clt[kConvClusterRec * (size_t)c + 4] = wmax2 > 0. ? (float)(0.5 * std::log(wmax2) + 1e-5) : -1.0e30f;
                    clt[kConvClusterRec * (size_t)c + 5] = wsum > 0. ? (float)(std::log(wsum) + 1e-5) : -1.0e30f;
                }
                if (!solve_only && knob("SHM_CONV_DROP_K") == nullptr && knob("SHM_CONV_DROP_K_FORMULA") == nullptr)
                    conv_drop_K = choose_drop_K(pk.data(), n_clusters_t, clt.data(), kConvClusterRec, lambda, n, cell, conv_drop_eps32, S, conv_drop_K);
                d_src_t.upload(pk, stream);
                d_clusters_t.upload(clt, stream);
            }
            if (!d_exptab.p) {
                std::vector<double> tab(2048);
                for (int j = 0; j < 2048; j++) tab[(size_t)j] = std::exp2((double)j / 2048.0);
                d_exptab.upload(tab, stream);
            }
            HIPCHK(hipStreamSynchronize(stream));
        }

        // z-slab plan (every rank computes the whole of it from the same inputs): equal planes, or -- shm_config.slab_plan = SHM_SLAB_PLAN_STEP1 -- planes
        // weighted by the Step-1 work the kernels' culling / tier rules leave in them (Step 1 is 80-97 % of a solve and does not shard evenly otherwise)
        {
            const bool weighted = cfg.slab_plan == SHM_SLAB_PLAN_STEP1 && total_slabs > 1 && !solve_only;
            slab_bounds.assign((size_t)total_slabs + 1, 0);
            if (weighted) {
                std::vector<double> w((size_t)n);
                step1_plane_weights_host(S, h_pos.data(), h_wn.data(), lambda, n, bbox_min, cell, sizeof(T) == 8 ? SHM_F64 : SHM_F32, conv_tier_log, w.data(),
                                         sizeof(T) == 4 && knob("SHM_CONV32_CLASSIC") == nullptr);
                plan_slabs_weighted(n, total_slabs, w.data(), sizeof(T) == 8 ? 4 : 8, slab_bounds);
            } else {
                for (int sidx = 0; sidx < total_slabs; sidx++) {
                    int32_t k0, k1;
                    shm_plan_slab(n, total_slabs, sidx, &k0, &k1);
                    slab_bounds[(size_t)sidx] = k0;
                    slab_bounds[(size_t)sidx + 1] = k1;
                }
            }
            slabs_equal = true;
            for (int sidx = 0; sidx < total_slabs; sidx++) {
                int32_t k0, k1;
                shm_plan_slab(n, total_slabs, sidx, &k0, &k1);
                slabs_equal = slabs_equal && slab_bounds[(size_t)sidx] == k0 && slab_bounds[(size_t)sidx + 1] == k1;
            }
        }
        vec = (n % vec_width<T>() == 0) ? vec_width<T>() : 1;
        // keep the device arrays across calls with the same grid size (the reference's `rebuild=false` reuse, :8): a repeated
        // computeDistance() then costs no hipMalloc/hipFree
        if (!(have_problem && n == alloc_n && (int)slabs.size() == cfg.local_slabs)) {
            slabs.clear();
            slabs.resize(cfg.local_slabs);
        }
        alloc_n = n;
        for (int ls = 0; ls < cfg.local_slabs; ls++) {
            Slab<T>& sl = slabs[ls];
            int32_t k0, k1;
            slab_range(first_slab + ls, &k0, &k1);
            sl.k0 = k0;
            sl.k1 = k1;
            sl.nzl = k1 - k0;
            sl.plane = (size_t)n * n;
            sl.nown = sl.plane * sl.nzl;
            sl.ntot = sl.plane * (sl.nzl + 2);
            if (sl.ntot >= ((size_t)1 << 32)) throw Error(SHM_ERR_INVALID, "slab too large for 32-bit local node indices; use more slabs");
            if (!solve_only)
                for (DevArray<T>* a : {&sl.Y0, &sl.Y1, &sl.Y2}) a->alloc(sl.ntot);
            for (DevArray<T>* a : {&sl.r, &sl.x, &sl.p, &sl.q}) a->alloc(sl.ntot);
            // ghosts of p are read only where a neighbour exists, but zero everything once for hygiene
            HIPCHK(hipMemsetAsync(sl.p.p, 0, sl.ntot * sizeof(T), stream));
            if (!solve_only) HIPCHK(hipMemsetAsync(sl.Y2.p, 0, sl.ntot * sizeof(T), stream));
            sl.partials.alloc(std::max<size_t>(kMaxPartials, (size_t)n * n / 8 + 8));
            sl.pq.alloc(1);
            sl.sc.alloc(SC_COUNT);
            HIPCHK(hipMemsetAsync(sl.sc.p, 0, SC_COUNT * sizeof(double), stream));
            sl.gp.n = n;
            sl.gp.nzl = sl.nzl;
            sl.gp.k0 = sl.k0;
            sl.gp.inv_h = 1. / cell;
            sl.gp.inv_h2 = 1. / (cell * cell);
        }
        HIPCHK(hipStreamSynchronize(stream));
        precond_ready = false;
        have_problem = true;
        have_conv = have_div = have_phi = have_constraints = false;
        // (round 6: the whole-grid solver of the gathered multi-rank solve is created when a solve first takes that path -- ensure_full() -- so that a run whose solves
        // all take the slab-distributed forms never allocates whole-grid arrays on every rank)
        full_problem_set = false;
        if (!(cfg.world > 1 && !solve_only && n >= 4 && n <= 1024)) full.reset();   // (a single slab has a fast Poisson solve for every n: fft_available() / gemm_dct())
        log("[shm] problem set: n=%d N=%zu S=%lld slabs=%d(local %d) vec=%d", n, N, (long long)S, total_slabs, cfg.local_slabs, vec);
    }

    // shm_opts.step1_arith: the tiered kernel unless the caller (or SHM_CONV_EXACT=1, read per call: tests flip it inside one process) asks for the reference's arithmetic
    // the whole-grid solver behind solve_gathered(): created and given the current problem on first use
    bool full_wanted() const { return cfg.world > 1 && !solve_only && n >= 4 && n <= 1024; }
    void ensure_full() {
        if (!full_wanted()) return;
        if (!full) {
            shm_config c = cfg;
            c.world = 1;
            c.rank = 0;
            c.local_slabs = 1;
            c.rccl_unique_id = nullptr;
            full.reset(new Solver<T>(c, true));
        }
        if (!full_problem_set) {
            shm_sources src{};
            src.S = S;
            src.pos = h_pos.data();
            src.wnormal = h_wn.data();
            src.area = h_area.data();
            src.lambda = lambda;
            shm_grid g{};
            g.n = n;
            for (int a = 0; a < 3; a++) g.bbox_min[a] = bbox_min[a];
            g.cell = cell;
            full->set_problem(src, g);
            full_problem_set = true;
        }
    }
    void select_step1_arith(int arith) {
        if (arith != SHM_STEP1_AUTO && arith != SHM_STEP1_EXACT_F64) throw Error(SHM_ERR_INVALID, "unknown step1_arith");
        conv_tiered = sizeof(T) == 8 && arith == SHM_STEP1_AUTO && knob("SHM_CONV_EXACT") == nullptr && tier_exponent_span_ok();
        // SHM_STEP1_EXACT_F64 (round 5, late): the same kernel with nothing far and nothing dropped -- every (node, source) pair through its fp64 body, which is the
        // leaner of the two fp64 bodies since this round (exponent by integer add, four pairs of a source in flight: 22 vector instructions per pair against the
        // all-fp64 kernel's 30) and leaves the set-up room beside it -- where EVERY pair of the grid stays inside the exponent span of a block's scale; the
        // all-fp64 kernel (gradual underflow by v_ldexp_f64) otherwise and behind SHM_CONV_EXACT_CLASSIC=1 (A/B knob).
        conv_tier_exact = sizeof(T) == 8 && arith == SHM_STEP1_EXACT_F64 && knob("SHM_CONV_EXACT_CLASSIC") == nullptr && tier_exponent_span_ok(true);
        conv_tiered = conv_tiered || conv_tier_exact;
        // fp32 handles (round 5): the same kernel with every kept pair in its packed-fp32 body and fp32 output; SHM_CONV32_CLASSIC=1: conv_normalize_kernel<float> (A/B)
        conv_tiered32 = sizeof(T) == 4 && knob("SHM_CONV32_CLASSIC") == nullptr && tier_exponent_span_ok();
    }
    // The tiered kernel's near tier works relative to one power of two per block and inserts a term's own power of two into the exponent field by an integer
    // add (yukawa_near): valid while no evaluated term of a block is more than 2^-990 below the block's scale.  A source that is not dropped lies at most
    // (skip + ln(w_s / w_near)) / lambda further from the block than the block's nearest source, and a block spans 2 rt: the evaluated exponents of a block span
    // at most lambda' (4 rt) + skip' + log2(w_max / w_min) bits (62 bits stand in for a nearest source of weight zero: the kernel's floor of 1e-37 on w_near^2).
    // Beyond that -- a cell of some 25 mean edge lengths, tCoef ~ 1e-3 -- Step 1 runs in the all-fp64 kernel, whose v_ldexp_f64 underflows gradually.
    bool tier_exponent_span_ok(bool every_pair = false) const {
        const double rt = std::sqrt(2 * 3.5 * 3.5 + 1.5 * 1.5) * cell;
        // (every_pair: nothing is dropped -- a block's evaluated sources lie up to the grid's diagonal further away than its nearest one; sources sit inside the grid's box)
        // (round 6: what is evaluated for certain lies inside the HARD drop threshold, the old rule on an eighth of the budget: ln 8 further out)
        const double reach = every_pair ? lambda * std::sqrt(3.0) * (double)(n - 1) * cell : std::min((sizeof(T) == 8 ? conv_tier_skip_base : conv_skip_base) + 2.0794415416798357, 1.0e6);
        const double bits = (4.0 * rt * lambda + reach + std::max(conv_w_span, 43.0)) * 1.4426950408889634 + 16.0;
        return bits < 990.0;
    }
    void need_problem() const {
        if (!have_problem) throw Error(SHM_ERR_STATE, "shm_grid_set_problem has not been called");
    }

    // ------------------------------------------------------------------------------------------
    // Steps 1+2
    // Order of the z-layers of blocks in the tiered kernel's work queues (round 5, late).  The queues used to run bottom to top, each XCD's a contiguous range of layers: the
    // last blocks handed out were whatever stood at the end of the slowest range, and a wave finishing its last block leaves its SIMD slot idle for the rest of the kernel
    // (2.2 % of the wave-cycles at 256^3, profiles/r05_sq_counters_conv.txt).  The blocks that cost most are the ones near the sources -- the grid's centre (buildGrid centres
    // the grid on the input) --, the ones at the faces drop most sources.  So: layers sorted by the distance of their GLOBAL plane from the grid's mid-plane, dealt round-robin to
    // the eight queues (every XCD gets the same mix and runs centre -> face), i.e. the blocks handed out last are the cheapest ones.  Within a layer nothing changes (x-adjacent
    // blocks stay neighbours in one queue: their halves of a 128-byte line of Y still meet in one L2).  SHM_TIER_LAYER_ORDER=0: bottom to top (A/B knob).
    const int* layer_order_for(Slab<T>& sl, int kk_begin, int layers, int npt) {
        const bool off = knob("SHM_TIER_LAYER_ORDER") != nullptr && atoi(knob("SHM_TIER_LAYER_ORDER")) == 0;   // (read per launch: a test flips it inside one process)
        if (off || layers < 16) return nullptr;
        const long long key = ((long long)kk_begin << 40) ^ ((long long)layers << 20) ^ ((long long)npt << 8) ^ (long long)sl.k0 * 0x9E3779B1LL;
        if (sl.layer_order.p && sl.layer_order_key == key) return sl.layer_order.p;
        std::vector<std::pair<double, int>> byc((size_t)layers);
        for (int l = 0; l < layers; l++) byc[(size_t)l] = {std::fabs((double)(sl.k0 + kk_begin - 1 + l * npt) + 0.5 * (npt - 1) - 0.5 * (n - 1)), l};
        std::stable_sort(byc.begin(), byc.end());
        std::vector<int> order;
        order.reserve((size_t)layers);
        for (int q = 0; q < 8; q++)
            for (int r = q; r < layers; r += 8) order.push_back(byc[(size_t)r].second);
        sl.layer_order.upload(order, stream);
        sl.layer_order_key = key;
        return sl.layer_order.p;
    }
    // Which far rule the tiered fp64 Step 1 classifies with (ConvParams::far_rule).  SHM_TIER_FAR_RULE=0/1: A/B knob.
    int far_rule_now() const {
        const char* e = knob("SHM_TIER_FAR_RULE");
        if (e) return atoi(e) != 0;
        return 0;   // (without a sample: the box rule; far_rule_plan decides where a sample runs)
    }
    // The differential far rule moves a tenth of the pairs from the fp64 body to the packed-fp32 one -- and on some inputs fills the tier so far that the a-posteriori test
    // sends a fifth of the blocks through the second pass (measured, Step 1 alone: bunny_small 256^3 -5.5 %, rocker 128^3 -17 %, SprayBottle.pc 256^3 -12.5 %; rocker 256^3
    // +4.5 %, rocker 512^3 +29 %; profiles/r05_far_rule.txt).  Which it is depends on how much of |X| cancels where, i.e. on the input.  So the kernel finds out on a SAMPLE:
    // eight layers of blocks spread over the planes run under the rule (real work: their Y stands) and report their far pairs and the pairs evaluated again; every queue then
    // works through one layer under the box rule -- work that needs no verdict, long enough for the sample to finish -- and the rest of the grid runs under the rule iff at
    // most 4 % of the sample's far pairs were evaluated again (every wave reads the same counters: far_rule_plan, conv_tiered_kernel).  One launch, no host round trip, nothing
    // tuned outside the solve; deterministic (same sample, counters and verdict in every solve of a problem).  Where one process holds the whole grid, the grid has >= 64 layers
    // of blocks (256^3 upwards) and the budget is the default one; everything else keeps the box rule.  A first version that read the sample's counters on the host between two
    // launches lost the gain to the sample launch's tail.
    // Does this Step 1 decide its far rule from a sample?  Builds (once per problem) the queue order of the layers and the rule of every position: each of the eight queues
    // starts with ONE sample layer (differential rule; spread over the planes), goes on with one of the eight most central remaining layers under the box rule -- work that
    // needs no verdict, long enough for every sample block to finish meanwhile --, and then runs its share of the rest, centre to faces, under the rule the sample earned.
    bool far_rule_plan(Slab<T>& sl, int planes, int npt) {
        if (sizeof(T) != 8 || !conv_tiered || conv_tier_exact || slabs.size() != 1 || total_slabs != 1 || knob("SHM_TIER_FAR_RULE") != nullptr || knob("SHM_TIER_NO_SAMPLE") != nullptr) return false;
        if (step1_budget > 0. && step1_budget != kTierBudget) return false;
        const int layers = (planes + npt - 1) / npt;
        if (layers < 64 || layers % 8 != 0 || planes % npt != 0) return false;
        const long long key = ((long long)layers << 32) ^ ((long long)npt << 16);
        if (sl.sample_key != key) {
            std::vector<char> role((size_t)layers, 2);
            std::vector<int> A, B1;
            const int nA = 8;   // sample layers: one per queue.  (Four -- every second queue -- where a layer holds 4096 blocks and more was measured at 512^3: rocker no cheaper, 584.9 against
                                // 585.3 ms, and the bunny's verdict flipped to the box rule, 190.2 against 188.6 ms: the smaller sample sits on other planes.)
            for (int a = 0; a < nA; a++) {
                const int l = (int)(((long long)(2 * a + 1) * layers) / (2 * nA));
                A.push_back(l);
                role[(size_t)l] = 1;
            }
            std::vector<std::pair<double, int>> byc;
            for (int l = 0; l < layers; l++)
                if (role[(size_t)l] != 1) byc.push_back({std::fabs((double)(sl.k0 + l * npt) + 0.5 * (npt - 1) - 0.5 * (n - 1)), l});
            std::stable_sort(byc.begin(), byc.end());
            for (int a = 0; a < 8; a++) B1.push_back(byc[(size_t)a].second);
            std::vector<int> order;
            std::vector<unsigned char> rule;
            size_t next_rest = 8;   // byc[0 .. 7] are the box layers; the rest is handed out centre to faces, a layer per queue in turn
            std::vector<std::vector<int>> rest(8);
            {
                std::vector<int> want(8);
                for (int q = 0; q < 8; q++) want[(size_t)q] = layers / 8 - 1 - ((nA == 8 || q % 2 == 0) ? 1 : 0);
                bool any = true;
                while (any && next_rest < byc.size()) {
                    any = false;
                    for (int q = 0; q < 8 && next_rest < byc.size(); q++)
                        if ((int)rest[(size_t)q].size() < want[(size_t)q]) {
                            rest[(size_t)q].push_back(byc[next_rest++].second);
                            any = true;
                        }
                }
            }
            for (int q = 0, a = 0; q < 8; q++) {
                if (nA == 8 || q % 2 == 0) {
                    order.push_back(A[(size_t)a++]);
                    rule.push_back(1);
                }
                order.push_back(B1[(size_t)q]);
                rule.push_back(0);
                for (int l : rest[(size_t)q]) {
                    order.push_back(l);
                    rule.push_back(2);
                }
            }
            if ((int)order.size() != layers) throw Error(SHM_ERR_INVALID, "far_rule_plan: internal count mismatch");
            sl.sample_blocks = nA * ((n + kTierTX - 1) / kTierTX) * ((n + kTierTY - 1) / kTierTY);
            sl.sample_order.upload(order, stream);
            sl.sample_rule.upload(rule, stream);
            sl.sample_ctr.alloc(4);
            sl.sample_key = key;
        }
        return true;
    }
    void launch_conv() {
        const bool slab_log = knob("SHM_CONV_SLAB_LOG") != nullptr;
        Event slab_ev[2];
        unsigned long long slab_pairs_seen[2] = {0, 0};
        for (Slab<T>& sl : slabs) {
            if (slab_log) {
                HIPCHK(hipStreamSynchronize(stream));
                slab_ev[0].record(stream);
            }
            ConvParams P;
            P.layer_order = nullptr;
            P.unit_rule = nullptr;
            P.sample_ctr = nullptr;
            P.sample_blocks = 0;
            P.far_rule = conv_tiered && !conv_tier_exact && sizeof(T) == 8 ? far_rule_now() : 0;
            P.n = n;
            P.kk_begin = 1;  // owned planes only: the ghost planes of Y are exchanged (exchange_Y_halos), not recomputed -- a ghost plane
            P.kk_end = sl.nzl + 1;  // would cost a whole 16-plane tile layer of Step 1 (41 % extra on 8 GPUs at 256^3)
            P.k0 = sl.k0;
            for (int a = 0; a < 3; a++) P.bbox_min[a] = bbox_min[a] - conv_ctr[a];   // Step 1 works in grid-centred coordinates (see set_problem)
            P.cell = cell;
            for (int a = 0; a < 3; a++) P.pad_pos[a] = P.bbox_min[a] - (double)n * cell;
            P.lambda = lambda;
            P.cexp = -lambda * 2954.639443740597;  // 2048 / ln 2
            P.S = n_clusters * conv_cluster<T>();
            P.n_clusters = n_clusters;
            P.far_gap = (float)conv_far_gap;
            // shm_opts.step1_budget b (default kTierBudget = 1e-8) moves the three thresholds that derive from the budget together: the far threshold G by -ln(b / 1e-8)
            // (terms e^-G below the dominant one carry the packed-fp32 error eps_far: G = 8 at 1e-8), the a-posteriori test (b / eps_far) and the drop threshold (b / 5)
            const double budget = step1_budget > 0. ? step1_budget : kTierBudget;
            const double g_shift = std::log(budget / kTierBudget);
            P.tier_log = conv_tier_exact ? 3.0e38f : conv_tiered32 ? -1.0e30f : (float)std::max(2.0, conv_tier_log - g_shift);   // (fp32 solve: every kept source that stays in the fp32 exponent range is "far")
            P.wscale = conv_wscale;
            {   // a-posteriori test of the packed-fp32 tier (shm_conv_tiered.hip.h): budget on Y / calibrated relative error of a far term as it shows up in X
                static const double redo_env = knob("SHM_CONV_REDO_RATIO") ? atof(knob("SHM_CONV_REDO_RATIO")) : -1.;   // A/B knob (0: never)
                const double ratio = redo_env >= 0. ? redo_env : budget / kTierEpsFar;
                P.far_redo_ratio = ratio > 0. ? (float)ratio : 3.0e38f;
            }
            P.skip_base = conv_tier_exact ? 3.0e38f : (float)std::min(conv_tiered ? conv_tier_skip_base - g_shift : conv_skip_base, 3.0e38);
            {   // the tiered kernels' drop rule (round 6; shm_conv_tiered.hip.h): eps follows the budget in the fp64 solve (a fifth of it, as before)
                const double eps = conv_tier_exact ? 0. : sizeof(T) == 8 ? conv_drop_eps64 * (budget / kTierBudget) : conv_drop_eps32;
                const double eps_soft = 0.875 * eps, tau = eps_soft / conv_drop_K, tau_hard = 0.125 * eps / (double)std::max<int64_t>(1, S);
                P.drop_eps_soft = (float)eps_soft;
                P.drop_ltau = eps > 0. ? (float)std::log2(tau) : -3.0e38f;
                P.drop_ltau_hard = eps > 0. ? (float)std::log2(tau_hard) : -3.0e38f;
                P.drop_tau_hard = (float)(tau_hard * 1.0001);
                P.drop_check = P.far_redo_ratio < 1.0e30f ? (float)((double)P.far_redo_ratio / budget) : (float)(1.0 / kTierEpsFar);   // 1 / eps_far
                static const double u0_env = knob("SHM_TIER_U0") ? atof(knob("SHM_TIER_U0")) : -1.;          // experiment knobs (defaults: kTierU0, kTierFlush)
                static const int flush_env = knob("SHM_TIER_FLUSH") ? atoi(knob("SHM_TIER_FLUSH")) : -1;
                P.tier_u0 = (float)(u0_env >= 0. ? u0_env : kTierU0);
                P.tier_flush = flush_env >= 0 ? flush_env : kTierFlush;
            }
            P.inv_lambda = (float)(1.0 / lambda);
            P.tiles_x = (n + kConvTile - 1) / kConvTile;
            P.tiles_y = P.tiles_x;
            // nodes per lane (a z-column sharing dx^2 + dy^2): 4 when the grid still yields a full wave of workgroups, else 2
            const int planes = P.kk_end - P.kk_begin;
            static const int npt_env = knob("SHM_TIER_NPT") ? atoi(knob("SHM_TIER_NPT")) : 0;   // A/B knob (round 5): 2 / 4 nodes per lane whatever the grid
            const bool npt4 = npt_env ? npt_env == 4 : (long long)P.tiles_x * P.tiles_y * ((planes + 15) / 16) >= conv_grid_cap;
            // fp32: 8 nodes per lane on grids large enough, culled as two 16-plane halves (see the kernel); fp64 stays at 4 (8 needs 256 VGPRs: 57 against
            // 44 ms at 256^3).  SHM_CONV_NPT4: A/B knob, and the reference for the bit-identity check of the two shapes (tools/skip_check.py)
            const bool npt4_env = knob("SHM_CONV_NPT4") != nullptr;   // (read per launch: the check flips it inside one process)
            const bool npt8 = sizeof(T) == 4 && !npt4_env && npt4 && (long long)P.tiles_x * P.tiles_y * ((planes + 31) / 32) >= conv_grid_cap;
            const int tile_z = npt8 ? 32 : npt4 ? 16 : 8;
            const double half_z = 0.5 * ((npt8 ? 16 : tile_z) - 1);   // extent of the unit that is culled and carries one exponent offset
            const double tile_diam = 2.0 * std::sqrt(2 * 3.5 * 3.5 + half_z * half_z) * cell;
            P.exact_offset = (lambda * tile_diam > 30.0) ? 1 : 0;  // tile-diameter bound looser than e^-30: per-node offsets
            const int tiles_z = (planes + tile_z - 1) / tile_z;
            P.n_tiles = P.tiles_x * P.tiles_y * tiles_z;
            // Step 1 strides a fixed grid of workgroups over the tiles.  The constraint set-up (second stream, dozens of short dependent kernels) can
            // only run in slots Step 1 leaves free: when every CU is full (the fp64 kernel's registers admit two workgroups per CU) a set-up kernel
            // waits for a Step-1 workgroup to finish a tile column, and the set-up stretches to Step 1's length and beyond (measured at 128^3: 3-5 ms of
            // exposed wait on a 6 ms Step 1).  So when Step 1 is short against the set-up (small grids; one slab of many on a multi-GPU run), an
            // eighth of the resident slots is left free for the set-up stream; long Step-1 launches keep the whole chip (the set-up hides anyway).
            unsigned grid = (unsigned)std::min(P.n_tiles, conv_grid_cap);
            {
                int occ = 0;
                const void* kfn = npt4 ? reinterpret_cast<const void*>(conv_normalize_kernel<T, 4>) : reinterpret_cast<const void*>(conv_normalize_kernel<T, 2>);
                if constexpr (sizeof(T) == 4)
                    if (npt8) kfn = reinterpret_cast<const void*>(conv_normalize_kernel<float, 8>);
                if (conv_tiered) kfn = npt4 ? reinterpret_cast<const void*>(conv_tiered_kernel<4, double, true>) : reinterpret_cast<const void*>(conv_tiered_kernel<2, double, true>);
                if (conv_tiered32) kfn = npt4 ? reinterpret_cast<const void*>(conv_tiered_kernel<4, float, false>) : reinterpret_cast<const void*>(conv_tiered_kernel<2, float, false>);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kBlock, 0) != hipSuccess || occ < 1) occ = 2;
                const unsigned resident = (unsigned)(occ * num_cus);
                const double pairs = (double)sl.nown * (double)S;
                const double conv_est_ms = pairs / (sizeof(T) == 8 ? 1.2e9 : 3.5e9);
                conv_est_total_ms = (&sl == &slabs[0] ? 0. : conv_est_total_ms) + conv_est_ms;
                const double setup_est_ms = 2.2e-3 * (double)std::min<int64_t>(S, (int64_t)8 * n * n);
                static const bool no_reserve = knob("SHM_CONV_NO_RESERVE") != nullptr;  // A/B knob
                if (conv_tiered || conv_tiered32) grid = std::min(grid, resident);   // its waves pull work from a queue: exactly the resident workgroups, nothing left to hand out
                if (!no_reserve && !o_fast_hint && conv_est_ms < 4.0 * setup_est_ms && grid > resident - resident / 8) grid = resident - resident / 8;
            }
            // Several launches over z chunks instead of one: the set-up stream's kernels are dispatched only where a Step-1 launch has no workgroups left to
            // hand out -- its tail -- so one long launch makes the ~135 dependent launches of the constraint set-up finish with Step 1 whatever their own
            // length (512^3: 251 ms for 6 ms of work; split in 8: 42 ms, Step 1 itself +1 %).  Chunks of ~2.5 ms of Step 1 give the set-up a window that often.
            // Only where the tiles cost about the same (the kernel spans the grid: lambda * side < 100, nothing is culled) -- with culling the persistent
            // workgroups of ONE launch balance the uneven tiles, and every extra launch adds an uneven tail (rocker 512^3 fp32, 16 launches: +10 %) -- and
            // where Step 1 is long enough to matter (>= 10 ms); at most 16 chunks (256^3: set-up done at 38 instead of 42.7 ms with 8, Step 1 unchanged).
            static const int split_env = knob("SHM_CONV_SPLIT") ? atoi(knob("SHM_CONV_SPLIT")) : 0;   // A/B knob (1: one launch)
            const bool uniform_tiles = lambda * cell * n < 100.;
            // (The tiered fp64 kernel needs none of this: its grid is exactly the resident workgroups -- nothing left to hand out -- and its 176 registers leave room on
            // every SIMD for a wave of the set-up kernels, which therefore run WHILE Step 1 runs; one launch, balanced by its work queue.)
            const int want_chunks = split_env > 0 ? split_env : (conv_tiered || conv_tiered32) ? 1 : (uniform_tiles && conv_est_total_ms >= 10. ? std::min(16, (int)std::lround(conv_est_total_ms / 2.5)) : 1);
            const int nchunks = std::max(1, std::min(tiles_z, want_chunks));
            const int chunk_planes = ((tiles_z + nchunks - 1) / nchunks) * tile_z;
            if (!d_pair_counters.p) d_pair_counters.alloc(3);
            if (!d_unit_counters.p) d_unit_counters.alloc(8 * kMaxConvLaunches);
            if (&sl == &slabs[0]) {
                HIPCHK(hipMemsetAsync(d_pair_counters.p, 0, 3 * sizeof(unsigned long long), stream));
                HIPCHK(hipMemsetAsync(d_unit_counters.p, 0, 8 * kMaxConvLaunches * sizeof(unsigned), stream));
                conv_launch_index = 0;
                conv_launches_last = 0;
            }
            unsigned long long* const cnt = d_pair_counters.p;
            // one launch over the local planes [1 + b0, 1 + b1) with the kernel shape `sel` (nodes per lane: 8 / 4 / 2; the tiered kernel ignores it)
            auto launch_range = [&](int b0, int b1, int sel, int rule = -1, const int* order = nullptr, int order_layers = 0) {
                if (b1 <= b0) return;
                conv_launches_last++;
                const int tz_sel = 4 * sel;
                ConvParams Pc = P;
                Pc.kk_begin = 1 + b0;
                Pc.kk_end = std::min(sl.nzl + 1, 1 + b1);
                Pc.n_tiles = P.tiles_x * P.tiles_y * ((Pc.kk_end - Pc.kk_begin + tz_sel - 1) / tz_sel);
                const dim3 g((unsigned)std::min<long long>(Pc.n_tiles, grid));
                if (conv_tiered || conv_tiered32) {
                    // the unit of work is a wave's sub-tile (8 x 8 x NPT nodes), pulled from a per-launch queue head by the waves of a grid no larger than what is resident
                    if (conv_launch_index >= kMaxConvLaunches) throw Error(SHM_ERR_INVALID, "too many Step-1 launches");
                    unsigned* const head = d_unit_counters.p + 8 * conv_launch_index++;   // eight queue heads (one per XCD) per launch
                    const int npt = npt4 ? 4 : 2;
                    Pc.tiles_x = (n + kTierTX - 1) / kTierTX;
                    Pc.tiles_y = (n + kTierTY - 1) / kTierTY;
                    const int layers = order ? order_layers : (Pc.kk_end - Pc.kk_begin + npt - 1) / npt;
                    Pc.n_tiles = Pc.tiles_x * Pc.tiles_y * layers;
                    Pc.layer_order = order ? order : layer_order_for(sl, Pc.kk_begin, layers, npt);
                    if (rule >= 0) Pc.far_rule = rule;
                    const dim3 gt((unsigned)std::min<long long>((Pc.n_tiles + 3) / 4, grid));
                    if constexpr (sizeof(T) == 8) {
                        if (npt4)
                            hipLaunchKernelGGL((conv_tiered_kernel<4, double, true>), gt, dim3(kBlock), 0, stream, Pc, d_src.p, d_clusters.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
                        else
                            hipLaunchKernelGGL((conv_tiered_kernel<2, double, true>), gt, dim3(kBlock), 0, stream, Pc, d_src.p, d_clusters.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
                    } else {
                        Pc.S = n_clusters_t * kTierCluster;
                        Pc.n_clusters = n_clusters_t;
                        if (npt4)
                            hipLaunchKernelGGL((conv_tiered_kernel<4, float, false>), gt, dim3(kBlock), 0, stream, Pc, d_src_t.p, d_clusters_t.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
                        else
                            hipLaunchKernelGGL((conv_tiered_kernel<2, float, false>), gt, dim3(kBlock), 0, stream, Pc, d_src_t.p, d_clusters_t.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
                    }
                    return;
                }
                // the tiles of a launch are handed out by a queue head (culled workloads: their costs differ severalfold); SHM_CONV_STATIC: static stride (A/B knob)
                static const bool static_tiles = knob("SHM_CONV_STATIC") != nullptr;
                unsigned* head = nullptr;
                if (!static_tiles && conv_launch_index < kMaxConvLaunches) head = d_unit_counters.p + 8 * conv_launch_index++;
                if constexpr (sizeof(T) == 4) {
                    if (sel == 8) {
                        hipLaunchKernelGGL((conv_normalize_kernel<float, 8>), g, dim3(kBlock), 0, stream, Pc, d_src.p, d_src32.p, d_clusters.p, d_exptab.p, sl.Y0.p, sl.Y1.p,
                                           sl.Y2.p, cnt, head);
                        return;
                    }
                }
                if (sel >= 4)
                    hipLaunchKernelGGL((conv_normalize_kernel<T, 4>), g, dim3(kBlock), 0, stream, Pc, d_src.p, d_src32.p, d_clusters.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
                else
                    hipLaunchKernelGGL((conv_normalize_kernel<T, 2>), g, dim3(kBlock), 0, stream, Pc, d_src.p, d_src32.p, d_clusters.p, d_exptab.p, sl.Y0.p, sl.Y1.p, sl.Y2.p, cnt, head);
            };
            const int sel0 = npt8 ? 8 : npt4 ? 4 : 2;
            if (sizeof(T) == 4 && !conv_tiered32 && nchunks == 1 && sel0 > 2 && planes % tile_z != 0) {
                // planes that do not fill the last layer of 32- (16-) plane tiles (a weighted slab plan cuts at multiples of 8 planes): that layer would run with most
                // of its lanes dead, so the remainder goes to the kernel shapes with fewer nodes per lane (same arithmetic; the 8-plane shape culls in smaller units)
                int b = (planes / tile_z) * tile_z;
                launch_range(0, b, sel0);
                for (int sel = sel0 / 2; sel >= 2; sel /= 2) {
                    const int tz_sel = 4 * sel, take = sel > 2 ? ((planes - b) / tz_sel) * tz_sel : planes - b;
                    launch_range(b, b + take, sel);
                    b += take;
                }
            } else if (far_rule_plan(sl, planes, npt4 ? 4 : 2)) {
                // one launch: sample layers, box layers, then the layers that run under the sample's verdict (far_rule_plan; the waves decide from the sample's counters themselves)
                const int npt = npt4 ? 4 : 2, layers = planes / npt;
                HIPCHK(hipMemsetAsync(sl.sample_ctr.p, 0, 4 * sizeof(unsigned long long), stream));
                P.unit_rule = sl.sample_rule.p;
                P.sample_ctr = sl.sample_ctr.p;
                P.sample_blocks = sl.sample_blocks;
                launch_range(0, planes, sel0, 0, sl.sample_order.p, layers);
                P.unit_rule = nullptr;
            } else {
                for (int b0 = 0; b0 < planes; b0 += chunk_planes) launch_range(b0, std::min(planes, b0 + chunk_planes), sel0);
            }
            if (slab_log) {   // measurement knob (tools/slab_plan_check.py): this slab's Step 1 alone -- time and evaluated pairs -- to stderr
                slab_ev[1].record(stream);
                HIPCHK(hipStreamSynchronize(stream));
                unsigned long long h[2] = {0, 0};
                HIPCHK(hipMemcpy(h, d_pair_counters.p, sizeof h, hipMemcpyDeviceToHost));
                fprintf(stderr, "[shm] step1 slab planes [%d,%d) ms %.3f pairs_fp64 %.6e pairs_fp32 %.6e\n", sl.k0, sl.k1, elapsed(slab_ev[0], slab_ev[1]),
                        (double)(h[0] - slab_pairs_seen[0]), (double)(h[1] - slab_pairs_seen[1]));
                slab_pairs_seen[0] = h[0];
                slab_pairs_seen[1] = h[1];
            }
        }
        HIPCHK(hipGetLastError());
        exchange_Y_halos();
        have_conv = true;
    }
    // ghost planes of Y0, Y1, Y2 from the neighbouring slabs (device copies / ncclSend+ncclRecv): the divergence reads Y2 one plane
    // below (and above, at the global top), the fast integration reads all three one plane below
    void exchange_Y_halos() {
        if (total_slabs <= 1) return;
        for (int sel : {ARR_Y0, ARR_Y1, ARR_Y2}) halo_exchange(sel);
    }

    void launch_div(int scrub) {
        static const bool classic = knob("SHM_DIV_CLASSIC") != nullptr;   // A/B knob: the one-node-per-thread kernel of rounds 1-3
        for (Slab<T>& sl : slabs) {
            if (classic || vec == 1) {   // (n not a multiple of the vector width: scalar kernel)
                hipLaunchKernelGGL((divergence_kernel<T>), dim3((unsigned)((n + kBlock - 1) / kBlock), (unsigned)n, (unsigned)sl.nzl), dim3(kBlock), 0, stream, sl.gp,
                                   sl.Y0.p, sl.Y1.p, sl.Y2.p, sl.r.p, scrub);
                sl.div_sum_blocks = 0;
                continue;
            }
            constexpr int V = vec_width<T>();
            const int lanes = (n + V - 1) / V;
            int LX = 1;
            while (LX < lanes && LX < kBlock) LX *= 2;
            const int xchunks = (lanes + LX - 1) / LX, RB = kBlock / LX, rowgroups = (n + RB - 1) / RB;
            // planes per workgroup: 4 measured best at 256^3 and 512^3 in both precisions (512^3 fp64: 0.79 ms against 0.84 with 32 and 0.80 with 2 --
            // many short marches keep more loads in flight than few deep ones; the first plane's extra load is a fifth of a march)
            int ZC = 4;
            static const int zc_env = knob("SHM_DIV_ZC") ? atoi(knob("SHM_DIV_ZC")) : 0;   // A/B knob
            if (zc_env > 0) ZC = zc_env;
            const unsigned nblk = (unsigned)((long long)xchunks * rowgroups * ((sl.nzl + ZC - 1) / ZC));
            sl.div_partials.alloc(nblk);
            sl.div_sum_blocks = (int)nblk;
            hipLaunchKernelGGL((divergence_march_kernel<T, V>), dim3(nblk), dim3(kBlock), 0, stream, sl.gp, LX, xchunks, rowgroups, ZC, sl.Y0.p, sl.Y1.p, sl.Y2.p,
                               sl.r.p, scrub, sl.div_partials.p);
        }
        HIPCHK(hipGetLastError());
        have_div = true;
    }

    void run_conv(int step1_arith) override {
        need_problem();
        HIPCHK(hipSetDevice(cfg.device));
        step1_budget = 0.;   // (the stage entry points run with the library's defaults, whatever the last solve's shm_opts asked for)
        dual_form_req = SHM_DUAL_AUTO;
        select_step1_arith(step1_arith);
        launch_conv();
        HIPCHK(hipStreamSynchronize(stream));
    }
    void run_divergence(int scrub) override {
        need_problem();
        if (!have_conv) throw Error(SHM_ERR_STATE, "run_conv must precede run_divergence");
        HIPCHK(hipSetDevice(cfg.device));
        launch_div(scrub);
        HIPCHK(hipStreamSynchronize(stream));
    }

    // ------------------------------------------------------------------------------------------
    // Constraint rows (:80-98 / :186-204), sequential over the sources like the reference.
    void build_rows() {
        rows.clear();
        std::unordered_set<uint64_t> used;
        used.reserve((size_t)S * 2);
        const double h = cell;
        for (int64_t s = 0; s < S; s++) {
            const double* b = &h_pos[3 * s];
            const size_t i = (size_t)std::floor((b[0] - bbox_min[0]) / h);
            const size_t j = (size_t)std::floor((b[1] - bbox_min[1]) / h);
            const size_t k = (size_t)std::floor((b[2] - bbox_min[2]) / h);
            const uint64_t cid = i + j * (uint64_t)n + k * (uint64_t)n * n;
            if (!used.insert(cid).second) continue;
            Row r;
            const double tx = (b[0] - (i * h + bbox_min[0])) / h;
            const double ty = (b[1] - (j * h + bbox_min[1])) / h;
            const double tz = (b[2] - (k * h + bbox_min[2])) / h;
            auto ix = [&](size_t a, size_t bb, size_t c) { return (int64_t)(a + bb * (size_t)n + c * (size_t)n * n); };
            r.nodes[0] = ix(i, j, k);
            r.nodes[1] = ix(i + 1, j, k);
            r.nodes[2] = ix(i, j + 1, k);
            r.nodes[3] = ix(i, j, k + 1);
            r.nodes[4] = ix(i + 1, j + 1, k);
            r.nodes[5] = ix(i + 1, j, k + 1);
            r.nodes[6] = ix(i, j + 1, k + 1);
            r.nodes[7] = ix(i + 1, j + 1, k + 1);
            r.coeffs[0] = (1. - tx) * (1. - ty) * (1. - tz);
            r.coeffs[1] = tx * (1. - ty) * (1. - tz);
            r.coeffs[2] = (1. - tx) * ty * (1. - tz);
            r.coeffs[3] = (1. - tx) * (1. - ty) * tz;
            r.coeffs[4] = tx * ty * (1. - tz);
            r.coeffs[5] = tx * (1. - ty) * tz;
            r.coeffs[6] = (1. - tx) * ty * tz;
            r.coeffs[7] = tx * ty * tz;
            r.cell[0] = (int)i; r.cell[1] = (int)j; r.cell[2] = (int)k;
            r.t[0] = tx; r.t[1] = ty; r.t[2] = tz;
            rows.push_back(r);
        }
        m = (int)rows.size();
        mp = ((m + kGJ - 1) / kGJ) * kGJ;
    }

    // shift items: every source contributes one bilinear evaluation per z-plane of its cell (:405-431); the owner of the
    // plane evaluates it, so the slabs sum to the reference's nested lerp.
    void build_shift_items(hipStream_t stream) {
        const size_t plane = (size_t)n * n;
        shift_host.resize(slabs.size());  // host copies outlive the asynchronous uploads
        for (size_t si = 0; si < slabs.size(); si++) {
            Slab<T>& sl = slabs[si];
            std::vector<ShiftItem>& items = shift_host[si];
            items.clear();
            for (int64_t s = 0; s < S; s++) {
                const double* b = &h_pos[3 * s];
                const int i = (int)std::floor((b[0] - bbox_min[0]) / cell);
                const int j = (int)std::floor((b[1] - bbox_min[1]) / cell);
                const int k = (int)std::floor((b[2] - bbox_min[2]) / cell);
                const double tx = (b[0] - (i * cell + bbox_min[0])) / cell;
                const double ty = (b[1] - (j * cell + bbox_min[1])) / cell;
                const double tz = (b[2] - (k * cell + bbox_min[2])) / cell;
                for (int dz = 0; dz < 2; dz++) {
                    const int kz = k + dz;
                    if (kz < sl.k0 || kz >= sl.k1) continue;
                    ShiftItem it;
                    it.node = (uint32_t)((size_t)i + (size_t)j * n + (size_t)(kz - sl.k0 + 1) * plane);
                    it.pad = 0.f;
                    it.tx = tx;
                    it.ty = ty;
                    it.weight = h_area[s] * (dz == 0 ? (1. - tz) : tz);
                    items.push_back(it);
                }
            }
            sl.n_shift = (int)items.size();
            sl.shift_items.upload(items, stream);
        }
    }
    // device tables of the per-slab reduction buffers for the loop-back sum ([0,ns): red, [ns,2ns): pq)
    void upload_red_tables(hipStream_t stream) {
        std::vector<double*> ptrs;
        for (Slab<T>& sl : slabs) ptrs.push_back(sl.red.p);
        for (Slab<T>& sl : slabs) ptrs.push_back(sl.pq.p);
        d_redptrs.upload(ptrs, stream);
        HIPCHK(hipStreamSynchronize(stream));  // ptrs is a stack object
    }
    void build_shift_items_only() {
        build_shift_items(stream);
        upload_red_tables(stream);
    }

    // Duration of the tiered fp64 Step 1 on this rank's nodes, from the sources alone (host, ~1 ns per (node sample, source), at most 2e6 of them): 32-128 pseudo-random nodes
    // (fixed sequence) against all S sources, each pair classified as the kernel classifies a node BLOCK against a source (distance beyond the node's nearest
    // source, less the block's diameter, against the far threshold G and the drop threshold), and the three shares priced with constants fitted to the 28
    // measured (data file, grid size) pairs of round 4 (tools/r04_ab.py, profiles/r04_all_files.txt): predicted / measured 0.68 ... 1.23, where nominal pairs at
    // a fixed rate -- conv_est_total_ms -- are off by up to 9x on the culled inputs.
    double estimate_step1_ms_tiered() const {
        if (S <= 0) return 0.;
        if (conv_tier_exact) {   // every pair through the fp64 body
            double nodes = 0.;
            for (const Slab<T>& sl : slabs) nodes += (double)sl.nown;
            return nodes * (double)S * 1.229e-9;
        }
        const int K = (int)std::max<int64_t>(32, std::min<int64_t>(128, 2000000 / S));
        const int64_t stride = std::max<int64_t>(1, (S * K + 1999999) / 2000000);   // <= 2e6 (sample, source) pairs whatever S (the floor of 32 samples alone let point
                                                                                     // clouds of 1e5-1e6 sources run tens of ms here): ~2 ms of host time, paid before the set-up is queued
        uint64_t st = 0x9E3779B97F4A7C15ULL;
        auto rnd = [&]() {
            st = st * 6364136223846793005ULL + 1442695040888963407ULL;
            return (double)(st >> 11) * (1.0 / 9007199254740992.0);
        };
        const double ext = (double)(n - 1) * cell, rt2 = 2.0 * 5.17 * cell;
        // (round 6: kept = inside the accumulated drop rule's candidate threshold, ln(K / eps_soft) -- was ln(S / eps); the 2e-9 stands for either precision as it did when the
        // constants below were fitted)
        const double r_near_gap = rt2 + conv_tier_log / lambda, r_keep_gap = rt2 + std::log(std::min((double)S, conv_drop_K) / (0.875 * 2e-9)) / lambda;
        std::vector<float> d2((size_t)S);
        size_t c_near = 0, c_keep = 0;
        for (int k = 0; k < K; k++) {
            const float x = (float)(bbox_min[0] + rnd() * ext), y = (float)(bbox_min[1] + rnd() * ext), z = (float)(bbox_min[2] + rnd() * ext);
            float dmin = 3.0e38f;
            for (int64_t s = 0; s < S; s += stride) {
                const float dx = x - (float)h_pos[3 * s], dy = y - (float)h_pos[3 * s + 1], dz = z - (float)h_pos[3 * s + 2];
                const float v = dx * dx + dy * dy + dz * dz;
                d2[(size_t)s] = v;
                dmin = std::min(dmin, v);
            }
            const double rn = std::sqrt((double)dmin);
            const float tn = (float)((rn + r_near_gap) * (rn + r_near_gap)), tk = (float)((rn + r_keep_gap) * (rn + r_keep_gap));
            for (int64_t s = 0; s < S; s += stride) {
                c_near += d2[(size_t)s] < tn;
                c_keep += d2[(size_t)s] < tk;
            }
        }
        const double sampled = (double)K * (double)((S + stride - 1) / stride);
        const double f_keep = (double)c_keep / sampled;
        const double f_near = conv_tiered32 ? 0. : (double)c_near / sampled;   // (fp32 solve: every kept pair goes through the packed-fp32 body)
        double nodes = 0.;
        for (const Slab<T>& sl : slabs) nodes += (double)sl.nown;
        // (per-pair costs of the round-4 kernel; the round-5 one is ~7 % faster, but the 0.38 of the rule that reads this estimate was drawn with these constants)
        return nodes * (double)S * (1.229e-9 * f_near + 1.84e-10 * (f_keep - f_near) + 3.3e-12);
    }

    // Per-slab CSR pieces, shift items, G = A A^T (sparse triplets -> dense on device -> inverted) and B = A K A^T.
    // Order: everything the inversion needs first (rows, G), then the Gauss-Jordan kernels are enqueued, and the rest of the host
    // work (per-slab lists, B, active-tile lists) runs while the GPU inverts; uploads come last (a pageable copy waits for the stream).
    void build_constraints() {
        hipStream_t stream = stream2;  // everything below runs beside the Step-1 kernel of the main stream
        const auto th0 = std::chrono::steady_clock::now();
        static const int prio_env = knob("SHM_SETUP_PRIO") ? atoi(knob("SHM_SETUP_PRIO")) : -1;   // A/B knob: 0 / 1
        setup_prio = prio_env >= 0 ? prio_env : (conv_tiered && conv_est_total_ms >= 150. && conv_est_total_ms < 1e29 ? 0 : 1);   // (estimate: 256^3 bunny 40, 512^3 320 / 160 ms)
        auto lap = [&](const char* what) { log("[shm]   setup %-28s %.2f ms", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count()); };
        gs_early = green_table_early_ok();
        if (gs_early) {
            enqueue_green_table(stream);
            if (!e_gs_done) e_gs_done.reset(new Event());
            e_gs_done->record(stream);
        }
        build_rows();
        lap("rows");
        const size_t plane = (size_t)n * n;
        // Direct dual solve (moderate m, explicit S): S itself is inverted on the set-up stream instead of G, and the dual system is solved with two dense
        // mat-vecs after Step 1 -- no G, no B, no iteration (see solve_dual), and none of the host tables
        // below that only they and the sparse sweeps of the iterative dual solver need (entries sorted by node, node -> rows hash, active tiles).
        // (shm_opts.dual_form -- `dual_form_req` -- overrides the rules below: DIRECT wherever the explicit S fits, or one of the two iterative forms)
        const bool no_direct = knob("SHM_DUAL_NO_DIRECT") != nullptr || dual_form_req == SHM_DUAL_EXPLICIT_S_CG || dual_form_req == SHM_DUAL_THROUGH_GRID;
        static const int direct_max_m_env = knob("SHM_DUAL_DIRECT_MAX_M") ? atoi(knob("SHM_DUAL_DIRECT_MAX_M")) : 4096;   // single-level Gauss-Jordan range
        const int direct_max_m = dual_form_req == SHM_DUAL_DIRECT ? 16384 : direct_max_m_env;
        dual_direct = dual_direct_requested && !no_direct && m <= direct_max_m;
        // Round 4: between 4096 and 16384 rows the direct solve pays exactly where this rank's Step 1 outlasts the inversion of S that runs beside it (measured,
        // tools/r04_ab.py, ms per solve iterative / direct: rocker 128^3 (m = 4 169) 23.9 / 19.8, SprayBottle 128^3 (4 141) 70.2 / 65.3, chair 256^3 (6 340) 58.7 / 57.8,
        // chair 512^3 (7 748) 373.6 / 367.5 -- and rocker 256^3 (9 110) 97.4 / 106.9, SprayBottle 256^3 (12 620) 174 / 221, knot 128^3 (12 155) 55.6 / 96.5).  The
        // line is drawn per problem from an estimate of Step 1 that knows what the tiers drop (estimate_step1_ms_tiered: +-25 % over the data files) and of the
        // set-up on an idle device (assembly 1.4e-7 m^2, inversion 5.5e-11 m^3, the Green's table); beside Step 1 the set-up runs at ~0.4 of that speed.
        static const bool direct_est_off = knob("SHM_DUAL_DIRECT_EST_OFF") != nullptr;   // A/B knob: the fixed limit alone
        if (!dual_direct && dual_direct_requested && !no_direct && !direct_est_off && (conv_tiered || conv_tiered32) && precond_available() && !gemm_dct() && n <= 512 &&
            m > direct_max_m && m <= 16384 && conv_est_total_ms < 1e29) {
            const double md = (double)m;
            const double table_ms = 10.0 * std::pow((double)n / 512.0, 4.0);
            const double setup_alone_ms = 5.5e-11 * md * md * md + 1.4e-7 * md * md + table_ms;
            // (the whole-grid solver of a multi-rank run hides its set-up behind the RANK's share of Step 1: conv_est_total_ms carries that share in nominal pairs)
            double nodes_here = 0.;
            for (const Slab<T>& sl : slabs) nodes_here += (double)sl.nown;
            const double share = std::min(1.0, conv_est_total_ms / std::max(1e-30, nodes_here * (double)S / 1.2e9));
            const double step1_ms = estimate_step1_ms_tiered() * share;
            dual_direct = setup_alone_ms <= 0.38 * step1_ms;
            log("[shm]   direct solve for m = %d: set-up alone ~%.1f ms, Step 1 ~%.1f ms -> %s", m, setup_alone_ms, step1_ms, dual_direct ? "direct" : "iterative");
        }
        // After the fp32 Step 1 -- which leaves the set-up's kernels no room: they run in its gaps and after it -- the direct solve's extra set-up (the
        // Green's table: three n^4 products, 10 ms at 512^3; the assembly of S) is paid in full, and at 512^3 it costs more than the iterations it replaces
        // (round 4, tools/r04_ab.py: bunny_small 512^3 fp32 112.8 ms direct / 107.6 iterative, bunny.pc 62.7 / 59.4; at 256^3: 15.8 / 18.5, 128^3: 4.7 / 7.3)
        const bool direct_always = knob("SHM_DUAL_DIRECT_ALWAYS") != nullptr || dual_form_req == SHM_DUAL_DIRECT;
        // (the all-fp64 kernel at 512^3: 321 direct / 329 iterative -- fp64 iterations cost twice as much, so only the fp32 solve changes)
        // Round 5: the fp32 solve's Step 1 is the tiered kernel too (two waves per SIMD): its set-up runs beside it like the fp64 solve's, and the direct solve wins
        // again (tools/r05_fp32_forms_ab.sh, ms per solve iterative / direct: bunny_small 512^3 110.0 / 102.2, bunny.pc 512^3 60.6 / 54.6, chair 512^3 214.3 / 209.4)
        if (sizeof(T) == 4 && !conv_tiered32 && n >= 512 && !gemm_dct() && !direct_always) dual_direct = false;
        dual_direct = dual_direct && schur_wanted();   // (schur_wanted() reads dual_direct: with it set only the structural conditions remain)
        const bool need_node_tables = !dual_direct;
        // ---- G = A A^T and B = A K A^T from the (node, row, coef) entries sorted by node: rows meet exactly at shared nodes.
        //      Sorted vectors instead of hash maps: the host part of the set-up is on the critical path of small / multi-GPU runs.
        struct Ent { int64_t node; int row; double coef; };
        std::vector<Ent> ents(need_node_tables ? (size_t)8 * m : 0);
        for (int r = 0; r < m && need_node_tables; r++)
            for (int e = 0; e < 8; e++) ents[(size_t)8 * r + e] = {rows[r].nodes[e], r, rows[r].coeffs[e]};
        std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.node != y.node ? x.node < y.node : x.row < y.row; });
        std::vector<int64_t> unode;       // distinct touched nodes, ascending
        std::vector<int> ustart;          // their entry ranges in `ents`
        for (size_t e = 0; e < ents.size(); e++)
            if (e == 0 || ents[e].node != ents[e - 1].node) {
                unode.push_back(ents[e].node);
                ustart.push_back((int)e);
            }
        ustart.push_back((int)ents.size());
        lap("sorted entries");
        // row by row with a dense scatter-accumulate scratch (value + owner stamp per column): no sorting, no hashing; the
        // column order inside a CSR row is irrelevant for the mat-vec
        // node -> group through a small open-addressing table (binary searching 56 stencil nodes per row dominated the set-up)
        size_t hbits = 4;
        while (((size_t)1 << hbits) < 4 * unode.size() + 16) hbits++;
        const size_t hmask = ((size_t)1 << hbits) - 1;
        std::vector<int64_t> hkey((size_t)1 << hbits, -1);
        std::vector<int> hval((size_t)1 << hbits, -1);
        auto hslot = [&](int64_t node) { return (size_t)(((uint64_t)node * 0x9E3779B97F4A7C15ULL) >> (64 - hbits)) & hmask; };
        for (size_t u = 0; u < unode.size(); u++) {
            size_t h = hslot(unode[u]);
            while (hkey[h] >= 0) h = (h + 1) & hmask;
            hkey[h] = unode[u];
            hval[h] = (int)u;
        }
        auto group_of = [&](int64_t node) -> int {
            size_t h = hslot(node);
            while (hkey[h] >= 0) {
                if (hkey[h] == node) return hval[h];
                h = (h + 1) & hmask;
            }
            return -1;
        };
        std::vector<double> accv((size_t)m, 0.);
        std::vector<int> stamp((size_t)m, -1), cols;
        auto add = [&](int tag, int col, double v) {
            if (stamp[(size_t)col] != tag) {
                stamp[(size_t)col] = tag;
                accv[(size_t)col] = v;
                cols.push_back(col);
            } else {
                accv[(size_t)col] += v;
            }
        };
        // G = A A^T in CSR on the host (rows sharing a node with row r), then either scattered into the dense m x m matrix that the blocked
        // Gauss-Jordan inverts in place, or -- large m -- split into boxes and a separator (two-level inverse, shm_twolevel.hip.h)
        std::vector<int> gptr((size_t)m + 1, 0), gcol;
        std::vector<double> gval;
        gcol.reserve((size_t)m * 32);
        gval.reserve((size_t)m * 32);
        std::vector<int> ugs((size_t)8 * m);
        for (int r = 0; r < m && !dual_direct; r++) {
            cols.clear();
            for (int e = 0; e < 8; e++) {
                const int ug = group_of(rows[r].nodes[e]);
                ugs[(size_t)8 * r + e] = ug;
                for (int y = ustart[ug]; y < ustart[ug + 1]; y++) add(2 * r, ents[y].row, rows[r].coeffs[e] * ents[y].coef);
            }
            for (int c : cols) {
                gcol.push_back(c);
                gval.push_back(accv[(size_t)c]);
            }
            gptr[(size_t)r + 1] = (int)gcol.size();
        }
        lap("G rows");
        DevArray<uint64_t> d_tidx;  // alive until the final synchronisation below
        DevArray<double> d_tval;
        static const int tl_min_m = knob("SHM_TL_MIN_M") ? atoi(knob("SHM_TL_MIN_M")) : 6144;  // dense inverse up to here (m^2 fp32 = 150 MB: L2 / MALL friendly, 3 launches)
        tl.on = !dual_direct && m > tl_min_m && build_two_level(gptr, gcol, gval, d_tidx, d_tval);
        std::vector<uint64_t> tidx;  // alive (like d_tidx / d_tval) until the final synchronisation below
        std::vector<double> tval;
        if (!tl.on && !dual_direct) {
            tidx.reserve(gcol.size() + (size_t)(mp - m));
            tval.reserve(gcol.size() + (size_t)(mp - m));
            for (int r = 0; r < m; r++)
                for (int e = gptr[(size_t)r]; e < gptr[(size_t)r + 1]; e++) {
                    tidx.push_back((uint64_t)r * (uint64_t)mp + (uint64_t)gcol[(size_t)e]);
                    tval.push_back(gval[(size_t)e]);
                }
            for (int a = m; a < mp; a++) {  // identity tail keeps the padded matrix SPD
                tidx.push_back((uint64_t)a * mp + a);
                tval.push_back(1.0);
            }
            ginv_rows = m;
            ginv_ld = mp;
            Ginv.alloc((size_t)mp * mp);
            HIPCHK(hipMemsetAsync(Ginv.p, 0, (size_t)mp * mp * sizeof(double), stream));
            d_tidx.upload(tidx, stream);
            d_tval.upload(tval, stream);
            hipLaunchKernelGGL(scatter_triplets_kernel, dim3(grid_for(tidx.size(), 4096)), dim3(kBlock), 0, stream, (size_t)tidx.size(), d_tidx.p,
                               d_tval.p, Ginv.p);
            HIPCHK(hipGetLastError());
        }
        prepare_schur();
        if (dual_direct) {
            enqueue_schur();
            lap("explicit S and its inversion enqueued");
        } else {
            enqueue_invert_G();
            lap("G uploaded, inversion enqueued");
            enqueue_schur();
            lap("explicit S enqueued");
        }

        // ---- host work that the inversion does not need, while the GPU inverts
        std::vector<int> bptr(m + 1, 0), bcol;
        std::vector<double> bval;
        bcol.reserve((size_t)m * 128);
        bval.reserve((size_t)m * 128);
        {
            const double ih2 = 1. / (cell * cell);
            const int64_t nn = n, pl = (int64_t)n * n;
            for (int r = 0; r < m && !dual_direct; r++) {  // B = A K A^T: K a_r lives on the 8 corners and their in-grid neighbours
                cols.clear();
                for (int e = 0; e < 8; e++) {
                    const int64_t c = rows[r].nodes[e];
                    const double cf = rows[r].coeffs[e];
                    const int64_t k = c / pl, j = (c - k * pl) / nn, i = c - k * pl - j * nn;
                    const int64_t nb[6] = {i > 0 ? c - 1 : -1, i < nn - 1 ? c + 1 : -1, j > 0 ? c - nn : -1, j < nn - 1 ? c + nn : -1,
                                           k > 0 ? c - pl : -1, k < nn - 1 ? c + pl : -1};
                    int deg = 0;
                    for (int q = 0; q < 6; q++) {
                        if (nb[q] < 0) continue;
                        deg++;
                        const int ub = group_of(nb[q]);
                        if (ub < 0) continue;  // K a_r reaches a node no constraint row touches
                        for (int y = ustart[ub]; y < ustart[ub + 1]; y++) add(2 * r + 1, ents[y].row, -cf * ih2 * ents[y].coef);
                    }
                    const int ug = ugs[(size_t)8 * r + e];
                    for (int y = ustart[ug]; y < ustart[ug + 1]; y++) add(2 * r + 1, ents[y].row, deg * cf * ih2 * ents[y].coef);
                }
                for (int c : cols) {
                    bcol.push_back(c);
                    bval.push_back(accv[(size_t)c]);
                }
                bptr[r + 1] = (int)bcol.size();
            }
        }
        lap("B rows");
        struct SlabLists {
            std::vector<int> row_ptr, node_ptr, ent_row;
            std::vector<uint32_t> ent_node, node_id;
            std::vector<double> ent_coef, nent_coef;
        };
        std::vector<SlabLists> lists(slabs.size());
        for (size_t si = 0; si < slabs.size(); si++) {
            Slab<T>& sl = slabs[si];
            SlabLists& L = lists[si];
            const int64_t lo = (int64_t)sl.k0 * (int64_t)plane, hi = (int64_t)sl.k1 * (int64_t)plane;
            const int64_t shiftoff = (int64_t)plane - lo;  // global node -> local ghost-layout index
            L.row_ptr.assign(m + 1, 0);
            std::vector<std::pair<uint32_t, std::pair<int, double>>> by_node;
            for (int r = 0; r < m; r++) {
                for (int e = 0; e < 8; e++) {
                    const int64_t g = rows[r].nodes[e];
                    if (g < lo || g >= hi) continue;
                    const uint32_t l = (uint32_t)(g + shiftoff);
                    L.ent_node.push_back(l);
                    L.ent_coef.push_back(rows[r].coeffs[e]);
                    by_node.push_back({l, {r, rows[r].coeffs[e]}});
                }
                L.row_ptr[r + 1] = (int)L.ent_node.size();
            }
            std::stable_sort(by_node.begin(), by_node.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            for (size_t a = 0; a < by_node.size(); a++) {
                if (a == 0 || by_node[a].first != by_node[a - 1].first) {
                    L.node_id.push_back(by_node[a].first);
                    L.node_ptr.push_back((int)a);
                }
                L.ent_row.push_back(by_node[a].second.first);
                L.nent_coef.push_back(by_node[a].second.second);
            }
            L.node_ptr.push_back((int)by_node.size());
            sl.n_touched = (int)L.node_id.size();
        }
        lap("per-slab lists");
        // ---- uploads (queued behind the inversion on the set-up stream)
        build_shift_items(stream);
        for (size_t si = 0; si < slabs.size(); si++) {
            Slab<T>& sl = slabs[si];
            SlabLists& L = lists[si];
            sl.row_ptr.upload(L.row_ptr, stream);
            sl.ent_node.upload(L.ent_node, stream);
            sl.ent_coef.upload(L.ent_coef, stream);
            sl.node_id.upload(L.node_id, stream);
            sl.node_ptr.upload(L.node_ptr, stream);
            sl.ent_row.upload(L.ent_row, stream);
            sl.nent_coef.upload(L.nent_coef, stream);
            sl.red.alloc((size_t)m + 1);
            sl.u.alloc((size_t)std::max(m, 1));
            sl.dv.alloc((size_t)7 * std::max(mp, 64));
            sl.touched_save.alloc((size_t)std::max(sl.n_touched, 1));
        }
        lap("slab uploads");
        if (total_slabs == 1 && fft_available() && need_node_tables) build_active_tiles(unode);   // (the sparse sweeps exist for the FFT transforms only)
        lap("active tiles");
        Bptr.upload(bptr, stream);
        Bcol.upload(bcol, stream);
        Bval.upload(bval, stream);
        have_B = !dual_direct;
        lap("B uploaded");
        upload_red_tables(stream);
        lap("B, reduction tables uploaded");
        last_host_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
        log("[shm] constraint set-up: host part %.2f ms (m=%d)", last_host_setup_ms, m);
        finish_invert_G();  // synchronises the set-up stream (d_tidx/d_tval go out of scope after it)
        last_setup_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
        have_constraints = true;
    }

    // Two-level split of G (shm_twolevel.hip.h): boxes of `box`^3 cells, separator = cells with a coordinate that is a multiple of `box`.
    // Returns false (caller falls back to the dense inverse) when a box would not fit the kernels' LDS staging even at box = 8.
    bool build_two_level(const std::vector<int>& gptr, const std::vector<int>& gcol, const std::vector<double>& gval, DevArray<uint64_t>& d_tidx,
                         DevArray<double>& d_tval) {
        hipStream_t stream = stream2;
        static const int box_env = knob("SHM_TL_BOX") ? atoi(knob("SHM_TL_BOX")) : 0;
        const int64_t nn = n, pl = (int64_t)n * n;
        std::vector<int> ci((size_t)m), cj((size_t)m), ck((size_t)m);
        for (int r = 0; r < m; r++) {
            const int64_t c = rows[(size_t)r].nodes[0];
            ck[(size_t)r] = (int)(c / pl);
            cj[(size_t)r] = (int)((c - (int64_t)ck[(size_t)r] * pl) / nn);
            ci[(size_t)r] = (int)(c - (int64_t)ck[(size_t)r] * pl - (int64_t)cj[(size_t)r] * nn);
        }
        std::vector<int> boxid, slot, ptrI, ptrS, rowsI, colsS, sepRow, colour_of;
        std::vector<size_t> offD, offE;
        int P = 0, nS = 0, tl_maxs = 0, tl_maxc = 0;
        for (int b : {box_env > 1 ? box_env : 16, 8, 4}) {
            tl.box = b;
            // box key -> compact id in order of first appearance (deterministic)
            std::unordered_map<uint64_t, int> ids;
            boxid.assign((size_t)m, -1);
            slot.assign((size_t)m, -1);
            sepRow.clear();
            colour_of.clear();
            std::vector<int> cnt;
            for (int r = 0; r < m; r++) {
                const int i = ci[(size_t)r], j = cj[(size_t)r], k = ck[(size_t)r];
                if (i % b == 0 || j % b == 0 || k % b == 0) {
                    slot[(size_t)r] = (int)sepRow.size();
                    sepRow.push_back(r);
                    continue;
                }
                const uint64_t key = (uint64_t)(i / b) | ((uint64_t)(j / b) << 20) | ((uint64_t)(k / b) << 40);
                auto it = ids.find(key);
                int id;
                if (it == ids.end()) {
                    id = (int)ids.size();
                    ids.emplace(key, id);
                    cnt.push_back(0);
                    colour_of.push_back(((i / b) & 1) | (((j / b) & 1) << 1) | (((k / b) & 1) << 2));
                } else id = it->second;
                boxid[(size_t)r] = id;
                cnt[(size_t)id]++;
            }
            P = (int)cnt.size();
            nS = (int)sepRow.size();
            if (P == 0 || nS == 0) return false;
            ptrI.assign((size_t)P + 1, 0);
            for (int a = 0; a < P; a++) ptrI[(size_t)a + 1] = ptrI[(size_t)a] + cnt[(size_t)a];
            rowsI.assign((size_t)ptrI[(size_t)P], 0);
            std::vector<int> fill(ptrI.begin(), ptrI.end() - 1);
            for (int r = 0; r < m; r++)
                if (boxid[(size_t)r] >= 0) {
                    slot[(size_t)r] = fill[(size_t)boxid[(size_t)r]] - ptrI[(size_t)boxid[(size_t)r]];  // local index inside the box
                    rowsI[(size_t)fill[(size_t)boxid[(size_t)r]]++] = r;
                }
            // separator columns of every box, in order of first appearance along its rows
            ptrS.assign((size_t)P + 1, 0);
            colsS.clear();
            std::vector<int> mark((size_t)nS, -1);
            int maxs = 0, maxc = 0;
            for (int a = 0; a < P; a++) {
                for (int t = ptrI[(size_t)a]; t < ptrI[(size_t)a + 1]; t++) {
                    const int r = rowsI[(size_t)t];
                    for (int e = gptr[(size_t)r]; e < gptr[(size_t)r + 1]; e++) {
                        const int c = gcol[(size_t)e];
                        if (boxid[(size_t)c] >= 0) continue;
                        if (mark[(size_t)slot[(size_t)c]] != a) {
                            mark[(size_t)slot[(size_t)c]] = a;
                            colsS.push_back(slot[(size_t)c]);
                        }
                    }
                }
                ptrS[(size_t)a + 1] = (int)colsS.size();
                maxs = std::max(maxs, cnt[(size_t)a]);
                maxc = std::max(maxc, ptrS[(size_t)a + 1] - ptrS[(size_t)a]);
            }
            tl_maxs = maxs;
            tl_maxc = maxc;
            if (maxs <= kTlMaxBox && maxc <= kTlMaxBox) break;
            if (b == 4) return false;
        }
        tl.P = P;
        tl.nS = nS;
        tl.nI = (int)rowsI.size();
        tl.nSp = ((nS + kGJ - 1) / kGJ) * kGJ;
        tl.ysz = (int)colsS.size();
        // dense blocks D_a, E_a and the separator block F (as triplets of the padded Schur matrix)
        offD.assign((size_t)P, 0);
        offE.assign((size_t)P, 0);
        size_t szD = 0, szE = 0;
        for (int a = 0; a < P; a++) {
            const size_t sa = (size_t)(ptrI[(size_t)a + 1] - ptrI[(size_t)a]), ca = (size_t)(ptrS[(size_t)a + 1] - ptrS[(size_t)a]);
            offD[(size_t)a] = szD;
            offE[(size_t)a] = szE;
            szD += (size_t)tl_ld((int)sa) * (size_t)tl_ld((int)sa);   // D_a padded to whole 64-row blocks (the batched blocked Gauss-Jordan)
            szE += sa * ca;
        }
        tl.szD = szD;
        tl.szE = szE;
        std::vector<double> hD(szD, 0.), hE(std::max<size_t>(szE, 1), 0.);
        std::vector<uint64_t> tidx;
        std::vector<double> tval;
        std::vector<int> lcol((size_t)nS, -1);  // separator slot -> local column of the box being filled
        for (int a = 0; a < P; a++) {
            const int s0 = ptrI[(size_t)a], sa = ptrI[(size_t)a + 1] - s0, c0 = ptrS[(size_t)a], ca = ptrS[(size_t)a + 1] - c0;
            for (int l = 0; l < ca; l++) lcol[(size_t)colsS[(size_t)(c0 + l)]] = l;
            for (int t = sa; t < tl_ld(sa); t++) hD[offD[(size_t)a] + (size_t)t * (size_t)tl_ld(sa) + (size_t)t] = 1.0;   // identity on the padded diagonal
            for (int t = 0; t < sa; t++) {
                const int r = rowsI[(size_t)(s0 + t)];
                for (int e = gptr[(size_t)r]; e < gptr[(size_t)r + 1]; e++) {
                    const int c = gcol[(size_t)e];
                    if (boxid[(size_t)c] >= 0) hD[offD[(size_t)a] + (size_t)t * (size_t)tl_ld(sa) + (size_t)slot[(size_t)c]] = gval[(size_t)e];  // same box (interiors of different boxes never couple)
                    else hE[offE[(size_t)a] + (size_t)t * ca + (size_t)lcol[(size_t)slot[(size_t)c]]] = gval[(size_t)e];
                }
            }
        }
        for (int g = 0; g < nS; g++) {
            const int r = sepRow[(size_t)g];
            for (int e = gptr[(size_t)r]; e < gptr[(size_t)r + 1]; e++) {
                const int c = gcol[(size_t)e];
                if (boxid[(size_t)c] >= 0) continue;
                tidx.push_back((uint64_t)g * (uint64_t)tl.nSp + (uint64_t)slot[(size_t)c]);
                tval.push_back(gval[(size_t)e]);
            }
        }
        for (int g = nS; g < tl.nSp; g++) {
            tidx.push_back((uint64_t)g * tl.nSp + g);
            tval.push_back(1.0);
        }
        // per separator row: the y-buffer slots of the boxes that border it, ascending (fixed summation order)
        std::vector<int> adj_ptr((size_t)nS + 1, 0), adj_idx(colsS.size());
        for (int v : colsS) adj_ptr[(size_t)v + 1]++;
        for (int g = 0; g < nS; g++) adj_ptr[(size_t)g + 1] += adj_ptr[(size_t)g];
        {
            std::vector<int> fillp(adj_ptr.begin(), adj_ptr.end() - 1);
            for (int y = 0; y < (int)colsS.size(); y++) adj_idx[(size_t)fillp[(size_t)colsS[(size_t)y]]++] = y;
        }
        // boxes by colour (parity of the box coordinates): boxes of one colour border disjoint separator rows
        std::vector<int> clist;
        for (int col = 0; col < 8; col++) {
            tl.colour_ptr[col] = (int)clist.size();
            for (int a = 0; a < P; a++)
                if (colour_of[(size_t)a] == col) clist.push_back(a);
        }
        tl.colour_ptr[8] = (int)clist.size();
        // ---- device: upload, invert the boxes, Schur complement into Ginv
        tl.ptrI.upload(ptrI, stream);
        tl.ptrS.upload(ptrS, stream);
        tl.rowsI.upload(rowsI, stream);
        {   // row -> box map and the (box, 64-column chunk) list of the row- / column-parallel application kernels
            std::vector<int> rowBox(rowsI.size()), chunkBox, chunkCol;
            for (int a = 0; a < P; a++) {
                for (int t = ptrI[(size_t)a]; t < ptrI[(size_t)a + 1]; t++) rowBox[(size_t)t] = a;
                for (int l0 = 0; l0 < ptrS[(size_t)a + 1] - ptrS[(size_t)a]; l0 += kWave) {
                    chunkBox.push_back(a);
                    chunkCol.push_back(l0);
                }
            }
            tl.nChunks = (int)chunkBox.size();
            tl.rowBox.upload(rowBox, stream);
            tl.chunkBox.upload(chunkBox, stream);
            tl.chunkCol.upload(chunkCol, stream);
            // set-up lists: (box, 16 rows of T) for all boxes; (box, 16 rows of the Schur update) per colour; offsets of the boxes' Gauss-Jordan panels
            std::vector<int> tBox, tRow, sBox, sRow;
            std::vector<size_t> offW((size_t)P, 0);
            size_t szW = 0;
            tl.nbMax = 0;
            for (int a = 0; a < P; a++) {
                const int sa = ptrI[(size_t)a + 1] - ptrI[(size_t)a];
                for (int r0 = 0; r0 < sa; r0 += kTlRowsPerWg) {
                    tBox.push_back(a);
                    tRow.push_back(r0);
                }
                offW[(size_t)a] = szW;
                szW += (size_t)kGJ * (size_t)tl_ld(sa);
                tl.nbMax = std::max(tl.nbMax, tl_ld(sa) / kGJ);
            }
            tl.nTChunks = (int)tBox.size();
            for (int col = 0; col < 8; col++) {
                tl.schur_ptr[col] = (int)sBox.size();
                for (int a = 0; a < P; a++) {
                    if (colour_of[(size_t)a] != col) continue;
                    for (int p0 = 0; p0 < ptrS[(size_t)a + 1] - ptrS[(size_t)a]; p0 += kTlRowsPerWg) {
                        sBox.push_back(a);
                        sRow.push_back(p0);
                    }
                }
            }
            tl.schur_ptr[8] = (int)sBox.size();
            tl.tBox.upload(tBox, stream);
            tl.tRow.upload(tRow, stream);
            tl.sBox.upload(sBox, stream);
            tl.sRow.upload(sRow, stream);
            tl.offW.upload(offW, stream);
            tl.gjP.alloc((size_t)P * kGJ * kGJ);
            tl.gjR.alloc(std::max<size_t>(szW, 1));
            tl.gjC.alloc(std::max<size_t>(szW, 1));
        }
        tl.colsS.upload(colsS, stream);
        tl.sepRow.upload(sepRow, stream);
        tl.adj_ptr.upload(adj_ptr, stream);
        tl.adj_idx.upload(adj_idx, stream);
        tl.colour_list.upload(clist, stream);
        tl.offD.upload(offD, stream);
        tl.offE.upload(offE, stream);
        tl.D.upload(hD, stream);
        tl.E.upload(hE, stream);
        tl.Tm.alloc(std::max<size_t>(szE, 1));
        tl.tbuf.alloc((size_t)tl.nI);
        tl.ybuf.alloc(std::max<size_t>(colsS.size(), 1));
        tl.vS.alloc((size_t)tl.nSp);
        tl.uS.alloc((size_t)tl.nSp);
        HIPCHK(hipMemsetAsync(tl.vS.p, 0, (size_t)tl.nSp * sizeof(double), stream));
        ginv_rows = nS;
        ginv_ld = tl.nSp;
        Ginv.alloc((size_t)tl.nSp * tl.nSp);
        HIPCHK(hipMemsetAsync(Ginv.p, 0, (size_t)tl.nSp * tl.nSp * sizeof(double), stream));
        d_tidx.upload(tidx, stream);
        d_tval.upload(tval, stream);
        hipLaunchKernelGGL(scatter_triplets_kernel, dim3(grid_for(tidx.size(), 4096)), dim3(kBlock), 0, stream, (size_t)tidx.size(), d_tidx.p, d_tval.p, Ginv.p);
        gjFlag.alloc(2);   // [0]: blocked Gauss-Jordan (enqueue_gj_invert), [1]: the boxes' inverses; both read by finish_invert_G()
        HIPCHK(hipMemsetAsync(gjFlag.p + 1, 0, sizeof(int), stream));
        {   // all boxes' D_a^-1 at once: the blocked Gauss-Jordan of enqueue_gj_invert(), batched over the boxes (blockIdx.y)
            const GjBatch Bt{tl.ptrI.p, tl.offD.p, tl.offW.p, tl.D.p, tl.gjP.p, tl.gjR.p, tl.gjC.p};
            const unsigned nb = (unsigned)tl.nbMax, uP = (unsigned)P;
            for (unsigned kb = 0; kb < nb; kb++) {
                hipLaunchKernelGGL(gj_pivot_kernel<4>, dim3(1, uP), dim3(256), 0, stream, (double*)nullptr, 0, (int)kb, (double*)nullptr, gjFlag.p + 1, setup_prio, Bt);
                hipLaunchKernelGGL(gj_panels_kernel, dim3(nb, uP), dim3(kBlock), 0, stream, (const double*)nullptr, 0, (int)kb, (const double*)nullptr, (double*)nullptr, 0,
                                   (double*)nullptr, kGJ, 0, setup_prio, Bt);
                hipLaunchKernelGGL((gj_update_kernel<GJ_ALL>), dim3(nb * (nb + 1) / 2, uP), dim3(kBlock), 0, stream, (double*)nullptr, 0, 0, (int)kb, 0, 1,
                                   (const double*)nullptr, 0, (const double*)nullptr, kGJ, 0, kGJ, setup_prio, Bt);
            }
            if (nb > 1) hipLaunchKernelGGL(gj_mirror_kernel, dim3(nb * (nb - 1) / 2, uP), dim3(kBlock), 0, stream, (double*)nullptr, 0, Bt);
        }
        if (tl.nTChunks > 0)
            hipLaunchKernelGGL(tl_T_kernel, dim3((unsigned)tl.nTChunks), dim3(kBlock), 0, stream, tl.view(), tl.tBox.p, tl.tRow.p, tl.D.p, tl.E.p, tl.Tm.p, setup_prio);
        for (int col = 0; col < 8; col++) {
            const int cntc = tl.schur_ptr[col + 1] - tl.schur_ptr[col];
            if (cntc > 0)
                hipLaunchKernelGGL(tl_schur_kernel, dim3((unsigned)cntc), dim3(kBlock), 0, stream, tl.view(), tl.sBox.p + tl.schur_ptr[col], tl.sRow.p + tl.schur_ptr[col],
                                   tl.E.p, tl.Tm.p, Ginv.p, tl.nSp, setup_prio);
        }
        // fp32 copies for the dual preconditioner
        tl.D32.alloc(std::max<size_t>(szD, 1));
        tl.E32.alloc(std::max<size_t>(szE, 1));
        tl.T32.alloc(std::max<size_t>(szE, 1));
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid_for(szD, 4096)), dim3(kBlock), 0, stream, szD, tl.D.p, tl.D32.p);
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid_for(szE, 4096)), dim3(kBlock), 0, stream, szE, tl.E.p, tl.E32.p);
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid_for(szE, 4096)), dim3(kBlock), 0, stream, szE, tl.Tm.p, tl.T32.p);
        HIPCHK(hipGetLastError());
        // (no synchronisation here: the uploads above were staged, and a non-positive pivot of a box is reported by finish_invert_G() with the others --
        // waiting for the set-up stream at this point stalls the rest of the host set-up behind a Step-1 kernel that leaves it no SIMD)
        log("[shm] two-level inverse of A A^T: box %d, %d boxes (%d interior rows, largest box %d rows x %d separator columns), separator %d rows", tl.box, P, tl.nI, tl_maxs, tl_maxc, nS);
        return true;
    }

    // u = (A A^T)^-1 w on `st`: dense mat-vec, or the four launches of the two-level inverse.  f32: the single-precision copies (dual
    // preconditioner); otherwise double (projector).
    void apply_Ginv(const double* w, double* u, bool f32, hipStream_t st) {
        if (m <= 0) return;
        if (!tl.on) {
            if (f32) launch_ginv_matvec<float>(st, m, m, mp, Ginv32.p, w, u);
            else launch_ginv_matvec<double>(st, m, m, mp, Ginv.p, w, u);
            return;
        }
        const TlBoxes V = tl.view();
        const unsigned grows = (unsigned)((tl.nI + kBlock / kWave - 1) / (kBlock / kWave));
        // Round 5 measured two shorter chains against the five launches below (rocker.obj 512^3, plain stencil CG, one box, profiles/r05_projection.txt):
        //   [t | y = T^T w | v_S by the last workgroup to arrive], S^-1, finish:  0.081 -> 0.266 ms per projection (the device-scope fence every workgroup needs before
        //                                                                          it takes its ticket writes the L2 back each time);
        //   [t | y = T^T w], S^-1 mat-vec gathering v_S itself (8 rows per workgroup), finish (SHM_TL_MERGED3=1):  0.079 -> 0.095 ms, loop 0.693 -> 0.680 of the roofline.
        // The launches are short because they are narrow, not because they are many: fewer of them did not pay.  Five it stays.
        static const bool merged3 = knob("SHM_TL_MERGED3") != nullptr;
        if (merged3 && (size_t)tl.nSp * sizeof(double) <= 60 * 1024) {
            // three launches: [t = D^-1 w_I | y = T^T w_I], u_S = S^-1 (w_S - gathered y), u_I = t - T u_S
            const unsigned g1 = grows + (unsigned)tl.nChunks;
            if (f32)
                hipLaunchKernelGGL((tl_rows_cols_kernel<float>), dim3(g1), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, (int)grows, tl.D32.p, tl.chunkBox.p, tl.chunkCol.p, tl.T32.p, w,
                                   tl.tbuf.p, tl.ybuf.p);
            else
                hipLaunchKernelGGL((tl_rows_cols_kernel<double>), dim3(g1), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, (int)grows, tl.D.p, tl.chunkBox.p, tl.chunkCol.p, tl.Tm.p, w,
                                   tl.tbuf.p, tl.ybuf.p);
            const unsigned g2 = (unsigned)((tl.nS + kTlSepRows - 1) / kTlSepRows);
            const size_t lds = (size_t)tl.nSp * sizeof(double);
            if (f32)
                hipLaunchKernelGGL((tl_sep_matvec_kernel<float>), dim3(g2), dim3(kBlock), lds, st, tl.nS, tl.nSp, Ginv32.p, tl.sepRow.p, tl.adj_ptr.p, tl.adj_idx.p, w, tl.ybuf.p, tl.uS.p);
            else
                hipLaunchKernelGGL((tl_sep_matvec_kernel<double>), dim3(g2), dim3(kBlock), lds, st, tl.nS, tl.nSp, Ginv.p, tl.sepRow.p, tl.adj_ptr.p, tl.adj_idx.p, w, tl.ybuf.p, tl.uS.p);
            const unsigned gfin = grows + (unsigned)((tl.nS + kBlock - 1) / kBlock);
            if (f32) hipLaunchKernelGGL((tl_finish_kernel<float>), dim3(gfin), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.nS, tl.sepRow.p, tl.T32.p, tl.tbuf.p, tl.uS.p, u);
            else hipLaunchKernelGGL((tl_finish_kernel<double>), dim3(gfin), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.nS, tl.sepRow.p, tl.Tm.p, tl.tbuf.p, tl.uS.p, u);
            return;
        }
        if (f32) hipLaunchKernelGGL((tl_rows_kernel<float>), dim3(grows), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.D32.p, w, tl.tbuf.p);
        else hipLaunchKernelGGL((tl_rows_kernel<double>), dim3(grows), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.D.p, w, tl.tbuf.p);
        if (tl.nChunks > 0) {
            if (f32) hipLaunchKernelGGL((tl_cols_kernel<float>), dim3((unsigned)tl.nChunks), dim3(kBlock), 0, st, V, tl.chunkBox.p, tl.chunkCol.p, tl.E32.p, tl.tbuf.p, tl.ybuf.p);
            else hipLaunchKernelGGL((tl_cols_kernel<double>), dim3((unsigned)tl.nChunks), dim3(kBlock), 0, st, V, tl.chunkBox.p, tl.chunkCol.p, tl.E.p, tl.tbuf.p, tl.ybuf.p);
        }
        hipLaunchKernelGGL(tl_gather_sep_kernel, dim3((unsigned)((tl.nS + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, tl.nS, tl.sepRow.p, tl.adj_ptr.p, tl.adj_idx.p, w,
                           tl.ybuf.p, tl.vS.p);
        if (f32) launch_ginv_matvec<float>(st, tl.nS, tl.nS, tl.nSp, Ginv32.p, tl.vS.p, tl.uS.p);
        else launch_ginv_matvec<double>(st, tl.nS, tl.nS, tl.nSp, Ginv.p, tl.vS.p, tl.uS.p);
        const unsigned gfin = grows + (unsigned)((tl.nS + kBlock - 1) / kBlock);
        if (f32) hipLaunchKernelGGL((tl_finish_kernel<float>), dim3(gfin), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.nS, tl.sepRow.p, tl.T32.p, tl.tbuf.p, tl.uS.p, u);
        else hipLaunchKernelGGL((tl_finish_kernel<double>), dim3(gfin), dim3(kBlock), 0, st, V, tl.rowBox.p, tl.nI, tl.nS, tl.sepRow.p, tl.Tm.p, tl.tbuf.p, tl.uS.p, u);
    }

    // in-place inverse of the SPD matrix M (mp x mp, mp a multiple of 64, identity tail) on the set-up stream: blocked Gauss-Jordan (shm_kernels.hip.h)
    // `refined_later`: the caller checks the inverse by a residual and refines (the direct dual solve with S^-1): the pivot blocks may then be inverted by the
    // 16-step block kernel, whose result differs from the scalar 64-step one at the 1e-11 level on ill-conditioned matrices (A A^T: near-dependent rows of
    // neighbouring cells; the projector test holds A P v to 1e-11 without refinement, so G keeps the scalar kernel)
    // SHM_SETUP_SKIP (experiment knob, bit mask): leave out the set-up's device work piece by piece -- 2: the assembly of S, 4: the Gauss-Jordan launches (1 = the Green's
    // table: SHM_SCHUR_KEEP_TABLE).  The solve's RESULT is then wrong; only Step 1's duration beside what is left means anything (tools/ab.py prints it).
    static int setup_skip() {
        static const int v = knob("SHM_SETUP_SKIP") ? atoi(knob("SHM_SETUP_SKIP")) : 0;
        return v;
    }
    void enqueue_gj_invert(double* M, int mp, bool refined_later = false) {
        hipStream_t stream = stream2;
        const int nb = mp / kGJ;
        static const int outer_env = knob("SHM_GJ_OUTER") ? atoi(knob("SHM_GJ_OUTER")) : 0;   // experiment knob: pivot blocks per outer block
        const int outer = outer_env > 0 ? std::min(outer_env, 8) : (nb >= 64 ? 4 : 1);
        gjP.alloc(kGJ * kGJ);
        gjR.alloc((size_t)outer * kGJ * mp);   // [outer * 64][mp]
        gjC.alloc((size_t)mp * outer * kGJ);   // [mp][outer * 64]
        gjFlag.alloc(2);
        HIPCHK(hipMemsetAsync(gjFlag.p, 0, sizeof(int), stream));
        if (setup_skip() & 4) return;   // (timing experiments: see setup_skip)
        const int c_ld = outer * kGJ;
        static const int pivot_env = knob("SHM_GJ_PIVOT_E") ? atoi(knob("SHM_GJ_PIVOT_E")) : 0;
        const bool classic = knob("SHM_GJ_CLASSIC") != nullptr;   // A/B knob, read per inversion (a test flips it inside one process): three dependent launches per pivot block (rounds 1-3)
        // one launch per pivot block (gj_step_kernel; it inverts the pivot tiles by the scalar elimination) unless a pivot kernel is asked for explicitly
        const bool stepped = outer == 1 && !classic && pivot_env == 0;
        const int pivot_e = pivot_env ? pivot_env : (refined_later ? 16 : 4);   // 16 = block Gauss-Jordan with 4 x 4 pivot blocks (round 3);   // A/B knob: 4 = 256 threads; 2 = 1024 threads (2 % faster on an idle GPU, but four 40-register waves per SIMD do not fit beside Step 1)
        auto launch_pivot = [&](int kb) {
            if (pivot_e == 16) hipLaunchKernelGGL(gj_pivot_block4_kernel, dim3(1), dim3(256), 0, stream, M, mp, kb, gjP.p, gjFlag.p, setup_prio);
            else if (pivot_e == 4) hipLaunchKernelGGL(gj_pivot_kernel<4>, dim3(1), dim3(256), 0, stream, M, mp, kb, gjP.p, gjFlag.p, setup_prio);
            else hipLaunchKernelGGL(gj_pivot_kernel<2>, dim3(1), dim3(1024), 0, stream, M, mp, kb, gjP.p, gjFlag.p, setup_prio);
        };
        if (stepped) {
            // one launch per pivot block: step k-1's update beside step k's pivot and panels (gj_step_kernel)
            gjR.alloc((size_t)2 * kGJ * mp);
            gjC.alloc((size_t)2 * mp * kGJ);
            for (int k = 0; k <= nb; k++) {
                const size_t nU = k == 0 ? 0 : (k < nb ? (size_t)(nb - 1) * nb / 2 : (size_t)nb * (nb + 1) / 2);
                const unsigned grid = (unsigned)((k < nb ? nb : 0) + nU);
                double* Rn = gjR.p + (size_t)(k & 1) * kGJ * mp;
                double* Cn = gjC.p + (size_t)(k & 1) * mp * kGJ;
                const double* Rp = gjR.p + (size_t)((k + 1) & 1) * kGJ * mp;
                const double* Cp = gjC.p + (size_t)((k + 1) & 1) * mp * kGJ;
                hipLaunchKernelGGL(gj_step_kernel, dim3(grid), dim3(kBlock), 0, stream, M, mp, nb, k, Rp, Cp, Rn, Cn, gjFlag.p, setup_prio);
            }
        } else if (outer == 1) {
            for (int kb = 0; kb < nb; kb++) {
                launch_pivot(kb);
                hipLaunchKernelGGL(gj_panels_kernel, dim3(nb), dim3(kBlock), 0, stream, M, mp, kb, gjP.p, gjR.p, 0, gjC.p, c_ld, 0, setup_prio);
                hipLaunchKernelGGL((gj_update_kernel<GJ_ALL>), dim3((unsigned)((size_t)nb * (nb + 1) / 2)), dim3(kBlock), 0, stream, M, mp, nb, kb, 0, 1,
                                   gjR.p, 0, gjC.p, c_ld, 0, kGJ, setup_prio);
            }
        } else {
            for (int k0 = 0; k0 < nb; k0 += outer) {
                const int nO = std::min(outer, nb - k0);
                for (int t = 0; t < nO; t++) {
                    const int kb = k0 + t;
                    launch_pivot(kb);
                    hipLaunchKernelGGL(gj_panels_kernel, dim3(nb), dim3(kBlock), 0, stream, M, mp, kb, gjP.p, gjR.p, t * kGJ, gjC.p, c_ld, t * kGJ, setup_prio);
                    hipLaunchKernelGGL((gj_update_kernel<GJ_CROSS>), dim3((unsigned)(nO * nb + nO * k0)), dim3(kBlock), 0, stream, M, mp, nb, kb, k0, nO,
                                       gjR.p, t * kGJ, gjC.p, c_ld, t * kGJ, kGJ, setup_prio);
                }
                const size_t nr = (size_t)(nb - nO);
                if (nr > 0)
                    hipLaunchKernelGGL((gj_update_kernel<GJ_REST>), dim3((unsigned)(nr * (nr + 1) / 2)), dim3(kBlock), 0, stream, M, mp, nb, -1, k0, nO,
                                       gjR.p, 0, gjC.p, c_ld, 0, nO * kGJ, setup_prio);
            }
        }
        if (nb > 1) hipLaunchKernelGGL(gj_mirror_kernel, dim3((unsigned)((size_t)nb * (nb - 1) / 2)), dim3(kBlock), 0, stream, M, mp);
        HIPCHK(hipGetLastError());
    }
    void enqueue_invert_G() {
        const int mp = ginv_ld;  // dense inverse: the padded m; two-level: the padded separator size
        enqueue_gj_invert(Ginv.p, mp);
        Ginv32.alloc((size_t)mp * mp);
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid_for((size_t)mp * mp, 4096)), dim3(kBlock), 0, stream2, (size_t)mp * mp, Ginv.p, Ginv32.p);
        HIPCHK(hipGetLastError());
    }
    // S = A K^+ A^T explicitly (shm_schur.hip.h), on the set-up stream.  For one slab, a DCT-sized grid and a moderate number of rows.
    //   prepare_schur(): host tables and their uploads -- BEFORE the inversion of G is queued (a pageable copy waits for everything queued on its stream)
    //   enqueue_schur(): the launches -- AFTER it (the inversion is a chain of short dependent kernels; these few throughput-bound ones fill the slots it
    //                    leaves idle)
    // The Green's table T depends on the grid alone (n, h) -- it is to K what the reference's poissonSolver factorisation is to L, built when the grid is
    // built (signed_heat_grid_solver.cpp:8-35, `rebuild`) -- but it is cheap enough to be rebuilt with every solve (see prepare_schur); S depends on the
    // sources and is assembled per solve.
    bool schur_wanted() const {
        const bool off = knob("SHM_DUAL_NO_DENSE_S") != nullptr || dual_form_req == SHM_DUAL_THROUGH_GRID;   // apply S through the grid (five sparse sweeps)
        // (beyond ~8000 rows the assembly costs Step 1 more time than the dense mat-vec saves the CG: rocker 512^3 fp32, m = 12 612: 29 ms of assembly for
        // 36 x 0.37 ms -- 493-508 against 491-501 ms per solve with S applied through the grid)
        // (beside the tiered fp64 Step 1 -- where the assembly is co-resident and hidden -- up to 16384 rows since round 4: rocker 512^3 fp64, m = 12 612: solve phase
        // 71.5 -> 35.9 ms, 631 -> 609 ms per solve; after the fp32 Step 1, which leaves the set-up's kernels no room, the same choice costs 401 -> 423 ms)
        static const int max_m_env = knob("SHM_DENSE_S_MAX_M") ? atoi(knob("SHM_DENSE_S_MAX_M")) : 0;
        // (an explicit shm_opts.dual_form request gets the documented 16384 rows whatever Step-1 kernel runs)
        const bool asked = dual_form_req == SHM_DUAL_DIRECT || dual_form_req == SHM_DUAL_EXPLICIT_S_CG;
        const int max_m = max_m_env > 0 ? max_m_env : ((conv_tiered || conv_tiered32 || asked) ? 16384 : 8192);
        // n not a power of two: applying S through the grid costs six dense products per CG iteration (shm_dct_gemm.hip.h: 5 ms at n = 362), so the explicit S
        // is worth its assembly up to the sizes its memory allows, whatever Step 1 hides
        if (gemm_dct()) return !off && m > 0 && m <= std::max(max_m, 16384);
        // Round 6 -- several z-slabs (ranks): S and its inverse are REPLICATED (assembled from the grid's Green's table and the global rows on every rank, beside that
        // rank's Step 1 like the rest of the set-up), and the solve touches the grid twice -- K^+ b and K^+ (A^T mu - b) on the slabs, two all-to-alls each --
        // with the whole dual system solved on m-vectors that every rank holds: no gather of D^T Y, no whole-grid solve per rank (solve_dual).  Wherever it fits.
        if (total_slabs > 1) return !off && fft_available() && m > 0 && m <= 16384 && n <= 512;
        if (off || total_slabs != 1 || !precond_available() || m <= 0 || m > max_m || n > 512) return false;
        // the assembly (216 table entries per matrix entry: ~2.2e-7 ms per m^2 on an idle device until round 4 -- 1.8 ms at m = 2842, 29 ms at m = 12 612; 1.1 ms at m = 2842 since the
        // windows along the last axis are fetched by two loads instead of three; the estimate below keeps the conservative constant) has to hide behind
        // this rank's Step 1 like the rest of the set-up; where Step 1 is short (<= 128^3, or a thin slab of a multi-GPU run) the sweeps through the grid are
        // cheap anyway (0.11 ms per iteration at 128^3) and the set-up is the critical path already
        // (the direct dual solve replaces the inversion of G, the host's B rows and the whole iteration by the assembly and the inversion of S: a gain at
        // every size it applies to -- 128^3: 12.1 -> 9.6 ms, 64^3: 4.5 -> 2.7 ms per solve -- so it is not subject to this test)
        const bool force = knob("SHM_DUAL_DENSE_S_ALWAYS") != nullptr || dual_form_req == SHM_DUAL_EXPLICIT_S_CG || dual_form_req == SHM_DUAL_DIRECT;
        const double schur_est_ms = 2.2e-7 * (double)m * (double)m;
        if (force || dual_direct) return true;
        // CG on the explicit S (no inversion): worth it only where an iteration through the grid costs clearly more than the dense mat-vecs -- which grows with
        // m^2 while the sparse sweeps grow with the grid.  Measured, fp64 (round 4, tools/r04_dense_s_256.py; ms per iteration through the grid / on the explicit
        // S): 512^3 rocker (m = 12 612) 1.37 / 0.69 -> 631 / 610 ms per solve; 256^3 rocker (9 110) 0.31 / 0.38, SprayBottle (12 620) 0.27 / 0.65, chair (6 340)
        // 0.23 / 0.245 -> 97 / 104, 174 / 195, 58.3 / 62.5 ms per solve (the assembly also costs Step 1 4-14 ms of shared SIMD time); 128^3: a tie.  Rounds 2-3
        // chose the explicit S for every m <= 8192 that Step 1 could hide.
        const double dense_iter_ms = 3.6e-9 * (double)m * (double)m * (sizeof(T) / 8.0) + 0.08;
        const double grid_iter_ms = std::max(0.15, 1.4 * std::pow((double)n / 512.0, 3.0) * (sizeof(T) / 8.0));
        // ... and only beside the tiered fp64 Step 1: after the fp32 (or all-fp64) kernel, which leaves the set-up's kernels no room, the assembly is exposed
        // (chair 512^3 fp32, m = 7 748: 243 ms through the grid, 253 with the explicit S and 17 ms of wait; 256^3 fp32: 38.1 / 43.5)
        return (conv_tiered || conv_tiered32) && conv_est_total_ms >= 3.0 * schur_est_ms && grid_iter_ms > 1.3 * dense_iter_ms;
    }
    // The Green's table of the grid (depends on n and h alone) on `st`.  Round 4: when the number of sources already guarantees the direct dual solve (m <= S <=
    // its limit), build_constraints() queues this BEFORE the host builds the constraint rows, so the table's kernels (0.9 ms at 256^3, 10 ms at 512^3) run while
    // the host works (1 ms) instead of after it; otherwise prepare_schur() queues it once m is known.
    bool green_table_early_ok() const {
        static const bool off = knob("SHM_GREEN_LATE") != nullptr;   // A/B knob
        static const int direct_max_m = knob("SHM_DUAL_DIRECT_MAX_M") ? atoi(knob("SHM_DUAL_DIRECT_MAX_M")) : 4096;
        if (dual_form_req == SHM_DUAL_EXPLICIT_S_CG || dual_form_req == SHM_DUAL_THROUGH_GRID) return false;
        return !off && dual_direct_requested && knob("SHM_DUAL_NO_DIRECT") == nullptr && knob("SHM_DUAL_NO_DENSE_S") == nullptr && (total_slabs == 1 || fft_available()) && precond_available() &&
               !gemm_dct() && n <= 512 && S > 0 && S <= direct_max_m && (sizeof(T) == 8 || conv_tiered32 || n < 512 || knob("SHM_DUAL_DIRECT_ALWAYS") != nullptr || dual_form_req == SHM_DUAL_DIRECT);
    }
    void enqueue_green_table(hipStream_t st) {
        const int P = n + 8;   // leading dimension of the last table index (rows stay 64-byte aligned)
        const size_t n1 = (size_t)n + 1;
        static const bool keep_table = knob("SHM_SCHUR_KEEP_TABLE") != nullptr;
        if (!keep_table) gs_n = 0;
        if (gs_n == n && gs_cell == cell) return;
        {
            const double pi = 3.14159265358979323846;
            h_gs_lam.resize(n);
            h_gs_ctab.resize(2 * (size_t)n);
            for (int k = 0; k < n; k++) h_gs_lam[k] = (2. - 2. * std::cos(pi * k / n)) / (cell * cell);   // the transforms' eigenvalues (setup_precond)
            for (int r = 0; r < 2 * n; r++) h_gs_ctab[r] = std::cos(pi * r / n);
            gs_lam.upload(h_gs_lam, st);
            gs_ctab.upload(h_gs_ctab, st);
            gs_T.alloc(n1 * n1 * P);
        }
        {
            // scratch of the three contractions: W0 (the symbol) is dead once the first product has been formed, so the second product's output W2 reuses
            // its storage (two scratch arrays of ~n^3 doubles beside the table instead of three: 3.3 instead of 4.4 GB at 512^3, kept for the next solve)
            DevArray<double>&W0 = gs_W0, &W1 = gs_W1, &W2 = gs_W0;
            gs_Cm.alloc(n1 * n);
            gs_Ct.alloc((size_t)n * P);
            W0.alloc(std::max((size_t)n * n * n, (size_t)n * n1 * P));
            W1.alloc((size_t)n * n * P);
            // Round 5: n = 2^k -- the three cosine contractions as FFT passes (shm_green_fft.hip.h: O(n^3 log n), 0.8 GB of traffic at 256^3) instead of dense
            // products on the fp64 matrix cores (6 n^4 flop: 1.2 ms at 256^3, 10 ms at 512^3 -- machine time taken from the Step-1 kernel they run beside).
            // SHM_GREEN_GEMM=1: the products (A/B; they also serve every n that is not a power of two)
            const bool by_fft = (n & (n - 1)) == 0 && n >= 16 && n <= 512 && knob("SHM_GREEN_GEMM") == nullptr;
            if (by_fft) {
                const double pi = 3.14159265358979323846;
                h_gs_tw.resize(2 * (size_t)n);   // [0, n): e^{-2 pi i t / n} ; [n, 2n): e^{-i pi k / n}
                for (int t = 0; t < n; t++) {
                    h_gs_tw[(size_t)t] = {std::cos(2. * pi * t / n), -std::sin(2. * pi * t / n)};
                    h_gs_tw[(size_t)n + t] = {std::cos(pi * t / n), -std::sin(pi * t / n)};
                }
                gs_tw.upload(h_gs_tw, st);
                // (the padded columns d3 in [n + 1, n + 8) of W1 are transformed like the others by the second and third pass: keep them finite)
                HIPCHK(hipMemsetAsync(W1.p, 0, (size_t)n * n * P * sizeof(double), st));
                hipLaunchKernelGGL(green_symbol_kernel, dim3((unsigned)std::min(n * n, 8 * num_cus)), dim3(kBlock), 0, st, n, gs_lam.p, W0.p);
                const long long Pl = P, n1l = (long long)n1;
                auto pass = [&](bool xpass, int ntiles, int tiles_a, DctAddr in, DctAddr out, const double* src, double* dst) {
                    CosiParams C;
                    C.ntiles = ntiles;
                    C.tiles_a = tiles_a;
                    C.in = in;
                    C.out = out;
                    const dim3 g((unsigned)std::min(ntiles, 16 * num_cus));
                    const Cplx<double>*tw = gs_tw.p, *om = gs_tw.p + n;
                    int l2 = 0;
                    while ((1 << l2) < n) l2++;
#define SHM_COSI_CASE(L)                                                                                                                              \
    case L:                                                                                                                                            \
        if (xpass) hipLaunchKernelGGL((cosi_lines_kernel<L, true>), g, dim3(kBlock), 0, st, C, src, dst, tw, om, setup_prio);                         \
        else hipLaunchKernelGGL((cosi_lines_kernel<L, false>), g, dim3(kBlock), 0, st, C, src, dst, tw, om, setup_prio);                              \
        break;
                    switch (l2) {
                        SHM_COSI_CASE(4) SHM_COSI_CASE(5) SHM_COSI_CASE(6) SHM_COSI_CASE(7) SHM_COSI_CASE(8) SHM_COSI_CASE(9)
                        default: throw Error(SHM_ERR_INVALID, "Green's table by FFT: n out of range");
                    }
#undef SHM_COSI_CASE
                };
                auto addr = [](long long a_stride, long long b_stride, long long line_stride, long long elem_stride) {
                    DctAddr A;
                    A.off = 0; A.a_stride = a_stride; A.b_stride = b_stride; A.line_stride = line_stride; A.elem_stride = elem_stride;
                    A.seg_stride = 0; A.seg_shift = 30; A.seg_mask = 0x3fffffff;
                    return A;
                };
                const int L = kCosiL;
                // W1[(k1,k2)][d3] from W0[(k1,k2)][k3]: tiles of L consecutive rows
                pass(true, n * n / L, n * n / L, addr((long long)L * n, 0, n, 1), addr((long long)L * Pl, 0, Pl, 1), W0.p, W1.p);
                // W2[k1][d2][d3] from W1[k1][k2][d3]: per k1, tiles of L consecutive d3
                pass(false, n * (P / L), P / L, addr(L, (long long)n * Pl, 1, Pl), addr(L, n1l * Pl, 1, Pl), W1.p, W2.p);
                // T[d1][(d2,d3)] from W2[k1][(d2,d3)]: tiles of L consecutive (d2,d3)
                pass(false, (int)(n1l * Pl / L), (int)(n1l * Pl / L), addr(L, 0, 1, n1l * Pl), addr(L, 0, 1, n1l * Pl), W2.p, gs_T.p);
                HIPCHK(hipGetLastError());
                gs_n = n;
                gs_cell = cell;
                return;
            }
            HIPCHK(hipMemsetAsync(gs_Ct.p, 0, (size_t)n * P * sizeof(double), st));
            hipLaunchKernelGGL(cosine_tables_kernel, dim3(grid_for(n1 * n, 1024)), dim3(kBlock), 0, st, n, P, gs_ctab.p, gs_Cm.p, gs_Ct.p);
            hipLaunchKernelGGL(green_symbol_kernel, dim3((unsigned)std::min(n * n, 8 * num_cus)), dim3(kBlock), 0, st, n, gs_lam.p, W0.p);
            // beside the tiered fp64 Step 1 (two 184-register waves per SIMD) only the narrow shape fits on a SIMD; otherwise the 128 x 128 tiles
            static const bool gemm_wide_env = knob("SHM_GREEN_WIDE") != nullptr;   // A/B knob
            const bool narrow = (conv_tiered || conv_tiered32 || knob("SHM_GREEN_NARROW") != nullptr) && !gemm_wide_env;
            auto tiles = [](size_t v) { return (unsigned)((v + kGemmT - 1) / kGemmT); };
            auto gemm = [&](unsigned batches, int M, int N, int K, const double* A, int lda, long long sA, const double* B, int ldb, long long sB, double* C, int ldc, long long sC) {
                static const int wn_env = knob("SHM_GREEN_WN") ? atoi(knob("SHM_GREEN_WN")) : 0;   // A/B knob (round 5)
                if (narrow && wn_env == 2)
                    hipLaunchKernelGGL(dgemm_rm_kernel<2>, dim3((unsigned)((N + 63) / 64), tiles((size_t)M), batches), dim3(kBlock), 0, st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, setup_prio);
                else if (narrow)
                    hipLaunchKernelGGL(dgemm_rm_kernel<1>, dim3((unsigned)((N + 31) / 32), tiles((size_t)M), batches), dim3(kBlock), 0, st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, setup_prio);
                else
                    hipLaunchKernelGGL(dgemm_rm_kernel<4>, dim3(tiles((size_t)N), tiles((size_t)M), batches), dim3(kBlock), 0, st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, setup_prio);
            };
            // W1[(k1,k2)][d3] = sum_k3 W0[(k1,k2)][k3] Ct[k3][d3]
            gemm(1, n * n, P, n, W0.p, n, 0LL, gs_Ct.p, P, 0LL, W1.p, P, 0LL);
            // W2[k1][d2][d3] = sum_k2 Cm[d2][k2] W1[k1][k2][d3]   (one product per k1)
            gemm((unsigned)n, (int)n1, P, n, gs_Cm.p, n, 0LL, W1.p, P, (long long)n * P, W2.p, P, (long long)(n1 * P));
            // T[d1][(d2,d3)] = sum_k1 Cm[d1][k1] W2[k1][(d2,d3)]
            gemm(1, (int)n1, (int)(n1 * P), n, gs_Cm.p, n, 0LL, W2.p, (int)(n1 * P), 0LL, gs_T.p, (int)(n1 * P), 0LL);
            HIPCHK(hipGetLastError());
            gs_n = n;
            gs_cell = cell;
        }
    }
    void prepare_schur() {
        hipStream_t st = stream2;
        have_S = false;
        if (!schur_wanted()) return;
        // T is rebuilt with every solve (0.85 ms at 256^3, 10 ms at 512^3, beside Step 1: +1 % / +2.5 % of a solve), so that a timed solve contains all of
        // its own work; SHM_SCHUR_KEEP_TABLE=1 keeps it while n and h stay the same (it depends on nothing else)
        if (!gs_early) enqueue_green_table(st);
        // rows in Morton order of their cells: the 16 x 16 tiles of the assembly then read neighbouring table entries
        std::vector<std::pair<uint64_t, int>> key((size_t)m);
        auto spread = [](uint64_t v) {
            v &= 0x1fffff;
            v = (v | v << 32) & 0x1f00000000ffffULL;
            v = (v | v << 16) & 0x1f0000ff0000ffULL;
            v = (v | v << 8) & 0x100f00f00f00f00fULL;
            v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
            v = (v | v << 2) & 0x1249249249249249ULL;
            return v;
        };
        for (int r = 0; r < m; r++) key[(size_t)r] = {spread((uint64_t)rows[r].cell[0]) | spread((uint64_t)rows[r].cell[1]) << 1 | spread((uint64_t)rows[r].cell[2]) << 2, r};
        std::sort(key.begin(), key.end());
        h_rowX.resize(4 * (size_t)m);
        h_rowT.resize(3 * (size_t)m);
        for (int q = 0; q < m; q++) {
            const int r = key[(size_t)q].second;
            for (int a = 0; a < 3; a++) {
                h_rowX[4 * (size_t)q + a] = rows[r].cell[a];
                h_rowT[3 * (size_t)q + a] = rows[r].t[a];
            }
            h_rowX[4 * (size_t)q + 3] = r;   // the row this sorted slot stands for
        }
        d_rowX.upload(h_rowX, st);
        d_rowT.upload(h_rowT, st);
        Sdense.alloc((size_t)mp * mp);
        if (!stream3) {
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, greatest));
            e_sch_in.reset(new Event());
            e_sch_done.reset(new Event());
        }
        e_sch_in->record(st);   // the tables are on the device
    }
    void enqueue_schur() {
        if (!schur_wanted()) return;
        // its own stream: the inversion of G on stream2 is a chain of ~135 short launches that each wait for a slot next to Step 1 (39 of Step 1's 40 ms at
        // 256^3); queued behind it the assembly would start when Step 1 is almost over and be exposed, beside it it is done long before.  (Direct dual solve:
        // there is no inversion of G; S is assembled and then inverted itself, all on stream2.)
        hipStream_t st = dual_direct ? stream2 : stream3;
        HIPCHK(hipStreamWaitEvent(st, e_sch_in->e, 0));
        const int P = n + 8;
        if (gs_early) HIPCHK(hipStreamWaitEvent(st, e_gs_done->e, 0));
        HIPCHK(hipMemsetAsync(Sdense.p, 0, (size_t)mp * mp * sizeof(double), st));
        const unsigned mt = (unsigned)((m + 15) / 16);
        if (setup_skip() & 2) hipLaunchKernelGGL(set_diagonal_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, Sdense.p, mp, 0, m, 1.0);   // (something invertible)
        else hipLaunchKernelGGL(schur_assemble_kernel, dim3(mt, mt), dim3(kBlock), 0, st, m, mp, n, P, d_rowX.p, d_rowT.p, gs_T.p, Sdense.p, setup_prio);
        HIPCHK(hipGetLastError());
        if (dual_direct) {
            Sinv.alloc((size_t)mp * mp);
            HIPCHK(hipMemcpyAsync(Sinv.p, Sdense.p, (size_t)mp * mp * sizeof(double), hipMemcpyDeviceToDevice, st));
            if (mp > m) hipLaunchKernelGGL(set_diagonal_kernel, dim3((unsigned)((mp - m + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, Sinv.p, mp, m, mp, 1.0);   // identity tail
            enqueue_gj_invert(Sinv.p, mp, true);
            Sinv_ones.alloc((size_t)2 * mp);
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, Sinv_ones.p + mp, m, 1.0);
            launch_ginv_matvec<double>(st, m, m, mp, Sinv.p, Sinv_ones.p + mp, Sinv_ones.p);
        } else {
            e_sch_done->record(st);
            HIPCHK(hipStreamWaitEvent(stream2, e_sch_done->e, 0));   // "set-up done" on stream2 (what the solve waits for) now includes S
        }
        have_S = true;
    }
    void finish_invert_G() {
        hipStream_t stream = stream2;
        int flag[2] = {0, 0};
        if (!tl.on) HIPCHK(hipMemsetAsync(gjFlag.p + 1, 0, sizeof(int), stream));   // (slot 1 belongs to the two-level build)
        HIPCHK(hipMemcpyAsync(flag, gjFlag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (flag[0] || flag[1]) throw Error(SHM_ERR_SINGULAR, "A A^T is not positive definite (duplicate or degenerate constraint rows)");
    }

    // ------------------------------------------------------------------------------------------
    // transport: sum a per-slab reduction vector over all slabs of all processes
    void allreduce(int which /*0: red, 1: pq*/, int count) {
        auto buf = [&](Slab<T>& sl) { return which == 0 ? sl.red.p : sl.pq.p; };
        if (slabs.size() > 1) {
            double** slot = d_redptrs.p + (which ? slabs.size() : 0);  // table built in build_constraints()
            hipLaunchKernelGGL(sum_slabs_kernel, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, (int)slabs.size(), slot, count);
        }
        if (comm) {
            Rccl& R = Rccl::get();
            R.chk(R.AllReduce(buf(slabs[0]), buf(slabs[0]), (size_t)count, Rccl::kFloat64, Rccl::kSum, comm, stream), "ncclAllReduce");
            for (size_t s = 1; s < slabs.size(); s++)
                HIPCHK(hipMemcpyAsync(buf(slabs[s]), buf(slabs[0]), (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
    }

    // transport: fill the ghost planes of `field` (p) from the neighbouring slabs
    void halo_exchange_p() { halo_exchange(ARR_P); }
    void halo_exchange(int sel) { halo_exchange(sel, stream); }
    void halo_exchange(int sel, hipStream_t stream) {
        const size_t pb = slabs[0].plane * sizeof(T);
        for (size_t s = 0; s + 1 < slabs.size(); s++) {
            Slab<T>&a = slabs[s], &b = slabs[s + 1];
            HIPCHK(hipMemcpyAsync(arr(b, sel), arr(a, sel) + (size_t)a.nzl * a.plane, pb, hipMemcpyDeviceToDevice, stream));                // a top -> b low ghost
            HIPCHK(hipMemcpyAsync(arr(a, sel) + (size_t)(a.nzl + 1) * a.plane, arr(b, sel) + b.plane, pb, hipMemcpyDeviceToDevice, stream));  // b bottom -> a high ghost
        }
        if (comm) {
            Rccl& R = Rccl::get();
            const int dt = sizeof(T) == 8 ? Rccl::kFloat64 : Rccl::kFloat32;
            Slab<T>&lo = slabs.front(), &hi = slabs.back();
            const size_t cnt = lo.plane;
            R.chk(R.GroupStart(), "ncclGroupStart");
            if (cfg.rank > 0) {
                R.chk(R.Send(arr(lo, sel) + lo.plane, cnt, dt, cfg.rank - 1, comm, stream), "ncclSend(lo)");
                R.chk(R.Recv(arr(lo, sel), cnt, dt, cfg.rank - 1, comm, stream), "ncclRecv(lo)");
            }
            if (cfg.rank < cfg.world - 1) {
                R.chk(R.Send(arr(hi, sel) + (size_t)hi.nzl * hi.plane, cnt, dt, cfg.rank + 1, comm, stream), "ncclSend(hi)");
                R.chk(R.Recv(arr(hi, sel) + (size_t)(hi.nzl + 1) * hi.plane, cnt, dt, cfg.rank + 1, comm, stream), "ncclRecv(hi)");
            }
            R.chk(R.GroupEnd(), "ncclGroupEnd");
        }
    }

    // ------------------------------------------------------------------------------------------
    struct StencilLaunch {
        dim3 grid, block;
        int xchunks, yblocks, zc;
    };
    StencilLaunch stencil_dims(const Slab<T>& sl) const {
        const int per_row = (n + vec - 1) / vec;
        int tx = 1;
        while (tx < per_row && tx < kBlock) tx <<= 1;
        const int tyb = kBlock / tx;
        const int xchunks = (per_row + tx - 1) / tx;
        const int yblocks = (n + tyb * kStRY - 1) / (tyb * kStRY);
        // planes per workgroup: as many as possible (less z-halo re-reading) while the grid still has >= ~4 workgroups per CU
        static const int zc_env = knob("SHM_STENCIL_ZC") ? atoi(knob("SHM_STENCIL_ZC")) : 0;
        int zc = 16;
        while (zc > 4 && (long long)xchunks * yblocks * ((sl.nzl + zc - 1) / zc) < 1024) zc >>= 1;
        if (zc_env > 0) zc = zc_env;
        const int zchunks = (sl.nzl + zc - 1) / zc;
        return {dim3((unsigned)(xchunks * yblocks * zchunks)), dim3((unsigned)tx, (unsigned)tyb), xchunks, yblocks, zc};
    }

    void launch_stencil(Slab<T>& sl) {
        const StencilLaunch L = stencil_dims(sl);
        if (vec == 1)
            hipLaunchKernelGGL((stencil_dot_kernel<T, 1>), L.grid, L.block, 0, stream, sl.gp, L.xchunks, L.yblocks, L.zc, sl.p.p, sl.q.p, sl.partials.p);
        else
            hipLaunchKernelGGL((stencil_dot_kernel<T, vec_width<T>()>), L.grid, L.block, 0, stream, sl.gp, L.xchunks, L.yblocks, L.zc, sl.p.p, sl.q.p, sl.partials.p);
    }

    template <int VEC> void launch_update_xr(Slab<T>& sl, int rho_slot, int grid) {
        hipLaunchKernelGGL((update_xr_kernel<T, VEC>), dim3(grid), dim3(kBlock), 0, stream, sl.nown / VEC, sl.plane, sl.sc.p, rho_slot, sl.pq.p,
                           sl.x.p, sl.p.p, sl.r.p, sl.q.p, sl.partials.p);
    }
    template <int VEC> void launch_update_p(Slab<T>& sl, int rho_old, int rho_new, int init, int use_uw, const T* zsrc, int grid) {
        hipLaunchKernelGGL((update_p_kernel<T, VEC>), dim3(grid), dim3(kBlock), 0, stream, sl.nown / VEC, sl.plane, sl.sc.p, rho_old, rho_new,
                           sl.red.p, init, use_uw, zsrc, sl.p.p);
    }
    void update_p_all(int rho_old, int rho_new, int init, bool pre, const std::vector<int>& nparts) {
        for (size_t s = 0; s < slabs.size(); s++) {
            const T* zsrc = pre ? slabs[s].z.p : slabs[s].r.p;
            if (vec == 1) launch_update_p<1>(slabs[s], rho_old, rho_new, init, pre ? 0 : 1, zsrc, nparts[s]);
            else launch_update_p<vec_width<T>()>(slabs[s], rho_old, rho_new, init, pre ? 0 : 1, zsrc, nparts[s]);
        }
    }
    template <int VEC> void launch_norm2(Slab<T>& sl, const T* v, int grid) {
        hipLaunchKernelGGL((norm2_kernel<T, VEC>), dim3(grid), dim3(kBlock), 0, stream, sl.nown / VEC, sl.plane, v, sl.partials.p);
    }

    // ------------------------------------------------------------------------------------------
    // Fused CG sweeps (shm_cg_fused.hip.h): a workgroup of 8 waves owns whole x rows, so the row must fit WX <= 8 waves of VEC-wide lanes.
    struct FusedCfg {
        int wx = 0, ry = 8, zc = 16, yblocks = 0, zchunks = 0, nw = 8, part = 0;
    };
    bool fused_available() const {
        static const bool off = knob("SHM_CG_CLASSIC") != nullptr;  // A/B knob: the round-1 four-kernel loop
        return !off && (n + vec - 1) / vec <= 8 * 64;
    }
    FusedCfg fused_cfg(const Slab<T>& sl) const {
        FusedCfg c;
        const int lanes = (n + vec - 1) / vec;
        c.wx = 1;
        while (c.wx * 64 < lanes) c.wx <<= 1;
        // waves per workgroup: 8 (two 512-thread workgroups per CU, out of phase), 16 where a row needs 4 or more waves side by side -- with 8 the
        // workgroup would own only 4 rows (wx = 4) or 2 (wx = 8) and re-read as many border rows from L2 as it owns: 512^3 fp64 (wx = 4), 8 -> 16 waves:
        // DIR 5.33 -> 5.60 TB/s, RES 5.57 -> 5.88, 1.558 -> 1.50 ms per iteration; where wy is already >= 4 (256^3, 512^3 fp32) 16 waves are no better
        static const int nw_env = knob("SHM_FUSED_WAVES") ? atoi(knob("SHM_FUSED_WAVES")) : 0;  // A/B knob: 4, 8 or 16 waves per workgroup
        static const int ry_env = knob("SHM_FUSED_RY") ? atoi(knob("SHM_FUSED_RY")) : 0;
        c.ry = (vec == 1) ? 4 : 2;
#ifdef SHM_AB_SHAPES
        if (ry_env == 4 || (ry_env == 2 && vec != 1)) c.ry = ry_env;   // (1 row per lane measured no better than 2)
        const bool four_waves = nw_env == 4 && c.wx <= 4;
#else
        (void)ry_env;
        const bool four_waves = false;
#endif
        c.nw = four_waves ? 4 : (c.ry <= 2 && (nw_env == 16 || (nw_env == 0 && c.wx >= 4))) ? 16 : 8;   // (16-wave kernels exist for <= 2 rows per lane)
        const int wy = c.nw / c.wx;
        static const int zc_env = knob("SHM_FUSED_ZC") ? atoi(knob("SHM_FUSED_ZC")) : 0;
        // rows per workgroup wy * ry: 8 rows per lane unless that leaves too few workgroups along y to fill the chip with deep z chunks
        // rows per lane.  The bordering rows / planes a workgroup re-reads are served by L2 (PMC: HBM traffic of both sweeps = 3.0 N T, the
        // algorithmic figure), so small row blocks cost nothing in HBM bytes and what matters is memory-level parallelism: 2 rows per lane keep
        // the kernel under 128 VGPRs = two workgroups per CU running out of phase (512^3 fp64: DIR 0.70 -> 0.60 ms, RES 0.64 -> 0.58 ms against
        // 4 rows per lane / one workgroup per CU; 8 rows per lane spill)
        c.yblocks = (n + wy * c.ry - 1) / (wy * c.ry);
        // z chunks for ONE round of resident workgroups (two 8-wave or one 16-wave workgroup per CU: 2 num_cus row blocks x z chunks), i.e. the deepest
        // chunks that still fill the chip: a chunk's first two planes are loaded before anything is computed, and every chunk re-reads its two bordering planes.
        // Round 2 ran two rounds with 8 waves; one measured better in round 3 (tools/r03_cg_probe.sh): 512^3 fp32 64 instead of 32 planes per chunk DIR
        // 0.699 -> 0.723, RES 0.725 -> 0.746 of the HBM peak; 256^3 fp64 16 instead of 8 planes: loop 0.569 -> 0.589; deeper than one round: worse.
        const int want = std::max(1, (2 * num_cus + c.yblocks - 1) / c.yblocks);
        c.zc = std::min(64, std::max(8, sl.nzl / want));
        if (zc_env > 0) c.zc = zc_env;
        c.zc = std::max(1, std::min(c.zc, sl.nzl));
        c.zchunks = (sl.nzl + c.zc - 1) / c.zc;
        while ((size_t)c.yblocks * c.zchunks > sl.partials.count && c.zc < sl.nzl) {  // one block partial per workgroup: never more than the buffer holds
            c.zc = std::min(sl.nzl, 2 * c.zc);
            c.zchunks = (sl.nzl + c.zc - 1) / c.zc;
        }
        return c;
    }
    enum FusedPart { FUSED_ALL = 0, FUSED_INTERIOR = 1, FUSED_BOUNDARY = 2 };
    template <int MODE, int VEC, int RY, int WX>
    void launch_fused_k(Slab<T>& sl, const FusedCfg& cc, int slot_old, int slot_new, int init, int use_uw, int alpha_slot, const T* zsrc, const T* pin, T* pout) {
        FusedCfg c = cc;
        FusedParams F;
        F.n = n; F.nzl = sl.nzl; F.k0 = sl.k0; F.zc = c.zc; F.yblocks = c.yblocks; F.inv_h2 = sl.gp.inv_h2;
        F.zc_first = 0; F.zc_stride = 1;
        // fold_pq (one GPU): the DIR sweep leaves its block partials of p'.Kp' in an array of their own and the next RES sweep sums them itself
        const bool dir_to_own = fold_pq && MODE == CGF_DIR;
        const int pq_np = (fold_pq && MODE == CGF_RES) ? fold_pq_np : 0;
        if (fold_pq && !sl.pq_partials.p) sl.pq_partials.alloc(sl.partials.count);
        if (c.part == FUSED_INTERIOR) { F.zc_first = 1; c.zchunks -= 2; }
        else if (c.part == FUSED_BOUNDARY) { F.zc_stride = c.zchunks - 1; c.zchunks = 2; }
#ifdef SHM_AB_SHAPES   // (round 6: the 4-wave shapes and, for vector widths > 1, the 4-rows-per-lane shapes exist for A/B runs only -- 52 of the ~120 instantiations of
                       // cg_fused_kernel, compiled in with -DSHM_AB_SHAPES; the launcher below never selects them otherwise)
        if constexpr (WX <= 4) {
            if (c.nw == 4) {
                hipLaunchKernelGGL((cg_fused_kernel<T, VEC, RY, WX, 4 / WX, MODE>), dim3((unsigned)(c.yblocks * c.zchunks)), dim3(256), 0, stream, F, sl.sc.p, slot_old,
                                   slot_new, sl.red.p, sl.pq.p, init, use_uw, alpha_slot, zsrc, pin, pout, sl.r.p, dir_to_own ? sl.pq_partials.p : sl.partials.p, sl.pq_partials.p, pq_np);
                return;
            }
        }
#endif
        if constexpr (RY <= 2) {
            if (c.nw == 16) {   // 16-wave workgroups: twice the rows per workgroup, half the re-read border rows (see fused_cfg)
                hipLaunchKernelGGL((cg_fused_kernel<T, VEC, RY, WX, 16 / WX, MODE>), dim3((unsigned)(c.yblocks * c.zchunks)), dim3(1024), 0, stream, F, sl.sc.p, slot_old,
                                   slot_new, sl.red.p, sl.pq.p, init, use_uw, alpha_slot, zsrc, pin, pout, sl.r.p, dir_to_own ? sl.pq_partials.p : sl.partials.p, sl.pq_partials.p, pq_np);
                return;
            }
        }
        hipLaunchKernelGGL((cg_fused_kernel<T, VEC, RY, WX, 8 / WX, MODE>), dim3((unsigned)(c.yblocks * c.zchunks)), dim3(512), 0, stream, F, sl.sc.p, slot_old,
                           slot_new, sl.red.p, sl.pq.p, init, use_uw, alpha_slot, zsrc, pin, pout, sl.r.p, dir_to_own ? sl.pq_partials.p : sl.partials.p, sl.pq_partials.p, pq_np);
    }
    template <int MODE, int VEC, int RY>
    void launch_fused_w(Slab<T>& sl, const FusedCfg& c, int slot_old, int slot_new, int init, int use_uw, int alpha_slot, const T* zsrc, const T* pin, T* pout) {
        switch (c.wx) {
            case 1: launch_fused_k<MODE, VEC, RY, 1>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout); break;
            case 2: launch_fused_k<MODE, VEC, RY, 2>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout); break;
            case 4: launch_fused_k<MODE, VEC, RY, 4>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout); break;
            default: launch_fused_k<MODE, VEC, RY, 8>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout); break;
        }
    }
    // returns the number of block partials the sweep leaves in sl.partials
    template <int MODE>
    int launch_fused(Slab<T>& sl, int slot_old, int slot_new, int init, int use_uw, int alpha_slot, const T* zsrc, const T* pin, T* pout, int part = FUSED_ALL) {
        FusedCfg c = fused_cfg(sl);
        c.part = part;
        if (vec == 1) launch_fused_w<MODE, 1, 4>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout);
#ifdef SHM_AB_SHAPES
        else if (c.ry == 4) launch_fused_w<MODE, vec_width<T>(), 4>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout);
#endif
        else launch_fused_w<MODE, vec_width<T>(), 2>(sl, c, slot_old, slot_new, init, use_uw, alpha_slot, zsrc, pin, pout);
        return c.yblocks * c.zchunks;
    }
    // half: -1 = the whole slab; 0 / 1 = its lower / upper half (split at a vector boundary)
    void launch_x_update2(Slab<T>& sl, int use_a, int use_b, hipStream_t st = nullptr, int half = -1) {
        if (!st) st = stream;
        const size_t nall = sl.nown / vec, nlow = nall / 2;
        const size_t v0 = half == 1 ? nlow : 0, nvec = half < 0 ? nall : (half == 0 ? nlow : nall - nlow);
        if (nvec == 0) return;
        const size_t off = sl.plane + v0 * (size_t)vec;
        const unsigned g = (unsigned)((nvec + (size_t)kXuTiles * kBlock - 1) / ((size_t)kXuTiles * kBlock));
        if (vec == 1)
            hipLaunchKernelGGL((cg_x_update2_kernel<T, 1>), dim3(g), dim3(kBlock), 0, st, nvec, off, sl.sc.p, use_a, use_b, sl.p.p, sl.q.p, sl.x.p);
        else
            hipLaunchKernelGGL((cg_x_update2_kernel<T, vec_width<T>()>), dim3(g), dim3(kBlock), 0, st, nvec, off, sl.sc.p, use_a, use_b, sl.p.p, sl.q.p,
                               sl.x.p);
    }

    int stream_grid(const Slab<T>& sl) const { return grid_for(sl.nown / vec, 2048); }

    // v <- P v on all slabs (v = r, or z when on_z); leaves red[0] = sum of the first nparts[s] `partials`,
    // sc[SC_UW] = u.w and optionally sc[SC_RR] (= red[0] - u.w = ||P v||^2 when the partials were those of ||v||^2).
    void launch_projection(const std::vector<int>& nparts, bool on_z = false, int save_rr = 0, hipStream_t st = nullptr) {
        hipStream_t stream = st ? st : this->stream;   // (another stream only on one slab without a communicator: no all-reduce below)
        for (size_t s = 0; s < slabs.size(); s++) {
            Slab<T>& sl = slabs[s];
            hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(1 + (8 * m + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, m, sl.row_ptr.p, sl.ent_node.p,
                               sl.ent_coef.p, on_z ? sl.z.p : sl.r.p, sl.partials.p, nparts[s], sl.red.p);
        }
        if (!st) allreduce(0, 1 + m);
        for (Slab<T>& sl : slabs) {
            // (SHM_PROJ_F32, round 5: the fp32 solve's projector with the fp32 copies of the matrices -- half the bytes per application, but the converged plain CG of
            // rocker 128^3 fp32 then lands at 1.1e-3 of the fp64 solve instead of 6e-5: the projector has to be exact to the solve's own rounding.  Not adopted.)
            static const bool proj_f32 = knob("SHM_PROJ_F32") != nullptr;
            apply_Ginv(sl.red.p + 1, sl.u.p, proj_f32 && sizeof(T) == 4, stream);
            const int nred = std::max(1, std::min(16, m / 1024));   // workgroups of the u.w reduction (one per ~1000 rows)
            if (!sl.proj_ticket.p) {
                sl.proj_ticket.alloc(1);
                HIPCHK(hipMemsetAsync(sl.proj_ticket.p, 0, sizeof(unsigned), stream));
            }
            hipLaunchKernelGGL((scatter_nodes_kernel<T>), dim3(nred + (sl.n_touched + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sl.n_touched,
                               sl.node_id.p, sl.node_ptr.p, sl.ent_row.p, sl.nent_coef.p, sl.u.p, sl.red.p + 1, m, sl.sc.p, save_rr,
                               on_z ? sl.z.p : sl.r.p, nred, sl.partials.p, sl.proj_ticket.p);   // (the partials were consumed by gather_rows_kernel above)
        }
    }

    // ------------------------------------------------------------------------------------------
    // DCT preconditioner
    // The O(n log n) line transforms (shm_dct.hip.h): n = 2^k in [16,1024]; with P > 1 slabs: P | n and both the z-slab and the y-pencil hold whole 16-line
    // tiles' worth of rows
    bool fft_available() const {
        if (n < 16 || n > 1024 || (n & (n - 1)) != 0) return false;
        if (total_slabs == 1) return true;
        return slabs_equal && (total_slabs & (total_slabs - 1)) == 0 && n % total_slabs == 0;
    }
    // The fast Poisson solve K^+ exists for every n: where the FFT sweeps do not apply (n not a power of two -- the reference's nx = (size_t)(2 * 2^(hCoef+3))
    // with a fractional hCoef, signed_heat_grid_solver.cpp:24) a single slab applies the transforms as dense products with the DCT matrix on the fp64 matrix
    // cores (shm_dct_gemm.hip.h).  n <= 1024 like the FFT path: two n^3 double work arrays beside the solver's own.
    bool gemm_dct() const { return !fft_available() && total_slabs == 1 && n >= 4 && n <= 1024; }
    bool precond_available() const { return fft_available() || gemm_dct(); }
    void setup_precond() {
        if (precond_ready) return;
        log2n = 0;
        while ((1 << log2n) < n) log2n++;
        std::vector<Cplx<TP>> tw(n), om(n);
        std::vector<TP> lam(n);
        const double pi = 3.14159265358979323846;
        for (int k = 0; k < n; k++) {
            tw[k] = {(TP)std::cos(-2. * pi * k / n), (TP)std::sin(-2. * pi * k / n)};
            om[k] = {(TP)std::cos(-pi * k / (2. * n)), (TP)std::sin(-pi * k / (2. * n))};
            lam[k] = (TP)((2. - 2. * std::cos(pi * k / n)) / (cell * cell));
        }
        d_tw.upload(tw, stream);
        d_om.upload(om, stream);
        d_lam.upload(lam, stream);
        std::vector<double> lam64(n);
        for (int k = 0; k < n; k++) lam64[k] = (2. - 2. * std::cos(pi * k / n)) / (cell * cell);
        d_lam64.upload(lam64, stream);
        for (Slab<T>& sl : slabs) {
            if (!gemm_dct()) sl.W1.alloc(sl.nown);
            if (total_slabs > 1) sl.W2.alloc(sl.nown);
            sl.z.alloc(sl.ntot);
            HIPCHK(hipMemsetAsync(sl.z.p, 0, sl.ntot * sizeof(T), stream));
        }
        if (gemm_dct()) {
            std::vector<double> ctab(4 * (size_t)n);
            for (int r = 0; r < 4 * n; r++) ctab[(size_t)r] = std::cos(pi * r / (2. * n));
            DevArray<double> d_ctab;
            d_ctab.upload(ctab, stream);
            gd_Cm.alloc((size_t)n * n);
            gd_Ct.alloc((size_t)n * n);
            gd_W1.alloc((size_t)n * n * n);
            gd_W2.alloc((size_t)n * n * n);
            hipLaunchKernelGGL(dct_matrix_kernel, dim3(grid_for((size_t)n * n, 1024)), dim3(kBlock), 0, stream, n, d_ctab.p, gd_Cm.p, gd_Ct.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));   // d_ctab is a local
        }
        HIPCHK(hipStreamSynchronize(stream));  // host vectors go out of scope
        precond_ready = true;
    }

    // C (M x N) = A (M x K) B (K x N), row-major, `batches` products with element strides sA / sB / sC (shm_schur.hip.h).  narrow: the 128 x 32 shape that
    // fits beside the tiered Step 1; otherwise 128 x 128 tiles.
    void launch_dgemm(hipStream_t st, bool narrow, int prio, unsigned batches, int M, int N, int K, const double* A, int lda, long long sA, const double* B, int ldb,
                      long long sB, double* C, int ldc, long long sC) {
        auto tiles = [](size_t v) { return (unsigned)((v + kGemmT - 1) / kGemmT); };
        if (narrow)
            hipLaunchKernelGGL(dgemm_rm_kernel<1>, dim3((unsigned)((N + 31) / 32), tiles((size_t)M), batches), dim3(kBlock), 0, st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, prio);
        else
            hipLaunchKernelGGL(dgemm_rm_kernel<4>, dim3(tiles((size_t)N), tiles((size_t)M), batches), dim3(kBlock), 0, st, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, prio);
    }

    // z' = K^+ in by six dense products with the DCT matrix (shm_dct_gemm.hip.h; one slab, any n).  The products run in double whatever T; an fp32 solve
    // converts on the way in and out.  dot: partial sums of in . z' in sl.partials, their count returned.
    int launch_precond_gemm(bool dot, int in_sel, int out_sel) {
        Slab<T>& sl = slabs[0];
        const size_t N3 = (size_t)n * n * n;
        const int nn = n, n2 = n * n;
        double *W1 = gd_W1.p, *W2 = gd_W2.p;
        const double* X;
        double* Z;
        if constexpr (sizeof(T) == 8) {
            X = reinterpret_cast<const double*>(arr(sl, in_sel)) + sl.plane;
            Z = reinterpret_cast<double*>(arr(sl, out_sel)) + sl.plane;
        } else {
            hipLaunchKernelGGL((convert_kernel<T, double>), dim3(grid_for(N3, 4096)), dim3(kBlock), 0, stream, N3, arr(sl, in_sel) + sl.plane, W2);
            X = W2;
            Z = W2;
        }
        auto gemm = [&](unsigned batches, int M, int Nc, int K, const double* A, int lda, long long sA, const double* B, int ldb, long long sB, double* C, int ldc, long long sC) {
            launch_dgemm(stream, false, 0, batches, M, Nc, K, A, lda, sA, B, ldb, sB, C, ldc, sC);
        };
        gemm(1, n2, nn, nn, X, nn, 0LL, gd_Ct.p, nn, 0LL, W1, nn, 0LL);                                   // x forward:  W1[(k,j)][a] = sum_i X[(k,j)][i] C[a][i]
        gemm((unsigned)n, nn, nn, nn, gd_Cm.p, nn, 0LL, W1, nn, (long long)n2, W2, nn, (long long)n2);                   // y forward, per plane k (X is dead from here on)
        gemm(1, nn, n2, nn, gd_Cm.p, nn, 0LL, W2, n2, 0LL, W1, n2, 0LL);                                   // z forward
        hipLaunchKernelGGL(spectral_scale_kernel, dim3(grid_for(N3, 4096)), dim3(kBlock), 0, stream, n, d_lam64.p, W1);
        gemm(1, nn, n2, nn, gd_Ct.p, nn, 0LL, W1, n2, 0LL, W2, n2, 0LL);                                   // z inverse
        gemm((unsigned)n, nn, nn, nn, gd_Ct.p, nn, 0LL, W2, nn, (long long)n2, W1, nn, (long long)n2);     // y inverse
        gemm(1, n2, nn, nn, W1, nn, 0LL, gd_Cm.p, nn, 0LL, Z, nn, 0LL);                                    // x inverse:  Z[(k,j)][i] = sum_a W1[(k,j)][a] C[a][i]
        if constexpr (sizeof(T) != 8) hipLaunchKernelGGL((convert_kernel<double, T>), dim3(grid_for(N3, 4096)), dim3(kBlock), 0, stream, N3, W2, arr(sl, out_sel) + sl.plane);
        int np = 0;
        if (dot) {
            np = grid_for(N3, 2048);
            hipLaunchKernelGGL((dot_partial_kernel<T>), dim3(np), dim3(kBlock), 0, stream, N3, arr(sl, in_sel) + sl.plane, arr(sl, out_sel) + sl.plane, sl.partials.p);
        }
        return np;
    }

    template <int MODE, typename TIn, typename TOut, bool DOT, int LOG2N, bool XPASS>
    void launch_dct_n(const DctParams& P, int ntiles, const TIn* in, TOut* out, const TOut* dotw, double* partials, const int* tile_list, const unsigned* elem_mask) {
        // segmented addressing (the packed all-to-all layout) occurs only in the y sweeps of a multi-slab transform
        constexpr bool kCanSeg = !XPASS && MODE != DCT_FUSED && !DOT;
        const bool seg = P.in.seg_shift < 30 || P.out.seg_shift < 30;
        if (seg && !kCanSeg) throw Error(SHM_ERR_INVALID, "DCT: segmented layout in a sweep that does not support it");
        if constexpr (kCanSeg) {
            if (seg) {
                launch_dct_k<MODE, TIn, TOut, DOT, LOG2N, XPASS, true>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask);
                return;
            }
        }
        launch_dct_k<MODE, TIn, TOut, DOT, LOG2N, XPASS, false>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask);
    }
    template <int MODE, typename TIn, typename TOut, bool DOT, int LOG2N, bool XPASS, bool SEG>
    void launch_dct_k(const DctParams& P, int ntiles, const TIn* in, TOut* out, const TOut* dotw, double* partials, const int* tile_list, const unsigned* elem_mask) {
        // Dense sweeps of the long transforms (n >= 256): the LDS-DMA double buffer of shm_dct.hip.h (PF, round 6) -- the next tile's input in flight, global -> LDS, under this
        // tile's FFT passes and stores.  Built, verified (76 GPU tests) and measured SLOWER than two or three plain workgroups per CU out of phase: 512^3 fp64 0.639 -> 0.687 ms per
        // sweep (0.46 -> 0.43 of the HBM peak), fp32 0.333 -> 0.411, 256^3 0.059 -> 0.070 (profiles/r06_dct_dma_rejected.txt): the sweeps are bound by their barrier-separated
        // LDS passes at the occupancy the LDS footprint allows, not by the bytes in flight -- the staging tile costs a workgroup per CU and adds a pass.  (Round 4's register
        // prefetch: the same verdict, profiles/r04_dct_prefetch_rejected.txt.)  Compiled only into -DSHM_DCT_PF builds; SHM_DCT_NO_PF=1 switches it off there (A/B).
        static const bool no_pf = knob("SHM_DCT_NO_PF") != nullptr;
#ifdef SHM_DCT_PF
        constexpr bool kCanPf = LOG2N >= 8 && !SEG && sizeof(TIn) == sizeof(TP) && ((size_t)(1 << LOG2N) * dct_lines_for(LOG2N, (int)sizeof(TP)) * sizeof(TIn)) % (16 * kBlock) == 0;
#else
        constexpr bool kCanPf = false;
#endif
        bool use_pf = false;
        if constexpr (kCanPf) {
            // (unit stride between the lines of a y / z tile, 16-byte aligned rows: what the plain layouts of launch_precond have)
            const bool aligned = XPASS ? (P.in.elem_stride == 1 && (P.in.line_stride * (long long)sizeof(TIn)) % 16 == 0)
                                       : (P.in.line_stride == 1 && (P.in.elem_stride * (long long)sizeof(TIn)) % 16 == 0 && (P.in.a_stride * (long long)sizeof(TIn)) % 16 == 0);
            use_pf = !no_pf && !tile_list && !elem_mask && aligned && (P.in.off * (long long)sizeof(TIn)) % 16 == 0 && (P.in.b_stride * (long long)sizeof(TIn)) % 16 == 0;
        }
        if constexpr (kCanPf) {
            if (use_pf) {
                launch_dct_pf<MODE, TIn, TOut, DOT, LOG2N, XPASS, SEG, true>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask);
                return;
            }
        }
        launch_dct_pf<MODE, TIn, TOut, DOT, LOG2N, XPASS, SEG, false>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask);
    }
    // (the kernel is named in the launch itself: a variant reached only through a function pointer assigned in a branch got no host stub from this compiler)
    template <int MODE, typename TIn, typename TOut, bool DOT, int LOG2N, bool XPASS, bool SEG, bool PF>
    void launch_dct_pf(const DctParams& P, int ntiles, const TIn* in, TOut* out, const TOut* dotw, double* partials, const int* tile_list, const unsigned* elem_mask) {
        static uint64_t configured = 0;  // per instantiation, one bit per device (the attribute is per device)
        constexpr size_t lds = dct_lds_bytes<LOG2N, (int)sizeof(TP)>() + (PF ? dct_stage_bytes<LOG2N, (int)sizeof(TP)>() : 0);
        const uint64_t dev_bit = 1ull << (cfg.device & 63);
        if (!(configured & dev_bit) || cfg.device >= 64) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dct_lines_kernel<TP, TIn, TOut, MODE, DOT, LOG2N, XPASS, SEG, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured |= dev_bit;
        }
        DctParams Q = P;
        Q.ntiles = ntiles;
#ifdef SHM_EXPERIMENT_KNOBS
        { static const int dbg = knob("SHM_DCT_SKIP") ? atoi(knob("SHM_DCT_SKIP")) : 0; Q.debug_skip = dbg; }
#else
        Q.debug_skip = 0;
#endif
        const int per_cu = std::max(1, (int)((size_t)(160 * 1024) / lds));
        const int grid = (int)std::min<long long>(ntiles, std::max<long long>(1, (long long)dct_grid_x16 * num_cus * per_cu / 16));
        hipLaunchKernelGGL((dct_lines_kernel<TP, TIn, TOut, MODE, DOT, LOG2N, XPASS, SEG, PF>), dim3((unsigned)grid), dim3(kBlock), lds, stream, Q, in, out, d_tw.p, d_om.p, d_lam.p, dotw,
                           partials, tile_list, elem_mask);
    }
    template <int MODE, typename TIn, typename TOut, bool DOT, bool XPASS>
    void launch_dct(const DctParams& P, int ntiles, const TIn* in, TOut* out, const TOut* dotw, double* partials, const int* tile_list = nullptr,
                    const unsigned* elem_mask = nullptr) {
        switch (log2n) {
            case 4: launch_dct_n<MODE, TIn, TOut, DOT, 4, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 5: launch_dct_n<MODE, TIn, TOut, DOT, 5, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 6: launch_dct_n<MODE, TIn, TOut, DOT, 6, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 7: launch_dct_n<MODE, TIn, TOut, DOT, 7, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 8: launch_dct_n<MODE, TIn, TOut, DOT, 8, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 9: launch_dct_n<MODE, TIn, TOut, DOT, 9, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            case 10: launch_dct_n<MODE, TIn, TOut, DOT, 10, XPASS>(P, ntiles, in, out, dotw, partials, tile_list, elem_mask); break;
            default: throw Error(SHM_ERR_INVALID, "DCT preconditioner: unsupported grid size");
        }
    }

    static DctAddr plain_addr(long long off, long long a_stride, long long b_stride, long long line_stride, long long elem_stride) {
        DctAddr A{};
        A.off = off; A.a_stride = a_stride; A.b_stride = b_stride; A.line_stride = line_stride; A.elem_stride = elem_stride;
        A.seg_stride = 0; A.seg_shift = 30; A.seg_mask = (1 << 30) - 1;
        return A;
    }

    // transport: block q of src[slab g] -> block g of dst[slab q] for all slab pairs (blocks of `blk` elements): the
    // transposition between z-slabs and y-pencils of the distributed DCT.  Local pairs are device copies; pairs on other
    // ranks travel as grouped ncclSend/ncclRecv (an all-to-all over xGMI).
    void alltoall_blocks(bool w1_to_w2, size_t blk) {
        auto srcp = [&](Slab<T>& sl) { return w1_to_w2 ? sl.W1.p : sl.W2.p; };
        auto dstp = [&](Slab<T>& sl) { return w1_to_w2 ? sl.W2.p : sl.W1.p; };
        const int ls = cfg.local_slabs;
        for (int a = 0; a < ls; a++)
            for (int b = 0; b < ls; b++)
                HIPCHK(hipMemcpyAsync(dstp(slabs[b]) + (size_t)(first_slab + a) * blk, srcp(slabs[a]) + (size_t)(first_slab + b) * blk, blk * sizeof(T),
                                      hipMemcpyDeviceToDevice, stream));
        if (comm) {
            Rccl& R = Rccl::get();
            const int dt = sizeof(T) == 8 ? Rccl::kFloat64 : Rccl::kFloat32;
            R.chk(R.GroupStart(), "ncclGroupStart");
            for (int peer = 0; peer < cfg.world; peer++) {
                if (peer == cfg.rank) continue;
                // sends ordered (my slab a, peer slab b); the peer posts its receives in that same order
                for (int a = 0; a < ls; a++)
                    for (int b = 0; b < ls; b++)
                        R.chk(R.Send(srcp(slabs[a]) + (size_t)(peer * ls + b) * blk, blk, dt, peer, comm, stream), "ncclSend(a2a)");
                for (int b = 0; b < ls; b++)      // peer's slab b sent ...
                    for (int a = 0; a < ls; a++)  // ... its block for my slab a
                        R.chk(R.Recv(dstp(slabs[a]) + (size_t)(peer * ls + b) * blk, blk, dt, peer, comm, stream), "ncclRecv(a2a)");
            }
            R.chk(R.GroupEnd(), "ncclGroupEnd");
        }
    }

    // z' = M^-1 r on every slab (no projection).  dot: also leave the partial sums of r.z' in sl.partials; returns
    // their count per slab.  One slab: x-fwd, y-fwd, z-fused, y-inv, x-inv in place on W1.  P slabs: the y sweeps write /
    // read the packed layout [dest slab][z_local][y_local][x] and two all-to-alls turn z-slabs into y-pencils and back.
    enum ArrSel { ARR_R = 0, ARR_Z = 1, ARR_P = 2, ARR_X = 3, ARR_Q = 4, ARR_Y0 = 5, ARR_Y1 = 6, ARR_Y2 = 7 };
    static T* arr(Slab<T>& sl, int sel) {
        switch (sel) {
            case ARR_R: return sl.r.p;
            case ARR_Z: return sl.z.p;
            case ARR_P: return sl.p.p;
            case ARR_X: return sl.x.p;
            case ARR_Y0: return sl.Y0.p;
            case ARR_Y1: return sl.Y1.p;
            case ARR_Y2: return sl.Y2.p;
            default: return sl.q.p;
        }
    }
    int launch_precond(bool dot, int in_sel = ARR_R, int out_sel = ARR_Z, double scale = 1.0 /* out = scale * K^+ in (transform sweeps only: it rides in the spectral step's normalisation) */) {
        if (gemm_dct()) {
            if (scale != 1.0) throw Error(SHM_ERR_INVALID, "launch_precond: no scale on the dense-product transforms");
            return launch_precond_gemm(dot, in_sel, out_sel);
        }
        const long long nn = n, plane = (long long)n * n;
        const int P = total_slabs;
        const int nzl = n / P, nyl = n / P;  // planes per slab == pencil rows per slab (P | n)
        const double inv8 = scale * 8.0 / ((double)n * n * n);
        const int kDctLines = dct_lines_for(log2n, (int)sizeof(TP));                      // real lines per tile (16, or 8 from n = 512 on)
        const int tiles_slab = (int)((long long)nzl * nn / kDctLines);  // tiles of one slab for the x and y sweeps
        int log2nyl = 0;
        while ((1 << log2nyl) < nyl) log2nyl++;
        const long long blk = (long long)nzl * nyl * nn;
        for (int pass = 0; pass < 7; pass++) {
            if (P == 1 && (pass == 2 || pass == 4)) continue;
            if (pass == 2) { alltoall_blocks(false, (size_t)blk); continue; }  // packed W2 (z-slabs) -> W1 (y-pencils [z][y_l][x])
            if (pass == 4) { alltoall_blocks(true, (size_t)blk); continue; }   // W1 (y-pencils) -> packed W2 (z-slabs)
            for (size_t si = 0; si < slabs.size(); si++) {
                Slab<T>& sl = slabs[si];
                const int g = first_slab + (int)si;
                DctParams Q{};
                Q.inv_n3_8 = inv8;
                Q.ky0 = 0;
                switch (pass) {
                    case 0:  // x-fwd: r (ghost layout) -> W1 natural [z_l][y][x]; tile = 16 consecutive rows
                        Q.tiles_a = tiles_slab;
                        Q.in = plain_addr((long long)sl.plane, (long long)kDctLines * nn, 0, nn, 1);
                        Q.out = plain_addr(0, (long long)kDctLines * nn, 0, nn, 1);
                        launch_dct<DCT_FWD, T, TP, false, true>(Q, tiles_slab, arr(sl, in_sel), sl.W1.p, (const TP*)nullptr, nullptr);
                        break;
                    case 1:  // y-fwd: W1 natural -> (P==1: W1 in place) | (P>1: W2 packed [q][z_l][y_l][x])
                        Q.tiles_a = n / kDctLines;
                        Q.in = plain_addr(0, kDctLines, plane, 1, nn);
                        if (P == 1) {
                            Q.out = Q.in;
                            launch_dct<DCT_FWD, TP, TP, false, false>(Q, tiles_slab, sl.W1.p, sl.W1.p, (const TP*)nullptr, nullptr);
                        } else {
                            Q.out = plain_addr(0, kDctLines, (long long)nyl * nn, 1, nn);
                            Q.out.seg_shift = log2nyl; Q.out.seg_mask = nyl - 1; Q.out.seg_stride = blk;
                            launch_dct<DCT_FWD, TP, TP, false, false>(Q, tiles_slab, sl.W1.p, sl.W2.p, (const TP*)nullptr, nullptr);
                        }
                        break;
                    case 3: {  // z-fused on the y-pencil [z (n)][y_l][x] of W1; tile = 16 consecutive x at one y_l
                        Q.tiles_a = n / kDctLines;
                        Q.in = plain_addr(0, kDctLines, nn, 1, (long long)nyl * nn);
                        Q.out = Q.in;
                        Q.ky0 = g * nyl;
                        const int tiles_z = (n / kDctLines) * nyl;
                        launch_dct<DCT_FUSED, TP, TP, false, false>(Q, tiles_z, sl.W1.p, sl.W1.p, (const TP*)nullptr, nullptr);
                        break;
                    }
                    case 5:  // y-inv: (P==1: W1 in place) | (P>1: W2 packed -> W1 natural)
                        Q.tiles_a = n / kDctLines;
                        Q.out = plain_addr(0, kDctLines, plane, 1, nn);
                        if (P == 1) {
                            Q.in = Q.out;
                            launch_dct<DCT_INV, TP, TP, false, false>(Q, tiles_slab, sl.W1.p, sl.W1.p, (const TP*)nullptr, nullptr);
                        } else {
                            Q.in = plain_addr(0, kDctLines, (long long)nyl * nn, 1, nn);
                            Q.in.seg_shift = log2nyl; Q.in.seg_mask = nyl - 1; Q.in.seg_stride = blk;
                            launch_dct<DCT_INV, TP, TP, false, false>(Q, tiles_slab, sl.W2.p, sl.W1.p, (const TP*)nullptr, nullptr);
                        }
                        break;
                    case 6:  // x-inv: W1 natural -> z (ghost layout) (+ partial r.z')
                        Q.tiles_a = tiles_slab;
                        Q.in = plain_addr(0, (long long)kDctLines * nn, 0, nn, 1);
                        Q.out = plain_addr((long long)sl.plane, (long long)kDctLines * nn, 0, nn, 1);
                        if (dot) launch_dct<DCT_INV, TP, T, true, true>(Q, tiles_slab, sl.W1.p, arr(sl, out_sel), arr(sl, in_sel), sl.partials.p);
                        else launch_dct<DCT_INV, TP, T, false, true>(Q, tiles_slab, sl.W1.p, arr(sl, out_sel), (const T*)nullptr, nullptr);
                        break;
                    default: break;
                }
            }
        }
        return dot ? tiles_slab : 0;
    }

    // ------------------------------------------------------------------------------------------
    // shift + phi (:110-111), shared by the constrained solve and the fast path; x holds -phi.
    void launch_shift_and_phi() {
        for (Slab<T>& sl : slabs) {
            const int g = std::max(1, std::min(256, (sl.n_shift + kBlock - 1) / kBlock));
            hipLaunchKernelGGL((shift_partial_kernel<T>), dim3(g), dim3(kBlock), 0, stream, sl.n_shift, sl.shift_items.p, n, sl.x.p, sl.partials.p);
            hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, sl.partials.p, g, sl.pq.p);
        }
        allreduce(1, 1);
        for (Slab<T>& sl : slabs)
            hipLaunchKernelGGL((write_phi_kernel<T>), dim3(grid_for(sl.nown, 4096)), dim3(kBlock), 0, stream, sl.nown, sl.plane, sl.x.p, sl.pq.p, area_sum,
                               sl.sc.p, sl.q.p);
        HIPCHK(hipGetLastError());
    }

    // fastIntegration: three families of prefix scans (see bfs_* kernels); slabs are chained bottom-up through the
    // top plane of x (device copy between local slabs, ncclSend/ncclRecv between ranks).
    void solve_fast(const shm_opts& o, shm_stats* st, Event& e_start, Event& e_conv, Event& e_div, std::chrono::steady_clock::time_point wall0) {
        (void)o;
        Event e_int, e_end;
        build_shift_items_only();
        Rccl* R = comm ? &Rccl::get() : nullptr;
        const int dt = sizeof(T) == 8 ? Rccl::kFloat64 : Rccl::kFloat32;
        for (size_t si = 0; si < slabs.size(); si++) {
            Slab<T>& sl = slabs[si];
            if (sl.k0 == 0) {
                hipLaunchKernelGGL((bfs_plane0_kernel<T>), dim3(1), dim3(std::min(1024, ((n + 63) / 64) * 64)), 0, stream, n, cell, bbox_min[0], bbox_min[1],
                                   sl.Y0.p, sl.Y1.p, sl.Y2.p, sl.x.p);
            } else if (si == 0) {
                R->chk(R->Recv(sl.x.p, sl.plane, dt, cfg.rank - 1, comm, stream), "ncclRecv(bfs)");  // low ghost <- top plane of the rank below
            }
            hipLaunchKernelGGL((bfs_z_kernel<T>), dim3((unsigned)((sl.plane + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sl.gp, cell, bbox_min[2], sl.Y0.p,
                               sl.Y1.p, sl.Y2.p, sl.x.p);
            const T* top = sl.x.p + (size_t)sl.nzl * sl.plane;
            if (si + 1 < slabs.size()) HIPCHK(hipMemcpyAsync(slabs[si + 1].x.p, top, sl.plane * sizeof(T), hipMemcpyDeviceToDevice, stream));
            else if (comm && cfg.rank < cfg.world - 1) R->chk(R->Send(top, sl.plane, dt, cfg.rank + 1, comm, stream), "ncclSend(bfs)");
        }
        HIPCHK(hipGetLastError());
        e_int.record(stream);
        launch_shift_and_phi();
        e_end.record(stream);
        HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        have_phi = true;
        if (st) {
            memset(st, 0, sizeof *st);
            st->n = n;
            st->S = S;
            st->shift = h_pinned[SC_SHIFT];
            st->ms_conv = elapsed(e_start, e_conv);
            st->ms_div = elapsed(e_conv, e_div);
            st->ms_pcg = elapsed(e_div, e_int);  // the integration that replaces the solve
            st->ms_shift = elapsed(e_int, e_end);
            st->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
            st->preconditioner = SHM_PRECOND_NONE;
        }
    }

    // K^+ applied to a vector that is non-zero only on the nodes the constraint rows touch, with the result needed only on those
    // nodes (one iteration of the dual solver): the x sweeps visit only the 16-row tiles that contain touched nodes, the y sweeps only
    // the z-planes that do; only the fused z sweep is dense.  S1 / S2 keep exact zeros outside the active tiles / planes (zeroed when
    // the constraint set is built, written only inside it), so the skipped lines transform to the zeros they would produce.
    void build_active_tiles(const std::vector<int64_t>& touched_nodes) {
        Slab<T>& sl = slabs[0];
        const int L = dct_lines_for(log2n_of(n), (int)sizeof(TP));
        const int64_t nn = n, pl = (int64_t)n * n;
        std::vector<int> ax, planes;
        for (int64_t g : touched_nodes) {
            const int64_t k = g / pl, j = (g - k * pl) / nn;
            ax.push_back((int)((k * nn + j) / L));
            planes.push_back((int)k);
        }
        std::sort(ax.begin(), ax.end());
        ax.erase(std::unique(ax.begin(), ax.end()), ax.end());
        std::sort(planes.begin(), planes.end());
        planes.erase(std::unique(planes.begin(), planes.end()), planes.end());
        std::vector<int> ay;
        const int tiles_a = n / L;
        for (int z : planes)
            for (int xc = 0; xc < tiles_a; xc++) ay.push_back(xc + z * tiles_a);
        std::vector<unsigned> zm((size_t)(n + 31) / 32, 0u);
        for (int z : planes) zm[(size_t)z >> 5] |= 1u << (z & 31);
        sl.act_z.upload(zm, stream2);
        sl.n_act_planes = (int)planes.size();
        sl.act_planes.upload(planes, stream2);
        sl.n_act_x = (int)ax.size();
        sl.n_act_y = (int)ay.size();
        sl.act_x.upload(ax, stream2);
        sl.act_y.upload(ay, stream2);
        sl.S1.alloc(sl.nown);
        sl.S2.alloc(sl.nown);
        sl.S4.alloc(sl.nown);
        sl.W1.alloc(sl.nown);
        HIPCHK(hipMemsetAsync(sl.S1.p, 0, sl.nown * sizeof(T), stream2));
        HIPCHK(hipStreamSynchronize(stream2));
    }
    static int log2n_of(int v) {
        int l = 0;
        while ((1 << l) < v) l++;
        return l;
    }
    void launch_precond_sparse(int in_sel, int out_sel) {
        Slab<T>& sl = slabs[0];
        const long long nn = n, plane = (long long)n * n;
        const int L = dct_lines_for(log2n, (int)sizeof(TP));
        const int tiles_all = (int)(plane * nn / L / nn);  // = n*n/L tiles per sweep
        DctParams Q{};
        Q.inv_n3_8 = 8.0 / ((double)n * n * n);
        Q.ky0 = 0;
        // x-fwd on the active tiles: in (ghost layout) -> S1
        Q.tiles_a = tiles_all;
        Q.in = plain_addr((long long)sl.plane, (long long)L * nn, 0, nn, 1);
        Q.out = plain_addr(0, (long long)L * nn, 0, nn, 1);
        launch_dct<DCT_FWD, T, TP, false, true>(Q, sl.n_act_x, arr(sl, in_sel), sl.S1.p, (const TP*)nullptr, nullptr, sl.act_x.p);
        // y-fwd on the active planes: S1 -> S2
        Q.tiles_a = n / L;
        Q.in = plain_addr(0, L, plane, 1, nn);
        Q.out = Q.in;
        launch_dct<DCT_FWD, TP, TP, false, false>(Q, sl.n_act_y, sl.S1.p, sl.S2.p, (const TP*)nullptr, nullptr, sl.act_y.p);
        // z-fused over all lines, but only the active planes are read (the others hold zeros) and written (the others are not needed): S2 -> W1
        Q.in = plain_addr(0, L, nn, 1, plane);
        Q.out = Q.in;
        // the recursive z step wins from n = 256 on (256^3: 79 -> 42 us, 512^3: 0.61 -> 0.09 ms per application); below that the transform of
        // all lines is cheaper than one more launch.  SHM_DUAL_Z_FFT: A/B knob for the transform version
        static const bool fft_z_env = knob("SHM_DUAL_Z_FFT") != nullptr;
        const bool fft_z = fft_z_env || n < 256;
        if (fft_z) launch_dct<DCT_FUSED, TP, TP, false, false>(Q, tiles_all, sl.S2.p, sl.W1.p, (const TP*)nullptr, nullptr, nullptr, sl.act_z.p);
        else {
            hipLaunchKernelGGL((zsolve_sparse_kernel<TP>), dim3((unsigned)((plane + kBlock - 1) / kBlock) + 1), dim3(kBlock), 0, stream, n, sl.n_act_planes,
                               sl.act_planes.p, d_lam64.p, sl.gp.inv_h2, sl.S2.p, sl.W1.p);   // + 1: the workgroup of the singular (0, 0) line
        }
        // y-inv on the active planes: W1 -> S4
        Q.in = plain_addr(0, L, plane, 1, nn);
        Q.out = Q.in;
        launch_dct<DCT_INV, TP, TP, false, false>(Q, sl.n_act_y, sl.W1.p, sl.S4.p, (const TP*)nullptr, nullptr, sl.act_y.p);
        // x-inv on the active tiles: S4 -> out (ghost layout)
        Q.tiles_a = tiles_all;
        Q.in = plain_addr(0, (long long)L * nn, 0, nn, 1);
        Q.out = plain_addr((long long)sl.plane, (long long)L * nn, 0, nn, 1);
        launch_dct<DCT_INV, TP, T, false, true>(Q, sl.n_act_x, sl.S4.p, arr(sl, out_sel), (const T*)nullptr, nullptr, sl.act_x.p);
    }

    // ------------------------------------------------------------------------------------------
    // Dual solver (see the block comment above the dual_* kernels): CG on S = A K^+ A^T for the multipliers.
    void solve_dual(const shm_opts& o, shm_stats* st, Event& e_start, Event& e_conv, Event& e_div, Event& e_setup, Event& e_s2a, Event& e_s2b,
                    std::chrono::steady_clock::time_point wall0) {
        Event e_pcg, e_end;
        const int check_every = o.check_every > 0 ? std::min(o.check_every, 4) : 2;
        auto mv = [&](Slab<T>& sl, int k) { return sl.dv.p + (size_t)k * mp; };
        enum { V_MU = 0, V_R = 1, V_P = 2, V_Z = 3, V_T1 = 4, V_T2 = 5, V_G = 6 };
        const std::vector<int> nopart(slabs.size(), 0);
        auto gather = [&](int sel) {  // red[1..m] = A * arr(sel), all-reduced
            for (Slab<T>& sl : slabs)
                hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(1 + (8 * m + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, m, sl.row_ptr.p, sl.ent_node.p,
                                   sl.ent_coef.p, arr(sl, sel), sl.partials.p, 0, sl.red.p);
            allreduce(0, 1 + m);
        };
        auto scatter = [&](int vec, int sel, int accumulate, double scale = 1.0, bool save = false) {  // arr(sel) (+)= scale A^T v
            for (Slab<T>& sl : slabs)
                hipLaunchKernelGGL((scatter_rows_to_nodes_kernel<T>), dim3((sl.n_touched + kBlock - 1) / kBlock + 1), dim3(kBlock), 0, stream, sl.n_touched,
                                   sl.node_id.p, sl.node_ptr.p, sl.ent_row.p, sl.nent_coef.p, mv(sl, vec), accumulate, arr(sl, sel), scale, save ? sl.touched_save.p : (T*)nullptr);
        };
        auto precondition = [&](int init) {  // z = Pm(G^-1 B G^-1 r); p = z (+ beta p)
            for (Slab<T>& sl : slabs) {
                apply_Ginv(mv(sl, V_R), mv(sl, V_T1), true, stream);
                hipLaunchKernelGGL(csr_matvec_kernel, dim3((m + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, m, Bptr.p, Bcol.p, Bval.p, mv(sl, V_T1), mv(sl, V_T2));
                apply_Ginv(mv(sl, V_T2), mv(sl, V_Z), true, stream);
                hipLaunchKernelGGL(dual_direction_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, init, mv(sl, V_R), mv(sl, V_Z), mv(sl, V_P), sl.sc.p);
            }
        };
        // ---- g = A K^+ b  and  sum(b)
        launch_precond(false, ARR_R, ARR_Z);
        gather(ARR_Z);
        for (Slab<T>& sl : slabs) {
            HIPCHK(hipMemcpyAsync(mv(sl, V_G), sl.red.p + 1, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, stream));
            if (sl.div_sum_blocks > 0) {   // the divergence kernel summed b as it wrote it (round 5: one pass over the grid less)
                hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, sl.div_partials.p, sl.div_sum_blocks, sl.pq.p);
                continue;
            }
            const int g = grid_for(sl.nown, 1024);
            hipLaunchKernelGGL((sum_kernel<T>), dim3(g), dim3(kBlock), 0, stream, sl.nown, sl.plane, sl.r.p, sl.partials.p);
            hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, sl.partials.p, g, sl.pq.p);
        }
        allreduce(1, 1);
        for (Slab<T>& sl : slabs) {
            hipLaunchKernelGGL(dual_init_mu_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, sl.pq.p, mv(sl, V_MU), sl.sc.p);
        }
        // ---- r = Pm(g - S mu), z, p
        const bool sparse_ok = total_slabs == 1 && slabs[0].n_act_x > 0 && !knob("SHM_DENSE_DCT");
        // explicit S (shm_schur.hip.h): one dense mat-vec instead of scatter, five sweeps, gather.  Several slabs / ranks (round 6): S (and S^-1, G^-1, B) are replicated and
        // every slab applies them to its own copy of the m-vectors -- the same numbers everywhere, no communication inside the dual iteration
        const bool dense_S = have_S;
        if (!dense_S)
            for (Slab<T>& sl : slabs) HIPCHK(hipMemsetAsync(sl.p.p, 0, sl.ntot * sizeof(T), stream));  // w = A^T nu lives in p: zero outside the touched nodes
        auto apply_S = [&](int vec) {   // red[1..m] = S v
            for (Slab<T>& sl : slabs) launch_ginv_matvec<double>(stream, m, m, mp, Sdense.p, mv(sl, vec), sl.red.p + 1);
        };
        if (dense_S) apply_S(V_MU);
        else {
            scatter(V_MU, ARR_P, 0);
            if (sparse_ok) launch_precond_sparse(ARR_P, ARR_Z);
            else launch_precond(false, ARR_P, ARR_Z);
            gather(ARR_Z);
        }
        for (Slab<T>& sl : slabs)
            hipLaunchKernelGGL(dual_init_residual_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, mv(sl, V_G), sl.red.p + 1, mv(sl, V_R), sl.sc.p, 1);
        const bool direct = dual_direct && dense_S;   // S^-1 is resident (build_constraints): no CG, no G^-1, no B
        if (!direct) precondition(1);
        HIPCHK(hipGetLastError());

        const int kMaxSamples = 32;
        std::vector<std::unique_ptr<Event>> ev;
        if (st) for (int a = 0; a < 3 * kMaxSamples; a++) ev.emplace_back(new Event());
        int nsamples = 0, it = 0;
        double rr0 = 0., rr = 0.;
        bool converged = false, breakdown = false;
        // ---- x = K^+ (A^T mu - b)   (the additive constant cancels in the shift), shift, phi
        // Transform sweeps: b - A^T mu is formed IN PLACE on the touched nodes of r and the sign rides in the spectral normalisation (-K^+ (b - A^T mu): bit for bit
        // the same numbers, floating-point subtraction being sign-symmetric), then the touched nodes get b back exactly.  (Rounds 2-4 wrote -b into q with a
        // whole-grid kernel first: 56 us of the 0.8 ms solve phase at 256^3.)
        auto finish = [&]() {
            if (gemm_dct()) {
                for (Slab<T>& sl : slabs)
                    hipLaunchKernelGGL((negate_kernel<T>), dim3(grid_for(sl.ntot, 4096)), dim3(kBlock), 0, stream, sl.ntot, sl.r.p, sl.q.p);
                scatter(V_MU, ARR_Q, 1);
                launch_precond(false, ARR_Q, ARR_X);
            } else {
                scatter(V_MU, ARR_R, 1, -1.0, true);
                launch_precond(false, ARR_R, ARR_X, -1.0);
                for (Slab<T>& sl : slabs)
                    hipLaunchKernelGGL((restore_nodes_kernel<T>), dim3((sl.n_touched + kBlock - 1) / kBlock + 1), dim3(kBlock), 0, stream, sl.n_touched, sl.node_id.p,
                                       sl.touched_save.p, sl.r.p);
            }
            gather(ARR_X);  // A x0: constant over the rows at convergence; its mean is the KKT solution's additive constant
            for (Slab<T>& sl : slabs) hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, sl.red.p + 1, m, sl.sc.p + SC_AXSUM);
            e_pcg.record(stream);
            launch_shift_and_phi();
            e_end.record(stream);
            HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        };
        bool finished = false;
        if (direct) {
            // Direct solve of the bordered system [[S, 1], [1^T, 0]] [delta; c] = [r; 0] for the correction of mu_0 (whose sum is already sum(b)):
            // u = S^-1 r, v = S^-1 1 (formed behind the inversion, in the set-up), delta = u - (1^T u / 1^T v) v;  then r = Pm(g - S mu) again with the explicit S.
            // The first pass IS the solution (cond(S) ~ 5e2 ... 7e4 on the bunny grids: 1e-12 and better); further passes are iterative refinement, taken only
            // while the residual test of the CG path -- the same one -- is not met.
            auto apply_Sinv = [&](const double* w, double* u) { launch_ginv_matvec<double>(stream, m, m, mp, Sinv.p, w, u); };
            // At most six passes, and never more than max_iters.  A tolerance below what the arithmetic can deliver (about eps * cond(S): cond is 5e2 ... 7e4
            // here) would otherwise end in SHM_ERR_NOCONV although mu is at rounding accuracy: when a pass no longer reduces the residual by at least a
            // factor of four and the residual already sits below 1e-9 of the right-hand side, the stagnation IS convergence; rel_residual reports what was reached.
            const int max_passes = std::min(o.max_iters, 6);
            double rr_prev = -1.;
            while (it < max_passes && !converged && !breakdown) {
                const bool sample = st && nsamples < kMaxSamples;
                if (sample) ev[3 * nsamples]->record(stream);
                for (Slab<T>& sl : slabs) {
                    apply_Sinv(mv(sl, V_R), mv(sl, V_T1));
                    hipLaunchKernelGGL(dual_bordered_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, mv(sl, V_T1), Sinv_ones.p, (const double*)nullptr, 1, mv(sl, V_MU));
                }
                if (sample) ev[3 * nsamples + 1]->record(stream);
                apply_S(V_MU);
                for (Slab<T>& sl : slabs)
                    hipLaunchKernelGGL(dual_init_residual_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, mv(sl, V_G), sl.red.p + 1, mv(sl, V_R), sl.sc.p, 0);
                if (sample) {
                    ev[3 * nsamples + 2]->record(stream);
                    nsamples++;
                }
                it++;
                HIPCHK(hipGetLastError());
                // The final stage is queued behind the pass without waiting for its residual: one host round trip per solve instead of two.  Should the pass
                // not have converged (not seen on any test case: the first pass is the solution), refinement passes follow and the final stage runs again.
                finish();
                finished = true;
                rr0 = h_pinned[SC_RR0];
                rr = h_pinned[SC_RR];
                if (!std::isfinite(rr) || !std::isfinite(rr0)) breakdown = true;
                else if (rr <= o.tol * o.tol * rr0) converged = true;
                else if (rr_prev >= 0. && rr > 0.0625 * rr_prev && rr <= 1e-18 * rr0) converged = true;   // stagnated at the rounding floor (residuals are squared here)
                rr_prev = rr;
                log("[shm] dual (direct) pass=%d rel_res=%.3e", it, rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.);
            }
        }
        while (!direct && it < o.max_iters && !converged && !breakdown) {
            const int batch_end = std::min(o.max_iters, it + check_every);
            for (; it < batch_end; it++) {
                const bool sample = st && nsamples < kMaxSamples;
                if (!dense_S) scatter(V_P, ARR_P, 0);
                if (sample) ev[3 * nsamples]->record(stream);
                if (dense_S) apply_S(V_P);
                else if (sparse_ok) launch_precond_sparse(ARR_P, ARR_Z);
                else launch_precond(false, ARR_P, ARR_Z);
                if (sample) ev[3 * nsamples + 1]->record(stream);
                if (!dense_S) gather(ARR_Z);
                for (Slab<T>& sl : slabs)
                    hipLaunchKernelGGL(dual_update_kernel, dim3(1), dim3(kDualBlock), 0, stream, m, sl.red.p + 1, mv(sl, V_P), mv(sl, V_MU), mv(sl, V_R), sl.sc.p);
                precondition(0);
                if (sample) {
                    ev[3 * nsamples + 2]->record(stream);
                    nsamples++;
                }
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            rr0 = h_pinned[SC_RR0];
            rr = h_pinned[SC_RR];
#ifdef SHM_EXPERIMENT_KNOBS
            static const int force_iters = knob("SHM_DUAL_FORCE_ITERS") ? atoi(knob("SHM_DUAL_FORCE_ITERS")) : 0;  // timing experiments only
#else
            const int force_iters = 0;
#endif
            if (force_iters > 0) converged = it >= force_iters;
            else if (!std::isfinite(rr) || !std::isfinite(rr0) || !std::isfinite(h_pinned[SC_RZ])) breakdown = true;
            else if (rr <= o.tol * o.tol * rr0) converged = true;
            log("[shm] dual it=%d rel_res=%.3e", it, rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.);
        }
        if (!finished) finish();
        have_phi = true;
        have_div = false;  // r still holds b, but q (= phi now) and p were reused
        if (st) {
            memset(st, 0, sizeof *st);
            st->n = n;
            st->m = m;
            st->S = S;
            st->iters = it;
            st->rel_residual = rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.;
            // the reference shifts phi_ref = -x_kkt = -(x0 - mean(A x0)); report ITS shift: shift(-x0) + mean(A x0)
            st->shift = h_pinned[SC_SHIFT] + h_pinned[SC_AXSUM] / (double)m;
            st->ms_conv = elapsed(e_start, e_conv);
            st->ms_div = elapsed(e_conv, e_div);
            HIPCHK(hipEventSynchronize(e_s2b.e));
            st->ms_setup = last_setup_wall_ms;  // host wall time from the start of the set-up to "(A A^T)^-1 ready" (it runs beside Step 1)
            st->ms_wait_setup = elapsed(e_div, e_setup);
            st->ms_pcg = elapsed(e_setup, e_pcg);
            st->ms_shift = elapsed(e_pcg, e_end);
            st->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
            double a_dct = 0., a_rest = 0.;
            for (int a = 0; a < nsamples; a++) {
                a_dct += elapsed(*ev[3 * a], *ev[3 * a + 1]);
                a_rest += elapsed(*ev[3 * a + 1], *ev[3 * a + 2]);
            }
            st->ms_precond_avg = nsamples ? a_dct / nsamples : 0.;
            st->ms_project_avg = nsamples ? a_rest / nsamples : 0.;  // gather, m-vector algebra, 2 G^-1 mat-vecs, B mat-vec
            st->kernel_samples = nsamples;
            st->preconditioner = SHM_PRECOND_DCT;
            st->solver = total_slabs > 1 ? SHM_SOLVER_DUAL_SLABS : SHM_SOLVER_DUAL;
            if (direct) {   // S^-1 and S once per pass (plus S^-1 1 once per solve)
                st->bytes_per_iter = 2.0 * (double)m * m * sizeof(double);
                st->cg_form = 2;
            } else if (dense_S) {  // S (fp64) once, the single-precision G^-1 twice
                st->bytes_per_iter = (double)m * m * (sizeof(double) + 2.0 * sizeof(float));
                st->cg_form = 3;
            } else if (sparse_ok) {  // bytes the five sparse sweeps actually move: active x tiles, active planes (y sweeps and the masked z sweep)
                const double L = dct_lines_for(log2n, (int)sizeof(TP)), tile_bytes = L * n * sizeof(TP);
                const double planes_active = (double)slabs[0].n_act_y / (n / L);
                st->bytes_per_iter = 4.0 * slabs[0].n_act_x * tile_bytes + 4.0 * slabs[0].n_act_y * tile_bytes + 2.0 * planes_active * n * n * sizeof(TP);
            } else {
                st->bytes_per_iter = (double)N * (3.0 * sizeof(T) + 8.0 * sizeof(TP)) - (double)N * sizeof(T);  // five dense DCT sweeps, no r.z read
            }
        }
        if (breakdown) throw Error(SHM_ERR_BREAKDOWN, fmt("dual CG broke down at iteration %d (rr=%g, rr0=%g)", it, rr, rr0));
        if (!converged) throw Error(SHM_ERR_NOCONV, fmt("dual CG: max_iters=%d reached, rel. residual %.3e > tol %.1e", o.max_iters,
                                                          std::sqrt(std::fabs(rr / rr0)), o.tol));
    }

    // ------------------------------------------------------------------------------------------
    // Several processes, dual solver: Steps 1-2 and the divergence ran on this rank's z-slabs (the bulk of the work, perfectly
    // parallel); the right-hand side b = D^T Y is gathered from all ranks (one N-vector in all, grouped ncclSend/ncclRecv over
    // xGMI) and every rank runs the single-GPU dual solve on the whole grid, whose constraint set-up overlapped Step 1 as usual.
    // Each rank keeps phi for its own planes.  (The dual iteration works on m-vectors and a handful of sparse transforms; it is
    // latency-bound on one GPU already, and its distributed form -- SHM_SOLVER_DUAL_SLABS -- pays two all-to-alls per iteration.)
    void solve_gathered(const shm_opts& o, shm_stats* st, Event& e_start, Event& e_conv, Event& e_div, std::chrono::steady_clock::time_point wall0) {
        Solver<T>& F = *full;
        Slab<T>& fs = F.slabs[0];
        Event c_s2a, c_s2b, e_gather, f_start, f_setup;
        c_s2a.record(F.stream2);
        F.conv_est_total_ms = conv_est_total_ms;   // what the whole-grid solver's set-up can hide behind is this rank's share of Step 1
        F.conv_tiered = conv_tiered || conv_tiered32;               // ... and beside WHICH Step-1 kernel it runs: the tiered one leaves room only for the narrow GEMM shape (round 4: the
                                                   // whole-grid solver never learnt this and queued the 256-register shape, which waits for Step 1's persistent waves to end)
        F.dual_direct_requested = true;
        F.dual_form_req = dual_form_req;
        F.build_constraints();
        F.dual_direct_requested = false;  // on the whole-grid solver's set-up stream: overlaps this rank's Step-1 kernel
        c_s2b.record(F.stream2);
        F.setup_precond();
        const size_t plane = (size_t)n * n;
        for (Slab<T>& sl : slabs)
            HIPCHK(hipMemcpyAsync(fs.r.p + (size_t)(sl.k0 + 1) * plane, sl.r.p + plane, sl.nown * sizeof(T), hipMemcpyDeviceToDevice, stream));
        if (comm) {
            Rccl& R = Rccl::get();
            const int dt = sizeof(T) == 8 ? Rccl::kFloat64 : Rccl::kFloat32;
            R.chk(R.GroupStart(), "ncclGroupStart");
            for (int peer = 0; peer < cfg.world; peer++) {
                if (peer == cfg.rank) continue;
                for (Slab<T>& sl : slabs) R.chk(R.Send(sl.r.p + plane, sl.nown, dt, peer, comm, stream), "ncclSend(gather b)");
                for (int b = 0; b < cfg.local_slabs; b++) {  // the peer sends its slabs in this order
                    int32_t k0, k1;
                    slab_range(peer * cfg.local_slabs + b, &k0, &k1);
                    R.chk(R.Recv(fs.r.p + (size_t)(k0 + 1) * plane, (size_t)(k1 - k0) * plane, dt, peer, comm, stream), "ncclRecv(gather b)");
                }
            }
            R.chk(R.GroupEnd(), "ncclGroupEnd");
        }
        e_gather.record(stream);
        HIPCHK(hipStreamWaitEvent(F.stream, e_gather.e, 0));
        f_start.record(F.stream);
        f_setup.record(F.stream);
        F.have_div = true;
        fs.div_sum_blocks = 0;   // (b came from the ranks' slabs: no partial sums of the whole grid)
        shm_stats cst;
        memset(&cst, 0, sizeof cst);
        // "did not converge" still leaves phi (include/shm_grid.h: SHM_ERR_NOCONV): finish the hand-over, then report it
        bool noconv = false;
        std::string noconv_msg;
        try {
            F.solve_dual(o, &cst, f_start, f_start, f_start, f_setup, c_s2a, c_s2b, wall0);  // synchronises F.stream; phi = F.q
        } catch (const Error& e) {
            if (e.code != SHM_ERR_NOCONV) throw;
            noconv = true;
            noconv_msg = e.what();
        }
        for (Slab<T>& sl : slabs)
            HIPCHK(hipMemcpyAsync(sl.q.p + plane, fs.q.p + (size_t)(sl.k0 + 1) * plane, sl.nown * sizeof(T), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        have_phi = true;
        if (st) {
            *st = cst;
            st->ms_conv = elapsed(e_start, e_conv);
            st->ms_div = elapsed(e_conv, e_div);
            st->ms_wait_setup = elapsed(e_div, e_gather);  // here: the gather of b (the set-up was waited for on the host before it)
            st->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        }
        if (noconv) throw Error(SHM_ERR_NOCONV, noconv_msg);
    }

    // ------------------------------------------------------------------------------------------
    // Projected (preconditioned) CG with the fused sweeps of shm_cg_fused.hip.h -- same recurrence as the classic loop in solve():
    //      x=0; r=P b; z=P M^-1 r; p0=-z;
    //      loop k { a_k = rho_k / (p_k . K p_k);  r = P(r + a_k K p_k);  z = P M^-1 r;  rho_{k+1} = r.z;  p_{k+1} = -z + (rho_{k+1}/rho_k) p_k }
    //      x = sum_k a_k p_k, accumulated two directions at a time (p_k lives in buffer k & 1: sl.p / sl.q)
    // N-sized launches per iteration: RES (r += a K p, ||r||^2), DIR (p' = -z + b p, p'.K p'), and x_update2 on odd k: 8 N T bytes.
    void solve_primal_fused(const shm_opts& o, shm_stats* st, bool pre, Event& e_start, Event& e_conv, Event& e_div, Event& e_setup, Event& e_s2a,
                            Event& e_s2b, std::chrono::steady_clock::time_point wall0) {
        Event e_pcg, e_end;
        std::vector<int> nparts(slabs.size()), zparts(slabs.size(), 0), kparts(slabs.size(), 0);
        fold_pq = total_slabs == 1 && !comm && knob("SHM_CG_NO_PQFOLD") == nullptr;   // (A/B knob: finalize_sum_kernel after every DIR sweep, as until round 4)
        fold_pq_np = 0;
        auto dirbuf = [&](Slab<T>& sl, int k) { return (k & 1) ? sl.q.p : sl.p.p; };
        const int zsel = pre ? ARR_Z : ARR_R;
        for (size_t s = 0; s < slabs.size(); s++) {
            Slab<T>& sl = slabs[s];
            HIPCHK(hipMemsetAsync(sl.x.p, 0, sl.ntot * sizeof(T), stream));
            nparts[s] = stream_grid(sl);
            if (vec == 1) launch_norm2<1>(sl, sl.r.p, nparts[s]);
            else launch_norm2<vec_width<T>()>(sl, sl.r.p, nparts[s]);
        }
        launch_projection(nparts, false, 2);
        if (pre) {
            const int np = launch_precond(true);
            for (auto& v : zparts) v = np;
            launch_projection(zparts, true, 0);
        }
        // Several slabs: the ghost planes of z travel on a second stream while the z chunks that do not touch them are swept (the first and the
        // last chunk of every slab follow once the planes have arrived).  One host thread issues everything in the same order on every rank; RCCL
        // orders the operations of one communicator across streams itself.  SHM_HALO_SERIAL: exchange first, then one sweep (A/B knob).
        // With a communicator (several processes) the overlap is OPT-IN (SHM_HALO_OVERLAP=1) until it has run on real multi-rank RCCL: it relies on RCCL
        // serialising the send/recv of stream_h against the all-reduces the same communicator issues on `stream`, which only the librccl double and the
        // loop-back transport (several slabs in one process: device copies) have exercised so far.  Read per solve: the tests flip it inside one process.
        const bool halo_serial = knob("SHM_HALO_SERIAL") != nullptr || (comm != nullptr && knob("SHM_HALO_OVERLAP") == nullptr);
        bool overlap = total_slabs > 1 && !halo_serial;
        for (Slab<T>& sl : slabs) overlap = overlap && fused_cfg(sl).zchunks >= 3;
        if (overlap && !stream_h) HIPCHK(hipStreamCreateWithFlags(&stream_h, hipStreamNonBlocking));
        Event e_ready, e_halo;
        auto run_dir = [&](int k, int slot_old, int slot_new, int init) {  // p_{k+1} (init: p_0) from z and p_k; leaves p'.Kp' in pq
            auto sweep = [&](int part) {
                for (size_t s = 0; s < slabs.size(); s++) {
                    Slab<T>& sl = slabs[s];
                    const int out = init ? 0 : k + 1;
                    kparts[s] = launch_fused<CGF_DIR>(sl, slot_old, slot_new, init, pre ? 0 : 1, 0, arr(sl, zsel), dirbuf(sl, k), dirbuf(sl, out), part);
                }
            };
            if (!overlap) {
                if (total_slabs > 1) halo_exchange(zsel);
                sweep(FUSED_ALL);
                return;
            }
            e_ready.record(stream);                                   // z is final (projection done)
            HIPCHK(hipStreamWaitEvent(stream_h, e_ready.e, 0));
            sweep(FUSED_INTERIOR);
            halo_exchange(zsel, stream_h);
            e_halo.record(stream_h);
            HIPCHK(hipStreamWaitEvent(stream, e_halo.e, 0));
            sweep(FUSED_BOUNDARY);
        };
        auto finalize_pq = [&]() {
            if (fold_pq) {   // the RES sweep sums the partials itself (cg_fused_kernel)
                fold_pq_np = kparts[0];
                return;
            }
            for (size_t s = 0; s < slabs.size(); s++)
                hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, slabs[s].partials.p, kparts[s], slabs[s].pq.p);
            allreduce(1, 1);
        };
        run_dir(0, SC_RHO_A, SC_RHO_A, 1);
        finalize_pq();
        HIPCHK(hipGetLastError());

        // Per-kernel durations are sampled on a few iterations only: a sampled iteration keeps everything on one stream (no projection beside the x update) and records
        // eight events, i.e. it runs ~40 us slower than the others -- 32 of them were 16 % of the 200-iteration stencil-PCG leg of bench.py and 40 % of a DCT-preconditioned
        // solve (round 5: 12; the kernels are steady, their averages do not move)
        const int kMaxSamples = 12, kEvPer = 8;
        std::vector<std::unique_ptr<Event>> ev;
        if (st) for (int a = 0; a < kEvPer * kMaxSamples; a++) ev.emplace_back(new Event());
        int nsamples = 0;
        const int sample_stride = pre ? 2 : 8;
        static const bool no_xoverlap = knob("SHM_CG_NO_XOVERLAP") != nullptr;   // A/B knob
        // (512^3, same box: rocker fp64 0.706 -> 0.717, fp32 0.680 -> 0.691 of the roofline, bunny fp64 0.720 -> 0.725, fp32 0.710 -> 0.722; profiles/r04_projection.txt)
        const bool xoverlap = !no_xoverlap && total_slabs == 1 && !comm;
        // Round 5: with the overlap on and no preconditioner, x is updated on HALF the grid in EVERY iteration instead of on all of it in every other one -- the upper half
        // on odd k, the lower half on even k, each time with the two latest directions (p_{k-1}, p_k: both buffers and both step lengths are valid once RES of
        // iteration k has run), so every a_j p_j reaches every node exactly once (the lower half takes a_0 p_0 alone at k = 0; the half that lags one step behind
        // at the end is flushed after the loop).  Same 4NT per two iterations -- but now EVERY projection (seven short dependent launches, 0.08 ms with the
        // two-level inverse) has 2NT of update to run beside, where the even iterations used to expose theirs in full.
        static const bool no_xsplit = knob("SHM_CG_NO_XSPLIT") != nullptr;   // A/B knob: the round-4 schedule
        const bool xsplit = xoverlap && !pre && !no_xsplit;
        Event e_xfork, e_xjoin;
        // The second stream is the set-up stream (idle during the loop, of the highest priority -- the projection's short kernels get their CUs at once -- and
        // known to run beside `stream`: the whole set-up does).  A stream of its own, of the same or of the lowest priority, was measured to serialise with the
        // main stream in every solver but the first of a process (HIP deals its few hardware queues out per process; the projection then queued up behind the
        // update: 0.31 instead of 0.08 ms per iteration, every sweep 15 % slower).
        hipStream_t const stream_x = stream2;
        int it = 0;
        double rr0 = 0., rr = 0.;
        bool converged = false, breakdown = false;
        while (it < o.max_iters && !converged && !breakdown) {
            const int batch_end = std::min(o.max_iters, it + o.check_every);
            for (; it < batch_end; it++) {
                const int slot_old = (it & 1) ? SC_RHO_B : SC_RHO_A, slot_new = (it & 1) ? SC_RHO_A : SC_RHO_B;
                const bool sample = st && nsamples < kMaxSamples && (it % sample_stride == 1);
                auto mark = [&](int k) {
                    if (sample) ev[kEvPer * nsamples + k]->record(stream);
                };
                mark(0);
                for (size_t s = 0; s < slabs.size(); s++)
                    nparts[s] = launch_fused<CGF_RES>(slabs[s], slot_old, slot_new, 0, 0, (it & 1) ? SC_ALPHA_B : SC_ALPHA_A, (const T*)nullptr,
                                                      dirbuf(slabs[s], it), (T*)nullptr);
                mark(1);
                // x += a_{k-1} p_{k-1} + a_k p_k (odd k) needs nothing of the projection, and the projection's kernels are m-sized and latency-bound: on one GPU the
                // two run side by side (fork after RES, which fixes a_k; join before DIR, which needs the projected residual and overwrites p_{k-1}).  Sampled
                // iterations keep everything on one stream so that the per-kernel durations of shm_stats stay what they say.
                const bool fork_x = xoverlap && ((it & 1) || xsplit) && !sample && !pre;
                if (fork_x) {
                    // the PROJECTION goes to the set-up stream (highest priority: its short kernels get their CUs at once), the update stays on the main stream
                    e_xfork.record(stream);
                    HIPCHK(hipStreamWaitEvent(stream_x, e_xfork.e, 0));
                    launch_projection(nparts, false, 1, stream_x);
                    e_xjoin.record(stream_x);
                    if (xsplit) launch_x_update2(slabs[0], 1, it > 0, nullptr, it & 1);   // (k = 0: the lower half takes a_0 p_0 alone)
                    else launch_x_update2(slabs[0], 1, 1);
                    HIPCHK(hipStreamWaitEvent(stream, e_xjoin.e, 0));
                } else {
                    launch_projection(nparts, false, 1);
                }
                mark(2);
                if (pre) launch_precond(true);
                mark(3);
                if (pre) launch_projection(zparts, true, 0);
                mark(4);
                if (!fork_x && xsplit) launch_x_update2(slabs[0], 1, it > 0, nullptr, it & 1);
                else if (!fork_x && (it & 1))
                    for (Slab<T>& sl : slabs) launch_x_update2(sl, 1, 1);
                mark(5);
                run_dir(it, slot_old, slot_new, 0);
                mark(6);
                finalize_pq();
                mark(7);
                if (sample) nsamples++;
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            rr0 = h_pinned[SC_RR0];
            rr = h_pinned[SC_RR];
            const double rho = h_pinned[(it & 1) ? SC_RHO_B : SC_RHO_A];
            if (!std::isfinite(rr) || !std::isfinite(rr0) || !std::isfinite(rho)) breakdown = true;
            else if (rr <= o.tol * o.tol * rr0) converged = true;
            log("[shm] it=%d rel_res=%.3e", it, rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.);
        }
        if (xsplit) {
            // the half that was not updated in the last iteration is one direction behind: the upper half after an even last index, the lower half after an odd one
            if (it & 1) launch_x_update2(slabs[0], 1, 0, nullptr, 1);
            else if (it > 0) launch_x_update2(slabs[0], 0, 1, nullptr, 0);
        } else if (it & 1)  // the last direction has an even index: its step is still missing from x
            for (Slab<T>& sl : slabs) launch_x_update2(sl, 1, 0);
        e_pcg.record(stream);
        launch_shift_and_phi();
        e_end.record(stream);
        HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        have_phi = true;
        have_div = false;  // q (a direction buffer, then phi) and r were reused
        const auto wall1 = std::chrono::steady_clock::now();
        if (st) {
            memset(st, 0, sizeof *st);
            st->n = n;
            st->m = m;
            st->S = S;
            st->iters = it;
            st->rel_residual = rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.;
            st->shift = h_pinned[SC_SHIFT];
            st->ms_conv = elapsed(e_start, e_conv);
            st->ms_div = elapsed(e_conv, e_div);
            HIPCHK(hipEventSynchronize(e_s2b.e));
            st->ms_setup = last_setup_wall_ms;  // host wall time from the start of the set-up to "(A A^T)^-1 ready" (it runs beside Step 1)
            st->ms_wait_setup = elapsed(e_div, e_setup);
            st->ms_pcg = elapsed(e_setup, e_pcg);
            st->ms_shift = elapsed(e_pcg, e_end);
            st->ms_total = std::chrono::duration<double, std::milli>(wall1 - wall0).count();
            double acc[5] = {0, 0, 0, 0, 0};
            for (int a = 0; a < nsamples; a++) {
                auto el = [&](int i, int j) { return (double)elapsed(*ev[kEvPer * a + i], *ev[kEvPer * a + j]); };
                acc[0] += el(5, 6);            // DIR sweep (incl. the halo exchange of z with several slabs)
                acc[1] += el(0, 1);            // RES sweep
                acc[2] += el(1, 2) + el(3, 4); // projections
                acc[3] += el(4, 5);            // x_update2 (sampled iterations are odd: one launch each)
                acc[4] += el(2, 3);            // DCT
            }
            const double inv = nsamples ? 1. / nsamples : 0.;
            st->ms_stencil_avg = acc[0] * inv;
            st->ms_update_xr_avg = acc[1] * inv;
            st->ms_project_avg = acc[2] * inv;
            st->ms_update_p_avg = acc[3] * inv;
            st->ms_precond_avg = pre ? acc[4] * inv : 0.;
            st->kernel_samples = nsamples;
            st->preconditioner = pre ? SHM_PRECOND_DCT : SHM_PRECOND_NONE;
            st->solver = SHM_SOLVER_PRIMAL;
            st->cg_form = xsplit ? 4 : 1;
            // 3NT (DIR) + 3NT (RES) + 4NT every other iteration (x_update2; cg_form 4: 2NT -- half the grid -- every iteration); the five DCT sweeps move 3T + 8TP more
            st->bytes_per_iter = 8.0 * (double)N * sizeof(T) + (pre ? (double)N * (3.0 * sizeof(T) + 8.0 * sizeof(TP)) : 0.);
        }
        if (breakdown) throw Error(SHM_ERR_BREAKDOWN, fmt("projected CG broke down at iteration %d (rr=%g, rr0=%g)", it, rr, rr0));
        if (!converged) throw Error(SHM_ERR_NOCONV, fmt("projected CG: max_iters=%d reached, rel. residual %.3e > tol %.1e", o.max_iters,
                                                          rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0., o.tol));
    }

    // ------------------------------------------------------------------------------------------
    // (node, source) pairs Step 1 evaluated in the last solve, per arithmetic: what bench.py computes the Step-1 roofline fraction from
    void report_pairs(shm_stats* st) noexcept {
        if (!st || !d_pair_counters.p) return;
        unsigned long long h[3] = {0, 0, 0};
        if (hipMemcpy(h, d_pair_counters.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
        st->pairs_fp64 = (double)h[0];
        st->pairs_fp32 = (double)h[1];
        st->pairs_redone = (double)h[2];
        st->conv_launches = conv_launches_last;
    }
    void solve(const shm_opts& o_in, shm_stats* st) override {
        struct Fin {
            Solver* s;
            shm_stats* st;
            ~Fin() { s->report_pairs(st); }
        } fin{this, st};
        solve_impl(o_in, st);
    }
    void solve_impl(const shm_opts& o_in, shm_stats* st) {
        need_problem();
        HIPCHK(hipSetDevice(cfg.device));
        shm_opts o = o_in;
        bool pre = false;
        if (o.preconditioner == SHM_PRECOND_DCT) {
            if (!precond_available()) throw Error(SHM_ERR_INVALID, "DCT preconditioner needs n = 2^k in [16,1024] and a power-of-two number of EQUAL z-slabs dividing n (shm_config.slab_plan = SHM_SLAB_PLAN_EQUAL); a single z-slab serves any n in [4,1024]");
            pre = true;
        } else if (o.preconditioner == SHM_PRECOND_AUTO) {
            pre = precond_available();
        } else if (o.preconditioner != SHM_PRECOND_NONE) {
            throw Error(SHM_ERR_INVALID, "unknown preconditioner");
        }
        if (!(o.tol > 0.)) o.tol = sizeof(T) == 8 ? 1e-8 : 1e-5;
        if (o.max_iters <= 0) o.max_iters = 20 * n;
        if (o.check_every <= 0) o.check_every = pre ? 4 : 32;

        Event e_start, e_conv, e_div, e_setup, e_pcg, e_end, e_s2a, e_s2b;
        const auto wall0 = std::chrono::steady_clock::now();
        if (o.dual_form < SHM_DUAL_AUTO || o.dual_form > SHM_DUAL_THROUGH_GRID) throw Error(SHM_ERR_INVALID, "unknown dual_form");
        if (o.step1_budget != 0. && !(o.step1_budget >= 1e-12 && o.step1_budget <= 1e-3)) throw Error(SHM_ERR_INVALID, "step1_budget must lie in [1e-12, 1e-3] (0: default 1e-8)");
        dual_form_req = o.dual_form;
        step1_budget = o.step1_budget;
        select_step1_arith(o.step1_arith);
        e_start.record(stream);
        // the arrival counters of the projection's ticketed reductions are "zero between launches" by the last workgroup's reset: a launch cut short (a fault) would
        // leave them non-zero for the rest of the process, so every solve starts from zero.  Projections on the two streams share these counters and the
        // partial-sum scratch: they are event-serialised (fork / join around the x update), never concurrent.
        for (Slab<T>& sl : slabs)
            if (sl.proj_ticket.p) HIPCHK(hipMemsetAsync(sl.proj_ticket.p, 0, sizeof(unsigned), stream));
        o_fast_hint = o.fast_integration != 0;
        launch_conv();
        e_conv.record(stream);
        launch_div(o.scrub_nonfinite);
        e_div.record(stream);
        if (o.fast_integration) {
            solve_fast(o, st, e_start, e_conv, e_div, wall0);
            return;
        }
        // Several ranks.  Round 6: where S fits (S <= 16384 sources bound the rows) and the grid is one of the sizes BASELINE.json names (256^3 ... 512^3, four or more equal
        // power-of-two slabs), AUTO takes the slab-distributed explicit-S forms of solve_dual -- S and its inverse replicated beside every rank's Step 1, K^+ on the slabs: no gather of
        // D^T Y, no whole-grid solve per rank.  Everything else (larger constraint sets, 1024^3, odd sizes, weighted plans; SHM_SOLVER_DUAL) keeps the gathered solve.
        // (from four ranks on: with two, each all-to-all crosses ONE xGMI link with a quarter of the grid, four times per solve, where the gather crosses it once with half --
        // tools/scaling_model.py, link constants assumed: 115 against 107 ms at 512^3 on two ranks, 54 against 58 on four, 30 against 36 on eight)
        const bool slab_forms_auto = full_wanted() && o.solver == SHM_SOLVER_AUTO && o.preconditioner != SHM_PRECOND_NONE && fft_available() && total_slabs >= 4 && n >= 256 && n <= 512 && S <= 16384 &&
                                     o.dual_form != SHM_DUAL_THROUGH_GRID && knob("SHM_MULTI_GATHERED") == nullptr;
        if (full_wanted() && !slab_forms_auto && (o.solver == SHM_SOLVER_AUTO || o.solver == SHM_SOLVER_DUAL) && o.preconditioner != SHM_PRECOND_NONE) {
            ensure_full();
            solve_gathered(o, st, e_start, e_conv, e_div, wall0);
            return;
        }
        static const bool setup_alone = knob("SHM_SETUP_ALONE") != nullptr;  // measurement knob: wait for Step 1 first, so that shm_stats.ms_setup is
        if (setup_alone) HIPCHK(hipStreamSynchronize(stream));                  // the set-up's time on an otherwise idle GPU (tools/scaling_model.py)
        e_s2a.record(stream2);
        dual_direct_requested = (o.solver == SHM_SOLVER_DUAL || o.solver == SHM_SOLVER_DUAL_SLABS || (o.solver == SHM_SOLVER_AUTO && o.preconditioner != SHM_PRECOND_NONE)) && precond_available();
        build_constraints();  // on stream2: overlaps the Step-1 kernel; returns once (A A^T)^-1 (or S^-1, for the direct dual solve) is ready
        dual_direct_requested = false;
        e_s2b.record(stream2);
        bool dual = false;
        if (o.solver == SHM_SOLVER_DUAL || o.solver == SHM_SOLVER_DUAL_SLABS) {
            if (!precond_available()) throw Error(SHM_ERR_INVALID, "the dual solver needs the DCT: n = 2^k in [16,1024] and a power-of-two number of EQUAL z-slabs dividing n (shm_config.slab_plan = SHM_SLAB_PLAN_EQUAL); a single z-slab serves any n in [4,1024]");
            dual = true;
        } else if (o.solver == SHM_SOLVER_AUTO) {
            dual = precond_available() && o.preconditioner != SHM_PRECOND_NONE;
        } else if (o.solver != SHM_SOLVER_PRIMAL) {
            throw Error(SHM_ERR_INVALID, "unknown solver");
        }
        if (pre || dual) setup_precond();
        e_setup.record(stream);
        if (dual) {
            solve_dual(o, st, e_start, e_conv, e_div, e_setup, e_s2a, e_s2b, wall0);
            return;
        }

        if (fused_available()) {
            solve_primal_fused(o, st, pre, e_start, e_conv, e_div, e_setup, e_s2a, e_s2b, wall0);
            return;
        }
        // ---- projected (preconditioned) CG, SURVEY 7.3 (the classic four-kernel loop: grids wider than 8 waves of vector lanes, or SHM_CG_CLASSIC):
        //      x=0; r=P b; z=P M^-1 r; p=-z; loop { q=Kp; a=rho/p.q; x+=a p; r=P(r+a q); z=P M^-1 r; rho'=r.z; p=-z+(rho'/rho) p }
        std::vector<int> nparts(slabs.size()), zparts(slabs.size(), 0);
        for (size_t s = 0; s < slabs.size(); s++) {
            Slab<T>& sl = slabs[s];
            HIPCHK(hipMemsetAsync(sl.x.p, 0, sl.ntot * sizeof(T), stream));
            nparts[s] = stream_grid(sl);
            if (vec == 1) launch_norm2<1>(sl, sl.r.p, nparts[s]);
            else launch_norm2<vec_width<T>()>(sl, sl.r.p, nparts[s]);
        }
        launch_projection(nparts, false, 2);
        if (pre) {
            const int np = launch_precond(true);
            for (auto& v : zparts) v = np;
            launch_projection(zparts, true, 0);
        }
        update_p_all(SC_RHO_A, SC_RHO_A, 1, pre, nparts);
        HIPCHK(hipGetLastError());

        // sampled per-kernel timing (events on the solver's stream; 8 events per sampled iteration)
        const int kMaxSamples = 32, kEvPer = 8;
        std::vector<std::unique_ptr<Event>> ev;
        if (st) for (int a = 0; a < kEvPer * kMaxSamples; a++) ev.emplace_back(new Event());
        int nsamples = 0;
        const int sample_stride = pre ? 2 : 8;

        int it = 0;
        double rr0 = 0., rr = 0.;
        bool converged = false, breakdown = false;
        while (it < o.max_iters && !converged && !breakdown) {
            const int batch_end = std::min(o.max_iters, it + o.check_every);
            for (; it < batch_end; it++) {
                const int slot_old = (it & 1) ? SC_RHO_B : SC_RHO_A, slot_new = (it & 1) ? SC_RHO_A : SC_RHO_B;
                halo_exchange_p();
                const bool sample = st && nsamples < kMaxSamples && (it % sample_stride == 1);
                auto mark = [&](int k) {
                    if (sample) ev[kEvPer * nsamples + k]->record(stream);
                };
                mark(0);
                for (Slab<T>& sl : slabs) launch_stencil(sl);
                mark(1);
                for (Slab<T>& sl : slabs) {
                    const StencilLaunch L = stencil_dims(sl);
                    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(kBlock), 0, stream, sl.partials.p, (int)L.grid.x, sl.pq.p);
                }
                allreduce(1, 1);
                mark(2);
                for (size_t s = 0; s < slabs.size(); s++) {
                    if (vec == 1) launch_update_xr<1>(slabs[s], slot_old, nparts[s]);
                    else launch_update_xr<vec_width<T>()>(slabs[s], slot_old, nparts[s]);
                }
                mark(3);
                launch_projection(nparts, false, 1);
                mark(4);
                if (pre) launch_precond(true);
                mark(5);
                if (pre) launch_projection(zparts, true, 0);
                mark(6);
                update_p_all(slot_old, slot_new, 0, pre, nparts);
                mark(7);
                if (sample) nsamples++;
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            rr0 = h_pinned[SC_RR0];
            rr = h_pinned[SC_RR];
            const double rho = h_pinned[(it & 1) ? SC_RHO_B : SC_RHO_A];
            if (!std::isfinite(rr) || !std::isfinite(rr0) || !std::isfinite(rho)) breakdown = true;
            else if (rr <= o.tol * o.tol * rr0) converged = true;  // rr = ||r'||^2 - u.w may round slightly below 0 at convergence
            log("[shm] it=%d rel_res=%.3e", it, std::sqrt(std::fabs(rr / rr0)));
        }
        e_pcg.record(stream);

        // ---- shift + phi (:110-111)
        launch_shift_and_phi();
        e_end.record(stream);
        HIPCHK(hipMemcpyAsync(h_pinned, slabs[0].sc.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        have_phi = true;
        const auto wall1 = std::chrono::steady_clock::now();

        if (st) {
            memset(st, 0, sizeof *st);
            st->n = n;
            st->m = m;
            st->S = S;
            st->iters = it;
            st->rel_residual = rr0 > 0. ? std::sqrt(std::fabs(rr / rr0)) : 0.;
            st->shift = h_pinned[SC_SHIFT];
            st->ms_conv = elapsed(e_start, e_conv);
            st->ms_div = elapsed(e_conv, e_div);
            HIPCHK(hipEventSynchronize(e_s2b.e));
            st->ms_setup = last_setup_wall_ms;  // host wall time from the start of the set-up to "(A A^T)^-1 ready" (it runs beside Step 1)  // runs on stream2 concurrently with ms_conv
            st->ms_wait_setup = elapsed(e_div, e_setup);
            st->ms_pcg = elapsed(e_setup, e_pcg);
            st->ms_shift = elapsed(e_pcg, e_end);
            st->ms_total = std::chrono::duration<double, std::milli>(wall1 - wall0).count();
            double acc[5] = {0, 0, 0, 0, 0};
            for (int a = 0; a < nsamples; a++) {
                auto el = [&](int i, int j) { return (double)elapsed(*ev[kEvPer * a + i], *ev[kEvPer * a + j]); };
                acc[0] += el(0, 1);
                acc[1] += el(2, 3);
                acc[2] += el(3, 4) + el(5, 6);
                acc[3] += el(6, 7);
                acc[4] += el(4, 5);
            }
            const double inv = nsamples ? 1. / nsamples : 0.;
            st->ms_stencil_avg = acc[0] * inv;
            st->ms_update_xr_avg = acc[1] * inv;
            st->ms_project_avg = acc[2] * inv;
            st->ms_update_p_avg = acc[3] * inv;
            st->ms_precond_avg = pre ? acc[4] * inv : 0.;
            st->kernel_samples = nsamples;
            st->preconditioner = pre ? SHM_PRECOND_DCT : SHM_PRECOND_NONE;
            st->solver = SHM_SOLVER_PRIMAL;
            // 11NT for the CG sweeps; the five DCT sweeps move 3T + 8TP more (read r twice + write z, 4 in-place sweeps of W)
            st->bytes_per_iter = 11.0 * (double)N * sizeof(T) + (pre ? (double)N * (3.0 * sizeof(T) + 8.0 * sizeof(TP)) : 0.);
        }
        if (breakdown) throw Error(SHM_ERR_BREAKDOWN, fmt("projected CG broke down at iteration %d (rr=%g, rr0=%g)", it, rr, rr0));
        if (!converged) throw Error(SHM_ERR_NOCONV, fmt("projected CG: max_iters=%d reached, rel. residual %.3e > tol %.1e", o.max_iters,
                                                          std::sqrt(std::fabs(rr / rr0)), o.tol));
    }

    // ------------------------------------------------------------------------------------------
    // global planes [ka, kb) of a field, clipped to what this process owns, packed in ascending plane order
    void copy_planes_to_host(int which, int ka, int kb, double* out) {
        size_t off = 0;
        for (Slab<T>& sl : slabs) {
            const int a = std::max(ka, sl.k0), b = std::min(kb, sl.k1);
            if (b <= a) continue;
            const T* srcp = nullptr;
            switch (which) {
                case SHM_FIELD_Y0: srcp = sl.Y0.p; break;
                case SHM_FIELD_Y1: srcp = sl.Y1.p; break;
                case SHM_FIELD_Y2: srcp = sl.Y2.p; break;
                case SHM_FIELD_DIV: srcp = sl.r.p; break;
                default: srcp = sl.q.p; break;
            }
            srcp += sl.plane * (size_t)(a - sl.k0 + 1);
            const size_t cnt = sl.plane * (size_t)(b - a);
            if (sizeof(T) == 8) {
                HIPCHK(hipMemcpyAsync(out + off, srcp, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
            } else {
                DevArray<double> tmp;
                tmp.alloc(cnt);
                hipLaunchKernelGGL((convert_kernel<T, double>), dim3(grid_for(cnt, 4096)), dim3(kBlock), 0, stream, cnt, srcp, tmp.p);
                HIPCHK(hipMemcpyAsync(out + off, tmp.p, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
            off += cnt;
        }
        HIPCHK(hipStreamSynchronize(stream));
    }
    void copy_owned_to_host(int which, double* out) {
        size_t off = 0;
        for (Slab<T>& sl : slabs) {
            const T* srcp = nullptr;
            switch (which) {
                case SHM_FIELD_Y0: srcp = sl.Y0.p; break;
                case SHM_FIELD_Y1: srcp = sl.Y1.p; break;
                case SHM_FIELD_Y2: srcp = sl.Y2.p; break;
                case SHM_FIELD_DIV: srcp = sl.r.p; break;
                default: srcp = sl.q.p; break;
            }
            srcp += sl.plane;
            if (sizeof(T) == 8) {
                HIPCHK(hipMemcpyAsync(out + off, srcp, sl.nown * sizeof(double), hipMemcpyDeviceToHost, stream));
            } else {
                DevArray<double> tmp;
                tmp.alloc(sl.nown);
                hipLaunchKernelGGL((convert_kernel<T, double>), dim3(grid_for(sl.nown, 4096)), dim3(kBlock), 0, stream, sl.nown, srcp, tmp.p);
                HIPCHK(hipMemcpyAsync(out + off, tmp.p, sl.nown * sizeof(double), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
            }
            off += sl.nown;
        }
        HIPCHK(hipStreamSynchronize(stream));
    }

    void owned_planes(int32_t* kb, int32_t* ke) override {
        need_problem();
        if (kb) *kb = slabs.front().k0;
        if (ke) *ke = slabs.back().k1;
    }
    void get_phi(double* out, int32_t* kb, int32_t* ke) override {
        need_problem();
        if (!have_phi) throw Error(SHM_ERR_STATE, "no phi: shm_grid_solve has not completed");
        HIPCHK(hipSetDevice(cfg.device));
        copy_owned_to_host(SHM_FIELD_PHI, out);
        if (kb) *kb = slabs.front().k0;
        if (ke) *ke = slabs.back().k1;
    }

    void get_field(shm_field f, double* out) override {
        need_problem();
        HIPCHK(hipSetDevice(cfg.device));
        if ((f == SHM_FIELD_Y0 || f == SHM_FIELD_Y1 || f == SHM_FIELD_Y2) && !have_conv) throw Error(SHM_ERR_STATE, "Y not computed");
        if (f == SHM_FIELD_DIV && !have_div) throw Error(SHM_ERR_STATE, "divergence not computed");
        if (f == SHM_FIELD_PHI && !have_phi) throw Error(SHM_ERR_STATE, "phi not computed");
        copy_owned_to_host((int)f, out);
    }

    void get_field_planes(shm_field f, int ka, int kb, double* out) override {
        need_problem();
        HIPCHK(hipSetDevice(cfg.device));
        if ((f == SHM_FIELD_Y0 || f == SHM_FIELD_Y1 || f == SHM_FIELD_Y2) && !have_conv) throw Error(SHM_ERR_STATE, "Y not computed");
        if (f == SHM_FIELD_DIV && !have_div) throw Error(SHM_ERR_STATE, "divergence not computed");
        if (f == SHM_FIELD_PHI && !have_phi) throw Error(SHM_ERR_STATE, "phi not computed");
        if (ka < slabs.front().k0 || kb > slabs.back().k1 || kb < ka) throw Error(SHM_ERR_INVALID, "plane range outside the planes this process owns");
        copy_planes_to_host((int)f, ka, kb, out);
    }

    void upload_owned(const double* in, int which) {
        // host N-vector -> slab array `which` (0: p with ghosts filled from the host copy, 1: r)
        std::vector<T> tmp;
        for (Slab<T>& sl : slabs) {
            tmp.assign(sl.ntot, (T)0);
            const int klo = std::max(0, sl.k0 - 1), khi = std::min(n, sl.k1 + 1);
            for (int k = klo; k < khi; k++)
                for (size_t a = 0; a < sl.plane; a++) tmp[(size_t)(k - sl.k0 + 1) * sl.plane + a] = (T)in[(size_t)k * sl.plane + a];
            T* dst = which == 0 ? sl.p.p : sl.r.p;
            HIPCHK(hipMemcpyAsync(dst, tmp.data(), sl.ntot * sizeof(T), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
    }

    void apply_laplacian(const double* u, double* out) override {
        need_problem();
        if (cfg.world != 1) throw Error(SHM_ERR_INVALID, "apply_laplacian is a single-process test entry point");
        HIPCHK(hipSetDevice(cfg.device));
        upload_owned(u, 0);
        for (Slab<T>& sl : slabs)
            hipLaunchKernelGGL((laplacian_kernel<T>), dim3(grid_for(sl.nown, 4096)), dim3(kBlock), 0, stream, sl.gp, sl.p.p, sl.q.p);
        HIPCHK(hipGetLastError());
        copy_owned_to_host(SHM_FIELD_PHI, out);
        have_phi = false;  // q now holds L u, not phi
    }

    void get_constraints(int64_t* nodes, double* coeffs, int32_t* m_out) override {
        need_problem();
        build_rows();
        for (int r = 0; r < m; r++)
            for (int e = 0; e < 8; e++) {
                nodes[8 * (size_t)r + e] = rows[r].nodes[e];
                coeffs[8 * (size_t)r + e] = rows[r].coeffs[e];
            }
        *m_out = m;
    }

    void get_schur(double* out, int32_t* m_out) override {   // test entry point: the explicit S = A K^+ A^T the dual solver would use (m x m, row-major)
        need_problem();
        if (cfg.world != 1) throw Error(SHM_ERR_INVALID, "get_schur is a single-process test entry point");
        HIPCHK(hipSetDevice(cfg.device));
        build_constraints();
        *m_out = m;
        if (!have_S) throw Error(SHM_ERR_STATE, "no explicit Schur complement for this problem (several slabs, n not a power of two or > 512, or too many rows)");
        HIPCHK(hipMemcpy2DAsync(out, (size_t)m * sizeof(double), Sdense.p, (size_t)mp * sizeof(double), (size_t)m * sizeof(double), (size_t)m, hipMemcpyDeviceToHost, stream2));
        HIPCHK(hipStreamSynchronize(stream2));
    }

    void apply_projector(double* v) override {
        need_problem();
        if (cfg.world != 1) throw Error(SHM_ERR_INVALID, "apply_projector is a single-process test entry point");
        HIPCHK(hipSetDevice(cfg.device));
        build_constraints();
        upload_owned(v, 1);
        std::vector<int> nparts(slabs.size(), 0);
        launch_projection(nparts);
        HIPCHK(hipGetLastError());
        have_div = true;
        copy_owned_to_host(SHM_FIELD_DIV, v);
        have_div = false;
    }

    std::vector<double> iso_vertices;
    std::vector<int64_t> iso_triangles;

    void isosurface(double iso, int method, int64_t* nv, int64_t* nt) override {
        need_problem();
        if (method != SHM_ISO_MARCHING_CUBES && method != SHM_ISO_MARCHING_TETS) throw Error(SHM_ERR_INVALID, fmt("isosurface: unknown method %d", method));
        const bool mc = method == SHM_ISO_MARCHING_CUBES;
        if (!have_phi) throw Error(SHM_ERR_STATE, "no phi: shm_grid_solve has not completed");
        HIPCHK(hipSetDevice(cfg.device));
        halo_exchange(ARR_Q);  // phi lives in q; cells of the top owned plane need the plane above
        iso_vertices.clear();
        iso_triangles.clear();
        std::unordered_map<uint64_t, int64_t> weld;
        for (Slab<T>& sl : slabs) {
            IsoParams P;
            P.n = n; P.nzl = sl.nzl; P.k0 = sl.k0;
            for (int a = 0; a < 3; a++) P.bbox_min[a] = bbox_min[a];
            P.cell = cell; P.iso = iso;
            DevArray<unsigned long long> counter, keys, sortk;
            DevArray<double> pos;
            counter.alloc(1);
            HIPCHK(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), stream));
            const size_t ncells = (size_t)(n - 1) * (n - 1) * (size_t)std::max(0, std::min(sl.nzl, n - 1 - sl.k0));
            const int grid = grid_for(ncells, 8192);
            if (mc) hipLaunchKernelGGL((iso_mc_kernel<T, false>), dim3(grid), dim3(kBlock), 0, stream, P, sl.q.p, counter.p, (double*)nullptr, (unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, 0ULL);
            else hipLaunchKernelGGL((iso_kernel<T, false>), dim3(grid), dim3(kBlock), 0, stream, P, sl.q.p, counter.p, (double*)nullptr, (unsigned long long*)nullptr,
                                    (unsigned long long*)nullptr, 0ULL);
            unsigned long long ntri = 0;
            HIPCHK(hipMemcpyAsync(&ntri, counter.p, sizeof ntri, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (ntri == 0) continue;
            pos.alloc(ntri * 9);
            keys.alloc(ntri * 3);
            sortk.alloc(ntri);
            HIPCHK(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), stream));
            if (mc) hipLaunchKernelGGL((iso_mc_kernel<T, true>), dim3(grid), dim3(kBlock), 0, stream, P, sl.q.p, counter.p, pos.p, keys.p, sortk.p, ntri);
            else hipLaunchKernelGGL((iso_kernel<T, true>), dim3(grid), dim3(kBlock), 0, stream, P, sl.q.p, counter.p, pos.p, keys.p, sortk.p, ntri);
            HIPCHK(hipGetLastError());
            std::vector<double> hpos(ntri * 9);
            std::vector<unsigned long long> hkeys(ntri * 3), hsort(ntri);
            HIPCHK(hipMemcpyAsync(hpos.data(), pos.p, hpos.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(hkeys.data(), keys.p, hkeys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(hsort.data(), sortk.p, hsort.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            // the append order is racy; the (cell, tet, triangle) key restores a deterministic order before welding
            std::vector<size_t> order(ntri);
            for (size_t a = 0; a < ntri; a++) order[a] = a;
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return hsort[x] < hsort[y]; });
            for (size_t a : order)
                for (int c = 0; c < 3; c++) {
                    auto ins = weld.insert({hkeys[a * 3 + c], (int64_t)(iso_vertices.size() / 3)});
                    if (ins.second)
                        for (int b = 0; b < 3; b++) iso_vertices.push_back(hpos[a * 9 + c * 3 + b]);
                    iso_triangles.push_back(ins.first->second);
                }
        }
        if (nv) *nv = (int64_t)(iso_vertices.size() / 3);
        if (nt) *nt = (int64_t)(iso_triangles.size() / 3);
    }

    void get_isosurface(double* vertices, int64_t* triangles) override {
        if (vertices && !iso_vertices.empty()) memcpy(vertices, iso_vertices.data(), iso_vertices.size() * sizeof(double));
        if (triangles && !iso_triangles.empty()) memcpy(triangles, iso_triangles.data(), iso_triangles.size() * sizeof(int64_t));
    }

    void apply_preconditioner(const double* v, double* out) override {
        need_problem();
        if (cfg.world != 1) throw Error(SHM_ERR_INVALID, "apply_preconditioner is a single-process test entry point");
        if (!precond_available()) throw Error(SHM_ERR_INVALID, "DCT preconditioner needs n = 2^k in [16,1024] and a power-of-two number of EQUAL z-slabs dividing n (shm_config.slab_plan = SHM_SLAB_PLAN_EQUAL); a single z-slab serves any n in [4,1024]");
        HIPCHK(hipSetDevice(cfg.device));
        setup_precond();
        upload_owned(v, 1);
        launch_precond(false);
        HIPCHK(hipGetLastError());
        for (Slab<T>& sl : slabs) HIPCHK(hipMemcpyAsync(sl.q.p, sl.z.p, sl.ntot * sizeof(T), hipMemcpyDeviceToDevice, stream));
        copy_owned_to_host(SHM_FIELD_PHI, out);
        have_conv = have_div = have_phi = false;
    }
};

}  // namespace shm
