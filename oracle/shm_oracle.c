/* CPU oracle (plain C) for the regular-grid Signed Heat Method solver.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- never by the product path.
 *
 * PARITY UNPINNED: the reference cannot be built in this image (geometry-central / Eigen / Polyscope
 * submodules empty, no network) and holds no tests or golden vectors.  This file restates the
 * reference's serial loops (citations are into /root/reference) and is pinned against the numpy/scipy
 * oracle that assembles the reference's literal sparse matrices and LU-solves the KKT system
 * (oracle/shm_oracle.py -> the .npz fixtures under tests/golden) and against BASELINE.md's spot values.
 *
 * The constrained solve (signed_heat_grid_solver.cpp:101-108, sparse LU of [[L,A^T],[A,0]]) is
 * restated as projected CG on null(A): mathematically the same solution, and the only
 * feasible form above 64^3.  Everything else follows the reference loop for loop.
 *
 * Node flattening: idx = i + j*n + k*n*n (signed_heat_grid_solver.cpp:505-508).
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC shm_oracle.c -o _build/libshm_oracle.so -lm
 * With OMP_NUM_THREADS=1 (or shmo_set_threads(1)) every loop runs in the reference's serial order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, k) ((size_t)(i) + (size_t)(j) * n + (size_t)(k) * n * n)

void shmo_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t > 0 ? t : 1);
#else
    (void)t;
#endif
}

int shmo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* signed_heat_3d.cpp:45-49 */
static inline double yukawa(double x0, double x1, double x2, double y0, double y1, double y2, double lambda) {
    double d0 = x0 - y0, d1 = x1 - y1, d2 = x2 - y2;
    double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    return exp(-lambda * r) / r;
}

/* Steps 1+2, signed_heat_grid_solver.cpp:48-65 (mesh) / :157-174 (points).
 * pos = barycenters or point positions [3S]; wn = N*A [3S] (the reference forms N*A first, :57/:166).
 * Only z-planes k in [k0,k1) are evaluated so that the CPU baseline can time a bounded sample; Y is
 * the full-size AoS array Y[3*idx+p].  Loop nest i -> j -> k as in the reference. */
void shmo_conv_normalize(int n_, const double* bbox_min, double cell, int S, const double* pos, const double* wn,
                         double lambda, int k0, int k1, double* Y) {
    const size_t n = (size_t)n_;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_; i++) {
        for (int j = 0; j < n_; j++) {
            for (int k = k0; k < k1; k++) {
                size_t idx = IDX(i, j, k);
                double x0 = i * cell + bbox_min[0], x1 = j * cell + bbox_min[1], x2 = k * cell + bbox_min[2];
                double a0 = 0., a1 = 0., a2 = 0.;
                for (int s = 0; s < S; s++) {
                    double g = yukawa(x0, x1, x2, pos[3 * s], pos[3 * s + 1], pos[3 * s + 2], lambda);
                    a0 += wn[3 * s] * g;
                    a1 += wn[3 * s + 1] * g;
                    a2 += wn[3 * s + 2] * g;
                }
                double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
                Y[3 * idx] = a0 / nrm;
                Y[3 * idx + 1] = a1 / nrm;
                Y[3 * idx + 2] = a2 / nrm;
            }
        }
    }
}

/* The same loops for the z-planes [k0,k1) only, written to a COMPACT array Yp[3*((k-k0)*n*n + j*n + i) + p]: what the full-size parity tests
 * compare sampled planes of the GPU's Y against at 512^3 / 1024^3, where a full-size AoS array would be 3-26 GB for two planes of use. */
void shmo_conv_normalize_planes(int n_, const double* bbox_min, double cell, int S, const double* pos, const double* wn,
                                double lambda, int k0, int k1, double* Yp) {
    const size_t n = (size_t)n_;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n_; i++) {
        for (int j = 0; j < n_; j++) {
            for (int k = k0; k < k1; k++) {
                size_t idx = (size_t)i + (size_t)j * n + (size_t)(k - k0) * n * n;
                double x0 = i * cell + bbox_min[0], x1 = j * cell + bbox_min[1], x2 = k * cell + bbox_min[2];
                double a0 = 0., a1 = 0., a2 = 0.;
                for (int s = 0; s < S; s++) {
                    double g = yukawa(x0, x1, x2, pos[3 * s], pos[3 * s + 1], pos[3 * s + 2], lambda);
                    a0 += wn[3 * s] * g;
                    a1 += wn[3 * s + 1] * g;
                    a2 += wn[3 * s + 2] * g;
                }
                double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
                Yp[3 * idx] = a0 / nrm;
                Yp[3 * idx + 1] = a1 / nrm;
                Yp[3 * idx + 2] = a2 / nrm;
            }
        }
    }
}

/* divYt = D^T * Y with D from gradient() (signed_heat_grid_solver.cpp:336-402), applied as a literal
 * scatter of the six triplets of every node; scrub = the non-finite -> 0 pass of the mesh overload (:72-74). */
void shmo_divergence(int n_, double cell, const double* Y, int scrub, double* b) {
    const size_t n = (size_t)n_, N = n * n * n;
    memset(b, 0, N * sizeof(double));
    const size_t stride[3] = {1, n, n * n};
    for (int k = 0; k < n_; k++)
        for (int j = 0; j < n_; j++)
            for (int i = 0; i < n_; i++) {
                size_t cur = IDX(i, j, k);
                int ijk[3] = {i, j, k};
                for (int p = 0; p < 3; p++) {
                    size_t nxt = cur + stride[p], c0 = cur;
                    if (ijk[p] == n_ - 1) { /* mirrored last row (:378-389) */
                        nxt = cur;
                        c0 = cur - stride[p];
                    }
                    double y = Y[3 * cur + p];
                    b[nxt] += y / cell;
                    b[c0] += -y / cell;
                }
            }
    if (scrub)
        for (size_t a = 0; a < N; a++)
            if (isinf(b[a]) || isnan(b[a])) b[a] = 0.;
}

/* out = L*u with L from laplacian() (signed_heat_grid_solver.cpp:278-334): seven triplets per row, an
 * out-of-grid neighbour redirected to the node itself (:299-319). */
void shmo_laplacian_apply(int n_, double cell, const double* u, double* out) {
    const size_t n = (size_t)n_;
    const double ih2 = 1. / (cell * cell);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < n_; k++)
        for (int j = 0; j < n_; j++)
            for (int i = 0; i < n_; i++) {
                size_t cur = IDX(i, j, k);
                size_t nx = (i == n_ - 1) ? cur : cur + 1, px = (i == 0) ? cur : cur - 1;
                size_t ny = (j == n_ - 1) ? cur : cur + n, py = (j == 0) ? cur : cur - n;
                size_t nz = (k == n_ - 1) ? cur : cur + n * n, pz = (k == 0) ? cur : cur - n * n;
                out[cur] = (u[nx] + u[ny] + u[nz] + u[px] + u[py] + u[pz] - 6. * u[cur]) * ih2;
            }
}

/* trilinearCoefficients, signed_heat_grid_solver.cpp:433-464 */
static void trilinear(int n_, const double* bbox_min, double h, const double* q, int64_t* nodes, double* c) {
    const size_t n = (size_t)n_;
    size_t i = (size_t)floor((q[0] - bbox_min[0]) / h);
    size_t j = (size_t)floor((q[1] - bbox_min[1]) / h);
    size_t k = (size_t)floor((q[2] - bbox_min[2]) / h);
    double tx = (q[0] - (i * h + bbox_min[0])) / h;
    double ty = (q[1] - (j * h + bbox_min[1])) / h;
    double tz = (q[2] - (k * h + bbox_min[2])) / h;
    nodes[0] = IDX(i, j, k);
    nodes[1] = IDX(i + 1, j, k);
    nodes[2] = IDX(i, j + 1, k);
    nodes[3] = IDX(i, j, k + 1);
    nodes[4] = IDX(i + 1, j + 1, k);
    nodes[5] = IDX(i + 1, j, k + 1);
    nodes[6] = IDX(i, j + 1, k + 1);
    nodes[7] = IDX(i + 1, j + 1, k + 1);
    c[0] = (1. - tx) * (1. - ty) * (1. - tz);
    c[1] = tx * (1. - ty) * (1. - tz);
    c[2] = (1. - tx) * ty * (1. - tz);
    c[3] = (1. - tx) * (1. - ty) * tz;
    c[4] = tx * ty * (1. - tz);
    c[5] = tx * (1. - ty) * tz;
    c[6] = (1. - tx) * ty * tz;
    c[7] = tx * ty * tz;
}

/* Constraint rows, signed_heat_grid_solver.cpp:80-98 / :186-204: one row per distinct cell in source
 * order.  nodes/coeffs must hold 8*S entries; returns m. */
int shmo_constraint_rows(int n_, const double* bbox_min, double cell, int S, const double* pos, int64_t* nodes,
                         double* coeffs) {
    const size_t n = (size_t)n_, N = n * n * n;
    unsigned char* used = (unsigned char*)calloc(N, 1);
    int m = 0;
    for (int s = 0; s < S; s++) {
        const double* b = pos + 3 * s;
        size_t i = (size_t)floor((b[0] - bbox_min[0]) / cell);
        size_t j = (size_t)floor((b[1] - bbox_min[1]) / cell);
        size_t k = (size_t)floor((b[2] - bbox_min[2]) / cell);
        size_t ci = IDX(i, j, k);
        if (used[ci]) continue;
        trilinear(n_, bbox_min, cell, b, nodes + 8 * (size_t)m, coeffs + 8 * (size_t)m);
        used[ci] = 1;
        m++;
    }
    free(used);
    return m;
}

/* evaluateFunction, signed_heat_grid_solver.cpp:405-431 */
static double eval_function(int n_, const double* bbox_min, double h, const double* u, const double* q) {
    const size_t n = (size_t)n_;
    int i = (int)floor((q[0] - bbox_min[0]) / h);
    int j = (int)floor((q[1] - bbox_min[1]) / h);
    int k = (int)floor((q[2] - bbox_min[2]) / h);
    double tx = (q[0] - (i * h + bbox_min[0])) / h;
    double ty = (q[1] - (j * h + bbox_min[1])) / h;
    double tz = (q[2] - (k * h + bbox_min[2])) / h;
    double v00 = u[IDX(i, j, k)] * (1. - tx) + u[IDX(i + 1, j, k)] * tx;
    double v01 = u[IDX(i, j, k + 1)] * (1. - tx) + u[IDX(i + 1, j, k + 1)] * tx;
    double v10 = u[IDX(i, j + 1, k)] * (1. - tx) + u[IDX(i + 1, j + 1, k)] * tx;
    double v11 = u[IDX(i, j + 1, k + 1)] * (1. - tx) + u[IDX(i + 1, j + 1, k + 1)] * tx;
    double v0 = v00 * (1. - ty) + v10 * ty;
    double v1 = v01 * (1. - ty) + v11 * ty;
    return v0 * (1. - tz) + v1 * tz;
}

/* evaluateAverageAlongSourceGeometry, signed_heat_grid_solver.cpp:466-496 */
double shmo_source_average(int n_, const double* bbox_min, double cell, const double* u, int S, const double* pos,
                           const double* area) {
    double shift = 0., norm = 0.;
    for (int s = 0; s < S; s++) {
        shift += area[s] * eval_function(n_, bbox_min, cell, u, pos + 3 * s);
        norm += area[s];
    }
    return shift / norm;
}

/* ---- (A A^T) dense Cholesky: the m x m SPD system inside the projector ------------------------- */
static int dense_cholesky(int m, double* G) { /* lower, in place, row-major */
    for (int j = 0; j < m; j++) {
        double d = G[(size_t)j * m + j];
        for (int k = 0; k < j; k++) d -= G[(size_t)j * m + k] * G[(size_t)j * m + k];
        if (!(d > 0.)) return -1;
        d = sqrt(d);
        G[(size_t)j * m + j] = d;
#pragma omp parallel for schedule(static)
        for (int i = j + 1; i < m; i++) {
            double s = G[(size_t)i * m + j];
            const double *ri = G + (size_t)i * m, *rj = G + (size_t)j * m;
            for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
            G[(size_t)i * m + j] = s / d;
        }
    }
    return 0;
}

static void cholesky_solve(int m, const double* G, double* v) {
    for (int i = 0; i < m; i++) {
        double s = v[i];
        const double* ri = G + (size_t)i * m;
        for (int k = 0; k < i; k++) s -= ri[k] * v[k];
        v[i] = s / ri[i];
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = v[i];
        for (int k = i + 1; k < m; k++) s -= G[(size_t)k * m + i] * v[k];
        v[i] = s / G[(size_t)i * m + i];
    }
}

typedef struct {
    int m;
    const int64_t* nodes;
    const double* coeffs;
    double* chol; /* m*m */
    double* w;    /* m */
} projector;

static int projector_init(projector* P, size_t N, int m, const int64_t* nodes, const double* coeffs) {
    P->m = m;
    P->nodes = nodes;
    P->coeffs = coeffs;
    P->chol = (double*)calloc((size_t)m * m, sizeof(double));
    P->w = (double*)malloc((size_t)m * sizeof(double));
    /* G = A A^T through a node -> (row, coeff) adjacency */
    size_t nnz = (size_t)8 * m;
    int64_t* order = (int64_t*)malloc(nnz * sizeof(int64_t));
    /* counting sort of entries by node is overkill at this size: use a hash-free O(nnz * deg) via
     * per-node linked lists */
    int64_t* head = (int64_t*)malloc(N * sizeof(int64_t));
    for (size_t a = 0; a < N; a++) head[a] = -1;
    for (size_t e = 0; e < nnz; e++) {
        order[e] = head[nodes[e]];
        head[nodes[e]] = (int64_t)e;
    }
    for (size_t e = 0; e < nnz; e++) {
        int r = (int)(e / 8);
        for (int64_t f = head[nodes[e]]; f >= 0; f = order[f]) {
            int c = (int)(f / 8);
            P->chol[(size_t)r * m + c] += coeffs[e] * coeffs[f];
        }
    }
    free(head);
    free(order);
    return dense_cholesky(m, P->chol);
}

static void projector_free(projector* P) {
    free(P->chol);
    free(P->w);
}

/* v <- v - A^T (A A^T)^-1 A v */
static void project(const projector* P, double* v) {
    const int m = P->m;
    for (int r = 0; r < m; r++) {
        double s = 0.;
        for (int e = 0; e < 8; e++) s += P->coeffs[8 * (size_t)r + e] * v[P->nodes[8 * (size_t)r + e]];
        P->w[r] = s;
    }
    cholesky_solve(m, P->chol, P->w);
    for (int r = 0; r < m; r++)
        for (int e = 0; e < 8; e++) v[P->nodes[8 * (size_t)r + e]] -= P->coeffs[8 * (size_t)r + e] * P->w[r];
}

static double dot(size_t N, const double* a, const double* b) {
    double s = 0.;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < N; i++) s += a[i] * b[i];
    return s;
}

/* Constrained solve (signed_heat_grid_solver.cpp:101-108) as projected CG; returns phi = -x.
 * stats[0]=iterations, stats[1]=final ||Pr||/||Pb||, stats[2]=max|A x|. Returns 0 on success. */
int shmo_constrained_solve(int n_, double cell, const double* b, int m, const int64_t* nodes, const double* coeffs,
                           double tol, int maxit, double* phi, double* stats) {
    const size_t n = (size_t)n_, N = n * n * n;
    projector P;
    if (projector_init(&P, N, m, nodes, coeffs) != 0) {
        projector_free(&P);
        return -1;
    }
    double* x = (double*)calloc(N, sizeof(double));
    double* r = (double*)malloc(N * sizeof(double));
    double* p = (double*)malloc(N * sizeof(double));
    double* q = (double*)malloc(N * sizeof(double));
    memcpy(r, b, N * sizeof(double));
    project(&P, r);
    for (size_t i = 0; i < N; i++) p[i] = -r[i];
    double rho = dot(N, r, r), rho0 = rho;
    int it = 0;
    if (!(rho0 == rho0) || rho0 > 1.7e308) {
        /* non-finite right-hand side (|X|^2 underflowed far from a finely sampled point cloud -> Y = +-inf, and the point overload has
         * no scrub, :180): the reference's sparse LU (:101-107) propagates it -- the KKT matrix is irreducible, so every entry of the
         * solution becomes NaN.  Report that instead of leaving the CG loop through a false NaN comparison with x = 0. */
        for (size_t i = 0; i < N; i++) phi[i] = NAN;
        if (stats) {
            stats[0] = 0;
            stats[1] = NAN;
            stats[2] = NAN;
        }
        free(x);
        free(r);
        free(p);
        free(q);
        projector_free(&P);
        return 2;
    }
    while (it < maxit && rho > tol * tol * rho0) {
        shmo_laplacian_apply(n_, cell, p, q); /* q = L p ; K = -L */
        double pq = -dot(N, p, q);
        double alpha = rho / pq;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) {
            x[i] += alpha * p[i];
            r[i] -= alpha * q[i]; /* r + alpha*K p */
        }
        project(&P, r);
        double rho_new = dot(N, r, r);
        double beta = rho_new / rho;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) p[i] = -r[i] + beta * p[i];
        rho = rho_new;
        it++;
    }
    double maxAx = 0.;
    for (int rr = 0; rr < m; rr++) {
        double s = 0.;
        for (int e = 0; e < 8; e++) s += coeffs[8 * (size_t)rr + e] * x[nodes[8 * (size_t)rr + e]];
        if (fabs(s) > maxAx) maxAx = fabs(s);
    }
    for (size_t i = 0; i < N; i++) phi[i] = -x[i];
    if (stats) {
        stats[0] = it;
        stats[1] = sqrt(rho / rho0);
        stats[2] = maxAx;
    }
    free(x);
    free(r);
    free(p);
    free(q);
    projector_free(&P);
    return 0;
}

/* integrateGreedily, signed_heat_grid_solver.cpp:224-275: FIFO BFS from (0,0,0); per axis the -1
 * neighbour is tried before the +1 neighbour; first visitor wins. */
void shmo_integrate_greedily(int n_, const double* bbox_min, double cell, const double* Y, double* phi) {
    const size_t n = (size_t)n_, N = n * n * n;
    unsigned char* visited = (unsigned char*)calloc(N, 1);
    int* queue = (int*)malloc(N * 3 * sizeof(int));
    size_t qh = 0, qt = 0;
    memset(phi, 0, N * sizeof(double));
    queue[0] = queue[1] = queue[2] = 0;
    qt = 1;
    visited[0] = 1;
    while (qh < qt) {
        int cur[3] = {queue[3 * qh], queue[3 * qh + 1], queue[3 * qh + 2]};
        qh++;
        size_t ci = IDX(cur[0], cur[1], cur[2]);
        double p[3] = {cur[0] * cell + bbox_min[0], cur[1] * cell + bbox_min[1], cur[2] * cell + bbox_min[2]};
        for (int a = 0; a < 3; a++) {
            for (int step = -1; step <= 1; step += 2) {
                if (step < 0 && cur[a] == 0) continue;
                if (step > 0 && cur[a] >= n_ - 1) continue;
                int nxt[3] = {cur[0], cur[1], cur[2]};
                nxt[a] += step;
                size_t ni = IDX(nxt[0], nxt[1], nxt[2]);
                if (visited[ni]) continue;
                double qpos[3] = {nxt[0] * cell + bbox_min[0], nxt[1] * cell + bbox_min[1], nxt[2] * cell + bbox_min[2]};
                double e0 = qpos[0] - p[0], e1 = qpos[1] - p[1], e2 = qpos[2] - p[2];
                double y0 = Y[3 * ni] + Y[3 * ci], y1 = Y[3 * ni + 1] + Y[3 * ci + 1], y2 = Y[3 * ni + 2] + Y[3 * ci + 2];
                double nn = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
                y0 /= nn;
                y1 /= nn;
                y2 /= nn;
                phi[ni] = phi[ci] + (y0 * e0 + y1 * e1 + y2 * e2);
                visited[ni] = 1;
                queue[3 * qt] = nxt[0];
                queue[3 * qt + 1] = nxt[1];
                queue[3 * qt + 2] = nxt[2];
                qt++;
            }
        }
    }
    free(queue);
    free(visited);
}

/* End to end, signed_heat_grid_solver.cpp:38-113 (scrub=1: mesh overload) / :146-221 (scrub=0: points).
 * stats: [0]=m [1]=cg iterations [2]=rel residual [3]=max|Ax| [4]=shift. */
int shmo_compute_distance(int n_, const double* bbox_min, double cell, int S, const double* pos, const double* wn,
                          const double* area, double lambda, int scrub, int fast, double tol, int maxit, double* phi,
                          double* stats) {
    const size_t n = (size_t)n_, N = n * n * n;
    double* Y = (double*)malloc(3 * N * sizeof(double));
    shmo_conv_normalize(n_, bbox_min, cell, S, pos, wn, lambda, 0, n_, Y);
    int rc = 0;
    double st[3] = {0, 0, 0};
    int m = 0;
    if (fast) {
        shmo_integrate_greedily(n_, bbox_min, cell, Y, phi);
    } else {
        double* b = (double*)malloc(N * sizeof(double));
        shmo_divergence(n_, cell, Y, scrub, b);
        int64_t* nodes = (int64_t*)malloc((size_t)8 * S * sizeof(int64_t));
        double* coeffs = (double*)malloc((size_t)8 * S * sizeof(double));
        m = shmo_constraint_rows(n_, bbox_min, cell, S, pos, nodes, coeffs);
        rc = shmo_constrained_solve(n_, cell, b, m, nodes, coeffs, tol, maxit, phi, st);
        free(b);
        free(nodes);
        free(coeffs);
    }
    double shift = shmo_source_average(n_, bbox_min, cell, phi, S, pos, area);
    for (size_t i = 0; i < N; i++) phi[i] -= shift;
    if (stats) {
        stats[0] = m;
        stats[1] = st[0];
        stats[2] = st[1];
        stats[3] = st[2];
        stats[4] = shift;
    }
    free(Y);
    return rc;
}
