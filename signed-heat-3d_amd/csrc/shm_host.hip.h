// Host-side pieces shared by the translation units of libshm_grid.so (round 6: the library is built from three of them -- the C ABI, Solver<double>, Solver<float> --
// compiled in parallel): error type, experiment knobs, RCCL loader, pinned staging pool, device arrays, the host twins of the Step-1 rules, the solver interface.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // prototypes and enum values only: librccl itself is dlopen'ed on first multi-process use

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/shm_grid.h"

namespace shm {

struct Error : std::runtime_error {
    shm_status code;
    Error(shm_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// Experiment knobs (environment variables that select measured-and-rejected variants, A/B shapes and instrumentation) are read only when SHM_DEBUG_KNOBS=1 is set:
// a product run cannot pick one up by accident (round 5; INTEGRATION.md section 4a lists them).  What a caller may legitimately choose is in shm_opts / shm_config.
static const char* knob(const char* name) {
    static const bool on = getenv("SHM_DEBUG_KNOBS") != nullptr && atoi(getenv("SHM_DEBUG_KNOBS")) != 0;
    return on ? getenv(name) : nullptr;
}

static std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            throw shm::Error(e_ == hipErrorOutOfMemory ? SHM_ERR_NOMEM : SHM_ERR_HIP,                        \
                             shm::fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)

// ---- RCCL, resolved lazily (only multi-process runs touch it) -----------------------------------
// The library is dlopen'ed at run time (single-GPU users never load it), but the prototypes and enum values come from the
// real header at build time: the function-pointer types below are decltype(&ncclXxx), so a signature drift in <rccl/rccl.h>
// is a compile error here, not a silent ABI mismatch.
struct Rccl {
    typedef ncclComm_t comm_t;
    typedef ncclUniqueId unique_id;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce_ = nullptr;
    decltype(&ncclSend) Send_ = nullptr;
    decltype(&ncclRecv) Recv_ = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString_ = nullptr;
    void* h = nullptr;
    enum { kFloat32 = ncclFloat32, kFloat64 = ncclFloat64, kSum = ncclSum };
    static_assert(sizeof(ncclUniqueId) == 128, "include/shm_grid.h documents a 128-byte unique id");
    static_assert(ncclFloat32 == 7 && ncclFloat64 == 8 && ncclSum == 0, "tests/native/rccl_mock.c hard-codes these values");

    int AllReduce(const void* s, void* r, size_t c, int dt, int op, comm_t comm, hipStream_t st) { return AllReduce_(s, r, c, (ncclDataType_t)dt, (ncclRedOp_t)op, comm, st); }
    int Send(const void* s, size_t c, int dt, int peer, comm_t comm, hipStream_t st) { return Send_(s, c, (ncclDataType_t)dt, peer, comm, st); }
    int Recv(void* r, size_t c, int dt, int peer, comm_t comm, hipStream_t st) { return Recv_(r, c, (ncclDataType_t)dt, peer, comm, st); }

    static Rccl& get() {
        static Rccl r;
        if (!r.h) r.load();
        return r;
    }
    void load() {
        // SHM_RCCL_LIB: tests substitute a shared-memory double (tests/native/rccl_mock.c) to run several ranks on one GPU
        const char* override_lib = getenv("SHM_RCCL_LIB");
        const char* names[] = {override_lib ? override_lib : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* nm : names) {
            lib = dlopen(nm, RTLD_NOW | (override_lib ? RTLD_LOCAL : RTLD_GLOBAL));
            if (lib || override_lib) break;
        }
        if (!lib) throw Error(SHM_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
        // resolve into a scratch copy: a failure leaves the singleton untouched (h stays null, the next call retries)
        Rccl t;
        const char* missing = nullptr;
        auto sym = [&](auto& field, const char* name) {
            *(void**)(&field) = dlsym(lib, name);
            if (!field && !missing) missing = name;
        };
        sym(t.GetUniqueId, "ncclGetUniqueId");
        sym(t.CommInitRank, "ncclCommInitRank");
        sym(t.CommDestroy, "ncclCommDestroy");
        sym(t.AllReduce_, "ncclAllReduce");
        sym(t.Send_, "ncclSend");
        sym(t.Recv_, "ncclRecv");
        sym(t.GroupStart, "ncclGroupStart");
        sym(t.GroupEnd, "ncclGroupEnd");
        sym(t.GetErrorString_, "ncclGetErrorString");
        if (missing) {
            dlclose(lib);
            throw Error(SHM_ERR_RCCL, std::string("librccl lacks symbol ") + missing);
        }
        t.h = lib;
        *this = t;
    }
    void chk(int rc, const char* what) {
        if (rc != 0) throw Error(SHM_ERR_RCCL, fmt("%s failed: %s", what, GetErrorString_ ? GetErrorString_((ncclResult_t)rc) : "?"));
    }
};

// Host -> device uploads go through pinned staging chunks: hipMemcpyAsync from pageable memory first waits for everything queued on its stream, and the
// set-up stream's work cannot progress while a Step-1 kernel that fills the SIMDs runs (fp32 and all-fp64 Step 1) -- the host part of the constraint
// set-up then stalled at its first upload until Step 1 had finished (rocker 512^3 fp32: 40 ms of exposed wait, 13 of them host work that had not started;
// round 3).  A chunk is reused once the event recorded behind its copy has completed.  Chunks and their events belong to ONE device (an event must be
// recorded on a stream of its own device): the pool is keyed by the device that is current at the upload, and a device's chunks are freed when the last
// solver on it is destroyed (Solver's constructor / destructor hold the reference).  Beyond kMaxBytes per device, or when pinned memory cannot be had,
// the plain copy is used.
struct PinnedPool {
    struct Chunk {
        void* p;
        size_t cap;
        hipEvent_t ev;
        bool busy;
        int device;
    };
    static constexpr size_t kMaxBytes = (size_t)2 << 30;
    std::vector<Chunk> chunks;
    std::unordered_map<int, size_t> total;   // bytes pinned per device
    std::unordered_map<int, int> users;      // solvers alive per device
    std::mutex mu;
    static PinnedPool& get() {
        static PinnedPool* P = new PinnedPool();   // (never destroyed: no HIP calls at process exit)
        return *P;
    }
    void acquire(int device) {
        std::lock_guard<std::mutex> lk(mu);
        users[device]++;
    }
    // the caller has made `device` current
    void release(int device) noexcept {
        std::lock_guard<std::mutex> lk(mu);
        if (--users[device] > 0) return;
        size_t keep = 0;
        for (Chunk& k : chunks) {
            if (k.device != device) {
                chunks[keep++] = k;
                continue;
            }
            if (k.busy) (void)hipEventSynchronize(k.ev);
            (void)hipEventDestroy(k.ev);
            (void)hipHostFree(k.p);
        }
        chunks.resize(keep);
        total[device] = 0;
    }
    void upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
        int device = 0;
        HIPCHK(hipGetDevice(&device));   // the stream's device: every caller runs under its solver's hipSetDevice
        std::lock_guard<std::mutex> lk(mu);
        Chunk* c = nullptr;
        for (Chunk& k : chunks) {
            if (k.device != device || k.cap < bytes) continue;
            if (k.busy) {
                if (hipEventQuery(k.ev) == hipSuccess) k.busy = false;
                else (void)hipGetLastError();   // hipErrorNotReady is not an error here
            }
            if (!k.busy && (!c || k.cap < c->cap)) c = &k;
        }
        if (!c && users[device] > 0 && total[device] + bytes <= kMaxBytes) {
            Chunk k{nullptr, std::max(bytes + bytes / 4, (size_t)1 << 20), nullptr, false, device};
            if (hipHostMalloc(&k.p, k.cap, hipHostMallocPortable) == hipSuccess && hipEventCreateWithFlags(&k.ev, hipEventDisableTiming) == hipSuccess) {
                chunks.push_back(k);
                total[device] += k.cap;
                c = &chunks.back();
            } else {
                (void)hipGetLastError();
                if (k.p) (void)hipHostFree(k.p);
            }
        }
        if (!c) {
            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
            return;
        }
        memcpy(c->p, src, bytes);
        HIPCHK(hipMemcpyAsync(dst, c->p, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c->ev, st));
        c->busy = true;
    }
};

template <typename T> struct DevArray {
    T* p = nullptr;
    size_t count = 0;
    DevArray() = default;
    DevArray(const DevArray&) = delete;
    DevArray& operator=(const DevArray&) = delete;
    DevArray(DevArray&& o) noexcept : p(o.p), count(o.count) {
        o.p = nullptr;
        o.count = 0;
    }
    ~DevArray() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    void alloc(size_t c) {
        if (c <= count && p) return;
        release();
        if (c == 0) c = 1;
        HIPCHK(hipMalloc((void**)&p, c * sizeof(T)));
        count = c;
    }
    void upload(const std::vector<T>& v, hipStream_t st) {
        alloc(v.size());
        if (!v.empty()) PinnedPool::get().upload(p, v.data(), v.size() * sizeof(T), st);   // the vector may die right after the call
    }
};

struct Event {
    hipEvent_t e = nullptr;
    Event() { HIPCHK(hipEventCreate(&e)); }
    ~Event() { if (e) (void)hipEventDestroy(e); }
    Event(const Event&) = delete;
    void record(hipStream_t s) { HIPCHK(hipEventRecord(e, s)); }
};
static float elapsed(Event& a, Event& b) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a.e, b.e));
    return ms;
}

// One constraint row (trilinearCoefficients, signed_heat_grid_solver.cpp:433-464).
struct Row {
    int64_t nodes[8];
    double coeffs[8];
    int cell[3];   // (i, j, k) of the cell and the trilinear parameters of the sample point in it: the separable form of coeffs
    double t[3];   // that the explicit Schur complement (shm_schur.hip.h) is assembled from
};


// ---- Step-1 work per z-plane, estimated on the host with the kernels' own culling / tier rules (per source, on a sample of node blocks) ----------------
// Equal-plane z-slabs are not equal work for Step 1 once sources are culled (fp32 configs: the end slabs keep more per block) or tiered (fp64: blocks near
// the object evaluate more pairs in fp64).  weights[k] is proportional to the cost of plane k; shm_plan_slab_weighted() cuts the planes by it.
//   fp64 (tiered kernel): block = 8 x 8 x 4 nodes, classified per source; near pairs cost 1, far (packed fp32) pairs 0.43 (measured ratio of the two tiers,
//   DESIGN.md section 4), dropped 0
//   fp32 through the tiered kernel (round 5; `tiered32`): the fp64 rule with the fp32 drop threshold and tier_log = -infinity -- every kept in-range pair costs 0.43
//   fp32 (conv_normalize_kernel<float>, SHM_CONV32_CLASSIC=1): block = 8 x 8 x 16 nodes (its culled unit), classified per CLUSTER of 32 Morton-sorted sources with the cluster's
//   bounding sphere and largest weight, like the kernel: kept clusters cost 32 pairs per node, skipped ones 0.  (A per-source rule predicts a 20 % imbalance
//   of equal slabs on rocker 512^3 where 3 % is measured: the spheres' radii, not the sources' distances, decide what that kernel skips.)
// K of the tiered kernels' drop rule (round 6): the number of sources a block is expected to find within an e-fold of the drop threshold (see Solver::set_problem)
static double drop_rule_K(int64_t S, const double* wn, double lambda) {
    double wsum = 0.;
    int64_t nz = 0;
    for (int64_t t = 0; t < S; t++) {
        const double w = std::sqrt(wn[3 * t] * wn[3 * t] + wn[3 * t + 1] * wn[3 * t + 1] + wn[3 * t + 2] * wn[3 * t + 2]);
        if (w > 0. && std::isfinite(w)) {
            wsum += w;
            nz++;
        }
    }
    const double abar = nz ? wsum / (double)nz : 0.;
    const double k_est = abar > 0. ? 3.0 * 157.0 / (abar * lambda * lambda) : 64.;
    return std::min(std::max(64.0, (double)S), std::max(64.0, k_est));
}

// K of the drop rule chosen per problem on a sample of blocks (round 6, late): the formula above is within a factor of three of the best K and that factor is worth 3 % of Step 1
// on the culled configurations (measured, SHM_CONV_DROP_K: rocker 512^3 fp32 254 ms at K = 350 against 262 at the formula's 1066; SprayBottle.pc 512^3 fp32 283 at 700 ... 1400
// against 307 at 350).  The host walks up to 48 blocks spread over the grid through the kernel's own scan -- clusters in storage order, 64 at a time, then the sources of the kept
// ones, candidates of a batch / cluster together or only the ones below the hard threshold -- for a ladder of K and keeps the K that drops most.  Deterministic (every rank
// derives the same K from the same sources); (samples x sources) <= 1.5e6: a few ms, only for S >= 4096.  src6: the sources as the kernel gets them (Morton order, grid-centred
// positions, UNscaled weights, padded with zero weights to whole clusters of 64); cl: their cluster records (centre, radius, ln largest weight, ln weight sum).
static double choose_drop_K(const double* src6, int n_clusters, const float* cl, int rec, double lambda, int n, double cell, double eps, int64_t S_true, double k_formula) {
    if (!(eps > 0.) || S_true < 4096 || n_clusters < 1) return k_formula;
    const int64_t Sp = (int64_t)n_clusters * 64;
    static const double fx[4] = {0.10, 0.37, 0.63, 0.90}, fz[3] = {0.15, 0.50, 0.85};
    const int want = (int)std::max<int64_t>(8, std::min<int64_t>(48, 1500000 / std::max<int64_t>(1, Sp)));
    const double ladder[] = {128., 192., 256., 384., 512., 768., 1024., 1536., 2048., 4096.};
    constexpr int NK = sizeof(ladder) / sizeof(ladder[0]);
    double dropped[NK] = {0};
    const double hx = 3.5 * cell, hy = 3.5 * cell, hz = 1.5 * cell, rt = std::sqrt(hx * hx + hy * hy + hz * hz), half = 0.5 * (double)(n - 1) * cell;
    const double eps_soft = 0.875 * eps, lhard = std::log(0.125 * eps / (double)S_true);
    std::vector<double> lb((size_t)Sp), bs((size_t)Sp), lbmax((size_t)n_clusters), bc((size_t)n_clusters);
    std::vector<char> cand_ok((size_t)n_clusters);
    int done = 0;
    for (int a = 0; a < 4 && done < want; a++)
        for (int b = 0; b < 4 && done < want; b++)
            for (int c3 = 0; c3 < 3 && done < want; c3++, done++) {
                // (a different pairing of the lattice coordinates per sample index, so that fewer than 48 samples still spread over the grid)
                const double c[3] = {(2. * fx[(a + c3) & 3] - 1.) * half, (2. * fx[(b + 2 * c3) & 3] - 1.) * half, (2. * fz[c3] - 1.) * half};
                double dmin = 1e300, wstar = 0., nstar[3] = {0, 0, 0};
                for (int64_t t = 0; t < Sp; t++) {
                    const double* q = src6 + 6 * t;
                    const double w = std::sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5]);
                    if (!(w > 0.)) continue;
                    const double dx = c[0] - q[0], dy = c[1] - q[1], dz = c[2] - q[2], d = std::sqrt(dx * dx + dy * dy + dz * dz);
                    if (d < dmin || (d == dmin && w > wstar)) {
                        dmin = d; wstar = w; nstar[0] = dx; nstar[1] = dy; nstar[2] = dz;
                    }
                }
                if (!(wstar > 0.)) continue;
                const double r_hi = std::sqrt((std::fabs(nstar[0]) + hx) * (std::fabs(nstar[0]) + hx) + (std::fabs(nstar[1]) + hy) * (std::fabs(nstar[1]) + hy) +
                                              (std::fabs(nstar[2]) + hz) * (std::fabs(nstar[2]) + hz));
                const double bxs = std::max(std::fabs(nstar[0]) - hx, 0.), bys = std::max(std::fabs(nstar[1]) - hy, 0.), bzs = std::max(std::fabs(nstar[2]) - hz, 0.);
                const double dbs = std::sqrt(bxs * bxs + bys * bys + bzs * bzs), inv_dstar = dmin > 0. && dbs > 0. ? 1. / dbs : 1e300, lwstar = std::log(wstar);
                for (int64_t t = 0; t < Sp; t++) {
                    const double* q = src6 + 6 * t;
                    const double w = std::sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5]);
                    if (!(w > 0.)) { lb[(size_t)t] = 1e300; bs[(size_t)t] = 0.; continue; }
                    const double ex = c[0] - q[0], ey = c[1] - q[1], ez = c[2] - q[2];
                    const double bx = std::max(std::fabs(ex) - hx, 0.), by = std::max(std::fabs(ey) - hy, 0.), bz = std::max(std::fabs(ez) - hz, 0.);
                    const double dist = std::sqrt(bx * bx + by * by + bz * bz), dc = std::sqrt(ex * ex + ey * ey + ez * ez);
                    const double dot = dc > 0. && dmin > 0. ? (ex * nstar[0] + ey * nstar[1] + ez * nstar[2]) / (dc * dmin) : 1.;
                    const double lip = std::sqrt(std::max(0., 2. - 2. * dot)) + rt * ((dist > 0. ? 1. / dist : 1e300) + inv_dstar);
                    const double gap = std::max(dist - r_hi, dc - dmin - rt * lip);
                    lb[(size_t)t] = std::log(w) - lwstar - lambda * gap;
                    bs[(size_t)t] = dist > 0. ? std::exp(lb[(size_t)t]) * r_hi / dist : 1e300;
                }
                for (int k = 0; k < n_clusters; k++) {
                    const float* r = cl + (size_t)k * rec;
                    const double gx = c[0] - r[0], gy = c[1] - r[1], gz = c[2] - r[2], gap = std::sqrt(gx * gx + gy * gy + gz * gz) - rt - r[3] - r_hi;
                    cand_ok[(size_t)k] = gap > 0.;
                    lbmax[(size_t)k] = (double)r[4] - lwstar - lambda * gap;
                    bc[(size_t)k] = gap > 0. ? std::exp((double)r[5] - lwstar - lambda * gap) * r_hi / (r_hi + gap) : 1e300;
                }
                for (int ki = 0; ki < NK; ki++) {
                    const double ltau = std::log(eps_soft / ladder[ki]);
                    double R = 0.;
                    int64_t nd = 0;
                    for (int k0 = 0; k0 < n_clusters; k0 += 64) {
                        const int k1 = std::min(n_clusters, k0 + 64);
                        double sum = 0.;
                        for (int k = k0; k < k1; k++)
                            if (cand_ok[(size_t)k] && lbmax[(size_t)k] <= ltau) sum += bc[(size_t)k];
                        const bool all = R + sum <= eps_soft;
                        if (all) R += sum;
                        for (int k = k0; k < k1; k++) {
                            const bool cd = cand_ok[(size_t)k] && lbmax[(size_t)k] <= (all ? ltau : std::min(ltau, lhard));
                            if (cd) { nd += 64; continue; }
                            double ssum = 0.;
                            for (int64_t t = (int64_t)k * 64; t < (int64_t)(k + 1) * 64; t++)
                                if (lb[(size_t)t] <= ltau) ssum += bs[(size_t)t];
                            const bool sall = R + ssum <= eps_soft;
                            if (sall) R += ssum;
                            for (int64_t t = (int64_t)k * 64; t < (int64_t)(k + 1) * 64; t++)
                                if (lb[(size_t)t] <= (sall ? ltau : std::min(ltau, lhard))) nd++;
                        }
                    }
                    dropped[ki] += (double)nd;
                }
            }
    int best = -1;
    for (int ki = 0; ki < NK; ki++)
        if (dropped[ki] > 0. && (best < 0 || dropped[ki] > dropped[best] * 1.002)) best = ki;   // (ties and near-ties: the smaller K)
    return best < 0 ? k_formula : std::min(std::max(64.0, (double)S_true), ladder[best]);
}

static void step1_plane_weights_host(int64_t S, const double* pos, const double* wn, double lambda, int n, const double* bbox_min, double cell, int precision,
                                     double tier_log, double* weights, bool tiered32 = false) {
    const bool per_source = precision == SHM_F64 || tiered32;   // the tiered kernel's classification: per (8 x 8 x 4 block, source)
    const bool f64 = per_source;                                // (below, "f64" selects that model)
    if (tiered32 && precision != SHM_F64) tier_log = -1.0e30;
    const int bz = f64 ? 4 : 16;
    const double half_z = 0.5 * (bz - 1);
    const double rt = std::sqrt(3.5 * 3.5 * 2 + half_z * half_z) * cell * 1.000001;
    const double drop_eps = precision == SHM_F64 ? 2e-9 : 6.0e-8;
    const double skip_base = std::log((double)S / drop_eps);   // (the classic kernels' drop threshold: Solver::set_problem)
    // tiered kernels (round 6): dropped by accumulated bound -- candidates below tau in groups of 64 while their bounds sum to <= eps_soft, below tau_hard always
    const double eps_soft = 0.875 * drop_eps, ln_tau = std::log(eps_soft / drop_rule_K(S, wn, lambda)), ln_tau_hard = std::log(0.125 * drop_eps / (double)S);
    const double far_cost = 0.43;
    std::vector<double> wmag((size_t)S);
    double wlo = 1e300, whi = 0.;
    for (int64_t s = 0; s < S; s++) {
        wmag[(size_t)s] = std::sqrt(wn[3 * s] * wn[3 * s] + wn[3 * s + 1] * wn[3 * s + 1] + wn[3 * s + 2] * wn[3 * s + 2]);
        if (wmag[(size_t)s] > 0.) wlo = std::min(wlo, wmag[(size_t)s]);
        whi = std::max(whi, wmag[(size_t)s]);
    }
    // (the kernel's per-source exponent-range test of the packed-fp32 tier, shm_conv_tiered.hip.h `in_range`, as a per-source rule on the block's centre)
    const double range_c = 1.4426950408889634 * lambda * 2.0 * rt - 113.0, lwhi = std::log2(std::max(whi, 1e-300));
    // fp32: the kernel's clusters (Morton order of the sources, 32 per cluster, bounding sphere about the mean, largest weight)
    constexpr int kCl = 32;
    std::vector<double> ccen, crad, clnw;
    if (!f64) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int64_t s = 0; s < S; s++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], pos[3 * s + a]);
                hi[a] = std::max(hi[a], pos[3 * s + a]);
            }
        const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
        auto spread = [](uint64_t v) {
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
            for (int a = 0; a < 3; a++) code |= spread((uint64_t)std::min(1048575.0, std::max(0.0, (pos[3 * s + a] - lo[a]) / ext * 1048575.0))) << a;
            order[(size_t)s] = {code, s};
        }
        std::sort(order.begin(), order.end());
        const int64_t ncl = (S + kCl - 1) / kCl;
        ccen.assign((size_t)ncl * 3, 0.);
        crad.assign((size_t)ncl, 0.);
        clnw.assign((size_t)ncl, -1e300);
        for (int64_t c = 0; c < ncl; c++) {
            const int64_t a0 = c * kCl, a1 = std::min<int64_t>(S, a0 + kCl);
            for (int64_t t = a0; t < a1; t++)
                for (int a = 0; a < 3; a++) ccen[(size_t)c * 3 + a] += pos[3 * order[(size_t)t].second + a] / (double)(a1 - a0);
            for (int64_t t = a0; t < a1; t++) {
                const int64_t sidx = order[(size_t)t].second;
                double d2 = 0.;
                for (int a = 0; a < 3; a++) d2 += (pos[3 * sidx + a] - ccen[(size_t)c * 3 + a]) * (pos[3 * sidx + a] - ccen[(size_t)c * 3 + a]);
                crad[(size_t)c] = std::max(crad[(size_t)c], std::sqrt(d2));
                if (wmag[(size_t)sidx] > 0.) clnw[(size_t)c] = std::max(clnw[(size_t)c], std::log(wmag[(size_t)sidx]));
            }
        }
    }
    const int tiles = (n + 7) / 8, layers = (n + bz - 1) / bz;
    const int K = std::min(tiles, 12);   // sampled blocks per axis and layer
    // the block layers are independent: spread over host threads (SprayBottle.pc 1024^3 fp64: 4e9 distance evaluations -- seconds on one core, inside
    // set_problem on every rank)
    auto do_layer = [&](int L, std::vector<double>& dist) {
        double acc = 0.;
        const double cz = (L * bz + half_z) * cell + bbox_min[2];
        for (int a = 0; a < K; a++)
            for (int b = 0; b < K; b++) {
                const int tx = (int)(((2 * a + 1) * (long long)tiles) / (2 * K)), ty = (int)(((2 * b + 1) * (long long)tiles) / (2 * K));
                const double cx = (tx * 8 + 3.5) * cell + bbox_min[0], cy = (ty * 8 + 3.5) * cell + bbox_min[1];
                double dmin = 1e300, wnear = 0., bnd[64], lbs[64];
                int64_t s_star = 0;
                for (int64_t s = 0; s < S; s++) {
                    const double dx = cx - pos[3 * s], dy = cy - pos[3 * s + 1], dz = cz - pos[3 * s + 2];
                    const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
                    dist[(size_t)s] = d;
                    if (wmag[(size_t)s] > 0. && (d < dmin || (d == dmin && wmag[(size_t)s] > wnear))) {
                        dmin = d;
                        wnear = wmag[(size_t)s];
                        s_star = s;
                    }
                }
                if (!(wnear > 0.)) continue;
                const double r_hi = dmin + rt, ln_near = std::log(wnear);
                double cost = 0.;
                if (f64) {
                    // (the kernel's bounds with the block's bounding sphere in place of its box; its differential bound of r_s - r_s* beside the box rule's: shm_conv_tiered.hip.h)
                    const double sx = cx - pos[3 * s_star], sy = cy - pos[3 * s_star + 1], sz = cz - pos[3 * s_star + 2];
                    const double inv_dstar_box = dmin - rt > 0. ? 1.0 / (dmin - rt) : 1e300;
                    double R = 0.;
                    for (int64_t g0 = 0; g0 < S; g0 += 64) {
                        const int64_t g1 = std::min<int64_t>(S, g0 + 64);
                        double gsum = 0.;
                        for (int64_t s = g0; s < g1; s++) {
                            bnd[(size_t)(s - g0)] = -1.;
                            if (!(wmag[(size_t)s] > 0.)) continue;
                            const double d = dist[(size_t)s], d_box = std::max(1e-300, d - rt);
                            const double ex = cx - pos[3 * s], ey = cy - pos[3 * s + 1], ez = cz - pos[3 * s + 2];
                            const double dot = d > 0. && dmin > 0. ? (ex * sx + ey * sy + ez * sz) / (d * dmin) : 1.0;
                            const double lip = std::sqrt(std::max(0., 2.0 - 2.0 * dot)) + rt * (1.0 / d_box + inv_dstar_box);
                            const double lhs_drop = lambda * std::max(d - rt - r_hi, d - dmin - rt * lip), rel = std::log(wmag[(size_t)s]) - ln_near;
                            lbs[(size_t)(s - g0)] = rel - lhs_drop;
                            if (lbs[(size_t)(s - g0)] <= ln_tau) {
                                bnd[(size_t)(s - g0)] = std::exp(lbs[(size_t)(s - g0)]) * r_hi / d_box;
                                gsum += bnd[(size_t)(s - g0)];
                            }
                        }
                        const bool soft_ok = R + gsum <= eps_soft;
                        if (soft_ok) R += gsum;
                        for (int64_t s = g0; s < g1; s++) {
                            if (!(wmag[(size_t)s] > 0.)) continue;
                            if (bnd[(size_t)(s - g0)] >= 0. && (soft_ok || lbs[(size_t)(s - g0)] <= ln_tau_hard)) continue;   // dropped
                            const double lhs = lambda * (dist[(size_t)s] - rt - r_hi), rel = std::log(wmag[(size_t)s]) - ln_near;
                            const double d_box = std::max(0., dist[(size_t)s] - rt), d0 = std::max(0., dmin - rt);
                            const bool in_range = 1.4426950408889634 * lambda * (d_box - d0) + range_c <= std::log2(wmag[(size_t)s]) - lwhi;
                            cost += lhs > tier_log + rel && in_range ? far_cost : 1.0;
                        }
                    }
                } else {
                    for (size_t c = 0; c < crad.size(); c++) {
                        const double dx = cx - ccen[3 * c], dy = cy - ccen[3 * c + 1], dz = cz - ccen[3 * c + 2];
                        const double gap = std::sqrt(dx * dx + dy * dy + dz * dz) - rt - crad[c] - r_hi;
                        if (!(gap * lambda > skip_base + clnw[c] - ln_near)) cost += kCl;
                    }
                }
                acc += cost;
            }
        const double per_plane = acc / ((double)K * K * (double)S) + 1e-3;   // + a floor: the per-block source scan and the stores cost something everywhere
        for (int k = L * bz; k < std::min(n, (L + 1) * bz); k++) weights[k] = per_plane;
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthr = (int)std::max(1u, std::min({hw ? hw : 1u, 16u, (unsigned)layers, (unsigned)(((double)layers * K * K * (double)S) / 2e6 + 1.)}));
    std::atomic<int> next{0};
    auto worker = [&]() {
        std::vector<double> dist((size_t)S);
        for (int L = next.fetch_add(1); L < layers; L = next.fetch_add(1)) do_layer(L, dist);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; t++) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
}

// Contiguous split of n planes into nslabs ranges whose boundaries are multiples of `granule` planes and whose weights are as equal as those boundaries allow
// (each boundary goes to the multiple of the granule nearest to its share of the cumulative weight; every slab keeps at least one granule).
static void plan_slabs_weighted(int n, int nslabs, const double* w, int granule, std::vector<int32_t>& bounds) {
    bounds.assign((size_t)nslabs + 1, 0);
    bounds[(size_t)nslabs] = n;
    if (granule < 1) granule = 1;
    if ((long long)granule * nslabs > n) granule = std::max(1, n / nslabs);
    std::vector<double> cum((size_t)n + 1, 0.);
    for (int k = 0; k < n; k++) cum[(size_t)k + 1] = cum[(size_t)k] + (w && w[k] > 0. ? w[k] : 0.);
    if (!(cum[(size_t)n] > 0.))
        for (int k = 0; k <= n; k++) cum[(size_t)k] = k;
    for (int i = 1; i < nslabs; i++) {
        const double target = cum[(size_t)n] * i / nslabs;
        const int lo = bounds[(size_t)i - 1] + granule, hi = n - (nslabs - i) * granule;   // leave a granule for every slab on either side
        int best = lo;
        double best_err = 1e300;
        for (int k = ((lo + granule - 1) / granule) * granule; k <= hi; k += granule) {
            const double e = std::fabs(cum[(size_t)k] - target);
            if (e < best_err) {
                best_err = e;
                best = k;
            }
        }
        bounds[(size_t)i] = std::min(std::max(best, lo), std::max(lo, hi));
    }
}

struct SolverBase {
    virtual ~SolverBase() = default;
    virtual void set_problem(const shm_sources&, const shm_grid&) = 0;
    virtual void solve(const shm_opts&, shm_stats*) = 0;
    virtual void get_phi(double*, int32_t*, int32_t*) = 0;
    virtual void owned_planes(int32_t*, int32_t*) = 0;
    virtual void run_conv(int step1_arith) = 0;
    virtual void run_divergence(int scrub) = 0;
    virtual void get_field(shm_field, double*) = 0;
    virtual void get_field_planes(shm_field, int, int, double*) = 0;
    virtual void apply_laplacian(const double*, double*) = 0;
    virtual void get_constraints(int64_t*, double*, int32_t*) = 0;
    virtual void apply_projector(double*) = 0;
    virtual void apply_preconditioner(const double*, double*) = 0;
    virtual void get_schur(double*, int32_t*) = 0;
    virtual void isosurface(double, int, int64_t*, int64_t*) = 0;
    virtual void get_isosurface(double*, int64_t*) = 0;
};

// one per precision, each in its own translation unit (shm_solver_f64.hip / shm_solver_f32.hip)
SolverBase* make_solver_f64(const shm_config& cfg);
SolverBase* make_solver_f32(const shm_config& cfg);

}  // namespace shm
