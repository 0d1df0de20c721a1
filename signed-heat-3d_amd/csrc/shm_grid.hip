// libshm_grid.so -- the C ABI (include/shm_grid.h) over shm::SolverBase; the solvers themselves live in shm_solver.hip.h (instantiated per precision in shm_solver_f64.hip / _f32.hip).
#include "shm_host.hip.h"


// =================================================================================================
// C ABI
// =================================================================================================
struct shm_solver {
    std::unique_ptr<shm::SolverBase> impl;
    std::string err;
    shm_config cfg;
};

static thread_local std::string g_create_error;

template <typename F> static shm_status guard(shm_solver* s, F&& f) {
    if (!s) return SHM_ERR_INVALID;
    try {
        f();
        s->err.clear();
        return SHM_OK;
    } catch (const shm::Error& e) {
        s->err = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        s->err = "host allocation failed";
        return SHM_ERR_NOMEM;
    } catch (const std::exception& e) {
        s->err = e.what();
        return SHM_ERR_INVALID;
    }
}

extern "C" {

int32_t shm_grid_abi_version(void) { return SHM_GRID_ABI_VERSION; }

void shm_plan_slab(int32_t n, int32_t nslabs, int32_t slab, int32_t* k0, int32_t* k1) {
    if (nslabs < 1 || n < 0 || slab < 0 || slab >= nslabs) {   // bad arguments: the empty range
        if (k0) *k0 = 0;
        if (k1) *k1 = 0;
        return;
    }
    // balanced contiguous split: the first (n % nslabs) slabs get one extra plane
    const int32_t q = n / nslabs, r = n % nslabs;
    const int32_t b = slab * q + (slab < r ? slab : r);
    if (k0) *k0 = b;
    if (k1) *k1 = b + q + (slab < r ? 1 : 0);
}

shm_status shm_step1_plane_weights(const shm_sources* src, const shm_grid* grid, int32_t precision, double* weights) {
    if (!src || !grid || !weights || src->S <= 0 || !src->pos || !src->wnormal || grid->n < 1 || !(grid->cell > 0.) || !(src->lambda > 0.) ||
        (precision != SHM_F64 && precision != SHM_F32))
        return SHM_ERR_INVALID;
    const char* tl = shm::knob("SHM_CONV_TIER_LOG");
    try {
        shm::step1_plane_weights_host(src->S, src->pos, src->wnormal, src->lambda, grid->n, grid->bbox_min, grid->cell, precision, tl ? atof(tl) : 8.0, weights,
                                      precision == SHM_F32 && shm::knob("SHM_CONV32_CLASSIC") == nullptr);
    } catch (const std::bad_alloc&) {
        return SHM_ERR_NOMEM;
    } catch (...) {
        return SHM_ERR_INVALID;
    }
    return SHM_OK;
}

void shm_plan_slab_weighted(int32_t n, int32_t nslabs, int32_t slab, const double* weights, int32_t granule, int32_t* k0, int32_t* k1) {
    // bad arguments (n < nslabs, nslabs < 1, slab out of range) or a failed allocation: the empty range [0, 0) -- nothing throws across the boundary
    if (k0) *k0 = 0;
    if (k1) *k1 = 0;
    if (nslabs < 1 || n < nslabs || slab < 0 || slab >= nslabs) return;
    try {
        std::vector<int32_t> b;
        shm::plan_slabs_weighted(n, nslabs, weights, granule, b);   // weights == NULL: equal weights
        if (k0) *k0 = b[(size_t)slab];
        if (k1) *k1 = b[(size_t)slab + 1];
    } catch (...) {
    }
}

shm_status shm_grid_create(const shm_config* cfg, shm_solver** out) {
    if (!cfg || !out) {
        g_create_error = "null argument";
        return SHM_ERR_INVALID;
    }
    *out = nullptr;
    try {
        shm_config c = *cfg;
        if (c.local_slabs <= 0) c.local_slabs = 1;
        if (c.world <= 0) c.world = 1;
        if (c.rank < 0 || c.rank >= c.world) throw shm::Error(SHM_ERR_INVALID, "rank out of range");
        if (c.precision != SHM_F64 && c.precision != SHM_F32) throw shm::Error(SHM_ERR_INVALID, "precision must be SHM_F64 or SHM_F32");
        std::unique_ptr<shm_solver> s(new shm_solver());
        s->cfg = c;
        s->impl.reset(c.precision == SHM_F64 ? shm::make_solver_f64(c) : shm::make_solver_f32(c));
        *out = s.release();
        g_create_error.clear();
        return SHM_OK;
    } catch (const shm::Error& e) {
        g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return SHM_ERR_INVALID;
    }
}

void shm_grid_destroy(shm_solver* s) { delete s; }

const char* shm_grid_last_error(const shm_solver* s) { return s ? s->err.c_str() : g_create_error.c_str(); }

shm_status shm_grid_set_problem(shm_solver* s, const shm_sources* src, const shm_grid* grid) {
    return guard(s, [&] {
        if (!src || !grid) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->set_problem(*src, *grid);
    });
}

shm_status shm_grid_solve(shm_solver* s, const shm_opts* opts, shm_stats* stats) {
    return guard(s, [&] {
        shm_opts o;
        memset(&o, 0, sizeof o);
        o.scrub_nonfinite = 1;
        if (opts) o = *opts;
        s->impl->solve(o, stats);
    });
}

shm_status shm_grid_get_phi(shm_solver* s, double* phi_out, int32_t* k_begin, int32_t* k_end) {
    return guard(s, [&] {
        if (!phi_out) throw shm::Error(SHM_ERR_INVALID, "null phi_out");
        s->impl->get_phi(phi_out, k_begin, k_end);
    });
}

shm_status shm_grid_owned_planes(shm_solver* s, int32_t* k_begin, int32_t* k_end) {
    return guard(s, [&] { s->impl->owned_planes(k_begin, k_end); });
}

shm_status shm_grid_compute_distance(shm_solver* s, const shm_sources* src, const shm_grid* grid, const shm_opts* opts, double* phi_out,
                                     shm_stats* stats) {
    shm_status rc = shm_grid_set_problem(s, src, grid);
    if (rc != SHM_OK) return rc;
    if (s->cfg.world != 1) {
        s->err = "shm_grid_compute_distance is the single-process entry point; use set_problem/solve/get_phi per rank";
        return SHM_ERR_INVALID;
    }
    rc = shm_grid_solve(s, opts, stats);
    if (rc != SHM_OK && rc != SHM_ERR_NOCONV) return rc;
    std::string keep = s->err;
    shm_status rc2 = shm_grid_get_phi(s, phi_out, nullptr, nullptr);
    if (rc2 != SHM_OK) return rc2;
    s->err = keep;
    return rc;
}

shm_status shm_grid_run_conv(shm_solver* s) { return guard(s, [&] { s->impl->run_conv(SHM_STEP1_AUTO); }); }
shm_status shm_grid_run_conv_arith(shm_solver* s, int32_t step1_arith) { return guard(s, [&] { s->impl->run_conv(step1_arith); }); }
shm_status shm_grid_run_divergence(shm_solver* s, int32_t scrub) { return guard(s, [&] { s->impl->run_divergence(scrub); }); }
shm_status shm_grid_get_field(shm_solver* s, shm_field f, double* out) {
    return guard(s, [&] {
        if (!out) throw shm::Error(SHM_ERR_INVALID, "null out");
        s->impl->get_field(f, out);
    });
}
shm_status shm_grid_get_field_planes(shm_solver* s, shm_field f, int32_t k_begin, int32_t k_end, double* out) {
    return guard(s, [&] {
        if (!out) throw shm::Error(SHM_ERR_INVALID, "null out");
        s->impl->get_field_planes(f, k_begin, k_end, out);
    });
}
shm_status shm_grid_apply_laplacian(shm_solver* s, const double* u, double* out) {
    return guard(s, [&] {
        if (!u || !out) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->apply_laplacian(u, out);
    });
}
shm_status shm_grid_get_constraints(shm_solver* s, int64_t* nodes, double* coeffs, int32_t* m) {
    return guard(s, [&] {
        if (!nodes || !coeffs || !m) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->get_constraints(nodes, coeffs, m);
    });
}
shm_status shm_grid_get_schur(shm_solver* s, double* out, int32_t* m) {
    return guard(s, [&] {
        if (!out || !m) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->get_schur(out, m);
    });
}
shm_status shm_grid_apply_projector(shm_solver* s, double* v) {
    return guard(s, [&] {
        if (!v) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->apply_projector(v);
    });
}

shm_status shm_grid_apply_preconditioner(shm_solver* s, const double* v, double* out) {
    return guard(s, [&] {
        if (!v || !out) throw shm::Error(SHM_ERR_INVALID, "null argument");
        s->impl->apply_preconditioner(v, out);
    });
}

shm_status shm_grid_isosurface(shm_solver* s, double isovalue, int64_t* n_vertices, int64_t* n_triangles) {
    return guard(s, [&] { s->impl->isosurface(isovalue, SHM_ISO_MARCHING_CUBES, n_vertices, n_triangles); });
}
shm_status shm_grid_isosurface_ex(shm_solver* s, double isovalue, int32_t method, int64_t* n_vertices, int64_t* n_triangles) {
    return guard(s, [&] { s->impl->isosurface(isovalue, method, n_vertices, n_triangles); });
}
shm_status shm_grid_get_isosurface(shm_solver* s, double* vertices, int64_t* triangles) {
    return guard(s, [&] { s->impl->get_isosurface(vertices, triangles); });
}

shm_status shm_comm_unique_id(void* out128) {
    if (!out128) return SHM_ERR_INVALID;
    try {
        shm::Rccl& R = shm::Rccl::get();
        shm::Rccl::unique_id id;
        R.chk(R.GetUniqueId(&id), "ncclGetUniqueId");
        memcpy(out128, &id, sizeof id);
        return SHM_OK;
    } catch (const shm::Error& e) {
        g_create_error = e.what();
        return e.code;
    }
}

}  // extern "C"
