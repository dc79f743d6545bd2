// C ABI of libmcgpu.so (include/mcgpu.h): context, device-buffer pool, error plumbing, timing,
// path-matrix handles and layout conversion.  The numerical kernels live in kernels_*.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "fastmath_tables.hpp"
#include "mcg_internal.hpp"

namespace mcg {

static thread_local std::string g_err;
Stats g_stats;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return status;
}

// ---- pool ------------------------------------------------------------------------------------
int pool_alloc(mcg_ctx* ctx, size_t bytes, void** out) {
    // best fit among cached buffers that are not more than 25% larger than requested
    int best = -1;
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        const size_t b = ctx->pool[i].bytes;
        if (b >= bytes && b <= bytes + bytes / 4 + 4096 && (best < 0 || b < ctx->pool[best].bytes)) best = (int)i;
    }
    if (best >= 0) {
        *out = ctx->pool[best].ptr;
        ctx->pool.erase(ctx->pool.begin() + best);
        return MCG_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop the cache and retry once
        for (auto& b : ctx->pool) (void)hipFree(b.ptr);
        ctx->pool.clear();
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(MCG_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    *out = p;
    return MCG_OK;
}

void pool_release(mcg_ctx* ctx, void* ptr, size_t bytes) {
    if (ptr) ctx->pool.push_back(PoolBuf{ptr, bytes});
}

int ensure_cap(mcg_ctx* ctx, double** buf, size_t* cap, size_t need_doubles) {
    if (*cap >= need_doubles && *buf) return MCG_OK;
    if (*buf) {
        MCG_HIP(hipStreamSynchronize(ctx->stream));
        MCG_HIP(hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    size_t n = need_doubles + need_doubles / 2 + 1024;
    MCG_HIP(hipMalloc((void**)buf, n * sizeof(double)));
    *cap = n;
    return MCG_OK;
}

// ---- timing ----------------------------------------------------------------------------------
TimedLaunch::TimedLaunch(mcg_ctx* c, int k, int64_t launches) : ctx(c), kernel(k), on(c->timing && ((c->timing_mask >> k) & 1u)) {
    if (!on) return;
    if (!ctx->ev_free.empty()) {
        ev = ctx->ev_free.back();
        ctx->ev_free.pop_back();
    } else {
        if (hipEventCreate(&ev.a) != hipSuccess) {
            on = false;
            return;
        }
        if (hipEventCreate(&ev.b) != hipSuccess) {
            (void)hipEventDestroy(ev.a);
            on = false;
            return;
        }
    }
    ev.launches = launches;
    (void)hipEventRecord(ev.a, ctx->stream);
}

TimedLaunch::~TimedLaunch() {
    if (!on) return;
    (void)hipEventRecord(ev.b, ctx->stream);
    ctx->ev_live.emplace_back(kernel, ev);
}

static int timing_collect(mcg_ctx* ctx) {
    if (ctx->ev_live.empty()) return MCG_OK;
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->ev_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, kv.second.a, kv.second.b) == hipSuccess) {
            ctx->t_total[kv.first] += ms;
            ctx->t_count[kv.first] += kv.second.launches;
        }
        ctx->ev_free.push_back(kv.second);
    }
    ctx->ev_live.clear();
    return MCG_OK;
}

// ---- path handles ----------------------------------------------------------------------------
int paths_new(mcg_ctx* ctx, int64_t n_paths, int n_steps, uint64_t path_begin, mcg_paths** out) {
    mcg_paths* P = new (std::nothrow) mcg_paths();
    if (!P) return fail(MCG_ERR_OOM, "host allocation failed");
    P->ctx = ctx;
    P->n_paths = n_paths;
    P->n_steps = n_steps;
    // rows start 2 KiB aligned and are padded to a whole 256-thread block, so generator kernels
    // store unconditionally (columns >= n_paths are scratch and never read back)
    P->ld = (n_paths + 255) / 256 * 256;
    if (P->ld == 0) P->ld = 256;
    P->path_begin = path_begin;
    P->bytes = (size_t)P->ld * (size_t)(n_steps + 1) * sizeof(double);
    void* p = nullptr;
    int rc = pool_alloc(ctx, P->bytes, &p);
    if (rc) {
        delete P;
        return rc;
    }
    P->data = (double*)p;
    ctx->live_paths.push_back(P);
    *out = P;
    return MCG_OK;
}

// ---- layout conversion -----------------------------------------------------------------------
// dst[c*dst_ld + r] = src[r*src_ld + c] for r < R, c < C, through a padded 32x32 LDS tile.
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ src, int64_t src_ld,
                                                   double* __restrict__ dst, int64_t dst_ld, int64_t R, int64_t C) {
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < R && c < C) tile[k][tx] = src[r * src_ld + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < R && c < C) dst[c * dst_ld + r] = tile[tx][k];
    }
}

static int transpose(mcg_ctx* ctx, const double* src, int64_t src_ld, double* dst, int64_t dst_ld, int64_t R,
                     int64_t C) {
    if (R == 0 || C == 0) return MCG_OK;
    // gridDim.y and gridDim.x are limited (65535 and 2^31-1 blocks): walk over-long dimensions in slabs
    constexpr int64_t MAX_Y = 65535ll * 32, MAX_X = 0x7fffffffll / 32 * 32;
    for (int64_t r0 = 0; r0 < R; r0 += MAX_Y) {
        const int64_t Rs = std::min(R - r0, MAX_Y);
        for (int64_t c0 = 0; c0 < C; c0 += MAX_X) {
            const int64_t Cs = std::min(C - c0, MAX_X);
            const int64_t gx = (Cs + 31) / 32, gy = (Rs + 31) / 32;
            TimedLaunch t(ctx, MCG_K_TRANSPOSE);
            hipLaunchKernelGGL(k_transpose, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ctx->stream,
                               src + r0 * src_ld + c0, src_ld, dst + c0 * dst_ld + r0, dst_ld, Rs, Cs);
        }
    }
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

// chunk of paths moved per staging round trip (bounds the temporary to ~512 MiB)
static int64_t chunk_paths(int n_cols) {
    int64_t c = (int64_t)(512ull << 20) / ((int64_t)n_cols * 8);
    c = c / 64 * 64;
    return c < 64 ? 64 : c;
}

}  // namespace mcg

using namespace mcg;

extern "C" {

const char* mcg_last_error(void) { return g_err.c_str(); }

const char* mcg_version(void) { return "mcgpu 0.1 (gfx950, philox4x32-10, fp64)"; }

int mcg_device_count(int* count) {
    if (!count) return fail(MCG_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return fail(MCG_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return MCG_OK;
}

static int init_impl(mcg_ctx** out, int device, bool adopt, void* external_stream) {
    if (!out) return fail(MCG_ERR_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MCG_ERR_NO_DEVICE,
                    "no HIP device available (%s); libmcgpu has no CPU fallback by design",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device < 0 || device >= n) return fail(MCG_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    MCG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MCG_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MCG_ERR_NO_DEVICE, "device %d is %s; libmcgpu is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    mcg_ctx* ctx = new (std::nothrow) mcg_ctx();
    if (!ctx) return fail(MCG_ERR_OOM, "host allocation failed");
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->coop_launch = true;
    if (adopt) {
        ctx->stream = (hipStream_t)external_stream;
        ctx->owns_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete ctx;
            return fail(MCG_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se));
        }
        ctx->owns_stream = true;
    }
    if (hipMalloc((void**)&ctx->scalars, SCALARS_DOUBLES * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&ctx->clk_stamps, sizeof(unsigned long long) * 2 * GBM_CLK_SLOTS) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_scalars, SCALARS_DOUBLES * sizeof(double)) != hipSuccess) {
        mcg_finalize(ctx);
        return fail(MCG_ERR_OOM, "workspace allocation failed");
    }
    // (the three generated tables back to back = fm::Tables, fastmath.hpp)
    constexpr size_t n_log = sizeof(fm::LOG_TAB_HOST) / sizeof(double), n_sc = sizeof(fm::SINCOS_TAB_HOST) / sizeof(double);
    if (hipMalloc((void**)&ctx->log_tab, sizeof(fm::LOG_TAB_HOST) + sizeof(fm::SINCOS_TAB_HOST) + sizeof(fm::EXP2_TAB_HOST)) != hipSuccess ||
        hipMemcpy(ctx->log_tab, fm::LOG_TAB_HOST, sizeof(fm::LOG_TAB_HOST), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->log_tab + n_log, fm::SINCOS_TAB_HOST, sizeof(fm::SINCOS_TAB_HOST), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->log_tab + n_log + n_sc, fm::EXP2_TAB_HOST, sizeof(fm::EXP2_TAB_HOST), hipMemcpyHostToDevice) != hipSuccess) {
        mcg_finalize(ctx);
        return fail(MCG_ERR_OOM, "table upload failed");
    }
    *out = ctx;
    return MCG_OK;
}

int mcg_init(mcg_ctx** out, int device) { return init_impl(out, device, false, nullptr); }

int mcg_init_on_stream(mcg_ctx** out, int device, void* stream) { return init_impl(out, device, true, stream); }

int mcg_finalize(mcg_ctx* ctx) {
    if (!ctx) return MCG_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    shm_release(ctx);
    for (mcg_paths* P : ctx->live_paths) {  // handles the caller still holds: orphan them (see mcg_paths_free)
        if (P->data) (void)hipFree(P->data);
        P->data = nullptr;
        P->ctx = nullptr;
    }
    ctx->live_paths.clear();
    for (auto& b : ctx->pool) (void)hipFree(b.ptr);
    ctx->pool.clear();
    for (auto& kv : ctx->ev_live) {
        (void)hipEventDestroy(kv.second.a);
        (void)hipEventDestroy(kv.second.b);
    }
    for (auto& ev : ctx->ev_free) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->fin_chunks) (void)hipFree(ctx->fin_chunks);
    if (ctx->scalars) (void)hipFree(ctx->scalars);
    if (ctx->clk_stamps) (void)hipFree(ctx->clk_stamps);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    if (ctx->weights) (void)hipFree(ctx->weights);
    if (ctx->lsm_v) (void)hipFree(ctx->lsm_v);
    if (ctx->log_tab) (void)hipFree(ctx->log_tab);
    if (ctx->batch_fork) (void)hipEventDestroy(ctx->batch_fork);
    if (ctx->batch_join) (void)hipEventDestroy(ctx->batch_join);
    if (ctx->batch_aux) (void)hipStreamDestroy(ctx->batch_aux);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MCG_OK;
}

int mcg_synchronize(mcg_ctx* ctx) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    return MCG_OK;
}

int mcg_trim(mcg_ctx* ctx) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& b : ctx->pool) (void)hipFree(b.ptr);
    ctx->pool.clear();
    return MCG_OK;
}

int mcg_set_allreduce(mcg_ctx* ctx, mcg_allreduce_fn fn, void* user) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    shm_release(ctx);  // a collective installed from outside replaces the node-local one, mailbox included
    ctx->allreduce = fn;
    ctx->allreduce_user = user;
    return MCG_OK;
}

// ---- generation --------------------------------------------------------------------------------
static int check_gen_args(mcg_ctx* ctx, double S0, double dt, int n_steps, int64_t n_paths, mcg_paths** out) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(MCG_ERR_INVALID, "out is NULL");
    if (n_steps < 1) return fail(MCG_ERR_INVALID, "n_steps must be >= 1 (got %d)", n_steps);
    if (n_paths < 0) return fail(MCG_ERR_INVALID, "n_paths must be >= 0 (got %lld)", (long long)n_paths);
    if (!(dt > 0.0)) return fail(MCG_ERR_INVALID, "dt must be > 0");
    if (!std::isfinite(S0)) return fail(MCG_ERR_INVALID, "S0 must be finite");
    MCG_HIP(hipSetDevice(ctx->device));
    return MCG_OK;
}

static int gen_gbm(mcg_ctx* ctx, uint64_t seed, double S0, double r, double sigma, double dt, int n_steps,
                   uint64_t path_begin, int64_t n_paths, bool payoff, double K, int is_call, mcg_paths** out) {
    int rc = check_gen_args(ctx, S0, dt, n_steps, n_paths, out);
    if (rc) return rc;
    if (!(sigma >= 0.0)) return fail(MCG_ERR_INVALID, "sigma must be >= 0");
    mcg_paths* P = nullptr;
    rc = paths_new(ctx, n_paths, n_steps, path_begin, &P);
    if (rc) return rc;
    if (n_paths > 0) {
        rc = launch_gbm(ctx, P, seed, S0, r, sigma, dt, payoff, K, is_call);
        if (rc) {
            mcg_paths_free(P);
            return rc;
        }
    }
    *out = P;
    return MCG_OK;
}

int mcg_paths_gbm(mcg_ctx* ctx, uint64_t seed, double S0, double r, double sigma, double dt, int n_steps,
                  uint64_t path_begin, int64_t n_paths, mcg_paths** out) {
    return gen_gbm(ctx, seed, S0, r, sigma, dt, n_steps, path_begin, n_paths, false, 0.0, 0, out);
}

int mcg_paths_gbm_payoff(mcg_ctx* ctx, uint64_t seed, double S0, double r, double sigma, double dt, int n_steps,
                         uint64_t path_begin, int64_t n_paths, double K, int is_call, mcg_paths** out) {
    return gen_gbm(ctx, seed, S0, r, sigma, dt, n_steps, path_begin, n_paths, true, K, is_call, out);
}

static int gen_rb(mcg_ctx* ctx, uint64_t seed, double S0, double r, double xi, double H, double eta, double rho,
                  double dt, int n_steps, uint64_t path_begin, int64_t n_paths, bool payoff, double K, int is_call,
                  mcg_paths** out) {
    int rc = check_gen_args(ctx, S0, dt, n_steps, n_paths, out);
    if (rc) return rc;
    if (!(xi >= 0.0)) return fail(MCG_ERR_INVALID, "xi must be >= 0");
    if (!(H >= 0.0) || !std::isfinite(H)) return fail(MCG_ERR_INVALID, "H must be >= 0");
    if (!std::isfinite(eta)) return fail(MCG_ERR_INVALID, "eta must be finite");
    if (!(std::fabs(rho) <= 1.0)) return fail(MCG_ERR_INVALID, "|rho| must be <= 1");
    if (n_steps > (1 << 24)) return fail(MCG_ERR_INVALID, "rBergomi n_steps must be <= 2^24 (got %d)", n_steps);  // (66 000 years of trading days)
    mcg_paths* P = nullptr;
    rc = paths_new(ctx, n_paths, n_steps, path_begin, &P);
    if (rc) return rc;
    if (n_paths > 0) {
        rc = launch_rbergomi(ctx, P, seed, S0, r, xi, H, eta, dt, payoff, K, is_call);
        if (rc) {
            mcg_paths_free(P);
            return rc;
        }
    }
    *out = P;
    return MCG_OK;
}

int mcg_paths_rbergomi(mcg_ctx* ctx, uint64_t seed, double S0, double r, double xi, double H, double eta, double rho,
                       double dt, int n_steps, uint64_t path_begin, int64_t n_paths, mcg_paths** out) {
    return gen_rb(ctx, seed, S0, r, xi, H, eta, rho, dt, n_steps, path_begin, n_paths, false, 0.0, 0, out);
}

int mcg_paths_rbergomi_payoff(mcg_ctx* ctx, uint64_t seed, double S0, double r, double xi, double H, double eta,
                              double rho, double dt, int n_steps, uint64_t path_begin, int64_t n_paths, double K,
                              int is_call, mcg_paths** out) {
    return gen_rb(ctx, seed, S0, r, xi, H, eta, rho, dt, n_steps, path_begin, n_paths, true, K, is_call, out);
}

// ---- host <-> device ---------------------------------------------------------------------------
int mcg_paths_from_host(mcg_ctx* ctx, const double* row_major, int64_t n_paths, int n_cols, mcg_paths** out) {
    if (!ctx || !out) return fail(MCG_ERR_INVALID, "ctx/out is NULL");
    if (n_paths < 1 || n_cols < 1 || !row_major)
        return fail(MCG_ERR_EMPTY_PATHS, "LSM::PredictOptionPrice: Empty pricePaths.");
    MCG_HIP(hipSetDevice(ctx->device));
    mcg_paths* P = nullptr;
    int rc = paths_new(ctx, n_paths, n_cols - 1, 0, &P);
    if (rc) return rc;
    const int64_t chunk = chunk_paths(n_cols);
    void* tmp = nullptr;
    const size_t tmp_bytes = (size_t)std::min<int64_t>(chunk, n_paths) * n_cols * sizeof(double);
    rc = pool_alloc(ctx, tmp_bytes, &tmp);
    if (rc) {
        mcg_paths_free(P);
        return rc;
    }
    for (int64_t p0 = 0; p0 < n_paths && rc == MCG_OK; p0 += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, n_paths - p0);
        hipError_t e = hipMemcpyAsync(tmp, row_major + p0 * n_cols, (size_t)cnt * n_cols * sizeof(double),
                                      hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            rc = fail(MCG_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e));
            break;
        }
        // src rows = paths (cnt), cols = n_cols; dst[j*ld + p0 + p]
        rc = transpose(ctx, (const double*)tmp, n_cols, P->data + p0, P->ld, cnt, n_cols);
        if (rc == MCG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(MCG_ERR_HIP, "sync failed");
    }
    pool_release(ctx, tmp, tmp_bytes);
    if (rc) {
        mcg_paths_free(P);
        return rc;
    }
    *out = P;
    return MCG_OK;
}

int mcg_paths_to_host(const mcg_paths* P, double* row_major_out) {
    if (!P || !row_major_out) return fail(MCG_ERR_INVALID, "paths/out is NULL");
    mcg_ctx* ctx = P->ctx;
    if (!ctx) return fail(MCG_ERR_INVALID, "paths outlived their ctx (mcg_finalize was called first)");
    if (P->n_paths == 0) return MCG_OK;
    MCG_HIP(hipSetDevice(ctx->device));
    const int n_cols = P->n_steps + 1;
    const int64_t chunk = chunk_paths(n_cols);
    void* tmp = nullptr;
    const size_t tmp_bytes = (size_t)std::min<int64_t>(chunk, P->n_paths) * n_cols * sizeof(double);
    int rc = pool_alloc(ctx, tmp_bytes, &tmp);
    if (rc) return rc;
    for (int64_t p0 = 0; p0 < P->n_paths && rc == MCG_OK; p0 += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, P->n_paths - p0);
        // src rows = steps (n_cols), cols = paths (cnt); dst[p*n_cols + j]
        rc = transpose(ctx, P->data + p0, P->ld, (double*)tmp, n_cols, n_cols, cnt);
        if (rc) break;
        hipError_t e = hipMemcpyAsync(row_major_out + p0 * n_cols, tmp, (size_t)cnt * n_cols * sizeof(double),
                                      hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(MCG_ERR_HIP, "D2H copy failed: %s", hipGetErrorString(e));
    }
    pool_release(ctx, tmp, tmp_bytes);
    return rc;
}

int mcg_paths_to_host_step_major(const mcg_paths* P, double* out) {
    if (!P || !out) return fail(MCG_ERR_INVALID, "paths/out is NULL");
    if (!P->ctx) return fail(MCG_ERR_INVALID, "paths outlived their ctx (mcg_finalize was called first)");
    if (P->n_paths == 0) return MCG_OK;
    MCG_HIP(hipSetDevice(P->ctx->device));
    MCG_HIP(hipMemcpy2DAsync(out, (size_t)P->n_paths * sizeof(double), P->data, (size_t)P->ld * sizeof(double),
                             (size_t)P->n_paths * sizeof(double), (size_t)(P->n_steps + 1), hipMemcpyDeviceToHost,
                             P->ctx->stream));
    MCG_HIP(hipStreamSynchronize(P->ctx->stream));
    return MCG_OK;
}

int mcg_paths_info(const mcg_paths* P, int64_t* n_paths, int* n_steps, int64_t* ld, void** device_ptr) {
    if (!P) return fail(MCG_ERR_INVALID, "paths is NULL");
    if (n_paths) *n_paths = P->n_paths;
    if (n_steps) *n_steps = P->n_steps;
    if (ld) *ld = P->ld;
    if (device_ptr) *device_ptr = P->data;
    return MCG_OK;
}

int mcg_paths_free(mcg_paths* P) {
    if (!P) return MCG_OK;
    if (mcg_ctx* ctx = P->ctx) {  // NULL: the ctx was finalized first and took the device memory with it
        if (P->data) pool_release(ctx, P->data, P->bytes);
        for (size_t i = ctx->live_paths.size(); i-- > 0;) {  // usually the most recent handle
            if (ctx->live_paths[i] == P) {
                ctx->live_paths[i] = ctx->live_paths.back();
                ctx->live_paths.pop_back();
                break;
            }
        }
    }
    delete P;
    return MCG_OK;
}

// ---- pricing -----------------------------------------------------------------------------------
int mcg_price_european(mcg_ctx* ctx, const mcg_paths* P, double K, double r, double T, int is_call, double* mean,
                       double* std_err) {
    if (!ctx || !P || !mean) return fail(MCG_ERR_INVALID, "ctx/paths/mean is NULL");
    if (P->ctx != ctx) return fail(MCG_ERR_INVALID, "paths belong to a different ctx");
    MCG_HIP(hipSetDevice(ctx->device));
    double s[3];
    if (P->has_sums && P->sums_K == K && (P->sums_is_call != 0) == (is_call != 0)) {
        s[0] = P->sums[0];
        s[1] = P->sums[1];
        s[2] = P->sums[2];
    } else {
        if (P->n_paths > 0) {
            int rc = launch_payoff_sums(ctx, P, K, is_call, s);
            if (rc) return rc;
        } else {
            // an empty shard still has to take part in the collective
            int rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, 2);
            if (rc) return rc;
            rc = finish_sums(ctx, 0, 0, s);
            if (rc) return rc;
        }
    }
    const double n = s[2];
    if (!(n >= 1.0)) return fail(MCG_ERR_EMPTY_PATHS, "no paths to price");
    const double disc = std::exp(-r * T);
    const double m = s[0] / n;
    const double var = n > 1.0 ? std::max(0.0, (s[1] - n * m * m) / (n - 1.0)) : 0.0;
    *mean = disc * m;
    if (std_err) *std_err = disc * std::sqrt(var / n);
    return MCG_OK;
}

int mcg_price_lsm(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                  int poly_order, double* mean, double* std_err) {
    if (!ctx || !P || !mean) return fail(MCG_ERR_INVALID, "ctx/paths/mean is NULL");
    if (P->ctx != ctx) return fail(MCG_ERR_INVALID, "paths belong to a different ctx");
    if (poly_order < 0 || poly_order > 15) return fail(MCG_ERR_INVALID, "poly_order must be in [0,15] (got %d)", poly_order);
    if (P->n_paths < 1 && !ctx->allreduce) return fail(MCG_ERR_EMPTY_PATHS, "LSM::PredictOptionPrice: Empty pricePaths.");
    MCG_HIP(hipSetDevice(ctx->device));
    return run_lsm(ctx, P, r, K, maturity, dt, is_call, poly_order, mean, std_err);
}

int mcg_lsm_one_launch_enabled(mcg_ctx* ctx, int* enabled) {
    if (!ctx || !enabled) return fail(MCG_ERR_INVALID, "ctx/enabled is NULL");
    *enabled = ctx->coop_launch ? 1 : 0;
    return MCG_OK;
}

int mcg_lsm_one_launch_reset(mcg_ctx* ctx) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ctx->coop_launch = true;
    ctx->coop_retry_in = 0;
    return MCG_OK;
}

int mcg_debug_lsm_hooks(mcg_ctx* ctx, long long spin_limit, int poll_delay) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    if (spin_limit > 0xffffffffLL || poll_delay < 0 || poll_delay > 1000) return fail(MCG_ERR_INVALID, "bad hook values");
    ctx->lsm_spin_limit = spin_limit;
    ctx->lsm_poll_delay = poll_delay;
    return MCG_OK;
}

int mcg_price_asymptotic(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                         double sigma, double dividend, double* price) {
    if (!ctx || !P || !price) return fail(MCG_ERR_INVALID, "ctx/paths/price is NULL");
    if (P->ctx != ctx) return fail(MCG_ERR_INVALID, "paths belong to a different ctx");
    *price = 0.0;
    if (P->n_paths < 1 && !ctx->allreduce) return MCG_OK;  // AsymptoticAnalysisPricer.cpp:47-49
    if (sigma <= 0.0) return fail(MCG_ERR_INVALID, "AsymptoticAnalysis: Volatility must be positive.");  // :50-52
    MCG_HIP(hipSetDevice(ctx->device));
    return run_asymptotic(ctx, P, r, K, maturity, dt, is_call, sigma, dividend, price);
}

int mcg_price_martingale(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                         int poly_order, int max_iterations, double* price, double* lower, double* upper) {
    if (!ctx || !P || !price) return fail(MCG_ERR_INVALID, "ctx/paths/price is NULL");
    if (P->ctx != ctx) return fail(MCG_ERR_INVALID, "paths belong to a different ctx");
    if (P->n_paths < 1 && !ctx->allreduce) return fail(MCG_ERR_EMPTY_PATHS, "MartingaleOptimization: Empty pricePaths.");
    if (max_iterations <= 0) return fail(MCG_ERR_INVALID, "MartingaleOptimization: maxIterations must be positive.");
    if (poly_order < 0 || poly_order > 15) return fail(MCG_ERR_INVALID, "poly_order must be in [0,15] (got %d)", poly_order);
    MCG_HIP(hipSetDevice(ctx->device));
    return run_martingale(ctx, P, r, K, maturity, dt, is_call, poly_order, max_iterations, price, lower, upper);
}

int mcg_price_branching(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                        int num_branches, const int* exercise_times, int n_ex, uint64_t seed, double* price, double* lower,
                        double* upper) {
    if (!ctx || !P || !price) return fail(MCG_ERR_INVALID, "ctx/paths/price is NULL");
    if (P->ctx != ctx) return fail(MCG_ERR_INVALID, "paths belong to a different ctx");
    if (P->n_paths < 1 && !ctx->allreduce) return fail(MCG_ERR_EMPTY_PATHS, "BranchingProcesses: Empty pricePaths.");  // :22-24
    if (!exercise_times || n_ex < 1) return fail(MCG_ERR_INVALID, "BranchingProcesses: No exercise times.");              // :25-27
    if (K <= 0.0) return fail(MCG_ERR_INVALID, "BranchingProcesses: Strike must be positive.");                           // :28-30
    MCG_HIP(hipSetDevice(ctx->device));
    return run_branching(ctx, P, r, K, maturity, dt, is_call, num_branches, exercise_times, n_ex, seed, price, lower, upper);
}

int mcg_batch_price_rows(mcg_ctx* ctx, const mcg_row* rows, int64_t n_rows, int n_paths, double r, double dt,
                         int num_branches, int poly_order, int max_iterations, uint64_t seed, double* out) {
    if (!ctx || !out) return fail(MCG_ERR_INVALID, "ctx/out is NULL");
    if (n_rows < 0 || (n_rows > 0 && !rows)) return fail(MCG_ERR_INVALID, "bad rows");
    // (the row kernels serve up to 256 paths per row and orders up to 4 -- the driver uses 250 and 2; anything beyond is
    // priced row by row through the single-contract entry points, run_batch_rows)
    if (n_paths < 1) return fail(MCG_ERR_INVALID, "n_paths must be >= 1 (got %d)", n_paths);
    if (poly_order < 0 || poly_order > 15) return fail(MCG_ERR_INVALID, "poly_order must be in [0,15] (got %d)", poly_order);
    if (max_iterations <= 0) return fail(MCG_ERR_INVALID, "MartingaleOptimization: maxIterations must be positive.");
    if (!(dt > 0.0)) return fail(MCG_ERR_INVALID, "dt must be > 0");
    if (n_rows > (int64_t)1 << 31) return fail(MCG_ERR_INVALID, "too many rows for one call (row ids are 32 bits of the Philox counter)");
    if (n_rows == 0) return MCG_OK;
    MCG_HIP(hipSetDevice(ctx->device));
    try {  // (the planner's host vectors: nothing may leave an extern "C" entry point as an exception)
        return run_batch_rows(ctx, rows, n_rows, n_paths, r, dt, num_branches, poly_order, max_iterations, seed, out, nullptr);
    } catch (const std::exception& e) {
        return fail(MCG_ERR_OOM, "mcg_batch_price_rows: %s", e.what());
    }
}

int mcg_batch_price_rows6(mcg_ctx* ctx, const mcg_row* rows, const double* features2, int64_t n_rows, int n_paths, double r,
                          double dt, int num_branches, int poly_order, int max_iterations, uint64_t seed, double* out6) {
    if (!ctx || !out6) return fail(MCG_ERR_INVALID, "ctx/out6 is NULL");
    if (n_rows < 0 || (n_rows > 0 && !rows)) return fail(MCG_ERR_INVALID, "bad rows");
    // (the argument checks are mcg_batch_price_rows's -- all of them BEFORE anything is allocated)
    if (n_paths < 1) return fail(MCG_ERR_INVALID, "n_paths must be >= 1 (got %d)", n_paths);
    if (poly_order < 0 || poly_order > 15) return fail(MCG_ERR_INVALID, "poly_order must be in [0,15] (got %d)", poly_order);
    if (max_iterations <= 0) return fail(MCG_ERR_INVALID, "MartingaleOptimization: maxIterations must be positive.");
    if (!(dt > 0.0)) return fail(MCG_ERR_INVALID, "dt must be > 0");
    if (n_rows > (int64_t)1 << 31) return fail(MCG_ERR_INVALID, "too many rows for one call (row ids are 32 bits of the Philox counter)");
    if (n_rows == 0) return MCG_OK;
    std::vector<double> four;
    std::vector<unsigned char> priced;
    try {  // nothing may leave an extern "C" entry point as an exception
        four.resize((size_t)n_rows * 4);
        priced.resize((size_t)n_rows);
    } catch (const std::exception&) {
        return fail(MCG_ERR_OOM, "mcg_batch_price_rows6: no host memory for %lld rows", (long long)n_rows);
    }
    MCG_HIP(hipSetDevice(ctx->device));
    int rc;
    try {
        rc = run_batch_rows(ctx, rows, n_rows, n_paths, r, dt, num_branches, poly_order, max_iterations, seed, four.data(), priced.data());
    } catch (const std::exception& e) {
        return fail(MCG_ERR_OOM, "mcg_batch_price_rows6: %s", e.what());
    }
    if (rc) return rc;
    for (int64_t i = 0; i < n_rows; ++i) {
        const double* p = &four[(size_t)i * 4];
        double* o = out6 + 6 * i;
        // PredictionGen.cpp:739-805: a row the driver skips, or whose paths hold an inf / nan (the row kernels' scan of the row's
        // block, run_batch_chunk), is written as ",0,0,0,0,0,0" -- features included; every other row carries what its
        // pricers returned, finite or not, as the driver prints it (:809-816)
        const bool ok = priced[(size_t)i] != 0;
        for (int c = 0; c < 4; ++c) o[c] = ok ? p[c] : 0.0;
        o[4] = ok && features2 ? features2[2 * i] : 0.0;
        o[5] = ok && features2 ? features2[2 * i + 1] : 0.0;
    }
    return MCG_OK;
}

int mcg_row_features(const double* hist, size_t n, double* twenty_day_vol, double* twenty_day_momentum) {
    if (!twenty_day_vol || !twenty_day_momentum || (n > 0 && !hist)) return fail(MCG_ERR_INVALID, "bad arguments");
    host_row_features(hist, n, twenty_day_vol, twenty_day_momentum);
    return MCG_OK;
}

int mcg_row_build(const double* hist, size_t n, double underlying_last, double dte, double strike_dist_pct, int option_type,
                  double dividend, mcg_row* row, double features2[2]) {
    if (!row || !features2 || (n > 0 && !hist)) return fail(MCG_ERR_INVALID, "bad arguments");
    return host_row_build(hist, n, underlying_last, dte, strike_dist_pct, option_type, dividend, row, features2);
}

int mcg_probe_write_ceiling(mcg_ctx* ctx, int64_t n_paths, int n_steps, int reps, double* gb_per_s, double* ms_per_launch) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    if (n_paths < 512 || n_steps < 1 || reps < 1 || reps > 1000) return fail(MCG_ERR_INVALID, "bad probe shape");
    MCG_HIP(hipSetDevice(ctx->device));
    return probe_write_ceiling(ctx, n_paths, n_steps, reps, gb_per_s, ms_per_launch);
}

int mcg_generator_clock_arm(mcg_ctx* ctx, int on) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ctx->clk_armed = on != 0;
    return MCG_OK;
}

int mcg_generator_clock(mcg_ctx* ctx, double* ghz_median, int* n_stamps, double* ghz_min, double* ghz_max) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    MCG_HIP(hipSetDevice(ctx->device));
    return generator_clock(ctx, ghz_median, n_stamps, ghz_min, ghz_max);
}

int mcg_stats(mcg_stats_t* out, int reset) {
    if (!out && !reset) return fail(MCG_ERR_INVALID, "out is NULL");
    std::atomic<int64_t>* src[] = {&g_stats.lsm_one_launch_sweeps, &g_stats.lsm_one_launch_timeouts, &g_stats.lsm_per_date_sweeps,
                                   &g_stats.lsm_per_date_launches, &g_stats.lsm_per_date_refits, &g_stats.lsm_per_date_faults,
                                   &g_stats.shm_barrier_failures, &g_stats.peer_mailbox_enabled, &g_stats.peer_mailbox_refused,
                                   &g_stats.batch_calls, &g_stats.batch_chunks, &g_stats.batch_rows, &g_stats.batch_rows_singly,
                                   &g_stats.batch_peak_workspace_bytes, &g_stats.peer_mailbox_kept,
                                   &g_stats.coalesced_rounds, &g_stats.coalesced_calls, &g_stats.coalesced_peak_calls_per_round,
                                   &g_stats.coalesced_fallbacks, &g_stats.coalesced_round_us, &g_stats.coalesced_device_wait_us,
                                   &g_stats.coalesced_wake_us, &g_stats.coalesced_prefetched, &g_stats.coalesced_prefetch_hits};
    static_assert(sizeof(mcg_stats_t) == sizeof(src) / sizeof(src[0]) * sizeof(int64_t), "mcg_stats_t lists the counters in this order");
    int64_t* dst = reinterpret_cast<int64_t*>(out);
    for (size_t k = 0; k < sizeof(src) / sizeof(src[0]); ++k) {
        const int64_t v = reset ? src[k]->exchange(0, std::memory_order_relaxed) : src[k]->load(std::memory_order_relaxed);
        if (out) dst[k] = v;
    }
    return MCG_OK;
}

int mcg_debug_lsm_date_fault(mcg_ctx* ctx, int mode, int date, int workgroup, int delay, long long spin_limit) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    if (mode < 0 || mode > 2 || delay < 0 || delay > 100000 || spin_limit > 0xffffffffLL) return fail(MCG_ERR_INVALID, "bad hook values");
    ctx->lsm_date_hook[0] = mode;
    ctx->lsm_date_hook[1] = date;
    ctx->lsm_date_hook[2] = workgroup;
    ctx->lsm_date_hook[3] = delay;
    ctx->lsm_date_spin_limit = spin_limit;
    return MCG_OK;
}

int mcg_debug_batch_budget(mcg_ctx* ctx, size_t bytes) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ctx->batch_budget = bytes;
    return MCG_OK;
}

// ---- host-only pieces --------------------------------------------------------------------------
int mcg_estimate_params(const double* hist, size_t n, double out5[5]) {
    if (!out5) return fail(MCG_ERR_INVALID, "out5 is NULL");
    return host_estimate_params(hist, n, out5);
}

int mcg_rbergomi_spectrum(double H, double eta, double dt, int n_steps, double* amp, double* comp, int* Mz) {
    if (n_steps < 1 || !amp || !comp) return fail(MCG_ERR_INVALID, "bad arguments");
    std::vector<double> k, c;
    int rc = host_rbergomi_spectrum(H, eta, dt, n_steps, k, c);
    if (rc) return rc;
    std::memcpy(amp, k.data(), k.size() * sizeof(double));
    std::memcpy(comp, c.data(), c.size() * sizeof(double));
    if (Mz) *Mz = (int)k.size();
    return MCG_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int mcg_timing_enable(mcg_ctx* ctx, int on) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ctx->timing = on != 0;
    return MCG_OK;
}

int mcg_timing_select(mcg_ctx* ctx, unsigned mask) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ctx->timing_mask = mask;
    return MCG_OK;
}

int mcg_timing_reset(mcg_ctx* ctx) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    int rc = timing_collect(ctx);
    for (int i = 0; i < MCG_K_COUNT; ++i) {
        ctx->t_total[i] = 0.0;
        ctx->t_count[i] = 0;
    }
    return rc;
}

int mcg_timing_get(mcg_ctx* ctx, int kernel, double* total_ms, int64_t* launches) {
    if (!ctx || kernel < 0 || kernel >= MCG_K_COUNT) return fail(MCG_ERR_INVALID, "bad ctx/kernel id");
    int rc = timing_collect(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->t_total[kernel];
    if (launches) *launches = ctx->t_count[kernel];
    return MCG_OK;
}

}  // extern "C"
