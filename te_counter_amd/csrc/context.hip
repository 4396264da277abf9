// Context lifetime, error reporting and pinned host memory of libtecount.so.
#include <cstdlib>

#include "tecount_internal.h"

static thread_local std::string g_create_err;

int tec_fail(tec_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_err = buf;
    return code;
}

extern "C" int tec_version(void) { return TEC_VERSION; }

extern "C" const char* tec_last_error(const tec_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int tec_create(int device_id, tec_ctx** out) {
    if (!out) return tec_fail(nullptr, TEC_EINVAL, "tec_create: out is NULL");
    *out = nullptr;
    // the BAM decoder keeps the inflate launches of several chunks in flight, each on a stream of its own: with four
    // hardware queues per process (HIP's default) this library's streams share queues and those launches run one after
    // the other.  Raised to sixteen (a process that holds several contexts and a tensor library has a dozen streams) unless the caller asked for more (effective only if HIP is not initialised yet)
    const char* hwq = getenv("GPU_MAX_HW_QUEUES");
    if (!hwq || atoi(hwq) < 16) setenv("GPU_MAX_HW_QUEUES", "16", 1);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return tec_fail(nullptr, TEC_EHIP, "tec_create: no HIP device visible (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return tec_fail(nullptr, TEC_EINVAL, "tec_create: device %d out of range (have %d)", device_id, n);
    tec_ctx* ctx = new tec_ctx();
    ctx->device = device_id;
    auto bail = [&](hipError_t err, const char* what) {
        int rc = tec_fail(nullptr, TEC_EHIP, "tec_create: %s: %s", what, hipGetErrorString(err));
        delete ctx;
        return rc;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&ctx->s_compute, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipMalloc(&ctx->d_stats, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(stats)");
    if ((e = hipMemset(ctx->d_stats, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(stats)");
    if (tec_bulk_init(ctx) != TEC_OK) {
        g_create_err = ctx->err;
        tec_destroy(ctx);
        return TEC_EHIP;
    }
    *out = ctx;
    return TEC_OK;
}

extern "C" void tec_destroy(tec_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    tec_bulk_release(ctx);
    tec_sc_release(ctx);
    tec_gbam_cache_release(ctx);
    tec_spill_release(ctx);
    if (ctx->d_int2ext) (void)hipFree(ctx->d_int2ext);
    if (ctx->d_index_blob) (void)hipFree(ctx->d_index_blob);
    if (ctx->d_counts) (void)hipFree(ctx->d_counts);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
    if (ctx->s_compute) (void)hipStreamDestroy(ctx->s_compute);
    delete ctx;
}

extern "C" int tec_alloc_pinned(tec_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return tec_fail(ctx, TEC_EINVAL, "tec_alloc_pinned: NULL argument");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    TEC_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return TEC_OK;
}

extern "C" int tec_free_pinned(tec_ctx* ctx, void* p) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_free_pinned: NULL context");
    if (p) TEC_HIP(ctx, hipHostFree(p));
    return TEC_OK;
}
