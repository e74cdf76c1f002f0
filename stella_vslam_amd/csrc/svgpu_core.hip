// Context lifetime, error reporting, scratch management.
#include "svgpu_internal.h"

int sv_set_error(svgpu_ctx* ctx, int status, const char* what, hipError_t e) {
    if (ctx) {
        ctx->last_error = what ? what : "";
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return status;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one process may drive several GPUs from several threads
// (one svgpu_ctx each), so the largest size granted so far is tracked per (device, kernel) under a mutex.
#include <map>
#include <mutex>
hipError_t sv_allow_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mtx;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mtx);
    size_t& g = granted[std::make_pair(dev, kernel)];
    if (bytes <= g) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) g = bytes;
    else (void)hipGetLastError();  // never leave a sticky error behind: an impossible size fails the launch that needs it
    return e;
}

int sv_ensure_stage(svgpu_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->stage_bytes) return SVGPU_OK;
    if (ctx->h_stage) {
        SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SV_HIP(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr;
        ctx->stage_bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    SV_HIP(ctx, hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault));
    ctx->stage_bytes = want;
    return SVGPU_OK;
}

int sv_ensure_scratch(svgpu_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return SVGPU_OK;
    if (ctx->d_scratch) {
        SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SV_HIP(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    SV_HIP(ctx, hipMalloc(&ctx->d_scratch, want));
    ctx->scratch_bytes = want;
    return SVGPU_OK;
}

static inline bool prof_wants(const SvProf& P, const char* name) { return P.name == "*" || P.name == name; }
void sv_prof_begin(svgpu_ctx* ctx, hipStream_t s, const char* name) {
    SvProf& P = ctx->prof;
    if (!prof_wants(P, name)) return;
    SvProfClass& K = P.cls[name];
    if (K.used + 2 > K.ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            K.ev.push_back(e);
        }
    }
    (void)hipEventRecord(K.ev[K.used], s);
}
void sv_prof_end(svgpu_ctx* ctx, hipStream_t s, const char* name) {
    SvProf& P = ctx->prof;
    if (!prof_wants(P, name)) return;
    SvProfClass& K = P.cls[name];
    if (K.used + 2 > K.ev.size()) return;
    (void)hipEventRecord(K.ev[K.used + 1], s);
    K.used += 2;
}
unsigned long long* sv_prof_counter(svgpu_ctx* ctx, const char* name) {
    SvProf& P = ctx->prof;
    if (!prof_wants(P, name)) return nullptr;
    if (!P.d_counter) {
        if (hipMalloc((void**)&P.d_counter, sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            P.d_counter = nullptr;
            return nullptr;
        }
        (void)hipMemset(P.d_counter, 0, sizeof(unsigned long long));
    }
    return P.d_counter;
}
static void prof_collect(SvProfClass& K) {
    for (size_t i = 0; i + 1 < K.used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, K.ev[i], K.ev[i + 1]) == hipSuccess) {
            K.total_ms += ms;
            K.launches += 1;
        }
    }
    K.used = 0;
}

extern "C" {

const char* svgpu_profile_kernels(void) {
    return "k_resize,k_blur,k_fast,k_select,k_describe,k_bf_binsort,k_bf_topk,k_bf_replay,k_cand,k_stereo,ba_linearize,ba_schur,ba_solve,ba_update,ba_chi2,k_pose_opt";
}

int svgpu_profile_select(svgpu_ctx* ctx, const char* kernel_name) {
    if (!ctx) return SVGPU_ERR_INVALID;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipDeviceSynchronize());
    ctx->prof.name = kernel_name ? kernel_name : "";
    for (auto& kv : ctx->prof.cls) {
        kv.second.used = 0;
        kv.second.total_ms = 0;
        kv.second.launches = 0;
    }
    if (ctx->prof.d_counter) SV_HIP(ctx, hipMemset(ctx->prof.d_counter, 0, sizeof(unsigned long long)));
    return SVGPU_OK;
}

int svgpu_profile_mfma_ops(svgpu_ctx* ctx, unsigned long long* int8_ops) {
    if (!ctx || !int8_ops) return SVGPU_ERR_INVALID;
    *int8_ops = 0;
    if (!ctx->prof.d_counter) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipDeviceSynchronize());
    unsigned long long tiles = 0;
    SV_HIP(ctx, hipMemcpy(&tiles, ctx->prof.d_counter, sizeof(tiles), hipMemcpyDeviceToHost));
    // one patch = 64 queries x 32 targets x (256 bit positions + the popcount k-step of 32), a multiply and an add each: 18 x
    // v_mfma_i32_32x32x32_i8 -- what the matrix pipe EXECUTES (the distances themselves are 16 / 18 of it)
    *int8_ops = tiles * 64ull * 32ull * 288ull * 2ull;
    return SVGPU_OK;
}

int svgpu_profile_read_class(svgpu_ctx* ctx, const char* kernel_name, double* total_ms, long long* launches) {
    if (!ctx || !kernel_name) return SVGPU_ERR_INVALID;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipDeviceSynchronize());
    double ms = 0;
    long long n = 0;
    auto it = ctx->prof.cls.find(kernel_name);
    if (it != ctx->prof.cls.end()) {
        prof_collect(it->second);
        ms = it->second.total_ms;
        n = it->second.launches;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return SVGPU_OK;
}

int svgpu_profile_read(svgpu_ctx* ctx, double* total_ms, long long* launches) {
    if (!ctx) return SVGPU_ERR_INVALID;
    if (ctx->prof.name.empty() || ctx->prof.name == "*") {
        if (total_ms) *total_ms = 0;
        if (launches) *launches = 0;
        return SVGPU_OK;
    }
    return svgpu_profile_read_class(ctx, ctx->prof.name.c_str(), total_ms, launches);
}

int svgpu_abi_version(void) { return SVGPU_ABI_VERSION; }

int svgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int svgpu_create_with_priority(int device, int priority, svgpu_ctx** out) {
    if (!out) return SVGPU_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return SVGPU_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SVGPU_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return SVGPU_ERR_HIP;
    svgpu_ctx* ctx = new (std::nothrow) svgpu_ctx();
    if (!ctx) return SVGPU_ERR_INVALID;
    ctx->device = device;
    int least = 0, greatest = 0;  // numerically: greatest priority = lowest number
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio = priority > 0 ? greatest : priority < 0 ? least : 0;
    // the auxiliary stream carries the blur of a batch beside its FAST pass (svgpu_orb_extract*): same priority
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio) != hipSuccess
        || hipStreamCreateWithPriority(&ctx->stream_aux, hipStreamNonBlocking, prio) != hipSuccess
        || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_stage[0], hipEventDisableTiming) != hipSuccess
        || hipEventCreateWithFlags(&ctx->ev_stage[1], hipEventDisableTiming) != hipSuccess
        || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        svgpu_destroy(ctx);
        return SVGPU_ERR_HIP;
    }
    *out = ctx;
    return SVGPU_OK;
}

int svgpu_create(int device, svgpu_ctx** out) { return svgpu_create_with_priority(device, 0, out); }

void svgpu_destroy(svgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_aux) (void)hipStreamSynchronize(ctx->stream_aux);
    sv_orb_release(ctx);
    for (auto& kv : ctx->prof.cls)
        for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
    if (ctx->prof.d_counter) (void)hipFree(ctx->prof.d_counter);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    sv_comm_release(ctx);
    sv_sky_release(ctx);
    if (ctx->ev_ba) (void)hipEventDestroy(ctx->ev_ba);
    if (ctx->ev_ba_copy) (void)hipEventDestroy(ctx->ev_ba_copy);
    if (ctx->ba_copy_stream) (void)hipStreamDestroy(ctx->ba_copy_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (hipEvent_t e : ctx->ev_stage)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream_aux) (void)hipStreamDestroy(ctx->stream_aux);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* svgpu_last_error(const svgpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

const char* svgpu_status_string(int status) {
    switch (status) {
        case SVGPU_OK: return "ok";
        case SVGPU_ERR_INVALID: return "invalid argument";
        case SVGPU_ERR_HIP: return "HIP runtime error";
        case SVGPU_ERR_CAPACITY: return "output capacity exceeded";
        case SVGPU_ERR_NOT_CONFIGURED: return "not configured";
        case SVGPU_ERR_NO_DEVICE: return "no HIP device";
        case SVGPU_ERR_NUMERIC: return "numerical failure";
        case SVGPU_STOPPED: return "stopped by caller";
        default: return "unknown status";
    }
}

int svgpu_synchronize(svgpu_ctx* ctx) {
    if (!ctx) return SVGPU_ERR_INVALID;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVGPU_OK;
}

void* svgpu_stream(svgpu_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

}  // extern "C"
