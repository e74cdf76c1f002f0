// Device-resident landmark table (include/svgpu.h svgpu_map_*): what the per-frame tracking loop reads of a data::landmark
// (pos_w_, mean_normal_, min / max valid distance, representative descriptor, has_observation / will_be_erased), as 96-byte
// records indexed by data::landmark::id_.  tracking_module.cc:554-594 (frame::can_observe per local landmark),
// match/projection.cc:13-207 and optimize/pose_optimizer_g2o.cc:88-107 read it through the tracker's kernels (track_kernels.hip);
// the mutators of data::landmark keep it current through svgpu_map_upsert / svgpu_map_erase.
#include <algorithm>

#include "svgpu_internal.h"

static_assert(sizeof(svgpu_landmark_record) == 96, "svgpu_landmark_record is a 96-byte record");

namespace {
inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

// staged = n records behind n ids; one thread per 16-byte piece of a record (6 per record): coalesced reads, 96-byte scattered writes.
// Entries are applied in order of i within a launch only if no id repeats: the host collapses repeats (last wins) before the upload.
__global__ __launch_bounds__(256) void k_map_scatter(const uint32_t* __restrict__ ids, const uint4* __restrict__ staged, int n, uint4* __restrict__ table, int cap) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n * 6) return;
    const int i = k / 6, part = k - 6 * i;
    const uint32_t id = ids[i];
    if (id < (uint32_t)cap) table[(size_t)id * 6 + part] = staged[k];
}
__global__ __launch_bounds__(256) void k_map_erase(const uint32_t* __restrict__ ids, int n, svgpu_landmark_record* __restrict__ table, int cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = ids[i];
    if (id < (uint32_t)cap) table[id].flags = 0;
}
__global__ __launch_bounds__(256) void k_map_gather(const uint32_t* __restrict__ ids, int n, const uint4* __restrict__ table, int cap, uint4* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n * 6) return;
    const int i = k / 6, part = k - 6 * i;
    const uint32_t id = ids[i];
    out[k] = id < (uint32_t)cap ? table[(size_t)id * 6 + part] : make_uint4(0, 0, 0, 0);
}

// stream s waits for every read recorded since the last writer did (mutex held).  An event may be re-recorded once a wait on it has been
// enqueued (the wait refers to the record that preceded it), so the events go back to the pool right away.
int map_wait_readers(svgpu_ctx* ctx, svgpu_map* m, hipStream_t s) {
    for (hipEvent_t e : m->reads_pending) SV_HIP(ctx, hipStreamWaitEvent(s, e, 0));
    m->reads_pool.insert(m->reads_pool.end(), m->reads_pending.begin(), m->reads_pending.end());
    m->reads_pending.clear();
    return SVGPU_OK;
}

// grows the table to hold `need` records (contents kept); waits for every stream that may still touch the old allocation
int map_grow(svgpu_ctx* ctx, svgpu_map* m, int need) {
    if (need <= m->cap) return SVGPU_OK;
    size_t cap = m->cap > 0 ? (size_t)m->cap : 4096;
    while (cap < (size_t)need) cap *= 2;
    if (cap > (size_t)1 << 30) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map: landmark id beyond 2^30");
    svgpu_landmark_record* fresh = nullptr;
    SV_HIP(ctx, hipMalloc((void**)&fresh, cap * sizeof(svgpu_landmark_record)));
    hipStream_t s = ctx->stream;
    if (m->wrote) SV_HIP(ctx, hipStreamWaitEvent(s, m->ev_write, 0));
    {
        const int rcw = map_wait_readers(ctx, m, s);
        if (rcw) {
            (void)hipFree(fresh);
            return rcw;
        }
    }
    if (m->cap > 0) SV_HIP(ctx, hipMemcpyAsync(fresh, m->rec, (size_t)m->cap * sizeof(svgpu_landmark_record), hipMemcpyDeviceToDevice, s));
    SV_HIP(ctx, hipMemsetAsync(fresh + m->cap, 0, (cap - (size_t)m->cap) * sizeof(svgpu_landmark_record), s));
    SV_HIP(ctx, hipStreamSynchronize(s));  // the old table is free of readers and writers from here on
    if (m->rec) SV_HIP(ctx, hipFree(m->rec));
    m->rec = fresh;
    m->cap = (int)cap;
    return SVGPU_OK;
}
}  // namespace

int sv_map_reader_begin(svgpu_ctx* ctx, svgpu_map* m, hipStream_t s) {
    if (m->wrote) SV_HIP(ctx, hipStreamWaitEvent(s, m->ev_write, 0));
    return SVGPU_OK;
}
int sv_map_reader_end(svgpu_ctx* ctx, svgpu_map* m, hipStream_t s) {
    if (m->reads_pending.size() >= 64) {  // a long stretch of reads without a write: forget the ones that have completed
        size_t kept = 0;
        for (hipEvent_t e : m->reads_pending) {
            if (hipEventQuery(e) == hipSuccess) m->reads_pool.push_back(e);
            else m->reads_pending[kept++] = e;
        }
        m->reads_pending.resize(kept);
    }
    hipEvent_t e = nullptr;
    if (!m->reads_pool.empty()) {
        e = m->reads_pool.back();
        m->reads_pool.pop_back();
    }
    else SV_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const hipError_t err = hipEventRecord(e, s);
    if (err != hipSuccess) {
        m->reads_pool.push_back(e);
        return sv_set_error(ctx, SVGPU_ERR_HIP, "hipEventRecord(read event)", err);
    }
    m->reads_pending.push_back(e);
    return SVGPU_OK;
}

extern "C" {

int svgpu_map_create(svgpu_ctx* ctx, svgpu_map** out) {
    if (!ctx || !out) return SVGPU_ERR_INVALID;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    svgpu_map* m = new (std::nothrow) svgpu_map();
    if (!m) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_create: out of memory");
    m->device = ctx->device;
    if (hipEventCreateWithFlags(&m->ev_write, hipEventDisableTiming) != hipSuccess) {
        svgpu_map_destroy(m);
        return sv_set_error(ctx, SVGPU_ERR_HIP, "svgpu_map_create: hipEventCreate");
    }
    *out = m;
    return SVGPU_OK;
}

void svgpu_map_destroy(svgpu_map* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->ev_write) {
        if (m->wrote) (void)hipEventSynchronize(m->ev_write);
        (void)hipEventDestroy(m->ev_write);
    }
    for (hipEvent_t e : m->reads_pending) {
        (void)hipEventSynchronize(e);
        (void)hipEventDestroy(e);
    }
    for (hipEvent_t e : m->reads_pool) (void)hipEventDestroy(e);
    if (m->rec) (void)hipFree(m->rec);
    delete m;
}

int svgpu_map_capacity(const svgpu_map* m) { return m ? m->cap : -1; }

int svgpu_map_upsert(svgpu_ctx* ctx, svgpu_map* m, int n, const uint32_t* ids, const svgpu_landmark_record* records) {
    if (!ctx || !m || n < 0 || (n > 0 && (!ids || !records))) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_upsert: bad arguments");
    if (m->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_upsert: the map lives on another device");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t max_id = 0;
    for (int i = 0; i < n; ++i) max_id = ids[i] > max_id ? ids[i] : max_id;
    if (max_id >= (1u << 30)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_upsert: landmark id beyond 2^30");
    std::lock_guard<std::mutex> lock(m->mtx);
    int rc = map_grow(ctx, m, (int)max_id + 1);
    if (rc) return rc;
    // page-locked image: ids | records, with repeated ids collapsed to their LAST entry (the scatter is unordered)
    const size_t o_rec = pad256((size_t)n * 4), total = o_rec + (size_t)n * sizeof(svgpu_landmark_record);
    if ((rc = sv_ensure_stage(ctx, total))) return rc;
    if ((rc = sv_ensure_scratch(ctx, total))) return rc;
    uint32_t* h_ids = (uint32_t*)ctx->h_stage;
    svgpu_landmark_record* h_rec = (svgpu_landmark_record*)(ctx->h_stage + o_rec);
    int kept = 0;
    {
        bool repeat = false;
        if (n > 1) {
            std::vector<uint32_t> sorted(ids, ids + n);
            std::sort(sorted.begin(), sorted.end());
            for (int i = 1; i < n && !repeat; ++i) repeat = sorted[i] == sorted[i - 1];
        }
        if (!repeat) {
            memcpy(h_ids, ids, (size_t)n * 4);
            memcpy(h_rec, records, (size_t)n * sizeof(svgpu_landmark_record));
            kept = n;
        }
        else {
            std::map<uint32_t, int> where;
            for (int i = 0; i < n; ++i) where[ids[i]] = i;
            for (const auto& kv : where) {
                h_ids[kept] = kv.first;
                h_rec[kept] = records[kv.second];
                ++kept;
            }
        }
    }
    hipStream_t s = ctx->stream;
    if ((rc = map_wait_readers(ctx, m, s))) return rc;
    if (m->wrote) SV_HIP(ctx, hipStreamWaitEvent(s, m->ev_write, 0));
    char* d = (char*)ctx->d_scratch;
    SV_HIP(ctx, hipMemcpyAsync(d, ctx->h_stage, o_rec + (size_t)kept * sizeof(svgpu_landmark_record), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_map_scatter, dim3((kept * 6 + 255) / 256), dim3(256), 0, s, (const uint32_t*)d, (const uint4*)(d + o_rec), kept, (uint4*)m->rec, m->cap);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipEventRecord(m->ev_write, s));
    m->wrote = true;
    SV_HIP(ctx, hipStreamSynchronize(s));  // the staging buffers belong to the context's next call
    return SVGPU_OK;
}

int svgpu_map_erase(svgpu_ctx* ctx, svgpu_map* m, int n, const uint32_t* ids) {
    if (!ctx || !m || n < 0 || (n > 0 && !ids)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_erase: bad arguments");
    if (m->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_erase: the map lives on another device");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(m->mtx);
    if (m->cap == 0) return SVGPU_OK;
    int rc = sv_ensure_stage(ctx, (size_t)n * 4);
    if (rc) return rc;
    if ((rc = sv_ensure_scratch(ctx, (size_t)n * 4))) return rc;
    memcpy(ctx->h_stage, ids, (size_t)n * 4);
    hipStream_t s = ctx->stream;
    if ((rc = map_wait_readers(ctx, m, s))) return rc;
    if (m->wrote) SV_HIP(ctx, hipStreamWaitEvent(s, m->ev_write, 0));
    SV_HIP(ctx, hipMemcpyAsync(ctx->d_scratch, ctx->h_stage, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_map_erase, dim3((n + 255) / 256), dim3(256), 0, s, (const uint32_t*)ctx->d_scratch, n, m->rec, m->cap);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipEventRecord(m->ev_write, s));
    m->wrote = true;
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_map_download(svgpu_ctx* ctx, const svgpu_map* cm, int n, const uint32_t* ids, svgpu_landmark_record* records) {
    svgpu_map* m = const_cast<svgpu_map*>(cm);
    if (!ctx || !m || n < 0 || (n > 0 && (!ids || !records))) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_download: bad arguments");
    if (m->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_map_download: the map lives on another device");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(m->mtx);
    const size_t o_rec = pad256((size_t)n * 4), total = o_rec + (size_t)n * sizeof(svgpu_landmark_record);
    int rc = sv_ensure_stage(ctx, total);
    if (rc) return rc;
    if ((rc = sv_ensure_scratch(ctx, total))) return rc;
    memcpy(ctx->h_stage, ids, (size_t)n * 4);
    hipStream_t s = ctx->stream;
    SvMapReadScope scope(ctx, m, s);
    if ((rc = scope.begin())) return rc;
    char* d = (char*)ctx->d_scratch;
    SV_HIP(ctx, hipMemcpyAsync(d, ctx->h_stage, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_map_gather, dim3((n * 6 + 255) / 256), dim3(256), 0, s, (const uint32_t*)d, n, (const uint4*)m->rec, m->cap, (uint4*)(d + o_rec));
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(ctx->h_stage + o_rec, d + o_rec, (size_t)n * sizeof(svgpu_landmark_record), hipMemcpyDeviceToHost, s));
    if ((rc = scope.end())) return rc;
    SV_HIP(ctx, hipStreamSynchronize(s));
    memcpy(records, ctx->h_stage + o_rec, (size_t)n * sizeof(svgpu_landmark_record));
    return SVGPU_OK;
}

}  // extern "C"
