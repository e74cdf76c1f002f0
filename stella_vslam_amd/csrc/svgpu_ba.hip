// Host side of local bundle adjustment: problem flattening, structure (CSR) construction, and the
// Levenberg-Marquardt control flow of g2o restated around the device kernels.
//   two-stage schedule                 optimize/local_bundle_adjuster_g2o.cc:306-348
//   OptimizationAlgorithmLevenberg     g2o (pinned 20230223_git): lambda init 1e-5 * max diag, rho test, 10 trials
//   terminate_action                   optimize/terminate_action.cc:36-76 (writes through the force-stop pointer)
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>

#include <chrono>
#include <cstdlib>
#include <dlfcn.h>
#include <sched.h>
#include <thread>

#include "svgpu_internal.h"
#include "ba_kernels.h"

void sv_ba_maxdiag(hipStream_t s, const BaDev& D);
void sv_ba_maxslot(hipStream_t s, const BaDev& D);
void sv_ba_owned_mark(hipStream_t s, const int* lm_off, int L, double* xch, double stop_vote);
void sv_ba_owned_check(hipStream_t s, const double* xch, int L, uint8_t* any_owner, double* verdict);
void sv_ba_points_share(hipStream_t s, const BaDev& D, double* points_out, double* xch, int dir);
void sv_ba_xs_move(hipStream_t s, const BaDev& D, const int* blk_idx, int nb, const int* slot_idx, int ns, double* buf, int dir);
bool sv_sky_partition_roles(int nP, const std::vector<int2>& blk_ab, int world, std::vector<int>& job_of, std::vector<int>& owner, int* ncuts, int* sep_blocks_n, long long* xch_doubles);
bool sv_sky_current_roles(svgpu_ctx* ctx, const std::vector<int>** job_of, const std::vector<int>** owner, const std::vector<int>** sep_blocks);
void sv_ba_zero_inactive(hipStream_t s, const BaDev& D);
size_t sv_ba_pairs_scratch_bytes(size_t pair_cap, int E, size_t nb_cap);
int sv_ba_build_pairs(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int2* pairs_out,
                      int* pair_l_out, std::vector<int>& dense_off_host);
int sv_ba_build_pairs_async(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int total,
                            int2* pairs_out, int* pair_l_out, int* dense_off_dev);
size_t sv_ba_pose_lists_scratch_bytes(size_t E);
size_t sv_ba_units_scratch_bytes(size_t num_pairs, size_t unit_cap);
int sv_ba_build_units(svgpu_ctx* ctx, hipStream_t s, const int* blk_off_dev, int NB, const int* pair_l_dev, int num_pairs, int L, int chunk_shift, int unit_cap, void* scratch,
                      size_t scratch_bytes, int4* unit_rec_out, int* blk_unit_off_out, int* num_units_out);
size_t sv_ba_renumber_scratch_bytes(size_t L);
int sv_ba_renumber_landmarks(svgpu_ctx* ctx, hipStream_t s, const int* lm_off_old, const int* e_pose_old, int L, int P, int E, void* scratch, size_t scratch_bytes,
                             int* order_out, int* lm_off_new, int* e_pose_new, int* e_point_new, int* src_edge);
void sv_ba_permute_measurements(hipStream_t s, const int* src_edge, int E, const float* uvr_old, const float* w_old, const float* hub_old, float* uvr_new, float* w_new, float* hub_new);
void sv_ba_permute_points(hipStream_t s, const int* order, int L, const double* pts_old, double* pts_new);
void sv_ba_permute_flags(hipStream_t s, const int* order, int L, const uint8_t* old_flags, uint8_t* new_flags);
int sv_ba_prepare_lists(svgpu_ctx* ctx, hipStream_t s, const int* e_pose_dev, int E, int P, void* scratch, size_t scratch_bytes, int* pe_off_dev, uint8_t* e_level_dev,
                        double* e_chi_dev, const void** sorted_out);
int sv_ba_prepare_pose_major(svgpu_ctx* ctx, hipStream_t s, const void* sorted, const int* e_point_dev, const float* e_uvr_dev, const float* e_w_dev, const float* e_huber_dev,
                             int E, int* pe_idx_dev, uint8_t* robust_dev, int* pm_point, float* pm_uvr, float* pm_w, float* pm_hub);
void sv_ba_build_pose_major(hipStream_t s, const int* pe_idx, const int* e_point, const float* e_uvr, const float* e_w, const float* e_hub, int E, int* pm_point,
                            float* pm_uvr, float* pm_w, float* pm_hub);
int sv_ba_build_pose_lists(svgpu_ctx* ctx, hipStream_t s, const int* e_pose_dev, const float* e_huber_dev, int E, int P, void* scratch, size_t scratch_bytes,
                           int* pe_off_dev, int* pe_idx_dev, uint8_t* robust_dev);

namespace {

struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(void* p) : base((char*)p) {}
    template <class T>
    T* take(size_t n) {
        T* r = (T*)(base + off);
        off += (n * sizeof(T) + 255) & ~size_t(255);
        return r;
    }
};
inline size_t pad(size_t b) { return (b + 255) & ~size_t(255); }

struct HostStructure {
    std::vector<int> pose_slot, slot_pose;
    std::vector<uint8_t> pt_free;
    std::vector<int> blk_off;
    std::vector<int2> blk_ab;
    bool envelope_ok = false;  // the envelope Cholesky has a plan for this block pattern
    size_t num_pairs = 0;  // the pairs themselves live on the device only
    int nP = 0, nL = 0;
};

// pose / landmark activity under the current edge levels (the cheap first half of build_structure)
// (no_levels: no edge is excluded yet -- the first stage of a call: the poses with an observation were marked by the staging pass and a
//  landmark is active when its edge range is not empty; saves a pass over all observations, ~0.5 ms of a config-5 call)
void activity_only(const svgpu_ba_problem& pr, const int* e_pose, const int* e_point,
                   const std::vector<uint8_t>& level, const std::vector<uint8_t>* pose_active_global, HostStructure& H,
                   const std::vector<uint8_t>* pose_seen_no_levels = nullptr, const int* lm_off = nullptr) {
    const int P = pr.num_poses, L = pr.num_points, E = pr.num_obs;
    std::vector<uint8_t> pa(P, 0), la(L, 0);
    if (pose_seen_no_levels && lm_off) {
        pa = *pose_seen_no_levels;
        for (int l = 0; l < L; ++l) la[l] = lm_off[l + 1] > lm_off[l];
    }
    else
        for (int e = 0; e < E; ++e)
            if (!level[e]) {
                pa[e_pose[e]] = 1;
                la[e_point[e]] = 1;
            }
    if (pose_active_global) pa = *pose_active_global;
    H.pose_slot.assign(P, -1);
    H.slot_pose.clear();
    for (int p = 0; p < P; ++p)
        if (pa[p] && !pr.pose_fixed[p]) {
            H.pose_slot[p] = (int)H.slot_pose.size();
            H.slot_pose.push_back(p);
        }
    H.nP = (int)H.slot_pose.size();
    H.pt_free.assign(L, 0);
    H.nL = 0;
    for (int l = 0; l < L; ++l)
        if (la[l] && !(pr.point_fixed && pr.point_fixed[l])) {
            H.pt_free[l] = 1;
            ++H.nL;
        }
}

// dense block offsets (from the device) -> kept blocks: every diagonal block (it carries Hpp + lambda I) and every non-empty
// off-diagonal block, in (a, b) order; the sorted pair array needs no compaction (empty blocks hold no pairs)
void compact_blocks(const std::vector<int>& dense_off, const std::vector<uint8_t>& present, HostStructure& H) {
    H.blk_ab.clear();
    H.blk_off.clear();
    size_t k = 0;
    for (int a = 0; a < H.nP; ++a)
        for (int b = a; b < H.nP; ++b, ++k)
            if (a == b || present[k]) {
                int2 ab;
                ab.x = a;
                ab.y = b;
                H.blk_ab.push_back(ab);
                H.blk_off.push_back(dense_off[k]);
            }
    H.blk_off.push_back(dense_off.empty() ? 0 : dense_off.back());
    H.num_pairs = (size_t)(dense_off.empty() ? 0 : dense_off.back());
}

}  // namespace

// ---- in-library collectives: RCCL resolved with dlopen on first use (no link-time dependency, nothing of Python in the data path)
namespace {
struct NcclId {
    char internal[128];  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};
struct Rccl {
    void* h = nullptr;
    int (*get_unique_id)(NcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, NcclId, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
    bool tried = false, ok = false;
};
Rccl g_rccl;
bool rccl_ready() {
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    // a process that already carries an RCCL (torch's) keeps using that copy
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names)
        if (!g_rccl.h) g_rccl.h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char* nm : names)
        if (!g_rccl.h) g_rccl.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.h) return false;
    g_rccl.get_unique_id = (int (*)(NcclId*))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (int (*)(void**, int, NcclId, int))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.comm_destroy = (int (*)(void*))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.get_error_string = (const char* (*)(int))dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
    return g_rccl.ok;
}
// svgpu_allreduce_fn bound to the context's communicator: sum of `count` doubles, in place, on `stream`
int rccl_allreduce_cb(void* user, double* dev_buf, size_t count, void* stream) {
    svgpu_ctx* ctx = (svgpu_ctx*)user;
    if (!ctx || !ctx->comm || !g_rccl.ok) return 1;
    return g_rccl.all_reduce(dev_buf, dev_buf, count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, (hipStream_t)stream);
}
}  // namespace

void sv_comm_release(svgpu_ctx* ctx) {
    if (ctx && ctx->comm && g_rccl.ok) g_rccl.comm_destroy(ctx->comm);
    if (ctx) ctx->comm = nullptr;
}

static int local_ba_impl(svgpu_ctx* ctx, const svgpu_ba_problem* pr, bool single_stage, int rank, int world,
                         svgpu_allreduce_fn allreduce, void* ar_user, volatile uint8_t* stop, double* pose_out, double* points_out, uint8_t* outlier_out,
                         svgpu_ba_stats* stats) {
    if (!ctx || !pr || !pose_out || !points_out) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: null argument");
    const int P = pr->num_poses, L = pr->num_points, E = pr->num_obs;
    if (P < 0 || L < 0 || E < 0 || (P > 0 && (!pr->pose_cw || !pr->pose_fixed || !pr->intrinsics)) || (L > 0 && !pr->points)
        || (E > 0 && (!pr->obs_pose || !pr->obs_point || !pr->obs_uvr || !pr->obs_inv_sigma_sq || (!outlier_out && !single_stage))))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: inconsistent problem");
    // the record gathers of k_ba_schur_rhs address the W / Hll / bl arrays with 32-bit byte offsets (144 bytes per observation, 48 per landmark)
    if ((size_t)E * 144 >= ((size_t)1 << 32) || (size_t)L * 48 >= ((size_t)1 << 32))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: more than 2^32 / 144 observations per rank (shard the landmarks over more ranks)");
    const bool sharded = allreduce != nullptr;
    if (sharded && (world < 1 || rank < 0 || rank >= world)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: bad rank/world");
    ctx->ba_ar_fn = allreduce;  // (the segmented envelope solve of ba_skyline.hip exchanges through the same all-reduce)
    ctx->ba_ar_user = ar_user;
    if (!sharded) {
        world = 1;
        rank = 0;
    }
    svgpu_ba_stats st;
    memset(&st, 0, sizeof(st));
    const bool trace = std::getenv("SVGPU_BA_TRACE") != nullptr;
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ba] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
        t_start = now;
    };
    // (the outputs start as the inputs: every early return leaves the estimate untouched.  At global-BA sizes the copy of the landmark
    //  positions -- 38 MB at 1.6 M landmarks -- is left to the host team below)
    const int team_min_obs = [] {  // (SVGPU_BA_TEAM_MIN_OBS, read per call: the tests take small problems through the team's set-up)
        const char* ev = std::getenv("SVGPU_BA_TEAM_MIN_OBS");
        return ev ? std::max(64, std::atoi(ev)) : 400000;
    }();
    const bool team_sized = E >= team_min_obs && !std::getenv("SVGPU_BA_ONE_THREAD");
    memcpy(pose_out, pr->pose_cw, sizeof(double) * 12 * (size_t)P);
    // (the flag is volatile and another thread raises it: ONE read decides both the copy and the return, so that no path leaves
    //  points_out unwritten -- ADVICE r5)
    const bool stopped_at_entry = !sharded && stop && *stop;  // local_bundle_adjuster_g2o.cc:308-310
    const bool team_copies_points = team_sized && !stopped_at_entry && E > 0 && P > 0 && L > 0;
    if (!team_copies_points) memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
    if (E > 0 && outlier_out) memset(outlier_out, 0, E);
    if (stats) *stats = st;
    if (!sharded) {
        if (stopped_at_entry) return SVGPU_STOPPED;
        if (E == 0 || P == 0 || L == 0) return SVGPU_OK;
    }
    else if (P == 0 || L == 0) return SVGPU_OK;  // same on every rank
    {
        const hipError_t e_dev = hipSetDevice(ctx->device);
        if (e_dev != hipSuccess) {  // an early return before the team exists: the estimate leaves as it came
            if (team_copies_points) memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
            SV_HIP(ctx, e_dev);
        }
    }
    hipStream_t s = ctx->stream;

    // ---- page-locked staging laid out exactly like the input block at the head of the device arena (one copy carries everything):
    //      poses | points | intrinsics | e_pose | e_point | e_uvr | e_w | e_huber | e_robust | lm_off | pe_off | pe_idx.  The observations
    //      are written sorted by landmark (stable; a plain copy when they arrive that way): landmark-major kernels then read
    //      contiguous runs.  pe_* = pose -> edge lists.  Behind it, the output block's host image.
    struct {
        size_t pose, points, intr, e_pose, e_point, e_uvr, e_w, e_hub, robust, lm_off, pe_off, pe_idx, total;
    } in;
    {
        size_t o = 0;
        auto put = [&](size_t bytes) {
            const size_t r = o;
            o += pad(bytes);
            return r;
        };
        in.pose = put(sizeof(double) * 12 * (size_t)P);
        in.points = put(sizeof(double) * 3 * (size_t)L);
        in.intr = put(sizeof(double) * 5 * (size_t)P);
        in.e_pose = put(4 * (size_t)E);
        in.e_point = put(4 * (size_t)E);
        in.e_uvr = put(12 * (size_t)E);
        in.e_w = put(4 * (size_t)E);
        in.e_hub = put(4 * (size_t)E);
        in.robust = put(E);
        in.lm_off = put(4 * (size_t)(L + 1));
        in.pe_off = put(4 * (size_t)(P + 1));
        in.pe_idx = put(4 * (size_t)E);
        in.total = o;
    }
    // structure block (host image only; its pieces go to separate device arrays): pt_free | pose_slot | slot_pose | blk_ab, prow_off, diag | prow_ent
    const size_t nb_cap_h = (size_t)P * (P + 1) / 2;
    const size_t st_pt_free = 0, st_pose_slot = pad(L), st_pe = st_pose_slot + pad(4 * (size_t)P), st_blk = st_pe + pad(4 * (size_t)P);
    const size_t st_prow = st_blk + pad(8 * nb_cap_h + 4 * (2 * (size_t)P + 2)), st_total = st_prow + pad(8 * 2 * (nb_cap_h + 1));
    const size_t out_ctl = 0, out_state = pad(sizeof(BaCtl)), out_outlier = out_state + pad(sizeof(double) * (12 * (size_t)P + 3 * (size_t)L));
    const size_t out_total = out_outlier + pad((size_t)E + 1);
    {
        const int r = sv_ensure_stage(ctx, in.total + out_total + st_total);
        if (r) {
            if (team_sized) memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
            return r;
        }
    }
    char* const hs = ctx->h_stage;
    char* const hs_out = hs + in.total;
    char* const hs_struct = hs_out + out_total;
    int* const lm_off = (int*)(hs + in.lm_off);
    int* const e_pose = (int*)(hs + in.e_pose);
    int* const e_point = (int*)(hs + in.e_point);
    float* const e_uvr = (float*)(hs + in.e_uvr);
    float* const e_w = (float*)(hs + in.e_w);
    float* const e_hub = (float*)(hs + in.e_hub);
    auto copy_measurements = [&](size_t a, size_t b) {
        memcpy(e_uvr + 3 * a, pr->obs_uvr + 3 * a, 12 * (b - a));
        memcpy(e_w + a, pr->obs_inv_sigma_sq + a, 4 * (b - a));
        if (pr->obs_huber_delta) memcpy(e_hub + a, pr->obs_huber_delta + a, 4 * (b - a));
        else memset(e_hub + a, 0, 4 * (b - a));
    };
    std::vector<int> perm;  // sorted position -> caller's observation index (empty = identity)
    std::vector<uint8_t> level((sharded || !single_stage) ? E : 0, 0);  // (host image of the edge levels: the stage boundary and the sharded set-up read it)
    bool no_levels = true;  // until the first gate: every edge is at level 0
    // host threads of the staging pass at global-BA sizes (SVGPU_BA_HOST_THREADS overrides, 1 .. 16)
    const int nth = [&] {
        if (!team_sized) return 1;
        const char* ev = std::getenv("SVGPU_BA_HOST_THREADS");
        const int v = ev ? std::atoi(ev) : 8;  // (config 5: 1.28 ms with 4 threads, 0.69 with 8, flat beyond)
        // (the team spins at its barriers: never more threads than CPUs this process may run on -- the affinity mask, which is what a
        //  container's cpuset leaves, not the machine's core count)
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t cpus;
        if (sched_getaffinity(0, sizeof(cpus), &cpus) == 0 && CPU_COUNT(&cpus) > 0) hw = hw == 0 ? (unsigned)CPU_COUNT(&cpus) : std::min(hw, (unsigned)CPU_COUNT(&cpus));
        const int cap = hw == 0 ? 16 : (int)std::min(16u, hw);
        return v < 1 ? 1 : (v > cap ? cap : v);
    }();
    // The host team of a global-BA sized call: nth threads (this one included) scan the observation indices -- range check, "already grouped
    // by landmark?" (the order local_bundle_adjuster_g2o.cc:168-227 and global_bundle_adjuster.cc:66-118 create their edges in) and,
    // valid in that case, the landmark offsets -- and stage the two index arrays; the nth - 1 workers then go on staging the measurements
    // and the landmark positions while this thread runs the device's structure work, and upload them on the copy stream once the arena
    // is known (`go`).  One spawn per call; the phases of the scan are separated by spin barriers (a few microseconds each).
    struct HostTeam {
        std::vector<std::thread> th;
        std::atomic<int> arrived{0};
        std::atomic<int> gate{0};    // 0: the team is being created, 1: run, 2: creation failed, leave
        std::atomic<int> staged{0};  // workers whose share of the measurements is in the staging image
        std::atomic<int> go{0};  // 0: the arena is not known yet, 1: upload, 2: abandon (the caller returns early or stages the observations itself)
        char* d_in = nullptr;
        hipStream_t s2 = nullptr;
        int err[16] = {0};
        bool running = false;
        void join() {
            for (auto& t : th)
                if (t.joinable()) t.join();
            th.clear();
            running = false;
        }
        void abandon_and_join() {
            int z = 0;
            go.compare_exchange_strong(z, 2);
            join();
        }
    } team;
    bool lm_major = true;
    std::vector<uint8_t> pose_seen(P, 0);  // a pose with an observation (the pose half of the first stage's activity pass)
    // (function scope: the workers run these closures until the team is joined -- TeamGuard below, declared behind them, goes first)
        int bad[16] = {0}, unsorted[16] = {0};
        std::vector<uint8_t> seen_q[16];
        const int32_t* const op = pr->obs_pose;
        const int32_t* const ol = pr->obs_point;
        auto copy_range = [&](size_t a, size_t b, bool indices_only) {
            memcpy(e_pose + a, pr->obs_pose + a, 4 * (b - a));
            memcpy(e_point + a, pr->obs_point + a, 4 * (b - a));
            if (indices_only) return;  // (the measurements follow in the team's second phase)
            copy_measurements(a, b);
        };
        auto barrier = [&](int k) {  // the k-th barrier of the call (k = 1, 2, ...): every thread of the team arrives once
            team.arrived.fetch_add(1);
            for (int spins = 0; team.arrived.load() < k * nth; ++spins)
                if (spins > 4000) std::this_thread::yield();
        };
        // Phase 1 of thread q: straight-line passes over its range (flag reductions the compiler vectorises), the landmark offsets as "end of
        // landmark l = index behind its last observation", stored where a landmark's run ends and closed over the empty landmarks by a running
        // maximum (shares of the offsets array, seeded by a backward look at the shares before)
        auto scan_range = [&](int q) {
            const size_t e0 = (size_t)E * q / nth, end = (size_t)E * (q + 1) / nth;
            const size_t k0 = ((size_t)L + 1) * q / nth, k1 = ((size_t)L + 1) * (q + 1) / nth;
            unsigned bd = 0, un = 0;
            for (size_t e = e0; e < end; ++e) bd |= (unsigned)((unsigned)op[e] >= (unsigned)P) | (unsigned)((unsigned)ol[e] >= (unsigned)L);
            for (size_t e = std::max<size_t>(e0, 1); e < end; ++e) un |= (unsigned)(ol[e] < ol[e - 1]);
            bad[q] = bd != 0, unsorted[q] = un != 0;
            if (!bd) {
                std::vector<uint8_t>& seen = q == 0 ? pose_seen : seen_q[q];
                if (q) seen.assign(P, 0);
                uint8_t* const sn = seen.data();
                for (size_t e = e0; e < end; ++e) sn[op[e]] = 1;
                copy_range(e0, end, nth > 1);
            }
            for (size_t k = k0; k < k1; ++k) lm_off[k] = 0;
            if (nth > 1) barrier(1);
            int any_bad = 0, any_un = 0;
            for (int t = 0; t < nth; ++t) any_bad |= bad[t], any_un |= unsorted[t];
            const bool ok = !any_bad && !any_un;
            if (ok && nth == 1)  // (one thread: stored unconditionally, the last observation of a landmark wins -- no branch to mispredict)
                for (size_t e = e0; e < end; ++e) lm_off[ol[e] + 1] = (int)(e + 1);
            else if (ok)
                for (size_t e = e0; e < end; ++e) {
                    const int l = ol[e], nxt = e + 1 < (size_t)E ? ol[e + 1] : -1;
                    if (nxt != l) lm_off[l + 1] = (int)(e + 1);
                }
            if (nth > 1) barrier(2);
            int v = 0;
            if (ok)
                for (size_t k = k0; k-- > 0;)
                    if (lm_off[k]) {
                        v = lm_off[k];
                        break;
                    }
            if (nth > 1) barrier(3);
            if (ok)
                for (size_t k = k0; k < k1; ++k) {
                    v = std::max(v, lm_off[k]);
                    lm_off[k] = v;
                }
            if (nth > 1) barrier(4);
        };
        // Phase 2 of worker q (1 .. nth - 1): its share of the measurements and of the landmark positions into the staging image, chunk by
        // chunk; a staged chunk is uploaded as soon as `go` says where to
        auto stage_measurements = [&](int q) {
            const int W = nth - 1, w = q - 1;
            const size_t CH = (size_t)128 << 10;   // observations per chunk: 2.6 MB of measurements
            const size_t LCH = (size_t)256 << 10;  // landmark positions: 6 MB per chunk
            const size_t e0 = (size_t)E * w / W, e1 = (size_t)E * (w + 1) / W, l0 = (size_t)L * w / W, l1 = (size_t)L * (w + 1) / W;
            memcpy(points_out + 3 * l0, pr->points + 3 * l0, 24 * (l1 - l0));  // (the caller's output starts as its input)
            hipError_t he = hipSetDevice(ctx->device);  // (the current device is a per-thread setting)
            size_t e_up = e0, l_up = l0;  // uploaded so far
            // (every hipMemcpyAsync holds the runtime's lock for ~15 us, and this thread's launches queue behind it: below 4 M observations the
            //  worker that finishes LAST uploads the whole image in four copies; beyond, every worker uploads its chunks as they are staged)
            const bool chunked = E >= (4 << 20);
            auto flush = [&](size_t e_staged, size_t l_staged) {
                if (!chunked || team.go.load() != 1 || he != hipSuccess) return;
                auto up = [&](size_t off, size_t bytes) {
                    if (bytes && he == hipSuccess) he = hipMemcpyAsync(team.d_in + off, hs + off, bytes, hipMemcpyHostToDevice, team.s2);
                };
                if (e_staged > e_up) {
                    up(in.e_uvr + 12 * e_up, 12 * (e_staged - e_up));
                    up(in.e_w + 4 * e_up, 4 * (e_staged - e_up));
                    up(in.e_hub + 4 * e_up, 4 * (e_staged - e_up));
                    e_up = e_staged;
                }
                if (l_staged > l_up) {
                    up(in.points + 24 * l_up, 24 * (l_staged - l_up));
                    l_up = l_staged;
                }
            };
            for (size_t a = e0; a < e1 && team.go.load() != 2; a += CH) {
                const size_t b = std::min(e1, a + CH);
                copy_measurements(a, b);
                flush(b, l0);
            }
            for (size_t a = l0; a < l1 && team.go.load() != 2; a += LCH) {
                const size_t b = std::min(l1, a + LCH);
                memcpy(hs + in.points + 24 * a, pr->points + 3 * a, 24 * (b - a));
                flush(e1, b);
            }
            const bool last = team.staged.fetch_add(1) + 1 == W;
            for (int spins = 0; team.go.load() == 0; ++spins)
                if (spins > 2000) std::this_thread::yield();
            if (team.go.load() == 1) {
                if (chunked) flush(e1, l1);
                else if (last) {  // (every share is staged: fetch_add orders the others' writes before this thread's reads)
                    const size_t offs[4] = {in.e_uvr, in.e_w, in.e_hub, in.points}, bytes[4] = {12 * (size_t)E, 4 * (size_t)E, 4 * (size_t)E, 24 * (size_t)L};
                    for (int k = 0; k < 4 && he == hipSuccess; ++k) he = hipMemcpyAsync(team.d_in + offs[k], hs + offs[k], bytes[k], hipMemcpyHostToDevice, team.s2);
                }
            }
            team.err[q] = (int)he;
        };
        struct TeamGuard {
            HostTeam& t;
            ~TeamGuard() { t.abandon_and_join(); }
        } team_guard{team};
        if (nth == 1) {
            if (team_sized) memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);  // (SVGPU_BA_HOST_THREADS=1)
            scan_range(0);
            if (!bad[0]) memcpy(hs + in.points, pr->points, sizeof(double) * 3 * (size_t)L);
        }
        else {
            team.running = true;
            // (the workers wait at a gate until the whole team exists: a thread the system refuses to create must not leave the others at a
            //  barrier that counts nth arrivals -- the call fails instead, the caller's estimate untouched)
            bool spawned = true;
            try {
                for (int q = 1; q < nth; ++q)
                    team.th.emplace_back([&, q] {
                        for (int spins = 0; team.gate.load() == 0; ++spins)
                            if (spins > 2000) std::this_thread::yield();
                        if (team.gate.load() != 1) return;
                        scan_range(q);
                        stage_measurements(q);
                    });
            } catch (...) {
                spawned = false;
            }
            team.gate.store(spawned ? 1 : 2);
            if (!spawned) {
                team.join();
                memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
                return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_global_ba: could not start the host threads of the set-up (SVGPU_BA_ONE_THREAD=1 runs it on the caller's thread)");
            }
            scan_range(0);  // (returns behind the last barrier: the whole scan is done)
        }
        int any_bad = 0, any_unsorted = 0;
        for (int q = 0; q < nth; ++q) any_bad |= bad[q], any_unsorted |= unsorted[q];
        if (any_bad) {
            if (team_sized) {
                team.go.store(2);
                team.join();
                memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
            }
            return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: observation index out of range");
        }
        for (int q = 1; q < nth; ++q)
            for (int p = 0; p < P; ++p) pose_seen[p] |= seen_q[q][p];
        lm_major = !any_unsorted;
        if (!lm_major && team.running) {  // (rare at this size: the permuted copy below writes the whole staging image itself)
            team.go.store(2);
            team.join();
            memcpy(hs + in.points, pr->points, sizeof(double) * 3 * (size_t)L);
        }
    if (!lm_major) {
        for (int l = 0; l <= L; ++l) lm_off[l] = 0;
        for (int e = 0; e < E; ++e) lm_off[pr->obs_point[e] + 1]++;
        for (int l = 0; l < L; ++l) lm_off[l + 1] += lm_off[l];
    }
    if (!lm_major) {
        perm.resize(E);
        std::vector<int> fill(lm_off, lm_off + L);
        for (int e = 0; e < E; ++e) perm[fill[pr->obs_point[e]]++] = e;
        for (int k = 0; k < E; ++k) {
            const int e = perm[k];
            e_pose[k] = pr->obs_pose[e];
            e_point[k] = pr->obs_point[e];
            e_uvr[3 * k] = pr->obs_uvr[3 * e];
            e_uvr[3 * k + 1] = pr->obs_uvr[3 * e + 1];
            e_uvr[3 * k + 2] = pr->obs_uvr[3 * e + 2];
            e_w[k] = pr->obs_inv_sigma_sq[e];
            e_hub[k] = pr->obs_huber_delta ? pr->obs_huber_delta[e] : 0.f;
        }
    }
    // (the pose -> edge lists and the robust-kernel flags are built on the device once the observations are there: sv_ba_build_pose_lists)
    memcpy(hs + in.pose, pr->pose_cw, sizeof(double) * 12 * (size_t)P);
    memcpy(hs + in.intr, pr->intrinsics, sizeof(double) * 5 * (size_t)P);

    lap("sort observations");
    // ---- solver choice (svgpu_ba_set_solver; SVGPU_BA_SOLVER overrides for experiments)
    int solver_opt = ctx->ba_solver;
    if (const char* ev = std::getenv("SVGPU_BA_SOLVER")) {
        if (!strcmp(ev, "cholesky")) solver_opt = SV_BA_SOLVER_CHOLESKY;
        else if (!strcmp(ev, "cholesky_mfma")) solver_opt = SV_BA_SOLVER_CHOLESKY_MFMA;
        else if (!strcmp(ev, "pcg")) solver_opt = SV_BA_SOLVER_PCG;
        else if (!strcmp(ev, "dense")) solver_opt = SV_BA_SOLVER_DENSE;
        else if (!strcmp(ev, "pcg_multi")) solver_opt = SV_BA_SOLVER_PCG_MULTI;
        else if (!strcmp(ev, "envelope")) solver_opt = SV_BA_SOLVER_ENVELOPE;
    }
    // ---- device arena
    const int nPmax = P, nmax = 6 * nPmax;
    const int nb_lm = (8 * L + 255) / 256 /* k_ba_update / k_ba_chi2: 8 lanes per landmark */, nb_chi = nb_lm, nb_pose = (P + 255) / 256;
    const size_t nb_cap = (size_t)P * (P + 1) / 2;
    // worst-case pair storage: sum over landmarks of k(k+1)/2 (+ duplicates never exceed k^2)
    size_t pair_cap = 0;
    for (int l = 0; l < L; ++l) {
        const size_t k = lm_off[l + 1] - lm_off[l];
        pair_cap += k * k;
    }
    const size_t sc_part_blocks = pair_cap / 64 + nb_cap + 16;  // NB * nshare <= pairs / 64 + NB (a share holds at least 64 pairs: one trip of a wave)
    const size_t xch_doubles = std::max(std::max((size_t)P, 4 * (size_t)L), nb_cap) + 8;
    const size_t nparts_max = (size_t)(P + 3) / 4 + 1;
    // the dense matrix of the tiled LL^T (ba_dense_tiled.hip): on request, and as a candidate of AUTO for windows of 24 - 256 keyframes on one rank
    const bool want_dense = solver_opt == SV_BA_SOLVER_DENSE || (solver_opt == SV_BA_SOLVER_AUTO && !sharded && P >= 24 && P <= 256 && !std::getenv("SVGPU_BA_NO_DENSE_TILED"));
    size_t need = 4 * pad(sizeof(double) * 12 * P) + 4 * pad(sizeof(double) * 3 * L) + 2 * pad(4 * (size_t)E) + pad(12 * (size_t)E)
                  + 2 * pad(4 * (size_t)E) + 2 * pad(E) + pad(8 * (size_t)E) + pad(40 * (size_t)P) + pad(4 * (size_t)P) + pad(L)
                  + pad(4 * (size_t)(L + 1)) + pad(4 * (size_t)(P + 1)) + pad(4 * (size_t)E) + pad(sizeof(double) * 18 * E)
                  + pad(sizeof(double) * 27 * 16 * (size_t)P) + pad(sizeof(double) * 36 * sc_part_blocks) + pad(sizeof(double) * 6 * 16 * (size_t)P)
                  + pad(sizeof(double) * 6 * L) + pad(sizeof(double) * 3 * L) + pad(sizeof(double) * 36 * P)
                  + pad(sizeof(double) * 6 * P) + pad(sizeof(double) * (36 * nb_cap + 6 * (size_t)P + 8)) + pad(sizeof(double) * nmax)
                  + (want_dense ? pad(sizeof(double) * ((size_t)(nmax + 1) * nmax + (size_t)(nmax / 48 + 1) * 48 * 48)) : 0)
                  + pad(sizeof(double) * (nb_chi + nb_lm + nb_pose + 8)) + pad(E + 1) + pad(8 * 42 * (size_t)P)
                  + pad(8 * (size_t)(64 + world + 1)) + pad(8 * xch_doubles) + pad(sizeof(BaCtl))
                  + pad(4 * (size_t)(P + 1)) + pad(8 * 2 * (nb_cap + 1)) + pad(4 * (size_t)P) + pad(8 * 36 * (size_t)P) + pad(8 * 6 * (size_t)nmax + 64)
                  + pad(8 * 2 * (size_t)nmax + 64) + pad(8 * 2 * 4 * nparts_max) + pad(64) + 8192;
    // the solve's own landmark numbering (ba_pairs.hip): one stage, the host team's sizes, observations grouped by landmark
    // (a sharded solve renumbers the landmarks of ITS shard: the numbering is private to the rank's kernels, every landmark-sized exchange
    //  between ranks -- ownership marks, the final positions -- stays in the caller's numbering)
    const bool renumber = team.running && single_stage && !std::getenv("SVGPU_BA_NO_RENUMBER");
    const bool chunk_units = renumber && pair_cap < ((size_t)1 << 31) && !std::getenv("SVGPU_BA_NO_UNITS");  // chunk-major units of the Schur kernel (ba_pairs.hip)
    const size_t unit_cap = chunk_units ? sc_part_blocks : 0;
    const size_t pair_scratch = std::max(std::max(std::max(sv_ba_pairs_scratch_bytes(pair_cap, E, nb_cap), sv_ba_pose_lists_scratch_bytes((size_t)E)), renumber ? sv_ba_renumber_scratch_bytes((size_t)L) : 0),
                                         chunk_units ? sv_ba_units_scratch_bytes(pair_cap, unit_cap) : 0);
    if (chunk_units) need += pad(16 * unit_cap) + pad(4 * (nb_cap + 2)) + pad(48 * unit_cap) + 1024;
    if (renumber) need += 3 * pad(4 * (size_t)E) + pad(12 * (size_t)E) + 2 * pad(4 * (size_t)E) + pad(4 * (size_t)(L + 1)) + pad(4 * (size_t)L) + pad(24 * (size_t)L) + pad(L) + 4096;
    need += pad(4 * (nb_cap + (size_t)P)) + pad(P) + pad(L);  // keyframe-segment exchange: block / slot lists, damping owners
    need += pad(8 * pair_cap) + pad(8 * nb_cap) + pad(4 * (nb_cap + 1)) + pad(pair_scratch) + in.total + out_total + 3 * pad(4 * (size_t)E) + pad(12 * (size_t)E) + pad(8 * (size_t)nb_lm) + pad(4 * pair_cap) + pad(64 * (size_t)nb_lm);
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    BaDev D;
    memset(&D, 0, sizeof(D));
    D.P = P;
    D.L = L;
    D.E = E;
    D.world = world;
    D.rank = rank;
    // input block (same layout as the staging buffer)
    D.pose_buf[0] = A.take<double>(12 * (size_t)P);
    D.pt_buf[0] = A.take<double>(3 * (size_t)L);
    double* d_intr = A.take<double>(5 * (size_t)P);
    int* d_e_pose = A.take<int>(E);
    int* d_e_point = A.take<int>(E);
    float* d_e_uvr = A.take<float>(3 * (size_t)E);
    float* d_e_w = A.take<float>(E);
    float* d_e_hub = A.take<float>(E);
    D.e_robust = A.take<uint8_t>(E);
    int* d_lm_off = A.take<int>(L + 1);
    int* d_pe_off = A.take<int>(P + 1);
    int* d_pe_idx = A.take<int>(E);
    if (A.off != in.total) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: internal layout mismatch");
    // output block: control block | poses | points | outlier flags
    char* const d_out = A.base + A.off;
    D.ctl = (BaCtl*)A.take<char>(sizeof(BaCtl));
    double* d_state_out = A.take<double>(12 * (size_t)P + 3 * (size_t)L);
    uint8_t* d_outlier = A.take<uint8_t>(E + 1);
    D.pose_buf[1] = A.take<double>(12 * (size_t)P);
    D.pt_buf[1] = A.take<double>(3 * (size_t)L);
    D.e_level = A.take<uint8_t>(E);
    D.e_chi = A.take<double>(E);
    int* d_pose_slot = A.take<int>(P);
    uint8_t* d_pt_free = A.take<uint8_t>(L);
    int* d_slot_pose = A.take<int>(P);
    D.W = A.take<double>(18 * (size_t)E);
    D.lp_part = A.take<double>(27 * 16 * (size_t)P);
    D.sc_part = A.take<double>(36 * sc_part_blocks);
    D.rhs_part = A.take<double>(6 * 16 * (size_t)P);
    D.Hll = A.take<double>(6 * (size_t)L);
    D.bl = A.take<double>(3 * (size_t)L);
    D.Hpp = A.take<double>(36 * (size_t)P);
    D.bp = A.take<double>(6 * (size_t)P);
    D.Sblk = A.take<double>(36 * nb_cap + 6 * (size_t)P + 8);
    D.S = want_dense ? A.take<double>((size_t)(nmax + 1) * nmax + (size_t)(nmax / 48 + 1) * 48 * 48) : nullptr;  // (+ the inverted diagonal tiles of ba_dense_tiled.hip)
    D.dp = A.take<double>(nmax);
    D.red = A.take<double>(nb_chi + nb_lm + nb_pose + 8);
    D.lm_max = A.take<double>(nb_lm);  // nb_lm == sv_ba_lm_blocks(L)
    static const bool dbg_stamps = std::getenv("SVGPU_BA_DBG") != nullptr;
    static const bool dbg_schur = dbg_stamps && !strcmp(std::getenv("SVGPU_BA_DBG"), "schur");
    static const bool dbg_chol = dbg_stamps && !strcmp(std::getenv("SVGPU_BA_DBG"), "chol");
    D.dbg_schur_on = dbg_schur ? 1 : (dbg_chol ? 2 : 0);
    D.dbg = dbg_stamps ? A.take<unsigned long long>(8 * (dbg_schur ? sc_part_blocks + 64 : (size_t)nb_lm)) : nullptr;
    double* d_HB_full = A.take<double>(42 * (size_t)P + (size_t)std::max(64, world));     // sharded: Hpp | bp summed over ranks | one slot per rank: its landmarks' largest diagonal (computeLambdaInit)
    double* d_sc = A.take<double>(64 + (size_t)world);            // sharded: [0..3] per-trial sums, [8..8+world) lambda-init slots
    double* d_xch = A.take<double>(xch_doubles);                  // sharded: pose-activity / block-presence / point exchange
    int* d_prow_off = A.take<int>(P + 1);
    int2* d_prow_ent = A.take<int2>(2 * (nb_cap + 1));
    int* d_diag_blk = A.take<int>(P);
    D.pcg_Minv = A.take<double>(36 * (size_t)P);
    D.pcg_rws = A.take<double>(6 * (size_t)nmax + 8);
    D.pcg_own = A.take<double>(2 * (size_t)nmax + 8);
    D.pcg_parts = A.take<double>(2 * 4 * nparts_max);
    D.pcg_scal = A.take<double>(8);
    int* d_pm_point = A.take<int>(E);
    float* d_pm_uvr = A.take<float>(3 * (size_t)E);
    float* d_pm_w = A.take<float>(E);
    float* d_pm_hub = A.take<float>(E);
    int2* d_blk_pairs = A.take<int2>(pair_cap);
    int* d_blk_pair_l = A.take<int>(pair_cap);
    int2* d_blk_ab = A.take<int2>(nb_cap);
    int* d_blk_off = A.take<int>(nb_cap + 1);
    char* d_pair_scratch = A.take<char>(pair_scratch);
    int* d_xs_idx = A.take<int>(nb_cap + (size_t)P);
    uint8_t* d_lam_slot = A.take<uint8_t>(P);
    uint8_t* d_any_owner = A.take<uint8_t>(L);  // sharded: the landmark has observations on some rank
    D.any_owner = d_any_owner;
    // renumbered solve: the uploaded arrays (caller's order) are the sources, these the arrays the kernels read
    int *rn_e_pose = nullptr, *rn_e_point = nullptr, *rn_src = nullptr, *rn_lm_off = nullptr, *rn_order = nullptr;
    float *rn_uvr = nullptr, *rn_w = nullptr, *rn_hub = nullptr;
    double* rn_pts = nullptr;
    uint8_t* rn_pt_free_in = nullptr;
    int4* d_unit_rec = nullptr;
    int* d_blk_unit_off = nullptr;
    double* d_rhs_unit = nullptr;
    if (chunk_units) {
        d_unit_rec = A.take<int4>(unit_cap);
        d_blk_unit_off = A.take<int>(nb_cap + 2);
        d_rhs_unit = A.take<double>(6 * unit_cap);
    }
    if (renumber) {
        rn_e_pose = A.take<int>(E), rn_e_point = A.take<int>(E), rn_src = A.take<int>(E);
        rn_uvr = A.take<float>(3 * (size_t)E), rn_w = A.take<float>(E), rn_hub = A.take<float>(E);
        rn_lm_off = A.take<int>(L + 1), rn_order = A.take<int>(L);
        rn_pts = A.take<double>(3 * (size_t)L);
        rn_pt_free_in = A.take<uint8_t>(L);
    }
    if (A.off > ctx->scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: internal arena overflow");
    D.e_pose = d_e_pose;
    D.e_point = d_e_point;
    D.e_uvr = d_e_uvr;
    D.e_w = d_e_w;
    D.e_huber = d_e_hub;
    D.intr = d_intr;
    D.any_equirect = 0;
    for (int p = 0; p < P; ++p)
        if (pr->intrinsics[5 * (size_t)p] == 0.0 && pr->intrinsics[5 * (size_t)p + 1] == 0.0) D.any_equirect = 1;
    D.pose_slot = d_pose_slot;
    D.pt_free = d_pt_free;
    D.lm_off = d_lm_off;
    D.pe_off = d_pe_off;
    D.pe_idx = d_pe_idx;
    D.slot_pose = d_slot_pose;
    D.blk_pairs = d_blk_pairs;
    D.blk_pair_l = d_blk_pair_l;
    D.blk_ab = d_blk_ab;
    D.blk_off = d_blk_off;
    D.prow_off = d_prow_off;
    D.prow_ent = d_prow_ent;
    D.diag_blk = d_diag_blk;
    D.maxslots = d_sc + 8;
    D.xsum = sharded ? d_sc : nullptr;
    D.red_chi_off = 0;
    D.red_chi_n = nb_chi;
    D.red_scale_off = nb_chi;
    D.red_scale_n = nb_lm + nb_pose;
    D.red_flag_off = nb_chi + nb_lm + nb_pose;
    // page-locked block: the control block's read-back copy + the word the caller's stop flag is mirrored into for the device
    const size_t pinned_need = (sizeof(BaCtl) + 7) / 8 + 16;
    if (ctx->pinned_doubles < pinned_need) {
        if (ctx->h_pinned) SV_HIP(ctx, hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->pinned_doubles = 0;
        SV_HIP(ctx, hipHostMalloc((void**)&ctx->h_pinned, sizeof(double) * pinned_need, hipHostMallocMapped));
        ctx->pinned_doubles = pinned_need;
    }
    if (!ctx->ev_ba) SV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ba, hipEventDisableTiming));
    BaCtl* const h_ctl = (BaCtl*)ctx->h_pinned;
    volatile int* const h_mirror = (volatile int*)(ctx->h_pinned + pinned_need - 8);
    *h_mirror = 0;
    {
        void* dptr = nullptr;
        SV_HIP(ctx, hipHostGetDevicePointer(&dptr, (void*)h_mirror, 0));
        D.stop_mirror = (const volatile int*)dptr;
    }

    uint8_t aux_flag = 0;  // g2o installs its own flag when the caller passes none (see svgpu.h)
    volatile uint8_t* flag = stop ? stop : &aux_flag;
    *h_mirror = *flag ? 1 : 0;
    // Wait for the stream.  While the device works, the caller's force_stop_flag (set asynchronously by the tracking thread,
    // mapping_module.h:232) is mirrored into the page-locked word k_ba_decide / k_ba_begin poll at every trial boundary.  The
    // host itself never branches on the caller's flag between two collectives of a sharded solve: every rank acts on the
    // all-reduced votes inside the control block only.
    auto wait_stream = [&]() -> int {
        SV_HIP(ctx, hipEventRecord(ctx->ev_ba, s));
        // (the mapping thread must not pin a host core the tracking thread needs: a short spin for the sub-100 us waits of a local
        //  window, then the poll yields / sleeps ~20 us between queries -- the stop flag is sampled every time)
        int polls = 0;
        for (;;) {
            if (*flag) *h_mirror = 1;
            const hipError_t q = hipEventQuery(ctx->ev_ba);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return sv_set_error(ctx, SVGPU_ERR_HIP, "hipEventQuery", q);
            if (++polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
            else if (polls > 200) std::this_thread::yield();
        }
        return SVGPU_OK;
    };
    auto read_ctl = [&]() -> int {
        SV_HIP(ctx, hipMemcpyAsync(h_ctl, D.ctl, sizeof(BaCtl), hipMemcpyDeviceToHost, s));
        return wait_stream();
    };

#define H2D(dst, src, bytes) SV_HIP(ctx, hipMemcpyAsync((void*)(dst), (src), (bytes), hipMemcpyHostToDevice, s))
    const void* sorted_edges = nullptr;  // the (pose, edge) records of sv_ba_prepare_lists, kept in the W buffer until the measurements are there
    char* const d_in = (char*)D.pose_buf[0];
    {
        const size_t want = sv_ba_pose_lists_scratch_bytes((size_t)E);
        const bool in_W = sizeof(double) * 18 * (size_t)E >= want;  // (the pair lists are built in the pair scratch: the records survive them only in W)
        void* sc = in_W ? (void*)D.W : (void*)d_pair_scratch;
        const size_t sc_bytes = in_W ? sizeof(double) * 18 * (size_t)E : pair_scratch;
        if (team.running && !in_W) {  // (cannot happen at these sizes; the team's work is then uploaded in one piece)
            team.go.store(2);
            team.join();
            copy_measurements(0, (size_t)E);
            memcpy(hs + in.points, pr->points, sizeof(double) * 3 * (size_t)L);
        }
        if (team.running) {
            if (!ctx->ba_copy_stream) SV_HIP(ctx, hipStreamCreateWithFlags(&ctx->ba_copy_stream, hipStreamNonBlocking));
            if (!ctx->ev_ba_copy) SV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ba_copy, hipEventDisableTiming));
            H2D(d_in + in.pose, hs + in.pose, in.points - in.pose);          // poses
            H2D(d_in + in.intr, hs + in.intr, in.e_uvr - in.intr);            // intrinsics | e_pose | e_point
            H2D(d_in + in.lm_off, hs + in.lm_off, in.pe_off - in.lm_off);     // landmark offsets
            // (the link is shared: the measurements queue behind the index arrays, which the structure kernels are waiting for)
            // SVGPU_BA_COPY_ON_MAIN: the measurements on the solve's own stream instead.  Measured over 150 config-5 calls: copy stream median
            // 7.50 ms, but 2 calls of 15 ms (the stream's first synchronisation arrives ~8 ms late: two DMA users of one process); own stream
            // 8.06 ms, no call above 8.9.  The default keeps the better mean (7.6 against 8.1); a caller that minds the tail sets the variable.
            const bool copy_on_main = std::getenv("SVGPU_BA_COPY_ON_MAIN") != nullptr;
            if (!copy_on_main) {
                SV_HIP(ctx, hipEventRecord(ctx->ev_ba_copy, s));
                SV_HIP(ctx, hipStreamWaitEvent(ctx->ba_copy_stream, ctx->ev_ba_copy, 0));
            }
            team.d_in = d_in;
            team.s2 = copy_on_main ? s : ctx->ba_copy_stream;
            team.go.store(1);
        }
        else H2D(d_in, hs, in.total);
        if (renumber) {
            const int rr = sv_ba_renumber_landmarks(ctx, s, d_lm_off, d_e_pose, L, P, E, d_pair_scratch, pair_scratch, rn_order, rn_lm_off, rn_e_pose, rn_e_point, rn_src);
            if (rr) return rr;
            D.e_pose = rn_e_pose, D.e_point = rn_e_point, D.lm_off = rn_lm_off;
            D.e_uvr = rn_uvr, D.e_w = rn_w, D.e_huber = rn_hub;
            D.pt_buf[0] = rn_pts;
            D.lm_order = rn_order;
            D.lm_off_caller = d_lm_off;
        }
        // pose -> edge lists on the device (they need the pose indices only)
        const int rp = sv_ba_prepare_lists(ctx, s, D.e_pose, E, P, sc, sc_bytes, d_pe_off, D.e_level, D.e_chi, &sorted_edges);  // (also clears e_level / e_chi)
        if (rp) return rp;
        D.pm_point = d_pm_point, D.pm_uvr = d_pm_uvr, D.pm_w = d_pm_w, D.pm_hub = d_pm_hub;
    }
    // the measurements are on the device (pipelined: the stream waits for the copy stream) -> pose-major copies + robust flags
    auto finish_observations = [&]() -> int {
        if (team.running) {
            team.join();
            for (int q = 0; q < nth; ++q)
                if (team.err[q]) return sv_set_error(ctx, SVGPU_ERR_HIP, "hipMemcpyAsync (measurements)", (hipError_t)team.err[q]);
            SV_HIP(ctx, hipEventRecord(ctx->ev_ba_copy, ctx->ba_copy_stream));
            SV_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_ba_copy, 0));
        }
        if (renumber) {  // measurements and positions into the solve's numbering
            sv_ba_permute_measurements(s, rn_src, E, d_e_uvr, d_e_w, d_e_hub, rn_uvr, rn_w, rn_hub);
            sv_ba_permute_points(s, rn_order, L, (const double*)(d_in + in.points), rn_pts);
        }
        const int rp = sv_ba_prepare_pose_major(ctx, s, sorted_edges, D.e_point, D.e_uvr, D.e_w, D.e_huber, E, d_pe_idx, D.e_robust, d_pm_point, d_pm_uvr, d_pm_w, d_pm_hub);
        sorted_edges = nullptr;
        return rp;
    };
    bool observations_finished = false;
    if (!team.running) {
        const int rp = finish_observations();
        if (rp) return rp;
        observations_finished = true;
    }
    BaCtl ctl0;
    memset(&ctl0, 0, sizeof(ctl0));
    ctl0.gain_thr = pr->gain_threshold;
    ctl0.pcg_tol2 = ctx->pcg_tol * ctx->pcg_tol;
    ctl0.phase = 2;
    H2D(D.ctl, &ctl0, sizeof(BaCtl));
    SV_HIP(ctx, hipMemsetAsync(d_sc, 0, 8 * (64 + (size_t)world), s));

    // ---- exchange helpers (sharded solve; no-ops otherwise)
    enum { XC_SETUP = 1, XC_POSE_BLOCKS = 2, XC_SYSTEM = 3, XC_SUMS = 6 };  // (4, 5: the separator / solution exchanges of ba_skyline.hip)
    for (long long& v : ctx->ba_xch) v = 0;
    ctx->ba_xch[0] = sharded ? 1 : 0;
    auto allreduce_dev = [&](double* dev, size_t n, int cat) -> int {
        if (!sharded || n == 0) return SVGPU_OK;
        ctx->ba_xch[cat] += 8 * (long long)n;
        ++ctx->ba_xch[7];
        return allreduce(ar_user, dev, n, (void*)s) == 0 ? SVGPU_OK : sv_set_error(ctx, SVGPU_ERR_HIP, "all-reduce callback failed");
    };
    std::vector<double> xch_host(sharded ? std::max((size_t)P, nb_cap) + 8 : 8);  // (the landmark-sized exchanges stay on the device)
    auto allreduce_host = [&](size_t n) -> int {  // xch_host[0..n) summed over the ranks (set-up steps only)
        H2D(d_xch, xch_host.data(), 8 * n);
        int r = allreduce_dev(d_xch, n, XC_SETUP);
        if (r) return r;
        SV_HIP(ctx, hipMemcpyAsync(xch_host.data(), d_xch, 8 * n, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        return SVGPU_OK;
    };
    // Whether "the caller passed a stop pointer" must mean the same on every rank: the early return and the skipped second stage below
    // branch on it, and a rank that left while the others entered the next collective would hang them.  A sharded solve therefore acts as
    // if every rank had a pointer as soon as ANY rank has one (the flag rides along with the contract check).
    bool stop_ptr_any = stop != nullptr;
    if (sharded) {
        // contract check on the device: a landmark's observations must not be split over ranks.  Every rank marks the landmarks it holds
        // observations of (its landmark offsets are already there), the marks are summed, a landmark with two owners fails the call on
        // every rank; what comes back is two words.  The sum also says which landmarks have an owner at all (the final exchange of the
        // points leaves the others alone).
        sv_ba_owned_mark(s, d_lm_off, L, d_xch, stop ? 1.0 : 0.0);
        int r = allreduce_dev(d_xch, (size_t)L + 1, XC_SETUP);
        if (r) return r;
        sv_ba_owned_check(s, d_xch, L, d_any_owner, d_sc + 32);
        double verdict[2] = {0.0, 0.0};
        SV_HIP(ctx, hipMemcpyAsync(verdict, d_sc + 32, 16, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        if (verdict[0] > 0.5) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: observations must be sharded by landmark");
        stop_ptr_any = verdict[1] > 0.5;
    }

    lap("arena + uploads");
    HostStructure HS;
    bool xs_on = false;  // keyframe-segment exchange (decided with the plan, the same on every rank)
    int xs_nb = 0, xs_ns = 0;
    bool have_lists = false;
    std::vector<uint8_t> pose_active;
    int solver = SV_BA_SOLVER_CHOLESKY;
    auto upload_structure = [&]() -> int {
        const std::vector<uint8_t>* pa_override = nullptr;
        if (sharded) {  // a pose is active if ANY rank holds an active observation of it
            for (int p = 0; p < P; ++p) xch_host[p] = 0;
            for (int e = 0; e < E; ++e)
                if (!level[e]) xch_host[e_pose[e]] = 1;
            int r = allreduce_host(P);
            if (r) return r;
            pose_active.assign(P, 0);
            for (int p = 0; p < P; ++p) pose_active[p] = xch_host[p] > 0.5;
            pa_override = &pose_active;
        }
        // Activity of poses / landmarks under the current levels (cheap), then the expensive part -- pose->edge lists
        // and (edge, edge) pair lists -- only when the free-pose numbering changed.  Lists built for an earlier stage
        // stay valid: edges excluded since then are skipped by level (lin_pose, rhs) or contribute W = Y = 0 (pairs).
        auto tb0 = std::chrono::steady_clock::now();
        auto sub = [&](const char* what) {
            if (trace) std::fprintf(stderr, "[ba]     %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
        };
        HostStructure probe;
        activity_only(*pr, e_pose, e_point, level, pa_override, probe, no_levels ? &pose_seen : nullptr, no_levels ? lm_off : nullptr);
        const bool reuse = have_lists && probe.pose_slot == HS.pose_slot;
        if (reuse) {
            HS.pt_free = probe.pt_free;
            HS.nL = probe.nL;
        }
        else {
            HS.pose_slot = probe.pose_slot;
            HS.slot_pose = probe.slot_pose;
            HS.pt_free = probe.pt_free;
            HS.nP = probe.nP;
            HS.nL = probe.nL;
        }
        D.nP = HS.nP;
        D.n = 6 * HS.nP;
        D.lin_split = sv_ba_lin_split(E, HS.nP);
        D.chol_in_lds = D.n <= 186 && sv_ba_chol_bytes(D.n) <= 160 * 1024 - 12 * 1024;
        solver = solver_opt;
        D.chol_mfma = 0;
        if (solver == SV_BA_SOLVER_CHOLESKY_MFMA) {  // the same dense on-chip solve, its MFMA form
            solver = SV_BA_SOLVER_CHOLESKY;
            D.chol_mfma = 1;
        }
        D.Hpp_full = sharded ? d_HB_full : D.Hpp;
        D.bp_full = sharded ? d_HB_full + 36 * (size_t)HS.nP : D.bp;
        D.scale_pose = (!sharded || rank == 0) ? 1 : 0;
        D.add_lambda = (!sharded || rank == 0) ? 1 : 0;
        memcpy(hs_struct + st_pt_free, HS.pt_free.data(), L);
        if (renumber) {
            H2D(rn_pt_free_in, hs_struct + st_pt_free, L);
            sv_ba_permute_flags(s, rn_order, L, rn_pt_free_in, d_pt_free);
        }
        else H2D(d_pt_free, hs_struct + st_pt_free, L);
        if (!reuse) {
            memcpy(hs_struct + st_pose_slot, HS.pose_slot.data(), 4 * (size_t)P);
            H2D(d_pose_slot, hs_struct + st_pose_slot, 4 * (size_t)P);
            // The pair lists are built on the device.  A local-BA sized problem (<= 48 free poses; the count is the same on every rank of
            // a sharded solve) keeps EVERY upper block -- blocks without a pair are zero blocks, the solvers of that size are dense
            // anyway, and no union of block patterns has to be agreed between ranks -- so nothing has to come back: the pair total stays
            // on the device (launches sized for the capacity) and the whole structure pipeline runs unattended.
            // Larger systems synchronise twice (pair total, block offsets) and keep only the blocks that hold a pair.
            const bool all_blocks = HS.nP <= 48;  // (the pair total stays on the device: the launches are sized for the capacity)
            std::vector<int> dense_off;
            if (all_blocks) {
                int rp = sv_ba_build_pairs_async(ctx, s, D, d_pair_scratch, pair_scratch, pair_cap, -1, d_blk_pairs, d_blk_pair_l, d_blk_off);
                if (rp) return rp;
                sub("pairs enqueued");
            }
            have_lists = true;
            int* const h_slot_pose = (int*)(hs_struct + st_pe);
            for (int sl = 0; sl < HS.nP; ++sl) h_slot_pose[sl] = HS.slot_pose[sl];
            if (HS.nP > 0) H2D(d_slot_pose, h_slot_pose, 4 * (size_t)HS.nP);
            if (all_blocks) {
                HS.blk_ab.clear();
                for (int a = 0; a < HS.nP; ++a)
                    for (int b = a; b < HS.nP; ++b) {
                        int2 ab;
                        ab.x = a;
                        ab.y = b;
                        HS.blk_ab.push_back(ab);
                    }
                HS.blk_off.clear();  // lives on the device only
                HS.num_pairs = pair_cap;  // (an upper bound: sum over the landmarks of k^2; sizes the Schur shares)
            }
            else {
                int rp = sv_ba_build_pairs(ctx, s, D, d_pair_scratch, pair_scratch, pair_cap, d_blk_pairs, d_blk_pair_l, dense_off);
                if (rp) return rp;
                sub("device pairs");
                std::vector<uint8_t> present(dense_off.size() ? dense_off.size() - 1 : 0, 0);
                for (size_t k = 0; k < present.size(); ++k) present[k] = dense_off[k + 1] > dense_off[k];
                if (sharded && !present.empty()) {  // the kept-block list must be the same on every rank: union of the local patterns
                    for (size_t k = 0; k < present.size(); ++k) xch_host[k] = present[k];
                    int r = allreduce_host(present.size());
                    if (r) return r;
                    for (size_t k = 0; k < present.size(); ++k) present[k] = xch_host[k] > 0.5;
                }
                compact_blocks(dense_off, present, HS);
                H2D(d_blk_off, HS.blk_off.data(), 4 * HS.blk_off.size());
            }
            // block rows for the PCG: row a lists its kept blocks (a, b) and, transposed, (b, a)
            const int NB = (int)HS.blk_ab.size();
            int2* const h_blk_ab = (int2*)(hs_struct + st_blk);
            int* const h_prow_off = (int*)(h_blk_ab + NB);
            int* const h_diag = h_prow_off + (HS.nP + 1);
            int2* const h_prow_ent = (int2*)(hs_struct + st_prow);
            for (int a = 0; a <= HS.nP; ++a) h_prow_off[a] = 0;
            for (int k = 0; k < NB; ++k) {
                const int2 ab = HS.blk_ab[k];
                h_blk_ab[k] = ab;
                h_prow_off[ab.x + 1]++;
                if (ab.x != ab.y) h_prow_off[ab.y + 1]++;
                else h_diag[ab.x] = k;
            }
            for (int a = 0; a < HS.nP; ++a) h_prow_off[a + 1] += h_prow_off[a];
            {
                // blk_ab is sorted by (a, b): filling row a with its transposed blocks (c, a), c < a, first (they arrive in c order)
                // and its own blocks (a, b), b >= a, afterwards leaves every row sorted by column
                std::vector<int> fill(h_prow_off, h_prow_off + HS.nP);
                for (int k = 0; k < NB; ++k) {
                    const int2 ab = HS.blk_ab[k];
                    if (ab.x != ab.y) {
                        int2 e2;
                        e2.x = k | (1 << 30);
                        e2.y = ab.x;
                        h_prow_ent[fill[ab.y]++] = e2;
                    }
                }
                for (int k = 0; k < NB; ++k) {
                    const int2 ab = HS.blk_ab[k];
                    int2 e1;
                    e1.x = k;
                    e1.y = ab.y;
                    h_prow_ent[fill[ab.x]++] = e1;
                }
            }
            if (NB > 0) H2D(d_blk_ab, h_blk_ab, 8 * (size_t)NB);
            H2D(d_prow_off, h_prow_off, 4 * (size_t)(HS.nP + 1));
            if (h_prow_off[HS.nP] > 0) H2D(d_prow_ent, h_prow_ent, 8 * (size_t)h_prow_off[HS.nP]);
            if (HS.nP > 0) H2D(d_diag_blk, h_diag, 4 * (size_t)HS.nP);
            sub("block rows");
        }
        else sv_ba_zero_inactive(s, D);
        D.NB = (int)HS.blk_ab.size();
        D.g = D.Sblk + 36 * (size_t)D.NB;
        D.pcg_nparts = (HS.nP + 3) / 4;
        // shares per block: ~192 pairs per wave (3 per lane); fewer, longer walks are slower (a lane's pairs are a chain of dependent
        // loads), measured 2.30 / 2.31 / 2.43 / 2.90 / 3.79 ms per config-3 call at 96 / 192 / 384 / 768 / 1536 pairs per wave
        static const size_t share_pairs = [] {
            const char* e = std::getenv("SVGPU_BA_SHARE_PAIRS");  // tuning aid: pairs per share of a block of the reduced system
            const long v = e ? std::atol(e) : 0;
            return (size_t)(v >= 64 ? v : 192);  // (>= 64: the partial-sum buffer is sized for it)
        }();
        D.nshare = D.NB > 0 ? (int)std::min<size_t>(16, std::max<size_t>(1, (HS.num_pairs / (size_t)D.NB + share_pairs - 1) / share_pairs)) : 1;
        if (D.NB > 0) {
            // A small system does not fill the chip with ~192-pair shares (config 3: 136 blocks x 7 shares for 1 792 unit slots): its shares shrink
            // down to one trip of a wave (64 pairs) while the units still fit one resident round -- 17.7 -> 13.4 us per launch there; a large
            // system keeps the long shares (config 5 at 64 pairs per share: 314 us instead of 189)
            const size_t avg = HS.num_pairs / (size_t)D.NB, ns_max = std::min<size_t>(16, std::max<size_t>(1, (avg + 63) / 64));
            while ((size_t)D.nshare < ns_max && (size_t)D.NB * (D.nshare + 1) <= 2304) ++D.nshare;
        }
        if (!reuse) {
            D.unit_rec = nullptr;
            D.num_units = 0;
            const size_t units_min = [] {  // (SVGPU_BA_UNITS_MIN, read per call: arithmetic shares below this many; the tests lower it)
                const char* ev = std::getenv("SVGPU_BA_UNITS_MIN");
                return ev ? (size_t)std::max(1, std::atoi(ev)) : (size_t)8192;
            }();
            if (chunk_units && HS.nP > 48 && (size_t)D.NB * D.nshare >= units_min && HS.num_pairs > 0) {
                // ~0.5 MB of W records per chunk of consecutive landmark ranks (SVGPU_BA_CHUNK_SHIFT overrides: chunk = 2^shift landmarks).  Measured,
                // Schur launch at config 5 / 9.6 M observations: 256 landmarks 125 / 930 us, 512: 122 / 880, 1 024: 132 / 882, 2 048: 158 / 973; arithmetic shares 144 / 1 537
                int shift = 4;
                while (((size_t)2 << shift) * 144 * ((size_t)E / (size_t)std::max(L, 1) + 1) <= ((size_t)1 << 19)) ++shift;
                if (const char* ev = std::getenv("SVGPU_BA_CHUNK_SHIFT")) shift = std::max(0, std::min(30, std::atoi(ev)));
                int U = 0;
                const int ru = sv_ba_build_units(ctx, s, d_blk_off, D.NB, d_blk_pair_l, (int)HS.num_pairs, L, shift, (int)unit_cap, d_pair_scratch, pair_scratch, d_unit_rec, d_blk_unit_off, &U);
                if (ru) return ru;
                if (U > 0) {
                    D.unit_rec = d_unit_rec, D.blk_unit_off = d_blk_unit_off, D.rhs_unit = d_rhs_unit, D.num_units = U;
                }
                if (trace) std::fprintf(stderr, "[ba]     chunk-major units: %d (chunks of %d landmarks)\n", U, 1 << shift);
                sub("units");
            }
        }
        // solver of this stage: PCG inside one workgroup's LDS when the blocks fit, else one launch per PCG iteration
        const bool lds_ok = sv_ba_pcg_lds_bytes(D) > 0;
        // AUTO: dense LL^T in LDS while it fits (n <= ~135: 70 us per trial against 88 us for the LDS-resident PCG at n = 96), the
        // LDS-resident PCG up to 512 unknowns / ~150 KB of blocks, the one-launch-per-iteration PCG beyond
        if (solver == SV_BA_SOLVER_AUTO && D.chol_in_lds) solver = SV_BA_SOLVER_CHOLESKY;
        if (solver == SV_BA_SOLVER_CHOLESKY && !D.chol_in_lds) solver = SV_BA_SOLVER_AUTO;
        // beyond one workgroup's LDS, a window whose block pattern is close to full (every keyframe of a local window shares landmarks with
        // most others): the tiled dense LL^T -- 2 launches per 48 columns, MFMA updates.  Measured per damping trial, auto before / tiled:
        // see DESIGN section 6 (windows of 40 - 100 keyframes).
        // (from where the register-tile solve no longer fits LDS -- n = 138: per trial 232 us at n = 156 against 386 for the LDS-resident PCG AUTO
        //  took there and 282 for the envelope factorisation; at n = 132, where the register-tile solve still fits, the two tie at ~200 us)
        if (solver == SV_BA_SOLVER_AUTO && D.S && !D.chol_in_lds && D.n <= 1536 && (size_t)D.NB * 5 >= (size_t)HS.nP * (HS.nP + 1)) solver = SV_BA_SOLVER_DENSE;  // >= 40 % of the upper blocks kept
        // beyond the on-chip solvers: the direct envelope factorisation while the envelope of the ordered block graph is small (keyframe
        // graphs are banded up to a few loop-closure rows), else -- or on request -- the PCG with one launch per iteration
        if ((solver == SV_BA_SOLVER_AUTO && !lds_ok) || solver == SV_BA_SOLVER_ENVELOPE) {
            if (!reuse) {
                bool ok = false;
                const int rs = sv_sky_plan(ctx, s, HS.nP, HS.blk_ab, (size_t)256 << 20, &ok, rank, world);
                if (rs) return rs;
                HS.envelope_ok = ok;
            }
            solver = HS.envelope_ok ? SV_BA_SOLVER_ENVELOPE : SV_BA_SOLVER_AUTO;
        }
        if (solver == SV_BA_SOLVER_AUTO || solver == SV_BA_SOLVER_PCG) solver = lds_ok ? SV_BA_SOLVER_PCG_LDS : SV_BA_SOLVER_PCG_MULTI;
        // Keyframe-segment exchange.  When the envelope solve is segmented and every observation this rank holds is of a keyframe in one of
        // ITS jobs' pieces (or of a separator keyframe) -- svgpu_ba_partition_keyframe_segments cuts shards that way -- the blocks and
        // right-hand side rows of a piece are complete on the rank that eliminates it and no other rank reads them: only the kept blocks
        // between two separator rows and the separator rows of g have to be summed.  One all-reduced flag makes the decision collective.
        if (!reuse) {
            xs_on = false;
            D.lam_slot = nullptr;
            const std::vector<int>*job_of = nullptr, *owner = nullptr, *sep_blocks = nullptr;
            const char* xe = std::getenv("SVGPU_BA_EXCHANGE");
            const bool candidate = sharded && world > 1 && solver == SV_BA_SOLVER_ENVELOPE && !(xe && !strcmp(xe, "full")) && sv_sky_current_roles(ctx, &job_of, &owner, &sep_blocks);
            if (sharded && world > 1) {
                double misplaced = candidate ? 0.0 : 1.0;
                if (candidate)
                    for (int e = 0; e < E && misplaced == 0.0; ++e) {
                        if (level[e]) continue;
                        const int sl = HS.pose_slot[e_pose[e]];
                        if (sl >= 0 && (*job_of)[sl] >= 0 && (*owner)[(*job_of)[sl]] != rank) misplaced = 1.0;
                    }
                xch_host[0] = misplaced;
                int r = allreduce_host(1);
                if (r) return r;
                std::vector<int> sep_slots;
                if (candidate)
                    for (int sl = 0; sl < HS.nP; ++sl)
                        if ((*job_of)[sl] < 0) sep_slots.push_back(sl);
                if (candidate && xch_host[0] < 0.5 && 36 * sep_blocks->size() + 6 * sep_slots.size() <= xch_doubles) {
                    xs_on = true;
                    xs_nb = (int)sep_blocks->size();
                    xs_ns = (int)sep_slots.size();
                    int* const h_idx = (int*)(hs_struct + st_blk);  // (the block-row image has been uploaded and the stream drained by the planner)
                    uint8_t* const h_lam = (uint8_t*)(h_idx + xs_nb + xs_ns);
                    SV_HIP(ctx, hipStreamSynchronize(s));
                    for (int k = 0; k < xs_nb; ++k) h_idx[k] = (*sep_blocks)[k];
                    for (int k = 0; k < xs_ns; ++k) h_idx[xs_nb + k] = sep_slots[k];
                    for (int sl = 0; sl < HS.nP; ++sl) h_lam[sl] = (*job_of)[sl] >= 0 ? (*owner)[(*job_of)[sl]] == rank : rank == 0;
                    H2D(d_xs_idx, h_idx, 4 * (size_t)(xs_nb + xs_ns));
                    H2D(d_lam_slot, h_lam, (size_t)HS.nP);
                    SV_HIP(ctx, hipStreamSynchronize(s));  // (the staging image is reused)
                    D.lam_slot = d_lam_slot;
                }
            }
            ctx->ba_xch[0] = !sharded ? 0 : (xs_on ? 2 : 1);
            if (trace && sharded) std::fprintf(stderr, "[ba]     exchange per trial: %s (%d separator blocks, %d separator rows of %d)\n", xs_on ? "keyframe segments" : "whole reduced system", xs_nb, xs_ns, HS.nP);
        }
        if (sv_ba_lin_split_max() > 16 || sv_ba_rhs_split() > 16) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: partial-sum buffers too small for the kernel splits");
        if (trace) std::fprintf(stderr, "[ba]   structure %s     %8.3f ms (%zu pairs, %zu blocks, n = %d, solver %d)\n", reuse ? "reused " : "rebuilt", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(), HS.num_pairs, HS.blk_ab.size(), D.n, solver);
        return SVGPU_OK;
    };

    // activeRobustChi2 of the estimate -> partial sums on the device (a sharded solve folds them and sums over the ranks,
    // the stop votes ride along)
    auto chi2_begin = [&](int store_cache) -> int {
        sv_ba_chi2(ctx, s, D, 0, store_cache, 0);
        if (sharded) {
            sv_ba_fold(s, D, d_sc, 0);
            return allreduce_dev(d_sc, 4, XC_SUMS);
        }
        return SVGPU_OK;
    };

    // one damping trial (preceded by the linearisation when the control block asks for one); everything is guarded by the control
    // block on the device, so a step enqueued behind a finished optimisation costs a dozen empty launches
    const int pcg_max_it = ctx->pcg_max_it > 0 ? ctx->pcg_max_it : 0;
    static const bool no_fusion = std::getenv("SVGPU_BA_NO_FUSION") != nullptr;  // A/B aid: the separate kernels
    auto enqueue_step = [&](bool* finished_seen) -> int {
        int r;
        const bool fused_tail = !no_fusion && !sharded && sv_ba_tail_ok(D);
        sv_ba_linearize(ctx, s, D, sharded ? 0 : 1);
        if (sharded) {  // pose blocks summed over the ranks before the damping is initialised from their diagonal
            // ONE collective (round 6; it was two): the rank's slot of the damping initialisation -- the largest diagonal entry of ITS landmarks,
            // which k_ba_lin has just left in the control block -- rides behind the pose blocks.  The pose part of the maximum is taken from the
            // SUMMED blocks afterwards, identically on every rank; k_ba_prepare takes the largest of that and the slots.
            if (HS.nP > 0) {
                SV_HIP(ctx, hipMemcpyAsync(d_HB_full, D.Hpp, 8 * 36 * (size_t)HS.nP, hipMemcpyDeviceToDevice, s));
                SV_HIP(ctx, hipMemcpyAsync(d_HB_full + 36 * (size_t)HS.nP, D.bp, 8 * 6 * (size_t)HS.nP, hipMemcpyDeviceToDevice, s));
            }
            D.maxslots = d_HB_full + 42 * (size_t)HS.nP;
            sv_ba_maxslot(s, D);
            if ((r = allreduce_dev(d_HB_full, 42 * (size_t)HS.nP + (size_t)world, XC_POSE_BLOCKS))) return r;
            sv_ba_maxdiag(s, D);
            sv_ba_prepare(s, D);
        }
        sv_ba_reduce(ctx, s, D);
        if (HS.nP > 0) {
            if (xs_on) {  // keyframe-segment shards: only what touches two separator rows is shared between ranks
                sv_ba_xs_move(s, D, d_xs_idx, xs_nb, d_xs_idx + xs_nb, xs_ns, d_xch, 0);
                if ((r = allreduce_dev(d_xch, 36 * (size_t)xs_nb + 6 * (size_t)xs_ns, XC_SYSTEM))) return r;
                sv_ba_xs_move(s, D, d_xs_idx, xs_nb, d_xs_idx + xs_nb, xs_ns, d_xch, 1);
            }
            else if ((r = allreduce_dev(D.Sblk, 36 * (size_t)D.NB + (size_t)D.n, XC_SYSTEM))) return r;
        }
        if (HS.nP > 0) {
            if (solver == SV_BA_SOLVER_CHOLESKY) sv_ba_solve(ctx, s, D);
            else if (solver == SV_BA_SOLVER_PCG_LDS) sv_ba_solve_pcg_lds(ctx, s, D);
            else if (solver == SV_BA_SOLVER_DENSE) sv_ba_solve_dense(ctx, s, D);
            else if (solver == SV_BA_SOLVER_ENVELOPE) {
                if ((r = sv_sky_solve(ctx, s, D))) return r;
            }
            else {
                // PCG: iterations are enqueued in chunks; the control block says when the solve (or the whole optimisation) is over
                sv_pcg_init(ctx, s, D);
                int it0 = 0, chunk = 96;
                for (;;) {
                    sv_pcg_iterate(ctx, s, D, it0, chunk);
                    it0 += chunk;
                    if ((r = read_ctl())) return r;
                    if (h_ctl->phase != 1) {
                        *finished_seen = h_ctl->phase == 2;
                        break;
                    }
                    if (h_ctl->pcg_done) break;
                    chunk = std::min(chunk + chunk / 2, 768);
                }
            }
        }
        if (fused_tail) sv_ba_tail(ctx, s, D);
        else {
            sv_ba_update(ctx, s, D);
            sv_ba_chi2(ctx, s, D, 1, 0, 1);
        }
        if (sharded) {
            sv_ba_fold(s, D, d_sc, 1);
            if ((r = allreduce_dev(d_sc, 4, XC_SUMS))) return r;
        }
        sv_ba_decide(s, D);
        return SVGPU_OK;
    };

    // SparseOptimizer::optimize(iterations) with the terminate_action hook; the Levenberg-Marquardt loop itself runs on the device
    int pcg_mi = 0;
    bool first_stage = true;
    // device_boundary: the stage follows k_ba_activity on the stream -- the structure of the previous stage is kept (same pose numbering, the
    // landmark activity already refreshed on the device) and the stage refuses to start if that turned out to be wrong (ctl.structure_changed)
    auto optimize = [&](int iterations, int* iters_done, bool device_boundary = false) -> int {
        *iters_done = 0;
        int r = SVGPU_OK;
        if (device_boundary) sv_ba_zero_inactive(s, D);
        else r = upload_structure();
        if (r) return r;
        if (!observations_finished) {
            if ((r = finish_observations())) return r;
            observations_finished = true;
        }
        const bool nothing = !sharded && HS.nP + HS.nL == 0;
        // the PCG iteration cap depends on the size of the reduced system of this stage
        pcg_mi = pcg_max_it > 0 ? pcg_max_it : std::max(2000, 4 * D.n);
        SV_HIP(ctx, hipMemcpyAsync(&D.ctl->pcg_max_it, &pcg_mi, sizeof(int), hipMemcpyHostToDevice, s));
        if ((r = chi2_begin(first_stage ? 1 : 0))) return r;
        first_stage = false;
        sv_ba_begin(s, D, nothing ? 0 : iterations, (int)(sharded ? 0 : (*flag ? 1 : 0)) | (device_boundary ? 2 : 0));
        // The steps of the whole stage are enqueued before the first read-back (the common case -- every first trial accepted --
        // costs ONE synchronisation per stage); rejected trials consume steps, so the loop tops up until the device reports phase 2.
        const int max_steps = 10 * std::max(iterations, 1);
        const bool trace2 = trace && std::getenv("SVGPU_BA_TRACE")[0] == '2';  // one step per read-back, printed
        int steps = 0, todo = nothing ? 0 : (trace2 ? 1 : iterations);
        for (;;) {
            bool finished = false;
            for (int k = 0; k < todo && !finished; ++k, ++steps)
                if ((r = enqueue_step(&finished))) return r;
            if ((r = read_ctl())) return r;
            if (trace2) std::fprintf(stderr, "[ba]     step %3d: it %d phase %d chi2 %.9g temp %.9g lambda %.6g rho %.4g qmax %d pcg_it %d fail %d\n", steps, h_ctl->it, h_ctl->phase, h_ctl->current_chi, h_ctl->temp_chi, h_ctl->lambda, h_ctl->rho, h_ctl->qmax, h_ctl->pcg_it, h_ctl->solve_failures);
            if (h_ctl->phase == 2 || steps >= max_steps) break;
            todo = trace2 ? 1 : std::max(1, std::min(h_ctl->it_max - h_ctl->it, max_steps - steps));
        }
        if (device_boundary && h_ctl->structure_changed) {
            // the stage refused to start (a pose lost its last edge): the control block still carries the PREVIOUS stage's iteration count and
            // terminate verdict -- nothing of it belongs to this stage; the caller rebuilds the structure on the host and runs the stage again
            *iters_done = 0;
            return SVGPU_OK;
        }
        *iters_done = h_ctl->it;
        // errors cached by the last computeActiveErrors (used by the gate and the outlier list)
        if (*iters_done > 0) sv_ba_chi2(ctx, s, D, 0, 1, 0);
        if (h_ctl->stopped_by_terminate) {
            *flag = 1;  // terminate_action writes through the optimizer's force-stop pointer
            st.stopped_by_terminate_action = 1;
        }
        else if (h_ctl->stop) *flag = 1;  // sharded: another rank's caller raised it
        return SVGPU_OK;
    };

    st.chi2_initial = 0;
    int it1 = 0, it2 = 0;
    rc = optimize(pr->num_first_iter, &it1);
    if (rc) return rc;
    if (D.dbg_schur_on) {  // SVGPU_BA_DBG=schur: life of the units of the last k_ba_schur_rhs launch (100 MHz stamps)
        const size_t nu = D.unit_rec ? (size_t)D.num_units : (size_t)D.NB * D.nshare;
        std::vector<unsigned long long> h(8 * nu);
        SV_HIP(ctx, hipMemcpyAsync(h.data(), D.dbg, 8 * h.size(), hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t u = 0; u < nu; ++u)
            if (h[8 * u + 4] < h[8 * u]) h[8 * u + 1] = h[8 * u + 2] = h[8 * u + 3] = h[8 * u + 4] = h[8 * u];  // an empty share: entry stamp only
        for (size_t u = 0; u < nu; ++u) t0 = std::min(t0, h[8 * u]), t1 = std::max(t1, h[8 * u + 4]);
        double sum[5] = {0, 0, 0, 0, 0};
        std::vector<int> per_cu(8 * 64, 0);
        std::vector<double> life(nu);
        for (size_t u = 0; u < nu; ++u) {
            for (int k = 1; k < 5; ++k) sum[k] += (double)(h[8 * u + k] - h[8 * u + k - 1]) * 0.01;
            life[u] = (double)(h[8 * u + 4] - h[8 * u]) * 0.01;
            sum[0] += life[u];
            const unsigned hw = (unsigned)h[8 * u + 5], xcc = (unsigned)(h[8 * u + 5] >> 32) & 15;
            per_cu[(xcc & 7) * 64 + (((hw >> 13) & 3) * 16 + ((hw >> 8) & 15))]++;
        }
        std::sort(life.begin(), life.end());
        int cus = 0, mxu = 0;
        for (int c : per_cu) cus += c > 0, mxu = std::max(mxu, c);
        std::fprintf(stderr, "[ba] schur stamps: %zu units, span %.1f us, mean life %.2f us (p10 %.2f p50 %.2f p90 %.2f max %.2f) -> mean resident %.0f waves; per unit: setup %.2f | first trip %.2f | other trips %.2f | reduce %.2f us; %d (XCC, SE, CU) slots used, max %d units on one\n",
                     nu, (double)(t1 - t0) * 0.01, sum[0] / nu, life[nu / 10], life[nu / 2], life[nu * 9 / 10], life[nu - 1], sum[0] / ((double)(t1 - t0) * 0.01),
                     sum[1] / nu, sum[2] / nu, sum[3] / nu, sum[4] / nu, cus, mxu);
        // start times per XCC: does every XCC receive units at the same rate?
        double last_start[8] = {0}, n_x[8] = {0};
        for (size_t u = 0; u < nu; ++u) {
            const unsigned xcc = (unsigned)(h[8 * u + 5] >> 32) & 7;
            last_start[xcc] = std::max(last_start[xcc], (double)(h[8 * u] - t0) * 0.01), n_x[xcc] += 1;
        }
        for (int x = 0; x < 8; ++x) std::fprintf(stderr, "[ba]   XCC %d: %.0f units, last start at %.1f us\n", x, n_x[x], last_start[x]);
    }
    if (D.dbg && D.dbg_schur_on == 2) {  // SVGPU_BA_DBG=chol: phase stamps of the last k_ba_chol_mfma (100 MHz): load | per panel factor, barrier, update | end
        std::vector<unsigned long long> h(32);
        SV_HIP(ctx, hipMemcpyAsync(h.data(), D.dbg, 8 * h.size(), hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        std::fprintf(stderr, "[ba] chol stamps (us):");
        for (int k = 1; k < 31; ++k)
            if (h[k] >= h[0] && h[k] - h[0] < 100000000ull) std::fprintf(stderr, " %d:%.2f", k, (double)(h[k] - h[0]) * 0.01);
        std::fprintf(stderr, "\n");
    }
    if (D.dbg && !D.dbg_schur_on) {  // SVGPU_BA_DBG: where the time of the last fused tail went (100 MHz stamps, relative to the first workgroup's entry)
        std::vector<unsigned long long> h(8 * (size_t)nb_lm);
        SV_HIP(ctx, hipMemcpyAsync(h.data(), D.dbg, 8 * h.size(), hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb_lm; ++b) t0 = std::min(t0, h[8 * (size_t)b]);
        double mx[7] = {0, 0, 0, 0, 0, 0, 0}, mn[7] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
        for (int b = 0; b < nb_lm; ++b)
            for (int k = 0; k < 7; ++k) {
                const double us = (double)(h[8 * (size_t)b + k] - t0) * 0.01;
                mx[k] = std::max(mx[k], us), mn[k] = std::min(mn[k], us);
            }
        std::fprintf(stderr, "[ba] tail stamps (us, min..max over %d workgroups): entry %.2f..%.2f | hop1 %.2f..%.2f | staged %.2f..%.2f | poses %.2f..%.2f | sums %.2f..%.2f\n",
                     nb_lm, mn[0], mx[0], mn[1], mx[1], mn[2], mx[2], mn[3], mx[3], mn[4], mx[4]);
    }
    st.chi2_initial = h_ctl->chi_begin;
    st.iters_stage1 = it1;
    lap("stage 1");
    if (sharded && stop_ptr_any && h_ctl->stop && it1 == 0 && !h_ctl->stopped_by_terminate) {  // raised before any work, agreed by all ranks
        st.lm_trials = h_ctl->lm_trials;
        if (stats) *stats = st;
        return SVGPU_STOPPED;
    }
    bool run_robust = !single_stage;
    // :317-321 (only the CALLER's flag is consulted here); sharded: the all-reduced vote and the all-reduced "a pointer exists", never a local fact
    if (sharded ? (stop_ptr_any && h_ctl->stop != 0) : (stop && *stop != 0)) run_robust = false;
    if (run_robust) {
        st.stage2_entered = 1;
        // The stage boundary stays on the device when nothing about it needs the host (one GPU, a reduced system whose blocks are all kept):
        // gate -> landmark activity + gated count + "did a pose lose its last edge" -> the second stage's kernels, enqueued without a
        // synchronisation; only a changed pose numbering (rare: a keyframe of the window left without a single inlier) comes back here.
        const bool host_boundary = std::getenv("SVGPU_BA_HOST_BOUNDARY") != nullptr;  // A/B aid (read per call: the tests switch it)
        bool on_device = !sharded && E > 0 && HS.nP > 0 && HS.nP <= 48 && have_lists && !host_boundary;
        if (on_device) {
            sv_ba_gate(s, D, 1, nullptr);
            sv_ba_activity(s, D, d_pt_free);
            rc = optimize(pr->num_second_iter, &it2, true);
            if (rc) return rc;
            st.num_gated = h_ctl->gated;
            if (h_ctl->structure_changed) on_device = false;  // the stage refused to start: the host path below rebuilds and runs it
        }
        if (!on_device) {
            if (E > 0) {
                if (!h_ctl->structure_changed) sv_ba_gate(s, D, 1, nullptr);  // (else the levels are already set)
                SV_HIP(ctx, hipMemcpyAsync(level.data(), D.e_level, E, hipMemcpyDeviceToHost, s));
                no_levels = false;
                SV_HIP(ctx, hipStreamSynchronize(s));
            }
            double gated = 0;
            for (int e = 0; e < E; ++e) gated += level[e];
            if (sharded) {
                xch_host[0] = gated;
                if ((rc = allreduce_host(1))) return rc;
                gated = xch_host[0];
            }
            st.num_gated = (int32_t)gated;
            rc = optimize(pr->num_second_iter, &it2);
            if (rc) return rc;
        }
        st.iters_stage2 = it2;
        lap("stage 2");
    }
    // ---- outlier list, final chi2, read-back: ONE copy of the output block (control block | poses | points | outlier flags)
    if (E > 0 && outlier_out) sv_ba_gate(s, D, 0, d_outlier);
    if ((rc = chi2_begin(0))) return rc;
    sv_ba_begin(s, D, 0, 0);  // folds the chi2 of the final estimate into the control block (phase 2: nothing else happens)
    sv_ba_pack_out(s, D, d_state_out);
    if (sharded) {  // every rank ends with every landmark: owners contribute their points, the rest zeros (an all-gather through the all-reduce), on the device
        sv_ba_points_share(s, D, d_state_out + 12 * (size_t)P, d_xch, 0);
        if ((rc = allreduce_dev(d_xch, 3 * (size_t)L, XC_SETUP))) return rc;
        sv_ba_points_share(s, D, d_state_out + 12 * (size_t)P, d_xch, 1);
    }
    // (a caller that takes no outlier flags -- global BA: 1.2 MB at config 5 -- gets the block without them)
    SV_HIP(ctx, hipMemcpyAsync(hs_out, d_out, outlier_out ? out_total : out_outlier, hipMemcpyDeviceToHost, s));
    if ((rc = wait_stream())) return rc;
    memcpy(h_ctl, hs_out + out_ctl, sizeof(BaCtl));
    memcpy(pose_out, hs_out + out_state, sizeof(double) * 12 * (size_t)P);
    {
        const char* const src = hs_out + out_state + sizeof(double) * 12 * (size_t)P;
        const size_t bytes = sizeof(double) * 3 * (size_t)L;
        if (bytes < ((size_t)16 << 20)) memcpy(points_out, src, bytes);
        else {  // (38 MB at 1.6 M landmarks: one core copies them in ~2 ms)
            const int nt = 4;
            std::vector<std::thread> th;
            auto share = [=](int q) { memcpy((char*)points_out + bytes * q / nt, src + bytes * q / nt, bytes * (q + 1) / nt - bytes * q / nt); };
            int started = 1;  // shares [0, started) have an owner (this thread takes 0)
            try {  // (no exception may cross the C ABI with joinable threads behind it: a share that finds no thread is copied here)
                th.reserve(nt - 1);
                for (int q = 1; q < nt; ++q) {
                    th.emplace_back(share, q);
                    started = q + 1;
                }
            }
            catch (...) {
            }
            share(0);
            for (int q = started; q < nt; ++q) share(q);
            for (auto& t : th) t.join();
        }
    }
    const uint8_t* const outl = (const uint8_t*)(hs_out + out_outlier);
    if (outlier_out)
        for (int k = 0; k < E; ++k) outlier_out[perm.empty() ? k : perm[k]] = outl[k];
    st.chi2_final = h_ctl->chi_begin;
    st.lm_trials = h_ctl->lm_trials;
    st.cholesky_failures = h_ctl->solve_failures;
    st.pcg_iterations = h_ctl->pcg_total_it;
    lap("read-back");
    st.lambda_final = h_ctl->lambda;
    ctx->ba_xch[8] = st.lm_trials;
    ctx->ba_xch[9] = st.iters_stage1 + st.iters_stage2;
    if (stats) *stats = st;
#undef H2D
    return SVGPU_OK;
}

extern "C" {

int svgpu_local_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out,
                   double* points_out, uint8_t* outlier_out, svgpu_ba_stats* stats) {
    return local_ba_impl(ctx, problem, false, 0, 1, nullptr, nullptr, stop, pose_out, points_out, outlier_out, stats);
}

// Shared tail of the two pose-optimizer entry points: inputs already on the device, results come back in ONE copy
// (pose | result[4] | outlier flags) through the page-locked staging buffer.
static int pose_optimize_run(svgpu_ctx* ctx, Arena& A, const double* pose_cw, int n, const double* d_pos, const float* d_uvr, const float* d_w,
                             const float* d_h, const double* intrinsics, int num_trials_robust, int num_trials, int num_each_iter,
                             int reset_stop_flag_each_round, char* h_out, double* pose_out, uint8_t* outlier_flags, int* num_valid, int* lm_iterations) {
    PoseOptDev P;
    memset(&P, 0, sizeof(P));
    char* const d_out = A.base + A.off;
    P.pose_out = A.take<double>(12);
    P.result = A.take<int>(4);
    P.outlier = A.take<uint8_t>(n);
    const size_t out_bytes = (size_t)((A.base + A.off) - d_out), off_result = pad(96), off_outlier = off_result + pad(16);
    P.level = A.take<uint8_t>(n);
    P.robust = A.take<uint8_t>(n);
    if (A.off > ctx->scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_pose_optimize: internal arena overflow");
    hipStream_t s = ctx->stream;
    P.n = n;
    P.pos_w = d_pos;
    P.uvr = d_uvr;
    P.inv_sigma_sq = d_w;
    P.huber = d_h;
    memcpy(P.intr, intrinsics, sizeof(double) * 5);
    memcpy(P.pose_in, pose_cw, sizeof(double) * 12);
    P.num_trials_robust = num_trials_robust;
    P.num_trials = num_trials;
    P.num_each_iter = num_each_iter;
    P.reset_flag_each_round = reset_stop_flag_each_round;
    P.gain_thr = 1e-3;  // terminateAction->setGainThreshold(1e-3), pose_optimizer_g2o.cc:55
    sv_pose_opt(ctx, s, P);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    memcpy(pose_out, h_out, sizeof(double) * 12);
    const int* result = (const int*)(h_out + off_result);
    memcpy(outlier_flags, h_out + off_outlier, n);
    *num_valid = result[0];
    if (lm_iterations) *lm_iterations = result[1];
    return SVGPU_OK;
}

int svgpu_pose_optimize(svgpu_ctx* ctx, const double* pose_cw, int n, const double* pos_w, const float* uvr,
                        const float* inv_sigma_sq, const float* huber_delta, const double* intrinsics, int num_trials_robust,
                        int num_trials, int num_each_iter, int reset_stop_flag_each_round, double* pose_out,
                        uint8_t* outlier_flags, int* num_valid, int* lm_iterations) {
    if (!ctx || !pose_cw || !pose_out || !num_valid || n < 0 || num_trials_robust < 0 || num_trials < 0 || num_each_iter < 0
        || (n > 0 && (!pos_w || !uvr || !inv_sigma_sq || !huber_delta || !intrinsics || !outlier_flags)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_pose_optimize: bad arguments");
    memcpy(pose_out, pose_cw, sizeof(double) * 12);
    *num_valid = 0;
    if (lm_iterations) *lm_iterations = 0;
    for (int i = 0; i < n; ++i) outlier_flags[i] = 0;
    if (n < 5) return SVGPU_OK;  // pose_optimizer_g2o.cc:109-111
    SV_HIP(ctx, hipSetDevice(ctx->device));
    // inputs: one page-locked image in device layout (pos_w | uvr | inv_sigma_sq | huber), one copy
    const size_t o_uvr = pad(24 * (size_t)n), o_w = o_uvr + pad(12 * (size_t)n), o_h = o_w + pad(4 * (size_t)n), in_total = o_h + pad(4 * (size_t)n);
    const size_t out_total = pad(96) + pad(16) + pad(n);
    int rc = sv_ensure_scratch(ctx, in_total + out_total + 2 * pad(n) + 1024);
    if (rc) return rc;
    if ((rc = sv_ensure_stage(ctx, in_total + out_total))) return rc;
    char* const hs = ctx->h_stage;
    memcpy(hs, pos_w, 24 * (size_t)n);
    memcpy(hs + o_uvr, uvr, 12 * (size_t)n);
    memcpy(hs + o_w, inv_sigma_sq, 4 * (size_t)n);
    memcpy(hs + o_h, huber_delta, 4 * (size_t)n);
    Arena A(ctx->d_scratch);
    char* const d_in = A.take<char>(in_total);
    SV_HIP(ctx, hipMemcpyAsync(d_in, hs, in_total, hipMemcpyHostToDevice, ctx->stream));
    return pose_optimize_run(ctx, A, pose_cw, n, (const double*)d_in, (const float*)(d_in + o_uvr), (const float*)(d_in + o_w), (const float*)(d_in + o_h),
                             intrinsics, num_trials_robust, num_trials, num_each_iter, reset_stop_flag_each_round, hs + in_total, pose_out,
                             outlier_flags, num_valid, lm_iterations);
}

int svgpu_pose_optimize_device(svgpu_ctx* ctx, const double* pose_cw, int n, const double* pos_w_dev, const float* uvr_dev,
                               const float* inv_sigma_sq_dev, const float* huber_delta_dev, const double* intrinsics, int num_trials_robust,
                               int num_trials, int num_each_iter, int reset_stop_flag_each_round, double* pose_out,
                               uint8_t* outlier_flags, int* num_valid, int* lm_iterations) {
    if (!ctx || !pose_cw || !pose_out || !num_valid || n < 0 || num_trials_robust < 0 || num_trials < 0 || num_each_iter < 0
        || (n > 0 && (!pos_w_dev || !uvr_dev || !inv_sigma_sq_dev || !huber_delta_dev || !intrinsics || !outlier_flags)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_pose_optimize_device: bad arguments");
    memcpy(pose_out, pose_cw, sizeof(double) * 12);
    *num_valid = 0;
    if (lm_iterations) *lm_iterations = 0;
    for (int i = 0; i < n; ++i) outlier_flags[i] = 0;
    if (n < 5) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t out_total = pad(96) + pad(16) + pad(n);
    int rc = sv_ensure_scratch(ctx, out_total + 2 * pad(n) + 1024);
    if (rc) return rc;
    if ((rc = sv_ensure_stage(ctx, out_total))) return rc;
    Arena A(ctx->d_scratch);
    return pose_optimize_run(ctx, A, pose_cw, n, pos_w_dev, uvr_dev, inv_sigma_sq_dev, huber_delta_dev, intrinsics, num_trials_robust, num_trials,
                             num_each_iter, reset_stop_flag_each_round, ctx->h_stage, pose_out, outlier_flags, num_valid, lm_iterations);
}

int svgpu_global_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out, double* points_out,
                    svgpu_ba_stats* stats) {
    return local_ba_impl(ctx, problem, true, 0, 1, nullptr, nullptr, stop, pose_out, points_out, nullptr, stats);
}

int svgpu_local_ba_sharded(svgpu_ctx* ctx, const svgpu_ba_problem* shard, int rank, int world, svgpu_allreduce_fn allreduce,
                           void* allreduce_user, volatile uint8_t* stop, double* pose_out, double* points_out,
                           uint8_t* outlier_out, svgpu_ba_stats* stats) {
    if (!ctx) return SVGPU_ERR_INVALID;
    if (!allreduce) {  // the context's own RCCL communicator (svgpu_comm_init)
        if (!ctx->comm) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: no all-reduce callback and no communicator (svgpu_comm_init)");
        if (rank != ctx->comm_rank || world != ctx->comm_world) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: rank / world differ from the communicator's");
        allreduce = rccl_allreduce_cb;
        allreduce_user = ctx;
    }
    return local_ba_impl(ctx, shard, false, rank, world, allreduce, allreduce_user, stop, pose_out, points_out, outlier_out, stats);
}

int svgpu_global_ba_sharded(svgpu_ctx* ctx, const svgpu_ba_problem* shard, int rank, int world, svgpu_allreduce_fn allreduce,
                            void* allreduce_user, volatile uint8_t* stop, double* pose_out, double* points_out, svgpu_ba_stats* stats) {
    if (!ctx) return SVGPU_ERR_INVALID;
    if (!allreduce) {
        if (!ctx->comm) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_global_ba_sharded: no all-reduce callback and no communicator (svgpu_comm_init)");
        if (rank != ctx->comm_rank || world != ctx->comm_world) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_global_ba_sharded: rank / world differ from the communicator's");
        allreduce = rccl_allreduce_cb;
        allreduce_user = ctx;
    }
    return local_ba_impl(ctx, shard, true, rank, world, allreduce, allreduce_user, stop, pose_out, points_out, nullptr, stats);
}

int svgpu_ba_last_exchange(svgpu_ctx* ctx, int64_t* info) {
    if (!ctx || !info) return SVGPU_ERR_INVALID;
    for (int k = 0; k < 10; ++k) info[k] = ctx->ba_xch[k];
    return SVGPU_OK;
}

// include/svgpu.h.  The block pattern is built the way the solve builds it (a kept block (a, b) = a free landmark observed from the free
// keyframes a and b; every diagonal block), the cuts come from the solve's own planner (sv_sky_partition_roles).
int svgpu_ba_partition_keyframe_segments(const svgpu_ba_problem* pr, int world, int32_t* landmark_rank, int32_t* info) {
    if (!pr || !landmark_rank || !info || world < 1) return SVGPU_ERR_INVALID;
    const int P = pr->num_poses, L = pr->num_points, E = pr->num_obs;
    for (int k = 0; k < 12; ++k) info[k] = 0;
    for (int l = 0; l < L; ++l) landmark_rank[l] = l % world;
    if (P <= 0 || L <= 0 || E <= 0 || world == 1) return SVGPU_OK;
    for (int e = 0; e < E; ++e)
        if (pr->obs_pose[e] < 0 || pr->obs_pose[e] >= P || pr->obs_point[e] < 0 || pr->obs_point[e] >= L) return SVGPU_ERR_INVALID;
    // free keyframes with an observation, numbered in pose order (activity_only above)
    std::vector<uint8_t> seen(P, 0);
    for (int e = 0; e < E; ++e) seen[pr->obs_pose[e]] = 1;
    std::vector<int> slot(P, -1);
    int nP = 0;
    for (int p = 0; p < P; ++p)
        if (seen[p] && !pr->pose_fixed[p]) slot[p] = nP++;
    info[6] = nP;
    if (nP < 4 || nP > 32768) return SVGPU_OK;  // (the pattern below is a dense BIT table, nP^2 / 8 bytes: 32 MB at 16 384 free keyframes, 128 MB at the 32 768 cap; the a/b scan behind it is O(nP^2); larger graphs keep the whole problem in one segment)
    // observations grouped by landmark
    std::vector<int> off(L + 1, 0), obs(E);
    for (int e = 0; e < E; ++e) ++off[pr->obs_point[e] + 1];
    for (int l = 0; l < L; ++l) off[l + 1] += off[l];
    {
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int e = 0; e < E; ++e) obs[fill[pr->obs_point[e]]++] = slot[pr->obs_pose[e]];
    }
    std::vector<uint64_t> present(((size_t)nP * nP + 63) / 64, 0);
    auto mark = [&](size_t k) { present[k >> 6] |= 1ull << (k & 63); };
    auto marked = [&](size_t k) -> bool { return (present[k >> 6] >> (k & 63)) & 1ull; };
    for (int l = 0; l < L; ++l) {
        if (pr->point_fixed && pr->point_fixed[l]) continue;
        for (int i = off[l]; i < off[l + 1]; ++i) {
            const int a = obs[i];
            if (a < 0) continue;
            for (int j = i + 1; j < off[l + 1]; ++j) {
                const int b = obs[j];
                if (b >= 0) mark((size_t)std::min(a, b) * nP + std::max(a, b));
            }
        }
    }
    std::vector<int2> blk_ab;
    for (int a = 0; a < nP; ++a)
        for (int b = a; b < nP; ++b)
            if (a == b || marked((size_t)a * nP + b)) {
                int2 ab;
                ab.x = a, ab.y = b;
                blk_ab.push_back(ab);
            }
    info[7] = (int)blk_ab.size();
    std::vector<int> job_of, owner;
    int ncuts = 0, sep_blocks_n = 0;
    long long xch_doubles = 0;
    if (!sv_sky_partition_roles(nP, blk_ab, world, job_of, owner, &ncuts, &sep_blocks_n, &xch_doubles)) return SVGPU_OK;
    // a landmark follows the piece of its keyframes.  (A FIXED landmark couples nothing and may be seen from two pieces: its observations
    // cannot be placed on one rank without feeding a foreign piece's pose blocks -- such a problem keeps l % world.)
    std::vector<int32_t> lr(L);
    int shared = 0, sep_only = 0;
    for (int l = 0; l < L; ++l) {
        int job = -1, on_sep = 0;
        for (int i = off[l]; i < off[l + 1]; ++i) {
            const int a = obs[i];
            if (a < 0) continue;
            if (job_of[a] < 0) on_sep = 1;
            else if (job < 0) job = job_of[a];
            else if (job != job_of[a]) return SVGPU_OK;
        }
        lr[l] = job >= 0 ? owner[job] : l % world;
        shared += on_sep;
        sep_only += on_sep && job < 0;
    }
    for (int l = 0; l < L; ++l) landmark_rank[l] = lr[l];
    int nsep = 0;
    for (int a = 0; a < nP; ++a) nsep += job_of[a] < 0;
    info[0] = 1, info[1] = (int)owner.size(), info[2] = ncuts, info[3] = nsep, info[4] = shared, info[5] = sep_only;
    info[8] = sep_blocks_n, info[9] = (int32_t)std::min<long long>(xch_doubles, 0x7fffffff);
    return SVGPU_OK;
}

int svgpu_ba_set_solver(svgpu_ctx* ctx, int solver, double pcg_tolerance, int pcg_max_iterations) {
    if (!ctx || solver < SVGPU_BA_SOLVER_AUTO || (solver > SVGPU_BA_SOLVER_PCG_MULTI && solver != SVGPU_BA_SOLVER_ENVELOPE && solver != SVGPU_BA_SOLVER_CHOLESKY_MFMA)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_ba_set_solver: bad solver");
    ctx->ba_solver = solver;
    ctx->pcg_tol = pcg_tolerance > 0 ? pcg_tolerance : 1e-10;
    ctx->pcg_max_it = pcg_max_iterations > 0 ? pcg_max_iterations : 0;
    return SVGPU_OK;
}

int svgpu_comm_unique_id(uint8_t* id128) {
    if (!id128) return SVGPU_ERR_INVALID;
    if (!rccl_ready()) return SVGPU_ERR_HIP;
    NcclId id;
    if (g_rccl.get_unique_id(&id) != 0) return SVGPU_ERR_HIP;
    memcpy(id128, id.internal, 128);
    return SVGPU_OK;
}

int svgpu_comm_init(svgpu_ctx* ctx, int rank, int world, const uint8_t* id128) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_comm_init: bad arguments");
    if (!rccl_ready()) return sv_set_error(ctx, SVGPU_ERR_HIP, "svgpu_comm_init: librccl.so could not be loaded");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    sv_comm_release(ctx);
    NcclId id;
    memcpy(id.internal, id128, 128);
    const int r = g_rccl.comm_init_rank(&ctx->comm, world, id, rank);
    if (r != 0) {
        ctx->comm = nullptr;
        return sv_set_error(ctx, SVGPU_ERR_HIP, g_rccl.get_error_string ? g_rccl.get_error_string(r) : "ncclCommInitRank failed");
    }
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return SVGPU_OK;
}

void svgpu_comm_destroy(svgpu_ctx* ctx) { sv_comm_release(ctx); }

int svgpu_comm_allreduce_f64(svgpu_ctx* ctx, double* dev_buf, size_t count, void* stream) {
    if (!ctx || !ctx->comm) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_comm_allreduce_f64: no communicator");
    return rccl_allreduce_cb(ctx, dev_buf, count, stream ? stream : (void*)ctx->stream) == 0 ? SVGPU_OK : sv_set_error(ctx, SVGPU_ERR_HIP, "ncclAllReduce failed");
}

}  // extern "C"
