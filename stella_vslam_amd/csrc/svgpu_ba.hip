// Host side of local bundle adjustment: problem flattening, structure (CSR) construction, and the
// Levenberg-Marquardt control flow of g2o restated around the device kernels.
//   two-stage schedule                 optimize/local_bundle_adjuster_g2o.cc:306-348
//   OptimizationAlgorithmLevenberg     g2o (pinned 20230223_git): lambda init 1e-5 * max diag, rho test, 10 trials
//   terminate_action                   optimize/terminate_action.cc:36-76 (writes through the force-stop pointer)
#include <algorithm>
#include <cfloat>
#include <cmath>

#include <chrono>
#include <cstdlib>

#include "svgpu_internal.h"
#include "ba_kernels.h"

void sv_ba_maxdiag(hipStream_t s, const BaDev& D);
void sv_ba_zero_inactive(hipStream_t s, const BaDev& D);
size_t sv_ba_pairs_scratch_bytes(size_t pair_cap, int L, size_t nb_cap);
int sv_ba_build_pairs(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int2* pairs_out,
                      std::vector<int>& dense_off_host);

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
    std::vector<int> pe_off, pe_idx;
    std::vector<int> blk_off;
    std::vector<int2> blk_ab;
    size_t num_pairs = 0;  // the pairs themselves live on the device only
    int nP = 0, nL = 0;
};

// pose / landmark activity under the current edge levels (the cheap first half of build_structure)
void activity_only(const svgpu_ba_problem& pr, const std::vector<int>& e_pose, const std::vector<int>& e_point,
                   const std::vector<uint8_t>& level, const std::vector<uint8_t>* pose_active_global, HostStructure& H) {
    const int P = pr.num_poses, L = pr.num_points, E = pr.num_obs;
    std::vector<uint8_t> pa(P, 0), la(L, 0);
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

// initializeOptimization(level 0): active vertices = endpoints of active edges; free = active and not fixed.
void build_structure(const svgpu_ba_problem& pr, const std::vector<int>& e_pose, const std::vector<int>& e_point,
                     const std::vector<int>& lm_off, const std::vector<uint8_t>& level, const std::vector<uint8_t>* pose_active_global,
                     HostStructure& H) {
    const int P = pr.num_poses, L = pr.num_points, E = pr.num_obs;
    std::vector<uint8_t> pa(P, 0), la(L, 0);
    for (int e = 0; e < E; ++e)
        if (!level[e]) {
            pa[e_pose[e]] = 1;
            la[e_point[e]] = 1;
        }
    if (pose_active_global) pa = *pose_active_global;  // sharded solve: activity summed over all ranks
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
    // pose -> active edges
    H.pe_off.assign(H.nP + 1, 0);
    for (int e = 0; e < E; ++e)
        if (!level[e] && H.pose_slot[e_pose[e]] >= 0) H.pe_off[H.pose_slot[e_pose[e]] + 1]++;
    for (int s = 0; s < H.nP; ++s) H.pe_off[s + 1] += H.pe_off[s];
    H.pe_idx.resize(H.pe_off[H.nP]);
    {
        std::vector<int> fill(H.pe_off.begin(), H.pe_off.end() - 1);
        for (int e = 0; e < E; ++e)
            if (!level[e] && H.pose_slot[e_pose[e]] >= 0) H.pe_idx[fill[H.pose_slot[e_pose[e]]]++] = e;
    }
    // the (edge, edge) pair lists of the upper blocks (a <= b) are built on the device: sv_ba_build_pairs + compact_blocks
}

// dense block offsets (from the device) -> kept blocks: every diagonal block (it carries Hpp + lambda I) and every non-empty
// off-diagonal block, in (a, b) order; the sorted pair array needs no compaction (empty blocks hold no pairs)
void compact_blocks(const std::vector<int>& dense_off, HostStructure& H) {
    H.blk_ab.clear();
    H.blk_off.clear();
    size_t k = 0;
    for (int a = 0; a < H.nP; ++a)
        for (int b = a; b < H.nP; ++b, ++k)
            if (a == b || dense_off[k + 1] > dense_off[k]) {
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

static int local_ba_impl(svgpu_ctx* ctx, const svgpu_ba_problem* pr, bool single_stage, int rank, int world,
                         svgpu_allreduce_fn allreduce, void* ar_user, volatile uint8_t* stop, double* pose_out, double* points_out, uint8_t* outlier_out,
                         svgpu_ba_stats* stats) {
    if (!ctx || !pr || !pose_out || !points_out) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: null argument");
    const int P = pr->num_poses, L = pr->num_points, E = pr->num_obs;
    if (P < 0 || L < 0 || E < 0 || (P > 0 && (!pr->pose_cw || !pr->pose_fixed || !pr->intrinsics)) || (L > 0 && !pr->points)
        || (E > 0 && (!pr->obs_pose || !pr->obs_point || !pr->obs_uvr || !pr->obs_inv_sigma_sq || (!outlier_out && !single_stage))))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: inconsistent problem");
    for (int e = 0; e < E; ++e)
        if (pr->obs_pose[e] < 0 || pr->obs_pose[e] >= P || pr->obs_point[e] < 0 || pr->obs_point[e] >= L)
            return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: observation index out of range");
    const bool sharded = allreduce != nullptr;
    if (sharded && (world < 1 || rank < 0 || rank >= world)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: bad rank/world");
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
    memcpy(pose_out, pr->pose_cw, sizeof(double) * 12 * (size_t)P);
    memcpy(points_out, pr->points, sizeof(double) * 3 * (size_t)L);
    if (E > 0 && outlier_out) memset(outlier_out, 0, E);
    if (stats) *stats = st;
    if (!sharded) {
        if (stop && *stop) return SVGPU_STOPPED;  // local_bundle_adjuster_g2o.cc:308-310
        if (E == 0 || P == 0 || L == 0) return SVGPU_OK;
    }
    else if (P == 0 || L == 0) return SVGPU_OK;  // same on every rank
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    // ---- sort the observations by landmark (stable): landmark-major kernels then read contiguous runs
    std::vector<int> lm_off(L + 1, 0), perm(E);
    for (int e = 0; e < E; ++e) lm_off[pr->obs_point[e] + 1]++;
    for (int l = 0; l < L; ++l) lm_off[l + 1] += lm_off[l];
    {
        std::vector<int> fill(lm_off.begin(), lm_off.end() - 1);
        for (int e = 0; e < E; ++e) perm[fill[pr->obs_point[e]]++] = e;
    }
    std::vector<int> e_pose(E), e_point(E);
    std::vector<float> e_uvr(3 * (size_t)E), e_w(E), e_hub(E);
    std::vector<uint8_t> level(E, 0), robust(E, 0);
    for (int k = 0; k < E; ++k) {
        const int e = perm[k];
        e_pose[k] = pr->obs_pose[e];
        e_point[k] = pr->obs_point[e];
        e_uvr[3 * k] = pr->obs_uvr[3 * e];
        e_uvr[3 * k + 1] = pr->obs_uvr[3 * e + 1];
        e_uvr[3 * k + 2] = pr->obs_uvr[3 * e + 2];
        e_w[k] = pr->obs_inv_sigma_sq[e];
        e_hub[k] = pr->obs_huber_delta ? pr->obs_huber_delta[e] : 0.f;
        robust[k] = e_hub[k] > 0.f;
    }

    lap("sort observations");
    // ---- device arena
    const int nPmax = P, nmax = 6 * nPmax;
    const int nb_chi = (E + 255) / 256, nb_lm = (8 * L + 255) / 256 /* k_ba_update_lm: 8 lanes per landmark */, nb_pose = (P + 255) / 256;
    const size_t pairs_max_guess = 0;  // pair lists are sized after build_structure (second arena piece)
    (void)pairs_max_guess;
    size_t need = 4 * pad(sizeof(double) * 12 * P) + 4 * pad(sizeof(double) * 3 * L) + 2 * pad(4 * (size_t)E) + pad(12 * (size_t)E)
                  + 2 * pad(4 * (size_t)E) + 2 * pad(E) + pad(8 * (size_t)E) + pad(40 * (size_t)P) + pad(4 * (size_t)P) + pad(L)
                  + pad(4 * (size_t)(L + 1)) + pad(4 * (size_t)(P + 1)) + pad(4 * (size_t)E) + 2 * pad(sizeof(double) * 18 * E) + pad(sizeof(double) * 27 * 4 * (size_t)P) + pad(4 * (size_t)P) + pad(sizeof(double) * 36 * 4 * ((size_t)P * (P + 1) / 2 + 1)) + pad(4 * ((size_t)P * (P + 1) / 2 + 1)) + pad(sizeof(double) * 6 * E)
                  + 3 * pad(sizeof(double) * 6 * L) + 2 * pad(sizeof(double) * 3 * L) + pad(sizeof(double) * 36 * P)
                  + pad(sizeof(double) * 6 * P) + pad(sizeof(double) * (size_t)(nmax + 1) * nmax) + pad(sizeof(double) * nmax)
                  + pad(sizeof(double) * (nb_chi + nb_lm + nb_pose + 8)) + pad(E + 1) + pad(8 * 42 * (size_t)P)
                  + pad(8 * (size_t)(64 + world + 1)) + pad(8 * ((size_t)(P > 4 * L ? P : 4 * L) + 8)) + 4096;
    // worst-case pair storage: sum over landmarks of k(k+1)/2 (+ duplicates never exceed k^2)
    size_t pair_cap = 0;
    for (int l = 0; l < L; ++l) {
        const size_t k = lm_off[l + 1] - lm_off[l];
        pair_cap += k * k;
    }
    const size_t nb_cap = (size_t)P * (P + 1) / 2;
    const size_t pair_scratch = sv_ba_pairs_scratch_bytes(pair_cap, L, nb_cap);
    need += pad(8 * pair_cap) + pad(8 * nb_cap) + pad(4 * (nb_cap + 1)) + pad(pair_scratch);
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    const size_t nb_cap_early = (size_t)P * (P + 1) / 2 + 1;
    BaDev D;
    memset(&D, 0, sizeof(D));
    D.P = P;
    D.L = L;
    D.E = E;
    D.pose_cur = A.take<double>(12 * (size_t)P);
    D.pose_trial = A.take<double>(12 * (size_t)P);
    D.pt_cur = A.take<double>(3 * (size_t)L);
    D.pt_trial = A.take<double>(3 * (size_t)L);
    int* d_e_pose = A.take<int>(E);
    int* d_e_point = A.take<int>(E);
    float* d_e_uvr = A.take<float>(3 * (size_t)E);
    float* d_e_w = A.take<float>(E);
    float* d_e_hub = A.take<float>(E);
    D.e_level = A.take<uint8_t>(E);
    D.e_robust = A.take<uint8_t>(E);
    D.e_chi = A.take<double>(E);
    double* d_intr = A.take<double>(5 * (size_t)P);
    int* d_pose_slot = A.take<int>(P);
    uint8_t* d_pt_free = A.take<uint8_t>(L);
    int* d_lm_off = A.take<int>(L + 1);
    int* d_pe_off = A.take<int>(P + 1);
    int* d_pe_idx = A.take<int>(E);
    D.W = A.take<double>(18 * (size_t)E);
    D.Y = A.take<double>(18 * (size_t)E);
    D.lp_part = A.take<double>(27 * 4 * (size_t)P);
    D.sc_part = A.take<double>(36 * 4 * nb_cap_early);
    D.GE = A.take<double>(6 * (size_t)E);
    D.Hll = A.take<double>(6 * (size_t)L);
    D.Dinv = A.take<double>(6 * (size_t)L);
    D.bl = A.take<double>(3 * (size_t)L);
    D.dl = A.take<double>(3 * (size_t)L);
    D.Hpp = A.take<double>(36 * (size_t)P);
    D.bp = A.take<double>(6 * (size_t)P);
    D.S = A.take<double>((size_t)(nmax + 1) * nmax);
    D.dp = A.take<double>(nmax);
    D.red = A.take<double>(nb_chi + nb_lm + nb_pose + 8);
    uint8_t* d_outlier = A.take<uint8_t>(E + 1);
    double* d_HB_full = A.take<double>(42 * (size_t)P);           // sharded: Hpp | bp summed over ranks
    double* d_sc = A.take<double>(64 + (size_t)(world > 0 ? world : 1));  // sharded: scalar exchange buffer
    double* d_xch = A.take<double>((size_t)(P > 4 * L ? P : 4 * L) + 8);    // sharded: pose-activity / point exchange
    int2* d_blk_pairs = A.take<int2>(pair_cap);
    int2* d_blk_ab = A.take<int2>(nb_cap);
    int* d_blk_off = A.take<int>(nb_cap + 1);
    char* d_pair_scratch = A.take<char>(pair_scratch);
    if (A.off > ctx->scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba: internal arena overflow");
    D.e_pose = d_e_pose;
    D.e_point = d_e_point;
    D.e_uvr = d_e_uvr;
    D.e_w = d_e_w;
    D.e_huber = d_e_hub;
    D.intr = d_intr;
    D.pose_slot = d_pose_slot;
    D.pt_free = d_pt_free;
    D.lm_off = d_lm_off;
    D.pe_off = d_pe_off;
    D.pe_idx = d_pe_idx;
    D.blk_pairs = d_blk_pairs;
    D.blk_ab = d_blk_ab;
    D.blk_off = d_blk_off;
    D.red_chi_off = 0;
    D.red_chi_n = nb_chi;
    D.red_scale_off = nb_chi;
    D.red_scale_n = nb_lm + nb_pose;
    D.red_flag_off = nb_chi + nb_lm + nb_pose;
    const int red_total = nb_chi + nb_lm + nb_pose + 8;
    // per-trial read-back (partial sums, flags) lands in page-locked memory: no staging copy on the D2H path
    if (ctx->pinned_doubles < (size_t)red_total) {
        if (ctx->h_pinned) SV_HIP(ctx, hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->pinned_doubles = 0;
        SV_HIP(ctx, hipHostMalloc((void**)&ctx->h_pinned, sizeof(double) * (size_t)red_total * 2, hipHostMallocDefault));
        ctx->pinned_doubles = (size_t)red_total * 2;
    }
    double* const red_host = ctx->h_pinned;

#define H2D(dst, src, bytes) SV_HIP(ctx, hipMemcpyAsync((void*)(dst), (src), (bytes), hipMemcpyHostToDevice, s))
    H2D(D.pose_cur, pr->pose_cw, sizeof(double) * 12 * (size_t)P);
    H2D(D.pt_cur, pr->points, sizeof(double) * 3 * (size_t)L);
    H2D(d_e_pose, e_pose.data(), 4 * (size_t)E);
    H2D(d_e_point, e_point.data(), 4 * (size_t)E);
    H2D(d_e_uvr, e_uvr.data(), 12 * (size_t)E);
    H2D(d_e_w, e_w.data(), 4 * (size_t)E);
    H2D(d_e_hub, e_hub.data(), 4 * (size_t)E);
    H2D(D.e_level, level.data(), E);
    H2D(D.e_robust, robust.data(), E);
    H2D(d_intr, pr->intrinsics, sizeof(double) * 5 * (size_t)P);
    H2D(d_lm_off, lm_off.data(), 4 * (size_t)(L + 1));
    SV_HIP(ctx, hipMemsetAsync(D.e_chi, 0, 8 * (size_t)E, s));

    // ---- exchange helpers (sharded solve; no-ops otherwise)
    auto allreduce_dev = [&](double* dev, size_t n) -> int {
        if (!sharded || n == 0) return SVGPU_OK;
        return allreduce(ar_user, dev, n, (void*)s) == 0 ? SVGPU_OK : sv_set_error(ctx, SVGPU_ERR_HIP, "all-reduce callback failed");
    };
    auto allreduce_host = [&](double* v, int n) -> int {  // sum n host doubles over the ranks
        if (!sharded) return SVGPU_OK;
        H2D(d_sc, v, 8 * (size_t)n);
        int r = allreduce_dev(d_sc, n);
        if (r) return r;
        SV_HIP(ctx, hipMemcpyAsync(v, d_sc, 8 * (size_t)n, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        return SVGPU_OK;
    };
    std::vector<double> xch_host((size_t)(P > 4 * L ? P : 4 * L) + 8);
    std::vector<uint8_t> owned(L, 0);  // landmarks whose observations live on this rank
    for (int k = 0; k < E; ++k) owned[e_point[k]] = 1;
    if (sharded) {  // contract check: a landmark's observations must not be split over ranks
        for (int l = 0; l < L; ++l) xch_host[l] = owned[l];
        H2D(d_xch, xch_host.data(), 8 * (size_t)L);
        int r = allreduce_dev(d_xch, L);
        if (r) return r;
        SV_HIP(ctx, hipMemcpyAsync(xch_host.data(), d_xch, 8 * (size_t)L, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        for (int l = 0; l < L; ++l)
            if (xch_host[l] > 1.5) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: observations must be sharded by landmark");
    }

    lap("arena + uploads");
    HostStructure HS;
    bool have_lists = false;
    std::vector<uint8_t> pose_active;
    auto upload_structure = [&]() -> int {
        const std::vector<uint8_t>* pa_override = nullptr;
        if (sharded) {  // a pose is active if ANY rank holds an active observation of it
            for (int p = 0; p < P; ++p) xch_host[p] = 0;
            for (int e = 0; e < E; ++e)
                if (!level[e]) xch_host[e_pose[e]] = 1;
            H2D(d_xch, xch_host.data(), 8 * (size_t)P);
            int r = allreduce_dev(d_xch, P);
            if (r) return r;
            SV_HIP(ctx, hipMemcpyAsync(xch_host.data(), d_xch, 8 * (size_t)P, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
            pose_active.assign(P, 0);
            for (int p = 0; p < P; ++p) pose_active[p] = xch_host[p] > 0.5;
            pa_override = &pose_active;
        }
        // Activity of poses / landmarks under the current levels (cheap), then the expensive part -- pose->edge lists
        // and (edge, edge) pair lists -- only when the free-pose numbering changed.  Lists built for an earlier stage
        // stay valid: edges excluded since then are skipped by level (lin_pose, rhs) or contribute W = Y = 0 (pairs).
        auto tb0 = std::chrono::steady_clock::now();
        HostStructure probe;
        activity_only(*pr, e_pose, e_point, level, pa_override, probe);
        const bool reuse = have_lists && probe.pose_slot == HS.pose_slot;
        if (reuse) {
            HS.pt_free = probe.pt_free;
            HS.nL = probe.nL;
        }
        else {
            build_structure(*pr, e_pose, e_point, lm_off, level, pa_override, HS);
            have_lists = true;
        }
        D.nP = HS.nP;
        D.n = 6 * HS.nP;
        D.chol_in_lds = D.n <= 192 && sizeof(double) * (size_t)(D.n + 1) * (D.n | 1) <= 160 * 1024 - 12 * 1024;
        D.Hpp_full = sharded ? d_HB_full : D.Hpp;
        D.bp_full = sharded ? d_HB_full + 36 * (size_t)HS.nP : D.bp;
        D.scale_pose = (!sharded || rank == 0) ? 1 : 0;
        H2D(d_pt_free, HS.pt_free.data(), L);
        if (!reuse) {
            H2D(d_pose_slot, HS.pose_slot.data(), 4 * (size_t)P);
            H2D(d_pe_off, HS.pe_off.data(), 4 * (size_t)(HS.nP + 1));
            if (!HS.pe_idx.empty()) H2D(d_pe_idx, HS.pe_idx.data(), 4 * HS.pe_idx.size());
            std::vector<int> dense_off;
            int rp = sv_ba_build_pairs(ctx, s, D, d_pair_scratch, pair_scratch, pair_cap, d_blk_pairs, dense_off);
            if (rp) return rp;
            compact_blocks(dense_off, HS);
            H2D(d_blk_off, HS.blk_off.data(), 4 * HS.blk_off.size());
            if (!HS.blk_ab.empty()) H2D(d_blk_ab, HS.blk_ab.data(), 8 * HS.blk_ab.size());
        }
        else sv_ba_zero_inactive(s, D);
        D.NB = (int)HS.blk_ab.size();
        if (trace) std::fprintf(stderr, "[ba]   structure %s     %8.3f ms (%zu pairs, %zu blocks)\n", reuse ? "reused " : "rebuilt", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(), HS.num_pairs, HS.blk_ab.size());
        return SVGPU_OK;
    };

    uint8_t aux_flag = 0;  // g2o installs its own flag when the caller passes none (see svgpu.h)
    volatile uint8_t* flag = stop ? stop : &aux_flag;

    // chi2 of the active set at the current / trial state (fixed-order sum of the per-block partials); in a
    // sharded solve the sum runs over all ranks and the stop flags are OR-reduced on the way
    auto chi2 = [&](int use_trial, int store_cache, double* out) -> int {
        double sum = 0;
        if (E > 0) {
            sv_ba_chi2(ctx, s, D, use_trial, store_cache);
            SV_HIP(ctx, hipMemcpyAsync(red_host, D.red + D.red_chi_off, 8 * (size_t)nb_chi, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
            for (int i = 0; i < nb_chi; ++i) sum += red_host[i];
        }
        if (sharded) {
            double v[2] = {sum, (double)(*flag ? 1 : 0)};
            int r = allreduce_host(v, 2);
            if (r) return r;
            sum = v[0];
            if (v[1] > 0.5) *flag = 1;
        }
        *out = sum;
        return SVGPU_OK;
    };

    double lambda = 0, last_chi = 0;

    // SparseOptimizer::optimize(iterations) with the terminate_action hook
    auto optimize = [&](int iterations, int* iters_done) -> int {
        *iters_done = 0;
        int r = upload_structure();
        if (r) return r;
        if (!sharded && HS.nP + HS.nL == 0) return SVGPU_OK;
        bool ok = true;
        double ni = 2;
        double current_chi = 0;
        for (int it = 0; it < iterations && ok; ++it) {
            if (!sharded && *flag) break;
            // activeRobustChi2 of the estimate.  After an accepted step it IS the trial's chi2 of the previous iteration (same
            // state, same fixed-order sum), so only the first iteration -- and the sharded solve, whose call also
            // OR-reduces the stop flags -- computes it again.
            if (it == 0 || sharded) {
                if ((r = chi2(0, 0, &current_chi))) return r;
            }
            if (*flag) break;  // sharded: the flag was just OR-reduced, every rank leaves together
            sv_ba_linearize(ctx, s, D);
            if (sharded && HS.nP > 0) {
                SV_HIP(ctx, hipMemcpyAsync(d_HB_full, D.Hpp, 8 * 36 * (size_t)HS.nP, hipMemcpyDeviceToDevice, s));
                SV_HIP(ctx, hipMemcpyAsync(d_HB_full + 36 * (size_t)HS.nP, D.bp, 8 * 6 * (size_t)HS.nP, hipMemcpyDeviceToDevice, s));
                if ((r = allreduce_dev(d_HB_full, 42 * (size_t)HS.nP))) return r;
            }
            if (it == 0) {  // computeLambdaInit
                SV_HIP(ctx, hipMemsetAsync(D.red + D.red_flag_off, 0, 16, s));
                sv_ba_maxdiag(s, D);
                double fl[2];
                SV_HIP(ctx, hipMemcpyAsync(fl, D.red + D.red_flag_off, 16, hipMemcpyDeviceToHost, s));
                SV_HIP(ctx, hipStreamSynchronize(s));
                double max_diag = fl[1];
                if (sharded) {  // max over ranks through a sum of one-hot slots
                    std::vector<double> slots(world, 0.0);
                    slots[rank] = max_diag;
                    if ((r = allreduce_host(slots.data(), world))) return r;
                    for (double v : slots) max_diag = std::max(max_diag, v);
                }
                lambda = 1e-5 * max_diag;
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                D.lambda = lambda;
                D.lambda_diag = (!sharded || rank == 0) ? lambda : 0.0;
                SV_HIP(ctx, hipMemsetAsync(D.red + D.red_flag_off, 0, 8, s));
                sv_ba_reduce(ctx, s, D);
                if (HS.nP > 0 && (r = allreduce_dev(D.S, (size_t)(D.n + 1) * D.n))) return r;
                sv_ba_solve(ctx, s, D);
                if (E > 0) sv_ba_chi2(ctx, s, D, 1, 0);
                SV_HIP(ctx, hipMemcpyAsync(red_host, D.red, 8 * (size_t)red_total, hipMemcpyDeviceToHost, s));
                SV_HIP(ctx, hipStreamSynchronize(s));
                double temp_chi = 0, scale = 0;
                if (E > 0)
                    for (int i = 0; i < nb_chi; ++i) temp_chi += red_host[D.red_chi_off + i];
                for (int i = 0; i < nb_lm + nb_pose; ++i) scale += red_host[D.red_scale_off + i];
                bool ok2 = red_host[D.red_flag_off] == 0.0;
                if (trace) std::fprintf(stderr, "[ba]     chol cycles diag %.0f panel %.0f trail %.0f back %.0f\n", red_host[D.red_flag_off + 2], red_host[D.red_flag_off + 3], red_host[D.red_flag_off + 4], red_host[D.red_flag_off + 5]);
                if (sharded) {
                    double v[4] = {temp_chi, scale, ok2 ? 0.0 : 1.0, (double)(*flag ? 1 : 0)};
                    if ((r = allreduce_host(v, 4))) return r;
                    temp_chi = v[0];
                    scale = v[1];
                    ok2 = v[2] < 0.5;
                    if (v[3] > 0.5) *flag = 1;
                }
                ++st.lm_trials;
                if (!ok2) {
                    temp_chi = DBL_MAX;
                    ++st.cholesky_failures;
                }
                rho = current_chi - temp_chi;
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(temp_chi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2;
                    current_chi = temp_chi;
                    std::swap(D.pose_cur, D.pose_trial);  // accept: the trial state becomes the estimate
                    std::swap(D.pt_cur, D.pt_trial);
                }
                else {
                    lambda *= ni;
                    ni *= 2;
                    if (!std::isfinite(lambda)) break;
                }
                ++qmax;
            } while (rho < 0 && qmax < 10 && !*flag);
            if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) ok = false;
            ++*iters_done;
            // postIteration: terminate_action (chi2 of the current estimate = current_chi)
            if (it == 0) last_chi = current_chi;
            else {
                const double gain = (last_chi - current_chi) / current_chi;
                last_chi = current_chi;
                if (gain >= 0 && gain < pr->gain_threshold) {
                    *flag = 1;
                    st.stopped_by_terminate_action = 1;
                }
            }
        }
        // errors cached by the last computeActiveErrors (used by the gate and the outlier list)
        double dummy;
        if (*iters_done > 0 && (r = chi2(0, 1, &dummy))) return r;
        return SVGPU_OK;
    };

    double chi0 = 0;
    {
        int r = upload_structure();
        if (r) return r;
        if ((r = chi2(0, 1, &chi0))) return r;
    }
    st.chi2_initial = chi0;
    lap("structure + chi2_0");
    if (sharded && stop && *stop) {  // the flag was OR-reduced inside chi2(): every rank returns together
        if (stats) *stats = st;
        return SVGPU_STOPPED;
    }
    int it1 = 0, it2 = 0;
    rc = optimize(pr->num_first_iter, &it1);
    if (rc) return rc;
    st.iters_stage1 = it1;
    lap("stage 1");
    if (sharded) {  // agree on the caller flags before the stage-2 decision
        double v[1] = {(double)((stop && *stop) ? 1 : 0)};
        if ((rc = allreduce_host(v, 1))) return rc;
        if (stop && v[0] > 0.5) *stop = 1;
    }
    bool run_robust = !single_stage;
    if (stop && *stop) run_robust = false;  // :317-321 (only the CALLER's flag is consulted here)
    if (run_robust) {
        st.stage2_entered = 1;
        if (E > 0) {
            sv_ba_gate(s, D, 1, nullptr);
            SV_HIP(ctx, hipMemcpyAsync(level.data(), D.e_level, E, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
        }
        double gated = 0;
        for (int e = 0; e < E; ++e) gated += level[e];
        if (sharded && (rc = allreduce_host(&gated, 1))) return rc;
        st.num_gated = (int32_t)gated;
        rc = optimize(pr->num_second_iter, &it2);
        if (rc) return rc;
        st.iters_stage2 = it2;
        lap("stage 2");
    }
    // ---- outlier list, final chi2, read-back
    std::vector<uint8_t> outl(E);
    if (E > 0 && outlier_out) {
        sv_ba_gate(s, D, 0, d_outlier);
        SV_HIP(ctx, hipMemcpyAsync(outl.data(), d_outlier, E, hipMemcpyDeviceToHost, s));
    }
    SV_HIP(ctx, hipMemcpyAsync(pose_out, D.pose_cur, sizeof(double) * 12 * (size_t)P, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(points_out, D.pt_cur, sizeof(double) * 3 * (size_t)L, hipMemcpyDeviceToHost, s));
    double chi1 = 0;
    {
        int r = chi2(0, 0, &chi1);
        if (r) return r;
    }
    if (sharded) {  // every rank ends with every landmark: owners contribute their points, the rest zeros
        for (int l = 0; l < L; ++l) {
            for (int k = 0; k < 3; ++k) xch_host[3 * (size_t)l + k] = owned[l] ? points_out[3 * (size_t)l + k] : 0.0;
            xch_host[3 * (size_t)L + l] = owned[l];
        }
        H2D(d_xch, xch_host.data(), 8 * 4 * (size_t)L);
        int r = allreduce_dev(d_xch, 4 * (size_t)L);
        if (r) return r;
        SV_HIP(ctx, hipMemcpyAsync(xch_host.data(), d_xch, 8 * 4 * (size_t)L, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        for (int l = 0; l < L; ++l)
            if (xch_host[3 * (size_t)L + l] > 0.5)
                for (int k = 0; k < 3; ++k) points_out[3 * (size_t)l + k] = xch_host[3 * (size_t)l + k];
    }
    if (outlier_out)
        for (int k = 0; k < E; ++k) outlier_out[perm[k]] = outl[k];
    st.chi2_final = chi1;
    lap("read-back");
    st.lambda_final = lambda;
    if (stats) *stats = st;
#undef H2D
    return SVGPU_OK;
}

extern "C" {

int svgpu_local_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out,
                   double* points_out, uint8_t* outlier_out, svgpu_ba_stats* stats) {
    return local_ba_impl(ctx, problem, false, 0, 1, nullptr, nullptr, stop, pose_out, points_out, outlier_out, stats);
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
    const size_t need = pad(24 * (size_t)n) + pad(12 * (size_t)n) + 2 * pad(4 * (size_t)n) + 3 * pad(n) + pad(96) + pad(16) + 1024;
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    PoseOptDev P;
    memset(&P, 0, sizeof(P));
    double* d_pos = A.take<double>(3 * (size_t)n);
    float* d_uvr = A.take<float>(3 * (size_t)n);
    float* d_w = A.take<float>(n);
    float* d_h = A.take<float>(n);
    P.outlier = A.take<uint8_t>(n);
    P.level = A.take<uint8_t>(n);
    P.robust = A.take<uint8_t>(n);
    P.pose_out = A.take<double>(12);
    P.result = A.take<int>(4);
    hipStream_t s = ctx->stream;
    SV_HIP(ctx, hipMemcpyAsync(d_pos, pos_w, 24 * (size_t)n, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_uvr, uvr, 12 * (size_t)n, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_w, inv_sigma_sq, 4 * (size_t)n, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_h, huber_delta, 4 * (size_t)n, hipMemcpyHostToDevice, s));
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
    int result[4] = {0, 0, 0, 0};
    SV_HIP(ctx, hipMemcpyAsync(pose_out, P.pose_out, sizeof(double) * 12, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(outlier_flags, P.outlier, n, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(result, P.result, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    *num_valid = result[0];
    if (lm_iterations) *lm_iterations = result[1];
    return SVGPU_OK;
}

int svgpu_global_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out, double* points_out,
                    svgpu_ba_stats* stats) {
    return local_ba_impl(ctx, problem, true, 0, 1, nullptr, nullptr, stop, pose_out, points_out, nullptr, stats);
}

int svgpu_local_ba_sharded(svgpu_ctx* ctx, const svgpu_ba_problem* shard, int rank, int world, svgpu_allreduce_fn allreduce,
                           void* allreduce_user, volatile uint8_t* stop, double* pose_out, double* points_out,
                           uint8_t* outlier_out, svgpu_ba_stats* stats) {
    if (!allreduce) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_local_ba_sharded: allreduce callback is required");
    return local_ba_impl(ctx, shard, false, rank, world, allreduce, allreduce_user, stop, pose_out, points_out, outlier_out, stats);
}

}  // extern "C"
