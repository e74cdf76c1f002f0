// Kernels of the tracked-frame chain (svgpu_track.hip): the steps tracking_module runs per image, fused so that each half of the chain
// is a handful of launches on ONE stream with no host round trip in between, and fed from the resident landmark table (svgpu_map)
// instead of host-flattened arrays.
//   k_track_frame   system.cc:384-395   undistort_keypoints + convert_keypoints_to_bearings + assign_keypoints_to_grid (keypoint count
//                                       read from device memory)
//   k_track_cand    data/frame.cc:59-85 (can_observe) / camera::*::reproject_to_image, data/common.cc:127-190 (get_keypoints_in_cell),
//                   match/projection.cc:40-86, 160-195 (gates + Hamming distances): one wave per query
// The sequential part (greedy claims) stays k_cand_replay_lds (match_kernels.hip), the optimisation k_pose_opt<EQ, true> (ba_kernels.hip).
// Every expression is the one the separate kernels evaluate (frame_device.h / match_device.h): the chain's results are bit-identical to
// svgpu_frame_observation + svgpu_match_current_and_last_frames / svgpu_reproject_landmarks + svgpu_match_in_cells on flattened inputs.
#include "svgpu_internal.h"
#include "track_kernels.h"
#include "frame_device.h"
#include "match_device.h"

namespace {
using namespace svfd;
using namespace svmd;

#define TRACK_FRAME_ROUNDS GRID_ONE_KPT_ROUNDS  // keypoints per thread of the one-workgroup frame kernel (8 192 keypoints)

__global__ __launch_bounds__(1024) void k_track_frame(TrackFrameProblem P) {
    const int tid = threadIdx.x;
    const int n = min(*P.n_dev, P.cap);
    if (tid == 0) {
        *P.n_host = n;
        if (P.counter_reset) *P.counter_reset = 0;
    }
#pragma unroll 1
    for (int i = tid; i < n; i += 1024) {
        ObsOut o;
        frame_obs_one(P.cam, P.kps[i], false, o);
        P.undist[i] = o.undist;
        P.xy[2 * i] = o.ux;
        P.xy[2 * i + 1] = o.uy;
        P.octave[i] = o.undist.octave;
        P.angle[i] = o.undist.angle;
        P.bearings[3 * i] = o.b0;
        P.bearings[3 * i + 1] = o.b1;
        P.bearings[3 * i + 2] = o.b2;
        if (P.depth_img) {  // system.cc:498-510: img_depth.at<float>(y, x) with the keypoint's float coordinates truncated to int
            const svgpu_keypoint kp = P.kps[i];
            const float depth = P.depth_img[(size_t)(int)kp.y * P.depth_pitch + (int)kp.x];
            float xr = -1.f, dp = -1.f;
            if (!(depth <= 0)) {
                dp = depth;
                xr = (float)((double)o.ux - P.focal_x_baseline / (double)depth);  // float - double / float: evaluated in double, stored as float
            }
            P.xright[i] = xr;
            P.depth_out[i] = dp;
        }
    }
    __syncthreads();  // the grid reads xy / octave back (workgroup scope: the only workgroup)
    grid_frame_one(P.G, n);
}

__device__ __forceinline__ void wave_lds_sync() {  // orders the wave's own LDS writes before its later LDS reads (no workgroup barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define TRACK_SORT_MAX 1024  // == CAND_SORT_MAX of match_kernels.hip: lists up to this many entries are left sorted by (distance, scan position)
#define TRACK_CACHE 4        // passing keypoints of a cell kept in registers between the counting and the distance loop

// List placement.  A list of at most TRACK_SLOT entries lives in its query's own slot (dist[q * TRACK_SLOT ...]): no allocation, no shared
// counter.  Longer lists (wide windows at the coarse levels: a few per frame) are appended behind the slots through ONE atomic counter
// (cand_off[nq], counting overflow entries only).  (First form: every list allocated from the counter after a counting walk -- 66 us per
// launch at 2 400 - 4 800 queries, whatever their number: thousands of same-address atomics and two dependent walks per query.)
template <int MODE>
__global__ __launch_bounds__(256) void k_track_cand(TrackCandProblem P) {
    __shared__ unsigned long long s_keys[4][TRACK_SORT_MAX];
    const svgpu_landmark_record* __restrict__ map = (const svgpu_landmark_record*)P.map;
    // which keypoints are closed from the start: `lm && lm->has_observation()` of the landmark the frame holds there (projection.cc:52-55)
    if (MODE == 1 && P.cur_lm) {
        const int k = blockIdx.x * 256 + threadIdx.x, nt = P.nt_dev ? min(*P.nt_dev, P.nt) : P.nt;
        if (k < nt) {
            const int id = P.cur_lm[k];
            P.occupied[k] = (id >= 0 && id < P.map_cap && (map[id].flags & SVGPU_LM_HAS_OBSERVATION)) ? 1 : 0;
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= P.nq) return;
    unsigned long long* s_key = s_keys[wave];
    // ---- the landmark of this query
    const int id = P.q_ids[q];
    const uint32_t flags = (id >= 0 && id < P.map_cap) ? map[id].flags : 0u;
    const bool present = (flags & SVGPU_LM_PRESENT) != 0, has_desc = (flags & SVGPU_LM_HAS_DESCRIPTOR) != 0;
    const svgpu_landmark_record* rec = map + (present ? id : 0);
    // ---- pose: kernel argument, or what the previous optimisation left on the device (trans_wc = -R^T t as Eigen evaluates it)
    double rot[9], tcw[3], twc[3];
    if (P.pose_dev) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) rot[3 * i + j] = P.pose_dev[4 * i + j];
            tcw[i] = P.pose_dev[4 * i + 3];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) twc[i] = ((-rot[i]) * tcw[0] + (-rot[3 + i]) * tcw[1]) + (-rot[6 + i]) * tcw[2];
    }
    else {
#pragma unroll
        for (int i = 0; i < 9; ++i) rot[i] = P.R.rot_cw[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tcw[i] = P.R.trans_cw[i], twc[i] = P.R.trans_wc[i];
    }
    ReprojOut o;
    const bool offered = MODE == 0 ? (present && has_desc) : present;
    reproject_one(P.R, rot, tcw, twc, offered, rec->pos_w[0], rec->pos_w[1], rec->pos_w[2], rec->mean_normal[0], rec->mean_normal[1], rec->mean_normal[2],
                  rec->min_valid_dist, rec->max_valid_dist, MODE == 0 ? P.q_octave[q] : 0, MODE == 0, o);
    const bool live = MODE == 0 ? o.vis : (o.vis && has_desc);
    if (lane == 0) {
        P.q_valid[q] = live ? 1 : 0;
        P.q_blocks[q] = (flags & SVGPU_LM_HAS_OBSERVATION) ? 1 : 0;
        if (MODE == 1) {
            P.visible[q] = o.vis ? 1 : 0;
            if (P.visible_host) P.visible_host[q] = o.vis ? 1 : 0;
            P.reproj[2 * q] = o.vis ? o.rx : 0.0;
            P.reproj[2 * q + 1] = o.vis ? o.ry : 0.0;
            P.x_right[q] = o.vis ? o.xr : 0.f;
            P.pred_level[q] = o.vis ? o.level : -1;
        }
    }
    int total = 0, list_off = q * TRACK_SLOT;
    if (live) {
        float ref_x, ref_y, margin;
        int min_level, max_level;
        reproject_window(P.R, o, ref_x, ref_y, margin, min_level, max_level);
        uint32_t qd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qd[k] = reinterpret_cast<const uint32_t*>(rec->descriptor)[k];
        const float q_angle = (MODE == 0 && P.check_orientation) ? P.q_angle[q] : 0.f;
        // ---- data::get_keypoints_in_cell (data/common.cc:127-190): the window's cells in column-major order over the lanes
        int lo_x = (int)floor((double)(ref_x - P.min_x - margin) * P.inv_w), hi_x = (int)ceil((double)(ref_x - P.min_x + margin) * P.inv_w);
        int lo_y = (int)floor((double)(ref_y - P.min_y - margin) * P.inv_h), hi_y = (int)ceil((double)(ref_y - P.min_y + margin) * P.inv_h);
        lo_x = max(lo_x, 0);
        lo_y = max(lo_y, 0);
        hi_x = min(hi_x, P.cols - 1);
        hi_y = min(hi_y, P.rows - 1);
        if (lo_x < P.cols && 0 <= hi_x && lo_y < P.rows && 0 <= hi_y && lo_x <= hi_x && lo_y <= hi_y) {
            const int ny = hi_y - lo_y + 1, ncell = (hi_x - lo_x + 1) * ny;
            // (mode 1: a keypoint that holds a landmark with observations is skipped by EVERY query's scan, projection.cc:52-55 -- it never
            //  enters a list.  Listed and left to the replay's owner table, 60 % of a tracked frame's targets were dead entries in front
            //  of the live ones, and the replay walked past them in every sweep.)
            auto passes = [&](int idx) {
                const int oct = P.t_octave[idx];
                if (oct < min_level || max_level < oct) return false;
                if (MODE == 1 && P.cur_lm) {
                    const int held = P.cur_lm[idx];
                    if (held >= 0 && held < P.map_cap && (map[held].flags & SVGPU_LM_HAS_OBSERVATION)) return false;
                }
                const float dx = P.t_xy[2 * idx] - ref_x, dy = P.t_xy[2 * idx + 1] - ref_y;
                return fabsf(dx) < margin && fabsf(dy) < margin;
            };
            // gates of projection.cc:52-62, 172-181 + the Hamming distance; 0xFFFFFFFF = the candidate can be left out of the list.  That
            // covers the reference's own `continue`s AND distances that cannot take part in any verdict: as a best only d <= thr counts
            // (:81, :198), as a second only one that can fail the ratio test of an acceptable best, lowe_ratio * d < thr (:82) -- unrelated
            // descriptors sit around 128 bits, so nine in ten candidates of a window end here instead of in a list the replay walks
            auto entry = [&](int t) -> uint32_t {
                if (P.t_xright && 0.f < P.t_xright[t]) {
                    const float err = fabsf(o.xr - P.t_xright[t]);
                    if (margin < err) return 0xFFFFFFFFu;
                }
                if (MODE == 0 && P.check_orientation && fabsf(angle_diff(q_angle, P.t_angle[t])) > 30.0f) return 0xFFFFFFFFu;
                const unsigned d = hamming256(qd, P.tdesc + (size_t)t * 8);
                if (P.thr < d && (MODE == 0 || !(P.lowe_ratio * (float)d < (float)P.thr))) return 0xFFFFFFFFu;
                return (d << 22) | (uint32_t)t;
            };
            // `sink(pos, e)` receives every listed candidate at its position in the reference's scan order
            auto walk = [&](auto&& sink) {
                int done = 0;
                for (int base = 0; base < ncell; base += 64) {
                    const int k = base + lane;
                    int n = 0, c = 0;
                    uint32_t cached[TRACK_CACHE];
                    if (k < ncell) {
                        c = (lo_x + k / ny) * P.rows + lo_y + k % ny;
                        for (int it = P.cell_off[c]; it < P.cell_off[c + 1]; ++it) {
                            const int idx = P.cell_items[it];
                            if (!passes(idx)) continue;
                            const uint32_t e = entry(idx);
                            if (e == 0xFFFFFFFFu) continue;
#pragma unroll
                            for (int u = 0; u < TRACK_CACHE; ++u)
                                if (n == u) cached[u] = e;
                            ++n;
                        }
                    }
                    // inclusive prefix sum over the lanes: DPP row scans
                    int incl = n;
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);  // row_bcast:15
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);  // row_bcast:31
                    int pos = done + incl - n;
                    if (n > 0 && n <= TRACK_CACHE) {
#pragma unroll
                        for (int u = 0; u < TRACK_CACHE; ++u)
                            if (u < n) sink(pos + u, cached[u]);
                    }
                    else if (n > TRACK_CACHE) {  // a crowded cell: walk it again
                        for (int it = P.cell_off[c]; it < P.cell_off[c + 1]; ++it) {
                            const int t = P.cell_items[it];
                            if (!passes(t)) continue;
                            const uint32_t e = entry(t);
                            if (e != 0xFFFFFFFFu) sink(pos++, e);
                        }
                    }
                    done += __builtin_amdgcn_readlane(incl, 63);
                }
                return done;
            };
            // ONE walk: entries go to the wave's LDS strip while they fit it
            total = walk([&](int pos, uint32_t e) {
                if (pos < TRACK_SORT_MAX) s_key[pos] = ((unsigned long long)(e >> 22) << 32) | ((unsigned long long)pos << 22) | (e & 0x3FFFFFu);
            });
            bool fits = true;
            if (total > TRACK_SLOT) {  // beyond the slot: appended behind the slots
                int off = 0;
                if (lane == 0) off = atomicAdd(P.counter, total);
                off = __builtin_amdgcn_readfirstlane(off);
                fits = off + total <= P.cap;  // beyond the capacity nothing is written: the host re-runs the chain with a larger one
                list_off = P.nq * TRACK_SLOT + off;
            }
            wave_lds_sync();
            if (fits && total <= 64) {  // (distance, scan position) order: gated entries last (k_cand_dist's order, which the replay's walks rely on)
                unsigned long long key = lane < total ? s_key[lane] : ~0ull;
#pragma unroll
                for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const unsigned long long other = __shfl_xor(key, j, 64);
                        const bool up = (lane & k) == 0, low = (lane & j) == 0;
                        const bool take_min = up == low;
                        key = take_min ? (key < other ? key : other) : (key < other ? other : key);
                    }
                if (lane < total) P.dist[list_off + lane] = key == ~0ull ? 0xFFFFFFFFu : ((uint32_t)(key >> 32) << 22) | (uint32_t)(key & 0x3FFFFFu);
            }
            else if (fits && total <= TRACK_SORT_MAX) {
                int npow2 = 128;
                while (npow2 < total) npow2 <<= 1;
                for (int i = total + lane; i < npow2; i += 64) s_key[i] = ~0ull;
                wave_lds_sync();
                for (int k = 2; k <= npow2; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < npow2; i += 64) {
                            const int p = i ^ j;
                            if (p > i) {
                                const unsigned long long a = s_key[i], b = s_key[p];
                                if ((a > b) == ((i & k) == 0)) {
                                    s_key[i] = b;
                                    s_key[p] = a;
                                }
                            }
                        }
                        wave_lds_sync();
                    }
                for (int i = lane; i < total; i += 64) {
                    const unsigned long long key = s_key[i];
                    P.dist[list_off + i] = key == ~0ull ? 0xFFFFFFFFu : ((uint32_t)(key >> 32) << 22) | (uint32_t)(key & 0x3FFFFFu);
                }
            }
            else if (fits)  // longer than the sorted form goes: a second walk writes the list as it is scanned (the replay walks such lists in full)
                walk([&](int pos, uint32_t e) { P.dist[list_off + pos] = e; });
        }
    }
    if (lane == 0) {
        P.cand_off[q] = list_off;
        P.cand_cnt[q] = total;
    }
}

}  // namespace

void sv_launch_track_frame(svgpu_ctx* ctx, hipStream_t s, const TrackFrameProblem& P) {
    SvProfScope ps(ctx, s, "k_track_frame");
    hipLaunchKernelGGL(k_track_frame, dim3(1), dim3(1024), 0, s, P);
}
void sv_launch_track_cand(svgpu_ctx* ctx, hipStream_t s, const TrackCandProblem& P) {
    SvProfScope ps(ctx, s, "k_track_cand");
    int blocks = (P.nq + 3) / 4;
    if (P.mode == 1 && P.cur_lm) blocks = std::max(blocks, (P.nt + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (P.mode == 0) hipLaunchKernelGGL(k_track_cand<0>, dim3(blocks), dim3(256), 0, s, P);
    else hipLaunchKernelGGL(k_track_cand<1>, dim3(blocks), dim3(256), 0, s, P);
}
