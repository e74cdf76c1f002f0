// Exercises the C++ adaptor classes (reference signatures) end to end on the GPU and checks them against the
// C-ABI entry points they wrap.  Built by host/Makefile, run by tests/test_gpu_host_adaptors.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "drop_in/flat_optimizers.h"
#include "matchers.h"
#include "orb_extractor.h"

using namespace stella_vslam_hip;

static cv::Mat synth(int w, int h, unsigned seed) {
    cv::Mat m(h, w, CV_8UC1);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) m.ptr(y)[x] = (uint8_t)(((x + 2 * y) >> 2) & 255);
    unsigned long long s = seed * 2654435761ull + 88172645463325252ull;
    auto rnd = [&]() {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 33);
    };
    for (int k = 0; k < 1500; ++k) {
        const int rw = 4 + rnd() % 37, rh = 4 + rnd() % 37, x0 = rnd() % w, y0 = rnd() % h, g = rnd() % 256;
        for (int y = y0; y < y0 + rh && y < h; ++y)
            for (int x = x0; x < x0 + rw && x < w; ++x) m.ptr(y)[x] = (uint8_t)g;
    }
    return m;
}

#define REQUIRE(c)                                                     \
    do {                                                               \
        if (!(c)) {                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                  \
        }                                                              \
    } while (0)

int main() {
    feature::orb_params params("ORB setting for test");
    feature::orb_extractor ext(&params, 800);
    cv::Mat img = synth(640, 480, 1), img2 = synth(640, 480, 1);
    for (int y = 0; y < 480; ++y)  // second frame: shifted by (3,1)
        for (int x = 0; x < 640; ++x) img2.ptr(y)[x] = img.ptr(std::min(y + 1, 479))[std::min(x + 3, 639)];
    std::vector<cv::KeyPoint> k1, k2;
    cv::Mat d1, d2, nomask;
    ext.extract(img, cv::_InputArray(), k1, d1);
    ext.extract(img2, cv::_InputArray(), k2, d2);
    REQUIRE(k1.size() > 1000 && (int)k1.size() == d1.rows && d1.cols == 32);
    REQUIRE(k2.size() > 1000 && (int)k2.size() == d2.rows);
    ext.sync_image_pyramid();
    REQUIRE(ext.image_pyramid_.size() == 8 && ext.image_pyramid_[1].cols == 533 && ext.image_pyramid_[7].rows == 134);
    REQUIRE(ext.image_pyramid_[0].data == img2.data);  // level 0 aliases the caller's image
    // empty image: silent return (orb_extractor.cc:30-32)
    std::vector<cv::KeyPoint> k0 = k1;
    cv::Mat d0, empty;
    ext.extract(empty, cv::_InputArray(), k0, d0);
    REQUIRE(k0.size() == k1.size());

    data::frame_observation f1, f2;
    f1.descriptors_ = d1;
    f1.undist_keypts_ = k1;
    f2.descriptors_ = d2;
    f2.undist_keypts_ = k2;
    match::robust rm(ext.context(), 0.8f, true);
    std::vector<std::pair<int, int>> matches;
    const unsigned nm = rm.brute_force_match(f2, f1, {}, matches);
    REQUIRE(nm == matches.size() && nm > 500);
    int good = 0;
    for (auto& m : matches) {
        const float dx = k2[m.first].pt.x - k1[m.second].pt.x, dy = k2[m.first].pt.y - k1[m.second].pt.y;
        good += std::fabs(dx + 3) < 2.5f * params.scale_factors_[k1[m.second].octave] && std::fabs(dy + 1) < 2.5f * params.scale_factors_[k1[m.second].octave];
    }
    REQUIRE(good > (int)(0.9 * nm));
    for (size_t i = 1; i < matches.size(); ++i) REQUIRE(matches[i - 1].first < matches[i].first);

    // tiny BA through the class interface: 3 cameras on a line looking at a plane of points
    optimize::flat_ba_problem p;
    const int P = 3, L = 60;
    for (int c = 0; c < P; ++c) {
        const double T[12] = {1, 0, 0, -0.3 * c, 0, 1, 0, 0, 0, 0, 1, 0};
        p.pose_cw.insert(p.pose_cw.end(), T, T + 12);
        p.pose_fixed.push_back(c < 2);
        const double K[5] = {458.654, 458.654, 367.215, 248.375, 0.0};
        p.intrinsics.insert(p.intrinsics.end(), K, K + 5);
    }
    for (int l = 0; l < L; ++l) {
        const double X[3] = {-1.0 + 0.2 * (l % 10), -0.6 + 0.2 * (l / 10), 4.0 + 0.05 * (l % 7)};
        for (int c = 0; c < P; ++c) {
            const double xc = X[0] - 0.3 * c, u = 458.654 * xc / X[2] + 367.215, v = 458.654 * X[1] / X[2] + 248.375;
            p.obs_pose.push_back(c);
            p.obs_point.push_back(l);
            p.obs_uvr.push_back((float)u);
            p.obs_uvr.push_back((float)v);
            p.obs_uvr.push_back(-1.f);
            p.obs_inv_sigma_sq.push_back(1.f);
            p.obs_huber_delta.push_back(std::sqrt(5.99146f));
        }
        p.points.push_back(X[0] + 0.01 * ((l * 7) % 5 - 2));
        p.points.push_back(X[1] - 0.01 * ((l * 3) % 5 - 2));
        p.points.push_back(X[2] + 0.02 * ((l * 5) % 5 - 2));
    }
    p.pose_cw[2 * 12 + 3] += 0.02;  // perturb the free camera
    optimize::local_bundle_adjuster_hip ba(ext.context());
    optimize::flat_ba_result r;
    bool stop = false;
    ba.optimize_flat(p, &stop, r);
    REQUIRE(r.status == SVGPU_OK && r.stats.chi2_final < 1e-3 * r.stats.chi2_initial + 1e-6);
    REQUIRE(std::fabs(r.pose_cw[2 * 12 + 3] + 0.6) < 1e-3);
    stop = true;
    ba.optimize_flat(p, &stop, r);
    REQUIRE(r.status == SVGPU_STOPPED && r.pose_cw == p.pose_cw);
    {   // global BA sibling: one LM run; a raised flag that the gain rule did not raise => "discard" (false)
        optimize::global_bundle_adjuster_hip gba(ext.context(), 10, true);
        optimize::flat_ba_result g;
        bool gstop = false;
        REQUIRE(gba.optimize_flat(p, &gstop, g) && g.stats.chi2_final < 1e-3 * g.stats.chi2_initial + 1e-6);
        REQUIRE(std::fabs(g.pose_cw[2 * 12 + 3] + 0.6) < 1e-3 && g.stats.iters_stage2 == 0);
        const bool by_gain_rule = g.stats.stopped_by_terminate_action != 0;
        REQUIRE(gstop == by_gain_rule);  // terminate_action raises the caller's flag when the gain rule stops the run
        gstop = true;
        REQUIRE(!gba.optimize_flat(p, &gstop, g));
    }

    // projection-family matcher with the candidate lists built on the device: frame-1 keypoints "reprojected" by the known shift
    // into frame 2, window 15 px x scale, levels +-1 (projection.cc:30-35); and the same queries through host-built lists
    {
        match::projection pm(ext.context(), 0.8f, true);
        match::projection::query_set q;
        q.descriptors = d1;
        std::vector<cv::Point2f> ref;
        std::vector<float> margins;
        std::vector<int> lo, hi;
        for (auto& k : k1) {
            q.angle.push_back(k.angle);
            ref.push_back(cv::Point2f{k.pt.x - 3.f, k.pt.y - 1.f});
            margins.push_back(15.f * params.scale_factors_[k.octave]);
            lo.push_back(std::max(0, k.octave - 1));
            hi.push_back(std::min(7, k.octave + 1));
        }
        const float bounds[4] = {0.f, 640.f, 0.f, 480.f};
        std::vector<int> m_cells;
        const unsigned nc = pm.match_in_cells(q, ref, margins, lo, hi, f2, {}, bounds, 64, 48, SVGPU_MATCH_RATIO_SAME_OCTAVE, 100, m_cells);
        REQUIRE(nc > 800 && m_cells.size() == k1.size());
        int consistent = 0;
        for (size_t i = 0; i < m_cells.size(); ++i)
            if (m_cells[i] >= 0) consistent += std::fabs(k2[m_cells[i]].pt.x - ref[i].x) < margins[i] && std::fabs(k2[m_cells[i]].pt.y - ref[i].y) < margins[i];
        REQUIRE(consistent == (int)nc);  // every match lies inside its query's window
        // area matcher (initialiser): level-0 keypoints only, window 60 px, all candidates in index order
        match::projection::query_set qa;
        qa.descriptors = d1;
        qa.cand_off.push_back(0);
        for (auto& k : k1) {
            qa.angle.push_back(k.angle);
            if (k.octave == 0)
                for (size_t j = 0; j < k2.size(); ++j)
                    if (k2[j].octave == 0 && std::fabs(k2[j].pt.x - k.pt.x) < 60.f && std::fabs(k2[j].pt.y - k.pt.y) < 60.f) qa.cand_idx.push_back((int)j);
            qa.cand_off.push_back((int)qa.cand_idx.size());
        }
        std::vector<int> m_area;
        const unsigned na = match::area(ext.context(), 0.9f, true).match_in_consistent_area(qa, f2, m_area);
        REQUIRE(na > 100 && m_area.size() == k1.size());
        std::vector<char> taken(k2.size(), 0);
        for (int t : m_area)
            if (t >= 0) {
                REQUIRE(!taken[t]);  // a target ends with exactly one holder
                taken[t] = 1;
            }
        std::printf("projection in cells: %u matches, area: %u matches\n", nc, na);
    }
    // stereo: right image = left shifted by 20 px -> the recovered disparity is 20
    {
        feature::orb_extractor ext_r(&params, 800);
        cv::Mat right(480, 640, CV_8UC1);
        for (int y = 0; y < 480; ++y)
            for (int x = 0; x < 640; ++x) right.ptr(y)[x] = img.ptr(y)[std::min(x + 20, 639)];
        std::vector<cv::KeyPoint> kl, kr;
        cv::Mat dl, dr;
        ext.extract(img, cv::_InputArray(), kl, dl);
        ext_r.extract(right, cv::_InputArray(), kr, dr);
        std::vector<float> xr, depth;
        match::stereo(&ext, &ext_r, kl, kr, dl, dr, 458.654f * 0.11f, 0.11f).compute(xr, depth);
        REQUIRE(xr.size() == kl.size());
        int ok = 0, tot = 0;
        for (size_t i = 0; i < xr.size(); ++i)
            if (xr[i] >= 0) {
                ++tot;
                ok += std::fabs((kl[i].pt.x - xr[i]) - 20.f) < 0.5f;
            }
        std::printf("stereo: %d of %d matched keypoints at the true disparity\n", ok, tot);
        REQUIRE(tot > 500 && 2 * ok > tot);  // the rectangle texture is self-similar along rows: the majority, not all, lock on
    }
    // frame observation + observability + fused landmark matcher: landmarks placed along the bearings of frame 1's keypoints
    // must reproject onto those keypoints and match them
    {
        camera::perspective cam(ext.context(), 640, 480, 458.654, 457.296, 320.0, 240.0, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0);
        REQUIRE(cam.img_bounds_.min_x_ < 0.f && cam.img_bounds_.max_x_ > 640.f);
        std::vector<cv::KeyPoint> und;
        std::vector<Vec3_t> brg, brg2;
        std::vector<int> cell_off, cell_items;
        cam.observe(k1, 64, 48, und, brg, cell_off, cell_items);
        REQUIRE(und.size() == k1.size() && brg.size() == k1.size() && cell_off.size() == 64 * 48 + 1 && (size_t)cell_off.back() == cell_items.size());
        REQUIRE(cell_items.size() > k1.size() / 2);
        cam.convert_keypoints_to_bearings(und, brg2);
        REQUIRE(std::memcmp(brg.data(), brg2.data(), brg.size() * sizeof(Vec3_t)) == 0);
        std::vector<cv::KeyPoint> und2;
        cam.undistort_keypoints(k1, und2);
        REQUIRE(std::memcmp(und.data(), und2.data(), und.size() * sizeof(cv::KeyPoint)) == 0);
        camera::landmark_set lms;
        const Mat33_t R = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const Vec3_t t = {0.1, -0.05, 0.2}, twc = {-0.1, 0.05, -0.2};
        const int n = (int)k1.size();
        lms.descriptors.create(n, 32, CV_8U);
        for (int i = 0; i < n; ++i) {
            const double depth = 3.0 + (i % 50) * 0.1;
            const Vec3_t pc = {brg[i][0] / brg[i][2] * depth, brg[i][1] / brg[i][2] * depth, depth};
            const Vec3_t pw = {pc[0] - t[0], pc[1] - t[1], pc[2] - t[2]};
            const double d = std::sqrt((pw[0] - twc[0]) * (pw[0] - twc[0]) + (pw[1] - twc[1]) * (pw[1] - twc[1]) + (pw[2] - twc[2]) * (pw[2] - twc[2]));
            lms.pos_w.push_back(pw);
            lms.mean_normal.push_back(Vec3_t{(pw[0] - twc[0]) / d, (pw[1] - twc[1]) / d, (pw[2] - twc[2]) / d});
            const float mx = (float)(d * 1.1 * params.scale_factors_[und[i].octave]);
            lms.max_valid_dist.push_back(mx);
            lms.min_valid_dist.push_back(mx / params.scale_factors_[7]);
            std::memcpy(lms.descriptors.ptr(i), d1.ptr(i), 32);
        }
        camera::observability ob;
        cam.can_observe(R, t, twc, lms, 0.5f, params.num_levels_, params.log_scale_factor_, ob);
        int vis = 0, close = 0;
        for (int i = 0; i < n; ++i)
            if (ob.visible[i]) {
                ++vis;
                close += std::fabs(ob.reproj[i][0] - und[i].pt.x) < 1e-3 && std::fabs(ob.reproj[i][1] - und[i].pt.y) < 1e-3 && ob.pred_scale_level[i] == std::min(7, und[i].octave + 1);
            }
        REQUIRE(vis > n * 9 / 10 && close == vis);
        data::frame_observation fo;
        fo.descriptors_ = d1;
        fo.undist_keypts_ = und;
        std::vector<int> mi;
        camera::observability ob2;
        const unsigned nmf = match::projection(ext.context(), 0.8f, true).match_frame_and_landmarks(cam, R, t, twc, lms, fo, {}, params, 64, 48, 5.0f, mi, ob2);
        int self = 0;
        for (int i = 0; i < n; ++i) self += mi[i] == i;
        REQUIRE(nmf > (unsigned)vis * 8 / 10 && (unsigned)self > nmf * 95 / 100 && ob2.visible == ob.visible);
        // landmark refresh: three observations per landmark = its keypoint's descriptor with 0 / 1 / 2 bits flipped -> row 1 (one
        // flip away from both others) has the smallest median? medians: row0 {0,1,2}->1, row1 {0,1,1}->1, row2 {0,1,2}->1: first wins
        {
            std::vector<int> off(1, 0), best;
            cv::Mat od(3 * 100, 32, CV_8U), rep;
            std::vector<Vec3_t> cams, pos, refc, mnrm;
            std::vector<float> rsf, mxd, mnd;
            for (int l = 0; l < 100; ++l) {
                for (int o = 0; o < 3; ++o) {
                    std::memcpy(od.ptr(3 * l + o), d1.ptr(l), 32);
                    if (o >= 1) od.ptr(3 * l + o)[0] ^= 1;
                    if (o == 2) od.ptr(3 * l + o)[5] ^= 4;
                    cams.push_back(Vec3_t{(double)o, 0.0, 0.0});
                }
                off.push_back(3 * (l + 1));
                pos.push_back(Vec3_t{1.0, 0.0, 4.0 + 0.01 * l});
                refc.push_back(Vec3_t{1.0, 0.0, 0.0});
                rsf.push_back(params.scale_factors_[2]);
            }
            data::compute_descriptors(ext.context(), off, od, best, rep);
            data::update_mean_normal_and_obs_scale_variance(ext.context(), off, cams, pos, refc, rsf, params.inv_scale_factors_[7], mnrm, mxd, mnd);
            bool okl = rep.rows == 100;
            for (int l = 0; l < 100 && okl; ++l)
                okl = best[l] == 0 && std::memcmp(rep.ptr(l), d1.ptr(l), 32) == 0 && std::fabs(mnrm[l][0]) < 1e-12 && mnrm[l][2] > 0.99
                      && mxd[l] == (float)((4.0 + 0.01 * l) * params.scale_factors_[2]) && mnd[l] == mxd[l] * params.inv_scale_factors_[7];
            REQUIRE(okl);
        }
        {   // BoW: a two-level vocabulary whose 2 x 3 leaves are keypoint descriptors; every descriptor used as a leaf must come back
            // as that leaf's word, through the inner node it hangs under
            std::vector<int> coff = {0, 2, 5, 8, 8, 8, 8, 8, 8, 8}, ch = {1, 2, 3, 4, 5, 6, 7, 8}, wid = {-1, -1, -1, 0, 1, 2, 3, 4, 5};
            std::vector<float> ww = {0, 0, 0, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
            cv::Mat nd(9, 32, CV_8U);
            std::memset(nd.ptr(0), 0, 9 * 32);
            const int leaf_kp[6] = {0, 10, 20, 30, 40, 50};
            for (int l = 0; l < 6; ++l) std::memcpy(nd.ptr(3 + l), d1.ptr(leaf_kp[l]), 32);
            std::memcpy(nd.ptr(1), d1.ptr(10), 32);  // inner nodes: one of their leaves
            std::memcpy(nd.ptr(2), d1.ptr(40), 32);
            data::bow_vocabulary_hip voc(ext.context(), coff, ch, nd, ww, wid, 2);
            cv::Mat qd(6, 32, CV_8U);
            for (int l = 0; l < 6; ++l) std::memcpy(qd.ptr(l), d1.ptr(leaf_kp[l]), 32);
            std::map<unsigned int, double> bv;
            std::map<unsigned int, std::vector<unsigned int>> fv;
            voc.compute_bow(qd, bv, fv, 1);
            double sum = 0;
            for (auto& kv : bv) sum += kv.second;
            size_t nfeat = 0;
            for (auto& kv : fv) nfeat += kv.second.size();
            REQUIRE(!bv.empty() && std::fabs(sum - 1.0) < 1e-12 && nfeat == 6 && fv.size() <= 2);
        }
        std::printf("frame observation: %d keypoints, %d landmarks visible, %u matched (%d onto their own keypoint)\n", n, vis, nmf, self);
    }
    // motion-only BA: perturbed camera 2 against the 60 exact observations of the scene above
    {
        std::vector<double> pos_w;
        std::vector<float> uvr, w, hub;
        for (int l = 0; l < L; ++l) {
            const double X[3] = {-1.0 + 0.2 * (l % 10), -0.6 + 0.2 * (l / 10), 4.0 + 0.05 * (l % 7)};
            pos_w.insert(pos_w.end(), X, X + 3);
            uvr.push_back((float)(458.654 * (X[0] - 0.6) / X[2] + 367.215));
            uvr.push_back((float)(458.654 * X[1] / X[2] + 248.375));
            uvr.push_back(-1.f);
            w.push_back(1.f);
            hub.push_back(std::sqrt(5.99146f));
        }
        const double T0[12] = {1, 0, 0, -0.57, 0, 1, 0, 0.01, 0, 0, 1, -0.02}, K[5] = {458.654, 458.654, 367.215, 248.375, 0.0};
        double T1[12];
        std::vector<uint8_t> flags;
        const unsigned good_obs = optimize::pose_optimizer_hip(ext.context()).optimize_flat(T0, pos_w, uvr, w, hub, K, T1, flags);
        REQUIRE(good_obs == (unsigned)L && std::fabs(T1[3] + 0.6) < 1e-4 && std::fabs(T1[7]) < 1e-4 && std::fabs(T1[11]) < 1e-4);
        std::printf("pose optimizer: %u inliers, t = (%.5f %.5f %.5f)\n", good_obs, T1[3], T1[7], T1[11]);
    }
    std::printf("adaptors ok: %zu / %zu keypoints, %u matches (%d consistent), BA chi2 %.3g -> %.3g\n", k1.size(), k2.size(), nm, good,
                r.stats.chi2_initial, r.stats.chi2_final);
    return 0;
}
