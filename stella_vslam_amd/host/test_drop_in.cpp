// Drives the drop-in classes (host/drop_in/hip_backend.h: the reference's own signatures) on a toy map built from the stand-in
// data:: classes: keyframes on an arc, landmarks with observations / descriptors / BoW nodes, clutter keypoints.  The checks are
// against the map's ground truth (which keypoint really observes which landmark) and against the object-graph side effects the
// reference's methods have (frm.add_landmark, duplicated / new-connection maps, erased outlier observations, refreshed landmark
// geometry).  The arithmetic itself is pinned by the Python parity tests through the C ABI; this test is about the flattening
// and the replay.  Built by host/Makefile, run by tests/test_gpu_host_adaptors.py.
#include <cmath>
#include <cstdio>
#include <random>

#include "drop_in/global_bundle_adjuster_hip.h"
#include "drop_in/hip_backend.h"
#include "drop_in/pose_optimizer_hip.h"

using namespace stella_vslam;
using lm_ptr = std::shared_ptr<data::landmark>;
using kf_ptr = std::shared_ptr<data::keyframe>;

#define REQUIRE(c)                                                              \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

static Mat44_t look_at(const double c[3], const double target[3]) {
    double z[3] = {target[0] - c[0], target[1] - c[1], target[2] - c[2]};
    const double nz = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    for (double& v : z) v /= nz;
    double x[3] = {z[2], 0.0, -z[0]};  // cross((0,1,0), z)
    const double nx = std::sqrt(x[0] * x[0] + x[2] * x[2]);
    for (double& v : x) v /= nx;
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    Mat44_t T = Mat44_t::Identity();
    for (int j = 0; j < 3; ++j) {
        T(0, j) = x[j];
        T(1, j) = y[j];
        T(2, j) = z[j];
    }
    for (int i = 0; i < 3; ++i) T(i, 3) = -(T(i, 0) * c[0] + T(i, 1) * c[1] + T(i, 2) * c[2]);
    return T;
}

struct toy_map {
    camera::perspective cam{camera::setup_type_t::Monocular, 752, 480, 458.654, 457.296, 367.215, 248.375, 0, 0, 0, 0, 0};
    feature::orb_params orb;
    data::map_database db;
    std::vector<kf_ptr> kfs;
    std::vector<lm_ptr> lms;
    std::vector<std::vector<int>> truth;  // truth[kf][keypoint] = landmark id or -1
    std::vector<Mat44_t> pose_gt;
    std::vector<Vec3_t> pos_gt;
};

static void project(const toy_map& M, const Mat44_t& T, const Vec3_t& p, double& u, double& v, double& z) {
    const double X = T(0, 0) * p(0) + T(0, 1) * p(1) + T(0, 2) * p(2) + T(0, 3), Y = T(1, 0) * p(0) + T(1, 1) * p(1) + T(1, 2) * p(2) + T(1, 3);
    z = T(2, 0) * p(0) + T(2, 1) * p(1) + T(2, 2) * p(2) + T(2, 3);
    u = M.cam.fx_ * X / z + M.cam.cx_;
    v = M.cam.fy_ * Y / z + M.cam.cy_;
}

static void build(toy_map& M, int n_kf, int n_lm, int n_clutter, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    M.cam.img_bounds_ = camera::image_bounds{0.f, 752.f, 0.f, 480.f};
    std::vector<std::array<uint8_t, 32>> lm_desc(n_lm);
    for (int l = 0; l < n_lm; ++l) {
        Vec3_t p;
        p(0) = -3.0 + 6.0 * U(rng), p(1) = -1.8 + 3.6 * U(rng), p(2) = 4.5 + 4.0 * U(rng);
        M.pos_gt.push_back(p);
        M.lms.push_back(std::make_shared<data::landmark>((unsigned)l, p));
        for (auto& b : lm_desc[l]) b = (uint8_t)(rng() & 255);
    }
    const double target[3] = {0, 0, 6.5};
    for (int k = 0; k < n_kf; ++k) {
        const double a = -0.25 + 0.5 * k / std::max(1, n_kf - 1);
        const double c[3] = {3.0 * std::sin(a), 0.05 * k, 3.0 - 3.0 * std::cos(a)};
        auto kf = std::make_shared<data::keyframe>((unsigned)k, &M.cam, &M.orb);
        const Mat44_t T = look_at(c, target);
        kf->set_pose_cw(T);
        M.pose_gt.push_back(T);
        std::vector<cv::KeyPoint> kps;
        std::vector<std::array<uint8_t, 32>> descs;
        std::vector<int> truth;
        for (int l = 0; l < n_lm; ++l) {
            double u, v, z;
            project(M, T, M.pos_gt[l], u, v, z);
            if (z < 0.5 || u < 8 || u > 744 || v < 8 || v > 472 || U(rng) > 0.8) continue;
            cv::KeyPoint kp;
            const double dist = std::sqrt(std::pow(M.pos_gt[l](0) - c[0], 2) + std::pow(M.pos_gt[l](1) - c[1], 2) + std::pow(M.pos_gt[l](2) - c[2], 2));
            kp.octave = std::min(7, std::max(0, (int)std::lround(std::log(9.0 / dist) / std::log(1.2)) + 1));
            kp.pt.x = (float)(u + 0.5 * N(rng) * M.orb.scale_factors_[kp.octave]);
            kp.pt.y = (float)(v + 0.5 * N(rng) * M.orb.scale_factors_[kp.octave]);
            kp.angle = (float)std::fmod(360.0 + 7.0 * l + 3.0 * N(rng), 360.0);
            auto d = lm_desc[l];
            for (int f = (int)(rng() % 24); f > 0; --f) {
                const unsigned bit = rng() & 255;
                d[bit >> 3] ^= (uint8_t)(1u << (bit & 7));
            }
            kps.push_back(kp);
            descs.push_back(d);
            truth.push_back(l);
        }
        for (int e = 0; e < n_clutter; ++e) {
            cv::KeyPoint kp;
            kp.pt.x = (float)(752 * U(rng)), kp.pt.y = (float)(480 * U(rng));
            kp.octave = (int)(rng() % 8), kp.angle = (float)(360 * U(rng));
            std::array<uint8_t, 32> d;
            for (auto& b : d) b = (uint8_t)(rng() & 255);
            kps.push_back(kp);
            descs.push_back(d);
            truth.push_back(-1);
        }
        const int n = (int)kps.size();
        kf->frm_obs_.undist_keypts_ = kps;
        kf->frm_obs_.descriptors_.create(n, 32, CV_8U);
        kf->frm_obs_.bearings_.resize(n);
        kf->landmarks_.assign(n, nullptr);
        for (int i = 0; i < n; ++i) {
            std::memcpy(kf->frm_obs_.descriptors_.ptr(i), descs[i].data(), 32);
            const double x = (kps[i].pt.x - M.cam.cx_) / M.cam.fx_, y = (kps[i].pt.y - M.cam.cy_) / M.cam.fy_, l2 = std::sqrt(x * x + y * y + 1.0);
            kf->frm_obs_.bearings_[i](0) = x / l2, kf->frm_obs_.bearings_[i](1) = y / l2, kf->frm_obs_.bearings_[i](2) = 1.0 / l2;
            const int node = truth[i] >= 0 ? truth[i] % 60 : (int)(rng() % 60);
            kf->bow_feat_vec_[(unsigned)node].push_back((unsigned)i);
            if (truth[i] >= 0) {
                kf->landmarks_[i] = M.lms[truth[i]];
                M.lms[truth[i]]->add_observation(kf, (unsigned)i);
                if (M.lms[truth[i]]->ref_keyfrm_.expired()) M.lms[truth[i]]->ref_keyfrm_ = kf;
            }
        }
        M.kfs.push_back(kf);
        M.truth.push_back(truth);
    }
    for (auto& lm : M.lms) {
        lm->compute_descriptor();
        lm->update_mean_normal_and_obs_scale_variance();
    }
}

static data::frame frame_like(toy_map& M, int k, bool with_pose) {
    data::frame f(1000 + k, &M.cam, &M.orb);
    f.frm_obs_ = M.kfs[k]->frm_obs_;
    f.bow_feat_vec_ = M.kfs[k]->bow_feat_vec_;
    f.landmarks_.assign(f.frm_obs_.undist_keypts_.size(), nullptr);
    if (with_pose) f.set_pose_cw(M.kfs[k]->get_pose_cw());
    return f;
}

int main() {
    toy_map M;
    build(M, 8, 900, 300, 11);
    const int A = 2, B = 3;
    int with_lm_B = 0;
    for (int t : M.truth[B]) with_lm_B += t >= 0;
    REQUIRE(with_lm_B > 300);

    {  // bow_tree::match_frame_and_keyframe: a frame with keyframe B's keypoints gets B's landmarks from keyframe A where both see them
        data::frame frm = frame_like(M, B, false);
        std::vector<lm_ptr> matched;
        const unsigned num = match::hip::bow_tree(0.75, true).match_frame_and_keyframe(M.kfs[A], frm, matched);
        REQUIRE(matched.size() == frm.frm_obs_.undist_keypts_.size());
        unsigned right = 0, wrong = 0;
        for (size_t i = 0; i < matched.size(); ++i)
            if (matched[i]) ((int)matched[i]->id_ == M.truth[B][i] ? right : wrong)++;
        REQUIRE(num == right + wrong && right > 150 && wrong * 20 < right);
    }
    {  // projection::match_current_and_last_frames: landmarks of "last frame" A land on the right keypoints of "current frame" B
        data::frame last = frame_like(M, A, true), curr = frame_like(M, B, true);
        last.landmarks_ = M.kfs[A]->get_landmarks();
        const unsigned num = match::hip::projection(0.9, true).match_current_and_last_frames(curr, last, 15.0f);
        unsigned right = 0, wrong = 0;
        for (size_t i = 0; i < curr.landmarks_.size(); ++i)
            if (curr.landmarks_[i]) ((int)curr.landmarks_[i]->id_ == M.truth[B][i] ? right : wrong)++;
        REQUIRE(num >= right + wrong && right > 200 && wrong * 20 < right);
        // projection::match_frame_and_landmarks on the not yet matched landmarks, with the caller-side reprojection maps
        std::vector<lm_ptr> local;
        eigen_alloc_unord_map<unsigned int, Vec2_t> lm_to_reproj;
        std::unordered_map<unsigned int, float> lm_to_x_right;
        std::unordered_map<unsigned int, unsigned int> lm_to_scale;
        std::set<unsigned> have;
        for (const auto& lm : curr.landmarks_)
            if (lm) have.insert(lm->id_);
        const Mat44_t T = curr.get_pose_cw();
        for (const auto& lm : M.lms) {
            if (have.count(lm->id_)) continue;
            double u, v, z;
            project(M, T, lm->get_pos_in_world(), u, v, z);
            if (z <= 0 || u <= 0 || u >= 752 || v <= 0 || v >= 480) continue;
            Vec2_t r;
            r(0) = u, r(1) = v;
            lm_to_reproj[lm->id_] = r;
            lm_to_x_right[lm->id_] = -1.f;
            const Vec3_t p = lm->get_pos_in_world();
            const double d = std::sqrt(std::pow(p(0) + (T(0, 0) * T(0, 3) + T(1, 0) * T(1, 3) + T(2, 0) * T(2, 3)), 2) + std::pow(p(1) + (T(0, 1) * T(0, 3) + T(1, 1) * T(1, 3) + T(2, 1) * T(2, 3)), 2)
                                       + std::pow(p(2) + (T(0, 2) * T(0, 3) + T(1, 2) * T(1, 3) + T(2, 2) * T(2, 3)), 2));
            lm_to_scale[lm->id_] = lm->predict_scale_level((float)d, (float)M.orb.num_levels_, M.orb.log_scale_factor_);
            local.push_back(lm);
        }
        const unsigned before = right + wrong;
        const unsigned num2 = match::hip::projection(0.8, true).match_frame_and_landmarks(curr, local, lm_to_reproj, lm_to_x_right, lm_to_scale, 5.0f);
        unsigned total = 0, r2 = 0;
        for (size_t i = 0; i < curr.landmarks_.size(); ++i)
            if (curr.landmarks_[i]) {
                ++total;
                r2 += (int)curr.landmarks_[i]->id_ == M.truth[B][i];
            }
        REQUIRE(total == before + num2 && r2 * 10 > total * 9);
    }
    {  // projection::match_frame_and_keyframe (relocalisation form) and match_by_Sim3_transform
        data::frame curr = frame_like(M, B, true);
        const unsigned num = match::hip::projection(0.9, true).match_frame_and_keyframe(curr, M.kfs[A], {}, 10.0f, 100);
        unsigned right = 0;
        for (size_t i = 0; i < curr.landmarks_.size(); ++i)
            if (curr.landmarks_[i] && (int)curr.landmarks_[i]->id_ == M.truth[B][i]) ++right;
        REQUIRE(num > 200 && right * 10 > num * 9);
        std::vector<lm_ptr> matched_in_kf(M.kfs[B]->frm_obs_.undist_keypts_.size(), nullptr);
        Mat44_t S = M.kfs[B]->get_pose_cw();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) S(i, j) *= 1.7;  // a Sim3 with scale 1.7 describes the same camera
        const unsigned n3 = match::hip::projection(0.9, true).match_by_Sim3_transform(M.kfs[B], S, M.lms, matched_in_kf, 7.5f);
        unsigned r3 = 0;
        for (size_t i = 0; i < matched_in_kf.size(); ++i)
            if (matched_in_kf[i] && (int)matched_in_kf[i]->id_ == M.truth[B][i]) ++r3;
        REQUIRE(n3 > 300 && r3 * 10 > n3 * 9);
        // match_keyframes_mutually with the true relative pose (s = 1): cross-checked matches only
        std::vector<lm_ptr> mutual(M.kfs[A]->frm_obs_.undist_keypts_.size(), nullptr);
        const Mat44_t TA = M.kfs[A]->get_pose_cw(), TB = M.kfs[B]->get_pose_cw();
        Mat33_t R12;
        Vec3_t t12;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R12(i, j) = TA(i, 0) * TB(j, 0) + TA(i, 1) * TB(j, 1) + TA(i, 2) * TB(j, 2);
        for (int i = 0; i < 3; ++i) t12(i) = TA(i, 3) - (R12(i, 0) * TB(0, 3) + R12(i, 1) * TB(1, 3) + R12(i, 2) * TB(2, 3));
        const float s12 = 1.0f;
        const unsigned nm = match::hip::projection(0.9, true).match_keyframes_mutually(M.kfs[A], M.kfs[B], mutual, s12, R12, t12, 7.5f);
        unsigned rm = 0;
        for (size_t i = 0; i < mutual.size(); ++i)
            if (mutual[i] && (int)mutual[i]->id_ == M.truth[A][i]) ++rm;
        REQUIRE(nm > 200 && rm * 10 > nm * 9);
    }
    {  // triangulation matchers: strip the landmarks of half of the keypoints of A and B, match the bare keypoints, compare with the truth
        auto a = M.kfs[A], b = M.kfs[B];
        const auto keep_a = a->landmarks_, keep_b = b->landmarks_;
        for (size_t i = 0; i < a->landmarks_.size(); i += 2) a->landmarks_[i] = nullptr;
        for (size_t i = 0; i < b->landmarks_.size(); ++i)
            if (M.truth[B][i] % 2 == 0 || i % 3 == 0) b->landmarks_[i] = nullptr;
        const Mat44_t TA = a->get_pose_cw(), TB = b->get_pose_cw();
        Mat33_t R12, E12;
        double t12[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R12(i, j) = TA(i, 0) * TB(j, 0) + TA(i, 1) * TB(j, 1) + TA(i, 2) * TB(j, 2);
        for (int i = 0; i < 3; ++i) t12[i] = TA(i, 3) - (R12(i, 0) * TB(0, 3) + R12(i, 1) * TB(1, 3) + R12(i, 2) * TB(2, 3));
        const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) E12(i, j) = tx[3 * i] * R12(0, j) + tx[3 * i + 1] * R12(1, j) + tx[3 * i + 2] * R12(2, j);
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<std::pair<unsigned, unsigned>> pairs;
            const unsigned num = variant ? match::hip::bow_tree(0.8, true).match_for_triangulation(a, b, E12, pairs, 0.02f)
                                         : match::hip::robust(0.8, true).match_for_triangulation(a, b, E12, pairs, 0.02f);
            unsigned right = 0;
            for (const auto& pr : pairs) {
                REQUIRE(!a->landmarks_[pr.first] && !b->landmarks_[pr.second]);
                right += M.truth[A][pr.first] >= 0 && M.truth[A][pr.first] == M.truth[B][pr.second];
            }
            REQUIRE(num == pairs.size() && num > 40 && right * 10 > num * 8);
        }
        a->landmarks_ = keep_a;
        b->landmarks_ = keep_b;
        // robust::brute_force_match: frame with A's keypoints against keyframe A itself -> identity matches on the landmark keypoints
        std::vector<std::pair<int, int>> matches;
        const unsigned nb = match::hip::robust(0.8, true).brute_force_match(a->frm_obs_, a, matches);
        unsigned ident = 0;
        for (const auto& m : matches) ident += m.first == m.second;
        REQUIRE(nb == matches.size() && nb > 300 && ident == nb);
    }
    {  // fuse::detect_duplication: landmarks B does not know yet become new connections at their true keypoint, known ones duplicates
        auto b = M.kfs[B];
        std::vector<lm_ptr> to_check;
        unsigned removed = 0;
        for (size_t i = 0; i < b->landmarks_.size(); ++i) {
            auto lm = b->landmarks_[i];
            if (!lm) continue;
            if (i % 2 == 0) {  // forget the association: the landmark is "not observed in the keyframe"
                lm->erase_observation(nullptr, b);
                b->landmarks_[i] = nullptr;
                to_check.push_back(lm);
                ++removed;
            }
        }
        // and clones of still-associated landmarks: same place and descriptor, new ids -> duplicates of what B already holds
        std::vector<lm_ptr> clones;
        for (size_t i = 1; i < b->landmarks_.size() && clones.size() < 60; i += 2)
            if (auto lm = b->landmarks_[i]) {
                auto c = std::make_shared<data::landmark>(100000 + lm->id_, lm->get_pos_in_world());
                c->descriptor_ = lm->descriptor_, c->mean_normal_ = lm->mean_normal_, c->min_valid_dist_ = lm->min_valid_dist_, c->max_valid_dist_ = lm->max_valid_dist_;
                c->add_observation(M.kfs[A], 0);
                c->add_observation(M.kfs[A + 3], 0);
                c->add_observation(M.kfs[A + 4], 0);
                clones.push_back(c);
                to_check.push_back(c);
            }
        std::unordered_map<lm_ptr, lm_ptr> duplicated;
        std::unordered_map<unsigned int, lm_ptr> new_connections;
        const unsigned nf = match::hip::fuse(0.6).detect_duplication(b, b->get_rot_cw(), b->get_trans_cw(), to_check, 3.0f, duplicated, new_connections, true);
        unsigned right_new = 0, right_dup = 0;
        for (const auto& kv : new_connections) right_new += M.truth[B][kv.first] == (int)kv.second->id_;
        for (const auto& kv : duplicated) right_dup += kv.first->id_ == 100000 + kv.second->id_;
        REQUIRE(nf == new_connections.size() + duplicated.size());
        REQUIRE(right_new * 10 > removed * 7 && right_new * 20 > new_connections.size() * 19);
        REQUIRE(right_dup * 10 > clones.size() * 7 && right_dup == duplicated.size());
        for (const auto& kv : new_connections) {  // restore the map
            b->landmarks_[kv.first] = kv.second;
            kv.second->add_observation(b, kv.first);
        }
        // unordered_set instantiation
        std::unordered_set<lm_ptr> as_set(clones.begin(), clones.end());
        REQUIRE(match::hip::fuse(0.6).detect_duplication(b, b->get_rot_cw(), b->get_trans_cw(), as_set, 3.0f, duplicated, new_connections, false) == duplicated.size());
    }
    {  // area::match_in_consistent_area (the initialiser): level-0 keypoints of A searched around their own position in a copy of A
        data::frame f1 = frame_like(M, A, false), f2 = frame_like(M, A, false);
        std::vector<cv::Point2f> prev;
        for (const auto& kp : f1.frm_obs_.undist_keypts_) prev.push_back(kp.pt);
        std::vector<int> m21;
        const unsigned na = match::hip::area(0.9, true).match_in_consistent_area(f1, f2, prev, m21, 50);
        unsigned lvl0 = 0, ident = 0;
        for (size_t i = 0; i < m21.size(); ++i) {
            lvl0 += f1.frm_obs_.undist_keypts_[i].octave == 0;
            ident += m21[i] == (int)i;
        }
        REQUIRE(na > 0 && na <= lvl0 && ident == na);
    }
    {  // local_bundle_adjuster_hip behind the factory line: perturbed local map, gross outlier observations
        std::mt19937 rng(5);
        std::normal_distribution<double> N(0.0, 1.0);
        YAML::Node yaml;
        yaml.kv["backend"] = "g2o";
        REQUIRE(optimize::hip_backend::create_local_bundle_adjuster(yaml) == nullptr);
        yaml.kv["backend"] = "hip";
        auto ba = optimize::hip_backend::create_local_bundle_adjuster(yaml);
        REQUIRE(ba != nullptr);
        auto curr = M.kfs[4];
        for (int k : {2, 3, 5}) curr->graph_node_->covisibilities_.push_back(M.kfs[k]);  // local: 2 3 4 5; fixed: everyone else who sees their landmarks
        double err_before = 0, err_after = 0;
        for (int k : {2, 3, 4, 5}) {
            Mat44_t T = M.kfs[k]->get_pose_cw();
            for (int i = 0; i < 3; ++i) T(i, 3) += 0.02 * N(rng);
            M.kfs[k]->set_pose_cw(T);
            for (int i = 0; i < 3; ++i) err_before += std::fabs(T(i, 3) - M.pose_gt[k](i, 3));
        }
        double perr_before = 0, perr_after = 0;
        for (auto& lm : M.lms) {
            Vec3_t p = lm->get_pos_in_world();
            for (int i = 0; i < 3; ++i) p(i) += 0.03 * N(rng);
            lm->set_pos_in_world(p);
        }
        // gross outliers: move 12 observed keypoints of keyframe 3 by 45 px
        std::vector<std::pair<int, lm_ptr>> bad;
        for (size_t i = 0; i < M.kfs[3]->landmarks_.size() && bad.size() < 12; i += 7)
            if (auto lm = M.kfs[3]->landmarks_[i])
                if (lm->num_observations() >= 4) {
                    M.kfs[3]->frm_obs_.undist_keypts_[i].pt.x += 45.f;
                    bad.emplace_back((int)i, lm);
                }
        for (size_t l = 0; l < M.lms.size(); ++l)
            for (int i = 0; i < 3; ++i) perr_before += std::fabs(M.lms[l]->get_pos_in_world()(i) - M.pos_gt[l](i));
        const unsigned refreshes_before = M.lms[bad[0].second->id_]->num_geometry_refreshes_;
        bool force_stop = false;
        ba->optimize(&M.db, curr, &force_stop);
        const auto* hipba = static_cast<const optimize::local_bundle_adjuster_hip*>(ba.get());
        REQUIRE(hipba->last_status_ == 0 && hipba->last_stats_.iters_stage1 > 0 && hipba->last_stats_.chi2_final < 0.2 * hipba->last_stats_.chi2_initial);
        for (int k : {2, 3, 4, 5})
            for (int i = 0; i < 3; ++i) err_after += std::fabs(M.kfs[k]->get_pose_cw()(i, 3) - M.pose_gt[k](i, 3));
        unsigned n_local = 0;
        for (size_t l = 0; l < M.lms.size(); ++l)
            for (int i = 0; i < 3; ++i) perr_after += std::fabs(M.lms[l]->get_pos_in_world()(i) - M.pos_gt[l](i));
        for (auto& lm : M.lms) n_local += lm->num_geometry_refreshes_ > 1;
        std::fprintf(stderr, "[ba] pose err %.4f -> %.4f, point err %.3f -> %.3f, refreshed landmarks %u, chi2 %.1f -> %.1f, iters %d + %d, gated %d\n", err_before, err_after,
                     perr_before, perr_after, n_local, hipba->last_stats_.chi2_initial, hipba->last_stats_.chi2_final, hipba->last_stats_.iters_stage1,
                     hipba->last_stats_.iters_stage2, hipba->last_stats_.num_gated);
        REQUIRE(err_after < 0.5 * err_before && perr_after < perr_before && n_local > 300);  // (depth along the narrow baseline stays weakly constrained)
        unsigned erased = 0;
        for (const auto& b : bad) erased += !M.kfs[3]->landmarks_[b.first] && !b.second->is_observed_in_keyframe(M.kfs[3]);
        REQUIRE(erased >= 10 && M.lms[bad[0].second->id_]->num_geometry_refreshes_ > refreshes_before);
        for (int k : {0, 1, 6, 7})  // fixed keyframes keep their poses bit for bit
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) REQUIRE(M.kfs[k]->get_pose_cw()(i, j) == M.pose_gt[k](i, j));
        // a raised force_stop_flag before the call leaves the map untouched (local_bundle_adjuster_g2o.cc:308-310)
        const Mat44_t keep = curr->get_pose_cw();
        force_stop = true;
        ba->optimize(&M.db, curr, &force_stop);
        for (int i = 0; i < 3; ++i) REQUIRE(curr->get_pose_cw()(i, 3) == keep(i, 3));
    }
    {  // pose_optimizer_hip behind the factory line, frame overload: a perturbed pose comes back to the keyframe's, gross outliers are flagged
        REQUIRE(optimize::hip_backend::create_pose_optimizer("g2o") == nullptr);
        auto po = optimize::hip_backend::create_pose_optimizer("hip");
        REQUIRE(po != nullptr);
        data::frame f = frame_like(M, 6, true);
        f.landmarks_ = M.kfs[6]->landmarks_;
        Mat44_t T0 = M.kfs[6]->get_pose_cw();
        for (int i = 0; i < 3; ++i) T0(i, 3) += 0.03 * (i + 1);
        f.set_pose_cw(T0);
        std::vector<int> moved;
        for (size_t i = 0; i < f.landmarks_.size() && moved.size() < 8; i += 5)
            if (f.landmarks_[i]) {
                f.frm_obs_.undist_keypts_[i].pt.y += 60.f;
                moved.push_back((int)i);
            }
        Mat44_t T1 = T0;
        std::vector<bool> flags;
        const unsigned good = po->optimize(f, T1, flags);
        double e0 = 0, e1 = 0;
        for (int i = 0; i < 3; ++i) e0 += std::fabs(T0(i, 3) - M.kfs[6]->get_pose_cw()(i, 3)), e1 += std::fabs(T1(i, 3) - M.kfs[6]->get_pose_cw()(i, 3));
        unsigned flagged = 0;
        for (int i : moved) flagged += flags.at(i) ? 1 : 0;
        std::fprintf(stderr, "[pose] %u good observations, translation error %.4f -> %.4f, %u of %zu moved keypoints flagged\n", good, e0, e1, flagged, moved.size());
        REQUIRE(flags.size() == f.frm_obs_.undist_keypts_.size() && good > 50 && e1 < 0.2 * e0 && flagged == moved.size());
    }
    {  // global_bundle_adjuster_hip::optimize over all keyframes with one free and one kept-fixed marker (global_bundle_adjuster.cc:131-181, 380-408)
        for (int k = 0; k < (int)M.kfs.size(); ++k) M.kfs[k]->graph_node_->spanning_root_ = k == 0;
        auto make_marker = [&](unsigned id, const Vec3_t& centre, bool keep_fixed, double noise) {
            auto mk = std::make_shared<data::marker>();
            mk->id_ = id, mk->keep_fixed_ = keep_fixed, mk->initialized_before_ = true;
            const double d[4][2] = {{-0.1, -0.1}, {0.1, -0.1}, {0.1, 0.1}, {-0.1, 0.1}};
            for (int c = 0; c < 4; ++c) {
                Vec3_t truth;
                truth(0) = centre(0) + d[c][0], truth(1) = centre(1) + d[c][1], truth(2) = centre(2);
                for (auto& kf : M.kfs) {
                    double u, v, z;
                    project(M, kf->get_pose_cw(), truth, u, v, z);
                    kf->markers_2d_[id].undist_corners_.resize(4);
                    kf->markers_2d_[id].undist_corners_[c].x = (float)u, kf->markers_2d_[id].undist_corners_[c].y = (float)v;
                    mk->observations_[kf->id_] = kf;
                }
                Vec3_t start;
                start(0) = truth(0) + noise, start(1) = truth(1) - noise, start(2) = truth(2) + noise;
                mk->corners_pos_w_.push_back(start);
            }
            for (auto& kf : M.kfs) kf->markers_.push_back(mk);
            return mk;
        };
        const Vec3_t c0 = M.lms[0]->get_pos_in_world(), c1 = M.lms[1]->get_pos_in_world();
        auto free_mk = make_marker(7, c0, false, 0.05), fixed_mk = make_marker(9, c1, true, 0.05);
        const auto fixed_before = fixed_mk->corners_pos_w_;
        optimize::global_bundle_adjuster_hip gba(10, true);
        std::unordered_set<unsigned int> okf, olm, omk;
        eigen_alloc_unord_map<unsigned int, Vec3_t> lm_pos;
        eigen_alloc_unord_map<unsigned int, Mat44_t> kf_pose;
        eigen_alloc_unord_map<unsigned int, std::array<Vec3_t, 4>> mk_pos;
        bool stop = false;
        REQUIRE(gba.optimize(M.kfs, okf, olm, omk, lm_pos, kf_pose, mk_pos, &stop));
        REQUIRE(okf.size() == M.kfs.size() && !olm.empty() && gba.last_stats_.chi2_final < gba.last_stats_.chi2_initial);
        REQUIRE(omk.count(7) == 1 && omk.count(9) == 0 && mk_pos.count(7) == 1 && mk_pos.count(9) == 0);  // a kept-fixed marker is neither moved nor reported
        // (one fixed keyframe of a monocular map leaves the gauge to drift: the corners are judged by their reprojection, not by position)
        double m0 = 0, m1 = 0;
        for (int c = 0; c < 4; ++c) {
            for (auto& kf : M.kfs) {
                const auto& obs = kf->markers_2d_.at(7).undist_corners_[c];
                double u, v, z;
                project(M, kf->get_pose_cw(), free_mk->corners_pos_w_[c], u, v, z);
                m0 += std::fabs(u - obs.x) + std::fabs(v - obs.y);
                project(M, kf_pose.at(kf->id_), mk_pos.at(7)[c], u, v, z);
                m1 += std::fabs(u - obs.x) + std::fabs(v - obs.y);
            }
            for (int i = 0; i < 3; ++i) REQUIRE(fixed_mk->corners_pos_w_[c](i) == fixed_before[c](i));
        }
        std::fprintf(stderr, "[gba] chi2 %.1f -> %.1f in %d iterations, free marker reprojection error %.1f -> %.1f px (sum)\n", gba.last_stats_.chi2_initial,
                     gba.last_stats_.chi2_final, gba.last_stats_.iters_stage1, m0, m1);
        REQUIRE(m1 < 0.2 * m0);
        const Mat44_t root = M.kfs[0]->get_pose_cw();  // the spanning root is a fixed vertex: its pose comes back bit for bit
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) REQUIRE(kf_pose.at(M.kfs[0]->id_)(i, j) == root(i, j));
        {  // the local bundle adjuster with the two markers in its window (local_bundle_adjuster_g2o.cc:246-304, 411-428)
            YAML::Node yaml;
            yaml.kv["backend"] = "hip";
            auto lba = optimize::hip_backend::create_local_bundle_adjuster(yaml);
            auto reproj = [&]() {
                double e = 0;
                for (int c = 0; c < 4; ++c)
                    for (int k : {2, 3, 4, 5}) {
                        const auto& obs = M.kfs[k]->markers_2d_.at(7).undist_corners_[c];
                        double u, v, z;
                        project(M, M.kfs[k]->get_pose_cw(), free_mk->corners_pos_w_[c], u, v, z);
                        e += std::fabs(u - obs.x) + std::fabs(v - obs.y);
                    }
                return e;
            };
            const double l0 = reproj();
            bool no_stop = false;
            lba->optimize(&M.db, M.kfs[4], &no_stop);
            const double l1 = reproj();
            std::fprintf(stderr, "[ba+markers] free marker reprojection error in the window %.1f -> %.1f px (sum)\n", l0, l1);
            REQUIRE(static_cast<const optimize::local_bundle_adjuster_hip*>(lba.get())->last_status_ == 0 && l1 < 0.2 * l0);
            for (int c = 0; c < 4; ++c)
                for (int i = 0; i < 3; ++i) REQUIRE(fixed_mk->corners_pos_w_[c](i) == fixed_before[c](i));
        }
        // optimize_for_initialization writes the map directly; with fix_markers the free marker keeps its corners
        const auto free_before = free_mk->corners_pos_w_;
        std::vector<std::shared_ptr<data::marker>> mks{free_mk, fixed_mk};
        const unsigned refreshes = M.lms[5]->num_geometry_refreshes_;
        gba.optimize_for_initialization(M.kfs, M.lms, mks, 1e-5f, true, nullptr);
        REQUIRE(gba.last_status_ == 0 && M.lms[5]->num_geometry_refreshes_ > refreshes);
        for (int c = 0; c < 4; ++c)
            for (int i = 0; i < 3; ++i) REQUIRE(free_mk->corners_pos_w_[c](i) == free_before[c](i));
    }
    std::printf("drop-in classes ok\n");
    return 0;
}
