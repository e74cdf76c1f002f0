// Stand-in for g2o::SE3Quat: unit quaternion + translation, every operation the reference's vertices / edges call FORWARDED to the oracle's
// own SE3 helpers (oracle/ba_oracle.c orc_dbg_*): the fixtures pin the reference's edge and vertex formulas, not g2o's SE3 arithmetic (which
// the oracle restates from g2o's published se3quat.h and is not pinned here).
#ifndef SVREF_G2O_SE3QUAT_H
#define SVREF_G2O_SE3QUAT_H
#include "stella_vslam/type.h"

extern "C" {
void orc_dbg_se3_from_Rt(const double* R_row_major, const double* t, double* q4, double* t3);
void orc_dbg_se3_exp_mul(const double* upd6, const double* q4, const double* t3, double* q4_out, double* t3_out);
void orc_dbg_se3_map(const double* q4, const double* t3, const double* p, double* out);
void orc_dbg_quat_to_R(const double* q4, double* R_row_major);
}

namespace g2o {
class SE3Quat {
public:
    SE3Quat() {}
    SE3Quat(const stella_vslam::Mat33_t& R, const stella_vslam::Vec3_t& t) {
        double Rr[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rr[3 * i + j] = R(i, j);
        orc_dbg_se3_from_Rt(Rr, t.data(), q, this->t);
    }
    struct Rot {
        const double* q;
        stella_vslam::Mat33_t toRotationMatrix() const {
            double Rr[9];
            orc_dbg_quat_to_R(q, Rr);
            stella_vslam::Mat33_t R;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) R(i, j) = Rr[3 * i + j];
            return R;
        }
    };
    Rot rotation() const { return Rot{q}; }
    stella_vslam::Vec3_t translation() const { return stella_vslam::Vec3_t(t[0], t[1], t[2]); }
    stella_vslam::Mat44_t to_homogeneous_matrix() const {
        stella_vslam::Mat44_t M = stella_vslam::Mat44_t::Identity();
        const stella_vslam::Mat33_t R = rotation().toRotationMatrix();
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M(i, j) = R(i, j);
            M(i, 3) = t[i];
        }
        return M;
    }
    stella_vslam::Vec3_t map(const stella_vslam::Vec3_t& p) const {
        stella_vslam::Vec3_t o;
        orc_dbg_se3_map(q, t, p.data(), o.data());
        return o;
    }
    // exp(update) is only ever used as the left factor of a product (shot_vertex.h:54): a deferred object, evaluated in operator*
    struct Exp {
        double u[6];
        SE3Quat operator*(const SE3Quat& T) const {
            SE3Quat O;
            orc_dbg_se3_exp_mul(u, T.q, T.t, O.q, O.t);
            return O;
        }
    };
    static Exp exp(const stella_vslam::Vec6_t& update) {
        Exp e;
        for (int i = 0; i < 6; ++i) e.u[i] = update(i);
        return e;
    }
    // read() / write() of shot_vertex (never run by the fixtures) need these to compile
    SE3Quat inverse() const {
        SE3Quat O;
        O.q[0] = -q[0], O.q[1] = -q[1], O.q[2] = -q[2], O.q[3] = q[3];
        const double zero[3] = {0, 0, 0};
        double Rt[3];
        orc_dbg_se3_map(O.q, zero, t, Rt);
        for (int i = 0; i < 3; ++i) O.t[i] = -Rt[i];
        return O;
    }
    void fromVector(const stella_vslam::Vec7_t& v) {
        for (int i = 0; i < 3; ++i) t[i] = v(i);
        for (int i = 0; i < 4; ++i) q[i] = v(3 + i);
    }
    double operator[](int i) const { return i < 3 ? t[i] : q[i - 3]; }
    double q[4] = {0, 0, 0, 1};  // x y z w
    double t[3] = {0, 0, 0};
};
}  // namespace g2o
#endif
