#!/usr/bin/env python3
"""Regenerates the fixtures of tests/golden/ (run from the repository root: python tests/golden/make_golden.py).

Two kinds of fixtures live here:
  reference_kats.json   the known-answer vectors the REFERENCE's own tests hold for this path, transcribed as data with
                        their source lines (the reference cannot be built in this container: no OpenCV / Eigen / g2o), i.e.
                        everything the reference pins: Hamming distances, scale tables, the trigonometric tolerance.
  *_oracle.npz          outputs of oracle/ (the CPU restatement) on seeded synthetic inputs.  They do NOT come from the
                        reference ("parity unpinned"); they freeze the oracle so that a later change to
                        it is noticed, and they let the GPU tests compare against committed bytes as well as against the
                        live oracle.
"""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from stella_vslam_amd import synthetic as S

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    kats = {
        "hamming": {"source": "test/stella_vslam/match/base.cc:11-57 (compute_descriptor_distance_32 and _64)",
                    "cases": [{"byte_1": 0b01010101, "byte_2": 0b01010101, "distance": 0},
                              {"byte_1": 0b01010101, "byte_2": 0b10101010, "distance": 256},
                              {"byte_1": 0b01100110, "byte_2": 0b00111100, "distance": 128}]},
        "scale_tables": {"source": "test/stella_vslam/feature/orb_params.cc:27-70 (EXPECT_FLOAT_EQ = 4 ulp)",
                         "num_levels": 10, "scale_factor": 1.26,
                         "rule": "scale_factors[l] ~ pow(sf, l); inv ~ pow(1/sf, l); sigma_sq[l] = s_l^2 with s_l = sf * s_(l-1) in fp32; inv_sigma_sq = 1 / sigma_sq"},
        "trigonometric": {"source": "test/stella_vslam/util/trigonometric.cc:8-20", "tolerance": 1e-3,
                          "range": "angles -2 pi .. 2 pi in steps of 0.01"},
        "extractor_invariants": {"source": "test/stella_vslam/feature/orb_extractor.cc:25-77,117-357",
                                 "rule": "600x600 white image with a black quadrant at (300,300): every keypoint within 2 x scale of the corner; no keypoint inside a mask rectangle; descriptors.rows == keypoints.size()"},
    }
    json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)

    seq = S.frame_sequence(2, 640, 480, seed=0x5EED)
    k0, d0, c0 = O.orb_extract(seq[0])
    k1, d1, c1 = O.orb_extract(seq[1])
    np.savez_compressed(os.path.join(HERE, "orb_640x480_oracle.npz"), kp0=k0, desc0=d0, counts0=c0, kp1=k1, desc1=d1, counts1=c1)
    m = O.brute_force_match(d1, k1["angle"], d0, k0["angle"], None, 0.8, True)
    np.savez_compressed(os.path.join(HERE, "bruteforce_oracle.npz"), matched_2_in_1=m)
    sc = S.ba_scene(num_kf=6, num_lm=300, obs_per_lm=4, num_fixed=2, seed=3)
    r = O.local_ba(sc)
    np.savez_compressed(os.path.join(HERE, "local_ba_oracle.npz"), pose_cw=r["pose_cw"], points=r["points"], outlier=r["outlier"],
                        stats=np.asarray(r["stats"], np.float64))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
