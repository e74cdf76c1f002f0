"""Python mirror of the flat part of stella_vslam::data (frame_observation) over the C ABI.

`frame_observation` is built the way system::create_*_frame builds it (system.cc:384-395): undistort the extractor's
keypoints, convert them to bearings and bin them into the matcher grid -- one device call, no host arithmetic.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import camera as _camera
from ._lib import lib
from .feature import Context


class frame_observation:
    """data/frame_observation.h:12-38.  `keypt_indices_in_cells_` is kept as CSR (cell = col * num_grid_rows_ + row):
    `cell_off_` (num_grid_cols_ * num_grid_rows_ + 1) and `cell_items_`; `indices_in_cell(col, row)` returns the
    reference's vector for one cell."""

    def __init__(self, camera: _camera.base, keypts, descriptors, num_grid_cols: int = 64, num_grid_rows: int = 48,
                 stereo_x_right=None, depths=None):
        self.num_grid_cols_, self.num_grid_rows_ = int(num_grid_cols), int(num_grid_rows)
        self.descriptors_ = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        und, brg, off, items = camera._observe(keypts, grid=(self.num_grid_cols_, self.num_grid_rows_))
        self.undist_keypts_, self.bearings_, self.cell_off_, self.cell_items_ = und, brg, off, items
        self.stereo_x_right_ = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, np.float32)
        self.depths_ = None if depths is None else np.ascontiguousarray(depths, np.float32)

    def indices_in_cell(self, col: int, row: int) -> np.ndarray:
        c = col * self.num_grid_rows_ + row
        return self.cell_items_[self.cell_off_[c]:self.cell_off_[c + 1]]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def landmarks_compute_descriptor(ctx: Context, obs_off, obs_desc):
    """data::landmark::compute_descriptor (data/landmark.cc:199-254) for every CSR row of observations at once:
    returns (best_obs, descriptors) -- the index inside each landmark's list and the n x 32 representative descriptors."""
    off = np.ascontiguousarray(obs_off, np.int32)
    d = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32)
    n = len(off) - 1
    best, out = np.zeros(n, np.int32), np.zeros((n, 32), np.uint8)
    ctx.check(lib().svgpu_landmarks_compute_descriptor(ctx.handle, n, _p(off), _p(d), _p(best), _p(out)), "svgpu_landmarks_compute_descriptor")
    return best, out


def landmarks_update_mean_normal_and_obs_scale_variance(ctx: Context, obs_off, obs_trans_wc, pos_w, ref_trans_wc, ref_scale_factor,
                                                        inv_scale_factor_last):
    """data::landmark::update_mean_normal_and_obs_scale_variance (data/landmark.cc:285-318) for every landmark at once:
    returns (mean_normal n x 3, max_valid_dist, min_valid_dist)."""
    off = np.ascontiguousarray(obs_off, np.int32)
    c = np.ascontiguousarray(obs_trans_wc, np.float64).reshape(-1, 3)
    p = np.ascontiguousarray(pos_w, np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(ref_trans_wc, np.float64).reshape(-1, 3)
    sfr = np.ascontiguousarray(ref_scale_factor, np.float32)
    n = len(off) - 1
    mnrm, mx, mn = np.zeros((n, 3), np.float64), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ctx.check(lib().svgpu_landmarks_update_geometry(ctx.handle, n, _p(off), _p(c), _p(p), _p(r), _p(sfr), C.c_float(inv_scale_factor_last),
                                                    _p(mnrm), _p(mx), _p(mn)), "svgpu_landmarks_update_geometry")
    return mnrm, mx, mn


class bow_vocabulary:
    """data/bow_vocabulary.h on a flat tree (node 0 = root; `child_off` / `children` CSR; `node_desc` n x 32; `node_weight`,
    `word_id` per node; `depth` = L).  transform() is compute_bow (data/bow_vocabulary.cc:18-24): the tree descent runs on the
    device, the two sparse maps are assembled here as DBoW2 does (TF-IDF weights summed per word, then L1-normalised)."""

    def __init__(self, ctx: Context, child_off, children, node_desc, node_weight, word_id, depth: int):
        self.ctx = ctx
        self.depth_ = int(depth)
        off, ch = np.ascontiguousarray(child_off, np.int32), np.ascontiguousarray(children, np.int32)
        nd = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        nw, wi = np.ascontiguousarray(node_weight, np.float32), np.ascontiguousarray(word_id, np.int32)
        self._h = C.c_void_p()
        ctx.check(lib().svgpu_bow_vocabulary_upload(ctx.handle, len(off) - 1, _p(off), _p(ch), _p(nd), _p(nw), _p(wi), C.byref(self._h)),
                  "svgpu_bow_vocabulary_upload")

    def descend(self, descriptors, levels_up: int = 4):
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word, weight, node = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.ctx.check(lib().svgpu_bow_transform(self.ctx.handle, self._h, _p(d), n, max(self.depth_ - levels_up, 0), _p(word), _p(weight), _p(node)),
                       "svgpu_bow_transform")
        return word, weight, node

    def transform(self, descriptors, levels_up: int = 4):
        """-> (bow_vec: {word_id: weight}, bow_feat_vec: {node_id: [feature indices]})"""
        word, weight, node = self.descend(descriptors, levels_up)
        bow_vec, feat = {}, {}
        for i in range(len(word)):
            if weight[i] > 0:  # DBoW2 skips zero-weight words (stop words)
                bow_vec[int(word[i])] = bow_vec.get(int(word[i]), 0.0) + float(weight[i])
                feat.setdefault(int(node[i]), []).append(i)
        norm = sum(abs(v) for _, v in sorted(bow_vec.items()))
        if norm > 0.0:
            bow_vec = {k: v / norm for k, v in bow_vec.items()}
        return dict(sorted(bow_vec.items())), dict(sorted(feat.items()))

    def close(self):
        if self._h:
            lib().svgpu_bow_vocabulary_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
