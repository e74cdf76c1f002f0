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


class resident_frame:
    """A data::frame_observation kept on the device (include/svgpu.h svgpu_frame_*): descriptors, undistorted keypoints, stereo
    x_right, bearings and the keypoint grid.  `adopt_extraction` builds it from what the extractor's last extract() left on the
    device (system.cc:384-395 without the upload), `upload` from host arrays (keyframes); `bind(ctx)` makes the NEXT
    projection-family matcher call of `ctx` read its keypoint side from it."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx.check(lib().svgpu_frame_create(ctx.handle, C.byref(self._h)), "svgpu_frame_create")

    def adopt_extraction(self, camera: _camera.base, num_grid_cols: int = 64, num_grid_rows: int = 48, extractor_ctx: Context | None = None):
        """-> (undist_keypts, bearings) as data::frame_observation holds them on the host"""
        from .feature import KEYPOINT_DTYPE as KP_DTYPE
        ectx = extractor_ctx or self.ctx
        n_max = lib().svgpu_orb_max_keypoints(ectx.handle)
        und = np.zeros(max(n_max, 1), KP_DTYPE)
        brg = np.zeros((max(n_max, 1), 3), np.float64)
        ectx.check(lib().svgpu_frame_adopt_extraction(ectx.handle, self._h, C.byref(camera.c_), num_grid_cols, num_grid_rows, _p(und), _p(brg)),
                   "svgpu_frame_adopt_extraction")
        n = self.size
        return und[:n].copy(), brg[:n].copy()

    def upload(self, camera: _camera.base, undist_keypts, descriptors, stereo_x_right=None, num_grid_cols: int = 64, num_grid_rows: int = 48):
        from .feature import KEYPOINT_DTYPE as KP_DTYPE
        k = np.ascontiguousarray(undist_keypts, KP_DTYPE)
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        xr = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, np.float32)
        self.ctx.check(lib().svgpu_frame_upload(self.ctx.handle, self._h, C.byref(camera.c_), _p(k), _p(d), _p(xr), len(k), num_grid_cols, num_grid_rows),
                       "svgpu_frame_upload")
        return self

    def set_stereo(self, stereo_x_right):
        xr = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, np.float32)
        self.ctx.check(lib().svgpu_frame_set_stereo(self.ctx.handle, self._h, _p(xr)), "svgpu_frame_set_stereo")
        return self

    @property
    def size(self) -> int:
        return int(lib().svgpu_frame_size(self._h))

    def bind(self, ctx: Context | None = None):
        c = ctx or self.ctx
        c.check(lib().svgpu_frame_bind(c.handle, self._h), "svgpu_frame_bind")
        return self

    def close(self):
        if self._h:
            lib().svgpu_frame_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
    device, the two sparse maps are assembled here.  Two frameworks, as in the reference's build switch (USE_DBOW2):
      framework = "fbow"   the DEFAULT build: fbow::Vocabulary::transform(descriptors, 4, bow_vec, bow_feat_vec) -- the store level counts
                           DOWN from the root, bow_feat_vec is keyed by FBoW's path code of the node, every word's weight is summed (no
                           stop-word skip) and the vector is normalised (`norm`: "L2" as upstream FBoW's transform does, or "L1")
      framework = "dbow2"  USE_DBOW2: TemplatedVocabulary::transform(features, bow_vec, bow_feat_vec, 4) -- levels UP from the leaves,
                           node index keys, zero-weight words skipped, L1 normalisation
    Both are restated from the libraries' published sources (neither is in the reference checkout): parity unpinned."""

    def __init__(self, ctx: Context, child_off, children, node_desc, node_weight, word_id, depth: int, framework: str = "dbow2", k: int | None = None,
                 norm: str | None = None):
        self.ctx = ctx
        self.depth_ = int(depth)
        if framework not in ("dbow2", "fbow"):
            raise ValueError(f"unknown BoW framework {framework!r}")
        self.framework_ = framework
        off, ch = np.ascontiguousarray(child_off, np.int32), np.ascontiguousarray(children, np.int32)
        self.k_ = int(k) if k is not None else int(np.diff(off).max(initial=2))
        self.norm_ = norm or ("L2" if framework == "fbow" else "L1")
        nd = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        nw, wi = np.ascontiguousarray(node_weight, np.float32), np.ascontiguousarray(word_id, np.int32)
        self._h = C.c_void_p()
        ctx.check(lib().svgpu_bow_vocabulary_upload(ctx.handle, len(off) - 1, _p(off), _p(ch), _p(nd), _p(nw), _p(wi), C.byref(self._h)),
                  "svgpu_bow_vocabulary_upload")

    @classmethod
    def load_fbow(cls, ctx: Context, path: str) -> "bow_vocabulary":
        """bow_vocabulary_util::load (data/bow_vocabulary.cc:26-47) for an .fbow file; see read_fbow for what is unverified."""
        t = read_fbow(path)
        return cls(ctx, t["child_off"], t["children"], t["node_desc"], t["node_weight"], t["word_id"], t["depth"], framework="fbow", k=t.get("k"))

    def descend(self, descriptors, levels_up: int = 4):
        """DBoW2 form: (word, weight, node index at depth L - levels_up).  FBoW form: use descend_fbow."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word, weight, node = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        level = max(self.depth_ - levels_up, 0)
        self.ctx.check(lib().svgpu_bow_transform(self.ctx.handle, self._h, _p(d), n, level, _p(word), _p(weight), _p(node)),
                       "svgpu_bow_transform")
        return word, weight, node

    def descend_fbow(self, descriptors, store_level: int = 4):
        """fbow::Vocabulary::transform's descent: (word, weight, bow_feat_vec key = path code at `store_level` from the root)."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word, weight, code = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        self.ctx.check(lib().svgpu_fbow_transform(self.ctx.handle, self._h, _p(d), n, int(store_level), self.k_, _p(word), _p(weight), _p(code)),
                       "svgpu_fbow_transform")
        return word, weight, code

    def transform(self, descriptors, levels_up: int = 4):
        """-> (bow_vec: {word_id: weight}, bow_feat_vec: {node key: [feature indices]}); `levels_up` is the literal 4 compute_bow passes."""
        bow_vec, feat = {}, {}
        if self.framework_ == "fbow":
            word, weight, node = self.descend_fbow(descriptors, levels_up)
            for i in range(len(word)):
                bow_vec[int(word[i])] = bow_vec.get(int(word[i]), 0.0) + float(weight[i])
                feat.setdefault(int(node[i]), []).append(i)
        else:
            word, weight, node = self.descend(descriptors, levels_up)
            for i in range(len(word)):
                if weight[i] > 0:  # DBoW2 skips zero-weight words (stop words)
                    bow_vec[int(word[i])] = bow_vec.get(int(word[i]), 0.0) + float(weight[i])
                    feat.setdefault(int(node[i]), []).append(i)
        if self.norm_ == "L2":
            norm = float(np.sqrt(sum(v * v for _, v in sorted(bow_vec.items()))))
        else:
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


# ------------------------------------------------------------------------------------------- .fbow vocabulary files
# data::bow_vocabulary_util::load (data/bow_vocabulary.cc:26-47) hands the path to fbow::Vocabulary::readFromFile in the default
# (FBoW) build.  FBoW is an un-vendored submodule (3rd/FBoW is empty in the checkout studied; the reference pins stella-cv/FBoW),
# so the layout below restates the published fbow/vocabulary.h and has NOT been checked against a real orb_vocab.fbow here
# (none is available offline): treat it as unverified until a real file has been read with it.
#   uint64 magic 55824124 | struct params (120 bytes) | _total_size bytes of blocks
#   params: char desc_name[50]; u32 alignment, nblocks; u64 desc_size_wp, block_size_wp, feature_off, child_off, total_size;
#           i32 desc_type, desc_size; u32 k                                   (natural C alignment, little endian)
#   block:  u16 N | u16 is_leaf | u32 parent | pad | N descriptors (desc_size_wp apart, from feature_off) |
#           N x {u32 id_or_childblock (MSB = leaf), f32 weight} from child_off
FBOW_MAGIC = 55824124
_FBOW_PARAMS = np.dtype([("desc_name", "S50"), ("_pad0", "V2"), ("alignment", "<u4"), ("nblocks", "<u4"), ("_pad1", "V4"), ("desc_size_wp", "<u8"),
                         ("block_size_wp", "<u8"), ("feature_off", "<u8"), ("child_off", "<u8"), ("total_size", "<u8"), ("desc_type", "<i4"),
                         ("desc_size", "<i4"), ("k", "<u4"), ("_pad2", "V4")])
assert _FBOW_PARAMS.itemsize == 120


def read_fbow(path: str) -> dict:
    """-> the flat tree svgpu_bow_vocabulary_upload takes (node 0 = a virtual root whose children are block 0's nodes; nodes in
    breadth-first order with siblings in block order, so that node ids at one depth sort like fbow's path-bit node ids)."""
    raw = np.fromfile(path, np.uint8)
    if len(raw) < 128 or int(raw[:8].view("<u8")[0]) != FBOW_MAGIC:
        raise ValueError(f"{path}: not an fbow vocabulary (bad signature)")
    P = raw[8:128].view(_FBOW_PARAMS)[0]
    if int(P["desc_size"]) != 32 or int(P["desc_type"]) != 0:
        raise ValueError(f"{path}: only 32-byte CV_8UC1 (ORB) vocabularies are supported (desc_size {P['desc_size']}, type {P['desc_type']})")
    bs, fo, co, dw, nb = (int(P[k]) for k in ("block_size_wp", "feature_off", "child_off", "desc_size_wp", "nblocks"))
    data = raw[128:128 + int(P["total_size"])]
    if len(data) != int(P["total_size"]) or nb * bs > len(data):
        raise ValueError(f"{path}: truncated")
    desc, weight, word, kids = [np.zeros(32, np.uint8)], [0.0], [-1], [[]]
    queue, depth, seen = [(0, 0, 0)], 0, set()      # (block, parent node, depth of the block's nodes - 1)
    while queue:
        blk, parent, d = queue.pop(0)
        if blk >= nb or blk in seen:
            raise ValueError(f"{path}: corrupt block graph at block {blk}")
        seen.add(blk)
        b = data[blk * bs:(blk + 1) * bs]
        n = int(b[:2].view("<u2")[0])
        info = b[co:co + 8 * n].view(np.dtype([("id", "<u4"), ("w", "<f4")]))
        depth = max(depth, d + 1)
        for i in range(n):
            node = len(desc)
            kids[parent].append(node)
            desc.append(b[fo + i * dw:fo + i * dw + 32].copy())
            kids.append([])
            if int(info["id"][i]) & 0x80000000:
                weight.append(float(info["w"][i]))
                word.append(int(info["id"][i]) & 0x7FFFFFFF)
            else:
                weight.append(0.0)
                word.append(-1)
                queue.append((int(info["id"][i]) & 0x7FFFFFFF, node, d + 1))
    off = np.zeros(len(desc) + 1, np.int32)
    off[1:] = np.cumsum([len(k) for k in kids])
    return dict(child_off=off, children=np.array([c for k in kids for c in k], np.int32), node_desc=np.stack(desc), node_weight=np.array(weight, np.float32),
                word_id=np.array(word, np.int32), depth=depth, k=int(P["k"]), desc_name=bytes(P["desc_name"]).split(b"\0")[0].decode())


def write_fbow(path: str, child_off, children, node_desc, node_weight, word_id, k: int, alignment: int = 8, desc_name: str = "orb") -> None:
    """Inverse of read_fbow (fixtures / converting a flat tree to a file the reference build can load)."""
    up = lambda v: (v + alignment - 1) // alignment * alignment
    dw, fo = up(32), up(8)
    co = fo + k * dw
    bs = up(co + 8 * k)
    blocks, queue = [], [(0, 0)]   # (node whose children form the block, parent block)
    index = {0: 0}
    while queue:
        node, parent = queue.pop(0)
        ch = [int(c) for c in children[child_off[node]:child_off[node + 1]]]
        assert 0 < len(ch) <= k
        b = np.zeros(bs, np.uint8)
        b[:2] = np.array([len(ch)], "<u2").view(np.uint8)
        leaf = all(child_off[c] == child_off[c + 1] for c in ch)
        b[2:4] = np.array([1 if leaf else 0], "<u2").view(np.uint8)
        b[4:8] = np.array([parent], "<u4").view(np.uint8)
        me = len(blocks)
        blocks.append(b)
        for i, c in enumerate(ch):
            b[fo + i * dw:fo + i * dw + 32] = node_desc[c]
            if child_off[c] == child_off[c + 1]:
                ident, w = 0x80000000 | int(word_id[c]), float(node_weight[c])
            else:
                index[c] = len(blocks) + len(queue)
                queue.append((c, me))
                ident, w = index[c], 0.0
            b[co + 8 * i:co + 8 * i + 4] = np.array([ident], "<u4").view(np.uint8)
            b[co + 8 * i + 4:co + 8 * i + 8] = np.array([w], "<f4").view(np.uint8)
    P = np.zeros(1, _FBOW_PARAMS)
    P["desc_name"], P["alignment"], P["nblocks"], P["desc_size_wp"], P["block_size_wp"] = desc_name.encode(), alignment, len(blocks), dw, bs
    P["feature_off"], P["child_off"], P["total_size"], P["desc_type"], P["desc_size"], P["k"] = fo, co, bs * len(blocks), 0, 32, k
    with open(path, "wb") as f:
        f.write(np.array([FBOW_MAGIC], "<u8").tobytes())
        f.write(P.tobytes())
        f.write(np.concatenate(blocks).tobytes())
