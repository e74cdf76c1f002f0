"""GPU parity: vocabulary-tree descent (compute_bow's transform) through the C ABI, identical to the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_bow import make_tree

pytestmark = pytest.mark.gpu


def test_bow_descent_and_maps():
    from stella_vslam_amd import data, feature
    ctx = feature.Context(0)
    rng = np.random.default_rng(1)
    tree = make_tree(rng, k=10, depth=4, prune=0.05)
    assert len(tree["node_desc"]) > 5000
    voc = data.bow_vocabulary(ctx, tree["child_off"], tree["children"], tree["node_desc"], tree["node_weight"], tree["word_id"], depth=4)
    leaves = np.flatnonzero(tree["word_id"] >= 0)
    n = 6001
    q = tree["node_desc"][rng.choice(leaves, n)].copy()
    fl = rng.integers(0, 256, (n, 8))
    for j in range(8):
        q[np.arange(n), fl[:, j] // 8] ^= (1 << (fl[:, j] % 8)).astype(np.uint8)
    q[::10] = rng.integers(0, 256, (len(q[::10]), 32), dtype=np.uint8)  # some descriptors unrelated to the vocabulary
    for levels_up in (4, 2, 1, 0, 9):
        w, wt, nid = voc.descend(q, levels_up)
        ow, owt, onid = O.bow_transform(tree, q, max(4 - levels_up, 0))
        assert np.array_equal(w, ow) and np.array_equal(wt, owt) and np.array_equal(nid, onid), levels_up
    bow_vec, feat = voc.transform(q, levels_up=2)
    w, wt, nid = O.bow_transform(tree, q, 2)
    keep = wt > 0
    assert sorted(bow_vec) == sorted(set(w[keep].tolist())) and abs(sum(bow_vec.values()) - 1.0) < 1e-9
    assert sorted(feat) == sorted(set(nid[keep].tolist()))
    assert sum(len(v) for v in feat.values()) == keep.sum() and all(v == sorted(v) for v in feat.values())
    e = voc.descend(np.zeros((0, 32), np.uint8))
    assert len(e[0]) == 0
    # malformed vocabularies are rejected at upload
    from stella_vslam_amd._lib import SvgpuError
    with pytest.raises(SvgpuError):
        data.bow_vocabulary(ctx, [0, 1], [0], np.zeros((1, 32), np.uint8), [1.0], [0], depth=1)   # child id 0 (a cycle through the root)


def test_fbow_descent_and_maps():
    """svgpu_fbow_transform == the oracle's restatement of fbow::Vocabulary::transform (level from the root, path-code keys), and the maps the
    binding assembles in its "fbow" framework."""
    from stella_vslam_amd import data, feature
    ctx = feature.Context(0)
    rng = np.random.default_rng(2)
    tree = make_tree(rng, k=10, depth=5, prune=0.08)
    voc = data.bow_vocabulary(ctx, tree["child_off"], tree["children"], tree["node_desc"], tree["node_weight"], tree["word_id"], depth=5, framework="fbow", k=10)
    leaves = np.flatnonzero(tree["word_id"] >= 0)
    n = 5003
    q = tree["node_desc"][rng.choice(leaves, n)].copy()
    q[::9] = rng.integers(0, 256, (len(q[::9]), 32), dtype=np.uint8)
    for level in (4, 0, 2, 6):
        w, wt, key = voc.descend_fbow(q, level)
        ow, owt, okey = O.fbow_transform(tree, q, level, 10)
        assert np.array_equal(w, ow) and np.array_equal(wt, owt) and np.array_equal(key, okey), level
    bow_vec, feat = voc.transform(q, 4)
    w, wt, key = O.fbow_transform(tree, q, 4, 10)
    assert sorted(bow_vec) == sorted(set(w.tolist())) and abs(sum(v * v for v in bow_vec.values()) - 1.0) < 1e-9   # every word counted, L2 norm
    assert sorted(feat) == sorted(set(int(k) for k in key)) and sum(len(v) for v in feat.values()) == n
