"""CPU: the oracle's vocabulary-tree descent (oracle/bow_oracle.c) against a literal Python walk (DBoW2
TemplatedVocabulary::transform semantics: min Hamming child, first wins ties, node id at level L - levelsup)."""
import numpy as np

from oracle import oracle as O


def make_tree(rng, k=10, depth=3, prune=0.1, dup=0.05):
    """Random k-ary vocabulary of the given depth; a fraction of the inner nodes is cut short (unbalanced tree: leaf above the
    last level) and a fraction of the children repeats a sibling's descriptor (distance ties)."""
    desc, weight, word, kids = [np.zeros(32, np.uint8)], [0.0], [-1], [[]]
    frontier = [(0, 0)]
    while frontier:
        node, lvl = frontier.pop(0)
        if lvl == depth or (lvl > 0 and rng.uniform() < prune):
            continue
        for c in range(k):
            d = desc[node].copy()
            flips = rng.integers(0, 256, 40 >> lvl if lvl < 3 else 5)
            for b in flips:
                d[b // 8] ^= 1 << (b % 8)
            if c > 0 and rng.uniform() < dup:
                d = desc[kids[node][rng.integers(0, c)]].copy()
            desc.append(d)
            weight.append(0.0)
            word.append(-1)
            kids.append([])
            kids[node].append(len(desc) - 1)
            frontier.append((len(desc) - 1, lvl + 1))
    nw = 0
    for i, ch in enumerate(kids):
        if not ch:
            word[i] = nw
            weight[i] = float(rng.uniform(0.0, 5.0)) if rng.uniform() > 0.02 else 0.0  # a few stop words
            nw += 1
    off = np.concatenate([[0], np.cumsum([len(c) for c in kids])]).astype(np.int32)
    children = np.array([c for ch in kids for c in ch], np.int32)
    return dict(child_off=off, children=children, node_desc=np.array(desc, np.uint8), node_weight=np.array(weight, np.float32),
                word_id=np.array(word, np.int32), depth=depth, kids=kids)


def literal(tree, d, node_level):
    bits = lambda a: np.unpackbits(a)
    cur, lvl, nid = 0, 0, 0
    while tree["kids"][cur]:
        lvl += 1
        ch = tree["kids"][cur]
        best, bd = ch[0], int((bits(d) != bits(tree["node_desc"][ch[0]])).sum())
        for c in ch[1:]:
            dd = int((bits(d) != bits(tree["node_desc"][c])).sum())
            if dd < bd:
                best, bd = c, dd
        cur = best
        if lvl == node_level:
            nid = cur
    return tree["word_id"][cur], tree["node_weight"][cur], nid


def test_descent_against_literal_walk():
    rng = np.random.default_rng(0)
    tree = make_tree(rng, k=10, depth=3)
    leaves = np.flatnonzero(tree["word_id"] >= 0)
    q = tree["node_desc"][rng.choice(leaves, 400)].copy()
    fl = rng.integers(0, 256, (400, 6))
    for j in range(6):
        q[np.arange(400), fl[:, j] // 8] ^= (1 << (fl[:, j] % 8)).astype(np.uint8)
    for node_level in (0, 1, 2, 3, 5):
        w, wt, nid = O.bow_transform(tree, q, node_level)
        for i in range(0, 400, 7):
            lw, lwt, ln = literal(tree, q[i], node_level)
            assert (w[i], wt[i], nid[i]) == (lw, lwt, ln)
        if node_level in (0, 5):
            assert (nid == 0).all()          # level 0 = root; a level deeper than the tree is never reached
    w, wt, nid = O.bow_transform(tree, q, 1)
    assert len(np.unique(nid)) >= 8 and len(np.unique(w)) > 100
    # a single-node vocabulary: every feature maps to the root word
    one = dict(child_off=[0, 0], children=np.zeros(0, np.int32), node_desc=np.zeros((1, 32), np.uint8), node_weight=[1.5], word_id=[0])
    w, wt, nid = O.bow_transform(one, q[:3], 2)
    assert (w == 0).all() and (wt == 1.5).all() and (nid == 0).all()


def test_fbow_file_round_trip(tmp_path):
    """.fbow reader (stella_vslam_amd/data.py:read_fbow, layout restated from fbow/vocabulary.h -- unverified against a real file):
    a written vocabulary reads back to the identical flat tree, whatever the alignment, and bad files are refused."""
    import pytest
    from stella_vslam_amd import data
    rng = np.random.default_rng(3)
    tree = make_tree(rng, k=6, depth=3, prune=0.15)
    for align in (8, 32):
        p = str(tmp_path / f"v{align}.fbow")
        data.write_fbow(p, tree["child_off"], tree["children"], tree["node_desc"], tree["node_weight"], tree["word_id"], k=6, alignment=align)
        back = data.read_fbow(p)
        for key in ("child_off", "children", "node_desc", "node_weight", "word_id"):
            assert np.array_equal(back[key], tree[key]), key
        assert back["depth"] == 3 and back["k"] == 6 and back["desc_name"] == "orb"
        q = rng.integers(0, 256, (50, 32), dtype=np.uint8)
        a, b = O.bow_transform(back, q, 2), O.bow_transform(tree, q, 2)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    raw = np.fromfile(p, np.uint8)
    bad = raw.copy()
    bad[0] ^= 1
    bad.tofile(str(tmp_path / "bad.fbow"))
    with pytest.raises(ValueError, match="signature"):
        data.read_fbow(str(tmp_path / "bad.fbow"))
    raw[:len(raw) - 100].tofile(str(tmp_path / "short.fbow"))
    with pytest.raises(ValueError, match="truncated"):
        data.read_fbow(str(tmp_path / "short.fbow"))


def literal_fbow(tree, d, store_level, k):
    """fbow::Vocabulary::_transform2 as a plain walk: the r2 key is the path code of the current block's node, filed at the store level
    (counted from the root) or, when a leaf comes first, under the block the leaf was found in."""
    nbits = int(np.ceil(np.log2(k)))
    bits = lambda a: np.unpackbits(a)
    cur, level, code, key = 0, 0, 0, 0
    while True:
        ch = tree["kids"][cur]
        dist = [int((bits(d) != bits(tree["node_desc"][c])).sum()) for c in ch]
        bi = int(np.argmin(dist))  # argmin returns the FIRST minimum
        if level == store_level:
            key = code
        child = ch[bi]
        if not tree["kids"][child]:
            if level < store_level:
                key = code
            return int(tree["word_id"][child]), float(tree["node_weight"][child]), key
        code = (code << nbits) | bi
        level += 1
        cur = child


def test_fbow_transform_restated():
    """The FBoW form of compute_bow (the reference's default build): level from the root, path-code keys, the leaf-above-the-store-level rule."""
    rng = np.random.default_rng(11)
    tree = make_tree(rng, k=6, depth=5, prune=0.15)
    leaves = np.flatnonzero(tree["word_id"] >= 0)
    q = tree["node_desc"][rng.choice(leaves, 300)].copy()
    q[::7] = rng.integers(0, 256, (len(q[::7]), 32), dtype=np.uint8)
    early = 0
    for store_level in (0, 1, 2, 4, 7):
        w, wt, key = O.fbow_transform(tree, q, store_level, 6)
        for i in range(len(q)):
            lw, lwt, lkey = literal_fbow(tree, q[i], store_level, 6)
            assert (w[i], np.float32(wt[i]), int(key[i])) == (lw, np.float32(lwt), lkey), (store_level, i)
        if store_level == 4:
            # path codes of depth-4 nodes use 3 bits per level: codes of features that stopped early are shorter
            early = int((key < (1 << 9)).sum())
    assert early > 0  # the pruned tree does have leaves above level 4
    # same words as the DBoW2 form (the descent is the same), different keys
    w2, _, _ = O.bow_transform(tree, q, 1)
    w, _, _ = O.fbow_transform(tree, q, 4, 6)
    assert np.array_equal(w, w2)
