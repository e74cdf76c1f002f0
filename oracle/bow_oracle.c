/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Never linked or called by the product path (stella_vslam_amd/libsvgpu.so).
 *
 * Vocabulary-tree descent of data::bow_vocabulary_util::compute_bow (data/bow_vocabulary.cc:18-24):
 *   USE_DBOW2 build : bow_vocab->transform(features, bow_vec, bow_feat_vec, 4)      (DBoW2 TemplatedVocabulary, FORB)
 *   default build   : bow_vocab->transform(descriptors, 4, bow_vec, bow_feat_vec)   (FBoW fork; submodule EMPTY in /root/reference)
 * Neither library is in the container; this restates the published DBoW2 algorithm (Galvez-Lopez & Tardos, T-RO 2012;
 * TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)): from the root, at every level move to the child
 * whose descriptor has the smallest Hamming distance (FIRST child wins ties: strict "<" scan starting from children[0]), record
 * the node reached at level `node_level` (DBoW2: L - levelsup, clamped at 0 = root), stop at a leaf, return its word id and
 * weight.  PARITY UNPINNED: no vocabulary file and no vectors in the reference's tests (bow_vocabulary.cc runs only with
 * $BOW_VOCAB set).  // VERIFY-AGAINST-DBOW2 / FBOW
 *
 * Tree layout (flat): node 0 = root; children of node i = children[child_off[i] .. child_off[i+1]) (empty => leaf);
 * node_desc n_nodes x 32; node_weight, word_id per node (leaves). */
#include <stddef.h>
#include <stdint.h>

static unsigned hamming32b(const uint8_t* a, const uint8_t* b) {
    const uint32_t* pa = (const uint32_t*)a;
    const uint32_t* pb = (const uint32_t*)b;
    unsigned d = 0;
    for (int i = 0; i < 8; ++i) d += (unsigned)__builtin_popcount(pa[i] ^ pb[i]);
    return d;
}

void orc_bow_transform(int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc, const float* node_weight,
                       const int32_t* word_id, int node_level, int n, const uint8_t* desc, int32_t* out_word, float* out_weight,
                       int32_t* out_node) {
    (void)n_nodes;
    for (int f = 0; f < n; ++f) {
        const uint8_t* d = desc + 32 * (size_t)f;
        int cur = 0, level = 0, nid = 0;
        while (child_off[cur + 1] > child_off[cur]) {
            ++level;
            const int beg = child_off[cur], end = child_off[cur + 1];
            int best = children[beg];
            unsigned best_d = hamming32b(d, node_desc + 32 * (size_t)best);
            for (int c = beg + 1; c < end; ++c) {
                const unsigned dd = hamming32b(d, node_desc + 32 * (size_t)children[c]);
                if (dd < best_d) {
                    best_d = dd;
                    best = children[c];
                }
            }
            cur = best;
            if (level == node_level) nid = cur;
        }
        out_word[f] = word_id[cur];
        out_weight[f] = node_weight[cur];
        out_node[f] = nid;
    }
}

/* fbow::Vocabulary::transform(features, level, fBow& r, fBow2& r2) -- the DEFAULT build of the reference (data/bow_vocabulary.cc:20-22 calls it
 * with level = 4).  FBoW (stella-cv/FBoW, un-pinned submodule, EMPTY in /root/reference) is restated from its published sources
 * (fbow.h `_transform2`), RECALLED, not citable here:  // VERIFY-AGAINST-FBOW
 *   - the descent starts in block 0 (our virtual root's children) with level = 0, curNode = 0;
 *   - in every block the child with the smallest distance wins, FIRST child keeps ties (strict "<" from child 0);
 *   - `if (level == storeLevel) r2[curNode].push_back(feature)` BEFORE the step: `level` counts DOWN FROM THE ROOT (not DBoW2's levels-up),
 *     and the key is the PATH CODE of the block's node: curNode = (curNode << nbits) | child index, nbits = ceil(log2(k));
 *   - a leaf ends the descent: r[word id] += weight, and a leaf met ABOVE the store level files the feature under the current block's code
 *     (`if (level < storeLevel) r2[curNode].push_back(feature)`).
 * Outputs per descriptor: word id, weight, the r2 key.  (r's normalisation is the binding's: see stella_vslam_amd/data.py.) */
void orc_fbow_transform(int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc, const float* node_weight,
                        const int32_t* word_id, int store_level, int k, int n, const uint8_t* desc, int32_t* out_word, float* out_weight,
                        uint32_t* out_node_code) {
    (void)n_nodes;
    int nbits = 0;
    while ((1 << nbits) < k) ++nbits;
    for (int f = 0; f < n; ++f) {
        const uint8_t* d = desc + 32 * (size_t)f;
        int cur = 0, level = 0;
        uint32_t code = 0, key = 0;
        int have_key = 0;
        while (child_off[cur + 1] > child_off[cur]) {
            const int beg = child_off[cur], end = child_off[cur + 1];
            int best = beg;
            unsigned best_d = hamming32b(d, node_desc + 32 * (size_t)children[beg]);
            for (int c = beg + 1; c < end; ++c) {
                const unsigned dd = hamming32b(d, node_desc + 32 * (size_t)children[c]);
                if (dd < best_d) {
                    best_d = dd;
                    best = c;
                }
            }
            if (level == store_level) key = code, have_key = 1;
            const int child = children[best];
            if (child_off[child + 1] == child_off[child]) { /* leaf */
                if (level < store_level) key = code, have_key = 1;
                cur = child;
                break;
            }
            code = (code << nbits) | (uint32_t)(best - beg);
            ++level;
            cur = child;
        }
        (void)have_key;
        out_word[f] = word_id[cur];
        out_weight[f] = node_weight[cur];
        out_node_code[f] = key;
    }
}
