// Vocabulary-tree descent of data::bow_vocabulary_util::compute_bow (data/bow_vocabulary.cc:18-24):
// per descriptor, from the root, move to the child with the smallest Hamming distance (first child wins ties), remember the
// node reached at `node_level`, stop at a leaf -> (word id, weight, node id).  The vocabulary is uploaded once and stays in
// HBM (a k = 10, L = 6 ORB vocabulary is 1.1 M nodes x 32 B = 36 MB); the accumulation into the sparse bow_vec /
// bow_feat_vec maps is the host adaptor's (they are std::map in the reference).
// 8 lanes per descriptor: lane j holds dword j of the descriptor and reads dword j of every child (one coalesced 32-byte
// row per child), distances are summed with three xor-shuffles; the upper tree levels are shared by all descriptors and live in L2.
#include "svgpu_internal.h"

struct svgpu_vocabulary {
    int device;
    int n_nodes;
    int32_t* child_off;
    int32_t* children;
    uint32_t* node_desc;
    float* node_weight;
    int32_t* word_id;
};

namespace {

inline size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }

__global__ void __launch_bounds__(256) k_bow_descend(int n, const uint32_t* __restrict__ desc, const int32_t* __restrict__ child_off,
                                                     const int32_t* __restrict__ children, const uint32_t* __restrict__ node_desc,
                                                     const float* __restrict__ node_weight, const int32_t* __restrict__ word_id,
                                                     int node_level, int32_t* __restrict__ out_word, float* __restrict__ out_weight,
                                                     int32_t* __restrict__ out_node) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = t >> 3, j = t & 7;
    const bool live = f < n;
    const uint32_t mine = live ? desc[(size_t)f * 8 + j] : 0u;
    int cur = 0, level = 0, nid = 0;
    // dead groups (f >= n) walk nothing: their loop condition is false from the start
    int beg = live ? child_off[0] : 0, end = live ? child_off[1] : 0;
    while (end > beg) {
        ++level;
        int best = 0;
        unsigned best_d = 0xFFFFFFFFu;
        for (int c = beg; c < end; ++c) {
            const int id = children[c];
            unsigned d = __popc(mine ^ node_desc[(size_t)id * 8 + j]);
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            if (d < best_d) {  // strict: the first child keeps ties
                best_d = d;
                best = id;
            }
        }
        cur = best;
        if (level == node_level) nid = cur;
        beg = child_off[cur];
        end = child_off[cur + 1];
    }
    if (live && j == 0) {
        out_word[f] = word_id[cur];
        out_weight[f] = node_weight[cur];
        out_node[f] = nid;
    }
}

// fbow::Vocabulary::transform(features, level, r, r2) (the reference's default BoW build, data/bow_vocabulary.cc:20-22): the same descent, but
// the r2 key is FBoW's: the PATH CODE (child indices, nbits each) of the node at `store_level` counted DOWN from the root, taken before the
// step of that level; a leaf met above the store level files the feature under the code of the block it was found in.
__global__ void __launch_bounds__(256) k_fbow_descend(int n, const uint32_t* __restrict__ desc, const int32_t* __restrict__ child_off,
                                                      const int32_t* __restrict__ children, const uint32_t* __restrict__ node_desc,
                                                      const float* __restrict__ node_weight, const int32_t* __restrict__ word_id, int store_level,
                                                      int nbits, int32_t* __restrict__ out_word, float* __restrict__ out_weight,
                                                      uint32_t* __restrict__ out_code) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = t >> 3, j = t & 7;
    const bool live = f < n;
    const uint32_t mine = live ? desc[(size_t)f * 8 + j] : 0u;
    int cur = 0, level = 0;
    uint32_t code = 0, key = 0;
    int beg = live ? child_off[0] : 0, end = live ? child_off[1] : 0;
    while (end > beg) {
        int best = beg;
        unsigned best_d = 0xFFFFFFFFu;
        for (int c = beg; c < end; ++c) {
            unsigned d = __popc(mine ^ node_desc[(size_t)children[c] * 8 + j]);
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            if (d < best_d) {  // strict: the first child keeps ties
                best_d = d;
                best = c;
            }
        }
        if (level == store_level) key = code;
        const int child = children[best];
        const int cb = child_off[child], ce = child_off[child + 1];
        cur = child;
        if (ce == cb) {  // a leaf ends the descent
            if (level < store_level) key = code;
            break;
        }
        code = (code << nbits) | (uint32_t)(best - beg);
        ++level;
        beg = cb;
        end = ce;
    }
    if (live && j == 0) {
        out_word[f] = word_id[cur];
        out_weight[f] = node_weight[cur];
        out_code[f] = key;
    }
}

}  // namespace

extern "C" {

int svgpu_bow_vocabulary_upload(svgpu_ctx* ctx, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                                const float* node_weight, const int32_t* word_id, svgpu_vocabulary** out) {
    if (!ctx || !out || n_nodes < 1 || !child_off || !node_desc || !node_weight || !word_id || child_off[0] != 0)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_vocabulary_upload: bad arguments");
    *out = nullptr;
    const int n_children = child_off[n_nodes];
    if (n_children < 0 || n_children > n_nodes || (n_children > 0 && !children)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_vocabulary_upload: bad tree");
    for (int i = 0; i < n_nodes; ++i)
        if (child_off[i + 1] < child_off[i]) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_vocabulary_upload: child_off must be non-decreasing");
    for (int c = 0; c < n_children; ++c)  // children point strictly downwards: no cycles, the descent terminates
        if (children[c] <= 0 || children[c] >= n_nodes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_vocabulary_upload: child id out of range");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    svgpu_vocabulary* V = new (std::nothrow) svgpu_vocabulary{};
    if (!V) return SVGPU_ERR_INVALID;
    V->device = ctx->device;
    V->n_nodes = n_nodes;
    const size_t bytes = pad((size_t)(n_nodes + 1) * 4) + pad((size_t)(n_children + 1) * 4) + pad((size_t)n_nodes * 32) + 2 * pad((size_t)n_nodes * 4);
    char* base = nullptr;
    if (hipMalloc(&base, bytes) != hipSuccess) {
        delete V;
        return sv_set_error(ctx, SVGPU_ERR_HIP, "svgpu_bow_vocabulary_upload: hipMalloc");
    }
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base + off; off += pad(b); return p; };
    V->child_off = (int32_t*)take((size_t)(n_nodes + 1) * 4);
    V->children = (int32_t*)take((size_t)(n_children + 1) * 4);
    V->node_desc = (uint32_t*)take((size_t)n_nodes * 32);
    V->node_weight = (float*)take((size_t)n_nodes * 4);
    V->word_id = (int32_t*)take((size_t)n_nodes * 4);
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemcpyAsync(V->child_off, child_off, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_children) e = hipMemcpyAsync(V->children, children, (size_t)n_children * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(V->node_desc, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(V->node_weight, node_weight, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(V->word_id, word_id, (size_t)n_nodes * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipFree(base);
        delete V;
        return sv_set_error(ctx, SVGPU_ERR_HIP, "svgpu_bow_vocabulary_upload: copy", e);
    }
    *out = V;
    return SVGPU_OK;
}

void svgpu_bow_vocabulary_free(svgpu_vocabulary* vocab) {
    if (!vocab) return;
    (void)hipSetDevice(vocab->device);
    if (vocab->child_off) (void)hipFree(vocab->child_off);  // base of the single allocation
    delete vocab;
}

static int bow_transform_impl(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int node_level, int fbow_k, int32_t* word_id,
                              float* weight, int32_t* node_id);
int svgpu_bow_transform(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int node_level, int32_t* word_id,
                        float* weight, int32_t* node_id) {
    return bow_transform_impl(ctx, vocab, desc, n, node_level, 0, word_id, weight, node_id);
}
int svgpu_fbow_transform(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int store_level, int k, int32_t* word_id,
                         float* weight, uint32_t* node_code) {
    if (k < 2 || store_level < 0) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_fbow_transform: bad arguments");
    return bow_transform_impl(ctx, vocab, desc, n, store_level, k, word_id, weight, reinterpret_cast<int32_t*>(node_code));
}
static int bow_transform_impl(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int node_level, int fbow_k, int32_t* word_id,
                              float* weight, int32_t* node_id) {
    if (!ctx || !vocab || n < 0 || vocab->device != ctx->device || (n > 0 && (!desc || !word_id || !weight || !node_id)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_transform: bad arguments");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = sv_ensure_scratch(ctx, pad((size_t)n * 32) + 3 * pad((size_t)n * 4) + 256);
    if (rc) return rc;
    char* base = (char*)ctx->d_scratch;
    uint32_t* d_desc = (uint32_t*)base;
    int32_t* d_word = (int32_t*)(base + pad((size_t)n * 32));
    float* d_w = (float*)((char*)d_word + pad((size_t)n * 4));
    int32_t* d_node = (int32_t*)((char*)d_w + pad((size_t)n * 4));
    SV_HIP(ctx, hipMemcpyAsync(d_desc, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    if (fbow_k > 0) {
        int nbits = 0;
        while ((1 << nbits) < fbow_k) ++nbits;
        hipLaunchKernelGGL(k_fbow_descend, dim3(((size_t)n * 8 + 255) / 256), dim3(256), 0, s, n, d_desc, vocab->child_off, vocab->children, vocab->node_desc,
                           vocab->node_weight, vocab->word_id, node_level, nbits, d_word, d_w, reinterpret_cast<uint32_t*>(d_node));
    }
    else
        hipLaunchKernelGGL(k_bow_descend, dim3(((size_t)n * 8 + 255) / 256), dim3(256), 0, s, n, d_desc, vocab->child_off, vocab->children, vocab->node_desc,
                           vocab->node_weight, vocab->word_id, node_level, d_word, d_w, d_node);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(word_id, d_word, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(weight, d_w, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(node_id, d_node, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

}  // extern "C"
