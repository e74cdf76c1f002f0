// Host side of the ORB front end: geometry / coefficient tables (bit-identical to the reference's
// host arithmetic), device workspace layout, launch sequence.
//   orb_params scale tables        feature/orb_params.cc:41-71
//   level sizes                    feature/orb_extractor.cc:157-159
//   cv::resize coefficient tables  OpenCV 4.x imgproc/src/resize.cpp (8-bit fixed point, 11-bit coefficients)
//   FAST cell lattice              feature/orb_extractor.cc:179-217
//   selection grid                 feature/orb_extractor.cc:292-305
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "svgpu_internal.h"

#pragma clang fp contract(off)
#pragma STDC FP_CONTRACT OFF

namespace {

inline int cv_floor_f(float v) {
    int i = (int)v;
    return i - (i > v);
}
inline int cv_round_f(float v) { return (int)lrintf(v); }  // round half to even (default rounding mode)
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <class T>
int upload(svgpu_ctx* ctx, T** dptr, const std::vector<T>& v) {
    if (*dptr) {
        SV_HIP(ctx, hipFree(*dptr));
        *dptr = nullptr;
    }
    const size_t n = v.empty() ? 1 : v.size();
    SV_HIP(ctx, hipMalloc((void**)dptr, n * sizeof(T)));
    if (!v.empty()) SV_HIP(ctx, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return SVGPU_OK;
}

template <class T>
void free_dev(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace

void sv_orb_release(svgpu_ctx* ctx) {
    free_dev(ctx->d_levels);
    free_dev(ctx->d_cells);
    free_dev(ctx->d_xofs);
    free_dev(ctx->d_xa);
    free_dev(ctx->d_yofs);
    free_dev(ctx->d_yb);
    free_dev(ctx->d_xg);
    free_dev(ctx->d_yrow);
    free_dev(ctx->d_band_rows);
    free_dev(ctx->d_gtab);
    free_dev(ctx->d_pyr);
    free_dev(ctx->d_blur);
    free_dev(ctx->d_keys);
    free_dev(ctx->d_sel);
    free_dev(ctx->d_cellpos);
    free_dev(ctx->d_dbands);
    free_dev(ctx->d_img);
    free_dev(ctx->d_mask);
    free_dev(ctx->d_kps);
    free_dev(ctx->d_desc);
    free_dev(ctx->d_counts);
    ctx->orb.configured = false;
    ctx->last_extract_n = -1;  // d_kps / d_desc are gone: nothing left to adopt (svgpu_frame_adopt_extraction)
    ctx->last_batch = 0;
}

extern "C" {

int svgpu_orb_scale_tables(float scale_factor, int num_levels, float* scale_factors, float* inv_scale_factors,
                           float* level_sigma_sq, float* inv_level_sigma_sq) {
    if (num_levels < 1) return SVGPU_ERR_INVALID;
    // orb_params.cc:41-71 -- four independent fp32 recurrences
    float s = 1.0f, inv = 1.0f;
    for (int l = 0; l < num_levels; ++l) {
        if (l > 0) {
            s = scale_factor * s;
            inv = (1.0f / scale_factor) * inv;
        }
        if (scale_factors) scale_factors[l] = s;
        if (inv_scale_factors) inv_scale_factors[l] = inv;
        if (level_sigma_sq) level_sigma_sq[l] = l == 0 ? 1.0f : s * s;
        if (inv_level_sigma_sq) inv_level_sigma_sq[l] = l == 0 ? 1.0f : 1.0f / (s * s);
    }
    return SVGPU_OK;
}

int svgpu_orb_configure(svgpu_ctx* ctx, int width, int height, int max_batch, float scale_factor, int num_levels,
                        int ini_fast_thr, int min_fast_thr, unsigned min_area) {
    if (!ctx) return SVGPU_ERR_INVALID;
    if (width < 8 || height < 8 || width > 16384 || height > 16384 || max_batch < 1 || num_levels < 1
        || num_levels > SV_MAX_LEVELS || !(scale_factor > 1.0f))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_orb_configure: bad geometry/parameters");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sv_orb_release(ctx);
    OrbConfig& C = ctx->orb;
    C = OrbConfig();
    C.width = width;
    C.height = height;
    C.max_batch = max_batch;
    C.blur_rows = max_batch <= BLUR_SMALL_BATCH ? BLUR_ROWS_SMALL : BLUR_ROWS;
    C.num_levels = num_levels;
    C.scale_factor = scale_factor;
    C.ini_thr = ini_fast_thr < 0 ? 0 : (ini_fast_thr > 255 ? 255 : ini_fast_thr);  // cv::FAST clamps (fast.cpp)
    C.min_thr = min_fast_thr < 0 ? 0 : (min_fast_thr > 255 ? 255 : min_fast_thr);
    C.min_area_sqrt = (unsigned)std::sqrt((double)min_area);  // orb_extractor.cc:20 (unsigned member)
    svgpu_orb_scale_tables(scale_factor, num_levels, C.scale_factors, nullptr, nullptr, nullptr);

    std::vector<short> xofs;
    std::vector<short2> xa, yofs, yb;
    std::vector<unsigned short> gtab;
    std::vector<uint32_t> xg;   // packed column-group records of k_pyramid_lds
    std::vector<short4> yrow;
    bool xg_ok = true;          // false: a level shrinks by more than 3x, the byte windows of the records do not fit
    size_t pyr_off = 0, blur_off = 0;
    int grid_first = 0, btile_first = 0;
    std::vector<std::pair<int, int>> grid_rows[SV_MAX_LEVELS];  // per level and selection-grid row: first / last level row its keypoints can lie on
    for (int l = 0; l < num_levels; ++l) {
        OrbLevel& L = C.levels[l];
        memset(&L, 0, sizeof(L));
        const float s = C.scale_factors[l];
        if (l == 0) {
            L.w = width;
            L.h = height;
        }
        else {  // orb_extractor.cc:157-159
            const double scale = (double)s;
            L.w = (int)std::round(width * 1.0 / scale);
            L.h = (int)std::round(height * 1.0 / scale);
        }
        if (L.w < 2 || L.h < 2) return sv_set_error(ctx, SVGPU_ERR_INVALID, "pyramid level smaller than 2 px");
        L.pitch = (int)align_up(L.w, 64);
        L.scale = s;
        L.kp_size = (float)(unsigned)(31u * s);  // orb_extractor.cc:274
        L.blur_off = (long long)blur_off;
        blur_off += align_up((size_t)L.pitch * L.h, 256);
        if (l > 0) {
            L.pyr_off = (long long)pyr_off;
            pyr_off += align_up((size_t)L.pitch * L.h, 256);
            // ---- resize tables (level l from level l-1)
            const OrbLevel& P = C.levels[l - 1];
            const double inv_scale_x = (double)L.w / P.w, inv_scale_y = (double)L.h / P.h;
            const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
            L.xtab_off = (int)xofs.size();
            L.ytab_off = (int)yofs.size();
            for (int dx = 0; dx < L.w; ++dx) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_f(fx);
                fx -= sx;
                if (sx < 0) {
                    fx = 0;
                    sx = 0;
                }
                if (sx >= P.w - 1) {
                    fx = 0;
                    sx = P.w - 1;
                }
                xofs.push_back((short)sx);
                short2 a;
                a.x = (short)cv_round_f((1.f - fx) * 2048);
                a.y = (short)cv_round_f(fx * 2048);
                xa.push_back(a);
            }
            for (int dy = 0; dy < L.h; ++dy) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_f(fy);
                fy -= sy;
                short2 o, bb;
                o.x = (short)(sy < 0 ? 0 : (sy > P.h - 1 ? P.h - 1 : sy));
                o.y = (short)(sy + 1 < 0 ? 0 : (sy + 1 > P.h - 1 ? P.h - 1 : sy + 1));
                bb.x = (short)cv_round_f((1.f - fy) * 2048);
                bb.y = (short)cv_round_f(fy * 2048);
                yofs.push_back(o);
                yb.push_back(bb);
            }
            // ---- packed records for k_pyramid_lds.  One record per 4 output columns c..c+3 (columns past w-1 repeat w-1):
            //      w0 = dword index of sx[c] | dword index of sx[c+2] << 16 (two 8-byte windows per source row),
            //      w1 = byte index inside its window of sx[c], sx[c+1] (window 0), sx[c+2], sx[c+3] (window 1),
            //      w2..w5 = (a0 | a1 << 16) of the four columns.  sx+1 is always the next byte (its weight is 0 when clamped).
            L.xg_off = (int)(xg.size() / 8);
            for (int c = 0; c < L.w; c += 4) {
                int sxs[4];
                for (int i = 0; i < 4; ++i) sxs[i] = xofs[L.xtab_off + std::min(c + i, L.w - 1)];
                const int base0 = sxs[0] >> 2, base2 = sxs[2] >> 2;
                const int k[4] = {sxs[0] - 4 * base0, sxs[1] - 4 * base0, sxs[2] - 4 * base2, sxs[3] - 4 * base2};
                for (int i = 0; i < 4; ++i) xg_ok = xg_ok && k[i] >= 0 && k[i] <= 6;
                xg.push_back((uint32_t)base0 | ((uint32_t)base2 << 16));
                xg.push_back((uint32_t)(k[0] & 255) | ((uint32_t)(k[1] & 255) << 8) | ((uint32_t)(k[2] & 255) << 16) | ((uint32_t)(k[3] & 255) << 24));
                for (int i = 0; i < 4; ++i) {
                    const short2 a = xa[L.xtab_off + std::min(c + i, L.w - 1)];
                    xg_ok = xg_ok && a.x >= 0 && a.y >= 0;
                    xg.push_back((uint32_t)(unsigned short)a.x | ((uint32_t)(unsigned short)a.y << 16));
                }
                xg.push_back(0);
                xg.push_back(0);
            }
            for (int dy = 0; dy < L.h; ++dy) {
                const short2 o = yofs[L.ytab_off + dy], c = yb[L.ytab_off + dy];
                xg_ok = xg_ok && c.x >= 0 && c.y >= 0;
                yrow.push_back(make_short4(o.x, o.y, c.x, c.y));
            }
        }
        // ---- blur tiles
        L.btile_first = btile_first;
        L.btiles_x = (L.w + BLUR_TW - 1) / BLUR_TW;
        L.btiles_y = (L.h + 4 * C.blur_rows - 1) / (4 * C.blur_rows);
        btile_first += L.btiles_x * L.btiles_y + ((L.h + 7) / 8 + 63) / 64;  // + edge tiles (64 strips of BLUR_EDGE_ROWS rows each)
        // ---- FAST cell lattice and selection grid
        L.cell_first = (int)C.cells.size();
        L.grid_first = grid_first;
        L.gtab_x_off = L.gtab_y_off = (int)gtab.size();
        if (L.w > 2 * SV_PATCH_RADIUS && L.h > 2 * SV_PATCH_RADIUS) {
            L.has_cells = 1;
            const unsigned min_bx = SV_PATCH_RADIUS, min_by = SV_PATCH_RADIUS;
            const unsigned max_bx = L.w - SV_PATCH_RADIUS, max_by = L.h - SV_PATCH_RADIUS;
            const unsigned rw = max_bx - min_bx, rh = max_by - min_by;
            const unsigned num_cols = rw / SV_CELL + 1, num_rows = rh / SV_CELL + 1;
            L.cells_x = (int)num_cols;
            for (unsigned i = 0; i < num_rows; ++i) {
                const unsigned min_y = min_by + i * SV_CELL;
                if (max_by - SV_OVERLAP <= min_y) continue;
                unsigned max_y = min_y + SV_CELL + SV_OVERLAP;
                if (max_by < max_y) max_y = max_by;
                for (unsigned j = 0; j < num_cols; ++j) {
                    const unsigned min_x = min_bx + j * SV_CELL;
                    if (max_bx - SV_OVERLAP <= min_x) continue;
                    unsigned max_x = min_x + SV_CELL + SV_OVERLAP;
                    if (max_bx < max_x) max_x = max_bx;
                    FastCell c;
                    c.min_x = (short)min_x;
                    c.min_y = (short)min_y;
                    c.w = (short)(max_x - min_x);
                    c.h = (short)(max_y - min_y);
                    c.lv = (short)l;
                    c.cj = (short)j;
                    c.order_base = (int)((i * num_cols + j) << 14);
                    C.cells.push_back(c);
                }
            }
            // distribute_keypoints (:292-305)
            const double scaled_min_area_sqrt = C.min_area_sqrt / s;  // fp32 division, widened
            const unsigned gx = (unsigned)std::ceil((int)rw / scaled_min_area_sqrt);
            const unsigned gy = (unsigned)std::ceil((int)rh / scaled_min_area_sqrt);
            const double delta_x = (double)(int)rw / gx, delta_y = (double)(int)rh / gy;
            L.grid_x = (int)gx;
            L.grid_y = (int)gy;
            L.gtab_x_off = (int)gtab.size();
            for (unsigned x = 0; x < rw; ++x) {
                unsigned ix = (unsigned)((float)x / delta_x);
                gtab.push_back((unsigned short)(ix < gx ? ix : gx - 1));
            }
            L.gtab_y_off = (int)gtab.size();
            grid_rows[l].assign(gy, std::make_pair(1 << 30, -1));
            for (unsigned y = 0; y < rh; ++y) {
                unsigned iy = (unsigned)((float)y / delta_y);
                iy = iy < gy ? iy : gy - 1;
                gtab.push_back((unsigned short)iy);
                grid_rows[l][iy].first = std::min(grid_rows[l][iy].first, (int)(min_by + y));
                grid_rows[l][iy].second = std::max(grid_rows[l][iy].second, (int)(min_by + y));
            }
            grid_first += (int)(gx * gy);
        }
        L.cell_count = (int)C.cells.size() - L.cell_first;
    }
    C.total_grid = grid_first;
    C.total_btiles = btile_first;
    // ---- bands of k_describe_bands: as many consecutive selection-grid rows of a level as fit the LDS budget (and DB_MAX_KP = 128 keypoints).
    //      A keypoint of grid row g lies on level rows [first(g), last(g)]; its patches need rows y - 15 .. y + 16 (un-blurred; row y + 16 carries
    //      zero weights but is read) and y - 18 .. y + 18 (blurred).  LDS pitch: the level width rounded up to 16-byte pieces, then to 32 mod 64
    //      (eight rows of dword reads then fall into 64 different banks); pieces beyond the level's own pitch read the next row's first bytes.
    {
        C.dbands.clear();
        C.dband_lds_bytes = 0;
        size_t budget = 48 * 1024;  // three 512-thread workgroups per CU
        if (const char* e = getenv("SVGPU_DESC_BAND_KB")) budget = (size_t)std::max(1, atoi(e)) * 1024;
        const size_t hard_limit = 80 * 1024;  // two workgroups per CU; wider images than that take k_describe
        bool ok = C.total_grid > 0 && getenv("SVGPU_DESCRIBE_LEGACY") == nullptr;
        for (int l = 0; l < num_levels && ok; ++l) {
            const OrbLevel& L = C.levels[l];
            if (!L.has_cells) continue;
            int lp = (L.w + 15) / 16 * 16;
            while (lp % 64 != 32) lp += 16;
            if (lp > 32000) ok = false;
            const int gy = L.grid_y, gx = L.grid_x;
            auto band_bytes = [&](int g0, int g1, DescBand* out) {  // rows of grid rows [g0, g1)
                int y0 = 1 << 30, y1 = -1;
                for (int g = g0; g < g1; ++g)
                    if (grid_rows[l][g].second >= 0) {
                        y0 = std::min(y0, grid_rows[l][g].first);
                        y1 = std::max(y1, grid_rows[l][g].second);
                    }
                if (y1 < 0) y0 = y1 = SV_PATCH_RADIUS;  // (grid rows no level row maps to: no keypoints either)
                const int nru = y1 - y0 + 32, nrb = y1 - y0 + 37;
                const int rpi = std::max(1, 64 / (lp / 16));  // a staging instruction carries whole groups of rpi rows: room for the last group
                const size_t bytes = (size_t)((std::max(nru, nrb) + rpi - 1) / rpi * rpi) * lp;
                if (out) {
                    out->lv = (short)l;
                    out->lp = (short)lp;
                    out->yu0 = (short)(y0 - 15);
                    out->nru = (short)nru;
                    out->yb0 = (short)(y0 - 18);
                    out->nrb = (short)nrb;
                    out->img_bytes = (int)bytes;
                    out->cell0 = L.grid_first + g0 * gx;
                    out->cell1 = L.grid_first + g1 * gx;
                }
                return bytes;
            };
            if (gx > 128) ok = false;
            for (int g0 = 0; g0 < gy && ok;) {
                int g1 = g0 + 1;
                if (band_bytes(g0, g1, nullptr) > hard_limit) ok = false;
                while (g1 < gy && (g1 + 1 - g0) * gx <= 128 && band_bytes(g0, g1 + 1, nullptr) <= budget) ++g1;
                DescBand bd;
                C.dband_lds_bytes = std::max(C.dband_lds_bytes, band_bytes(g0, g1, &bd));
                C.dbands.push_back(bd);
                g0 = g1;
            }
        }
        if (!ok) C.dbands.clear();
        // heaviest bands first (level 0 stages the most bytes per keypoint): the tail of the launch is made of the light ones
        std::stable_sort(C.dbands.begin(), C.dbands.end(), [](const DescBand& a, const DescBand& b) { return a.img_bytes > b.img_bytes; });
        if (!C.dbands.empty()) C.dband_lds_bytes += 2 * 128 * sizeof(int2);  // + the per-keypoint arrays (DB_MAX_KP)
    }
    C.pyr_frame_bytes = pyr_off ? pyr_off : 256;
    C.blur_frame_bytes = blur_off;

    // ---- pyramid bands: band k owns rows [k*h/K, (k+1)*h/K) of every level; bottom-up it also needs the source rows
    //      of everything it computes at the next level (two taps per output row, clamped -- the yofs table).
    //      K is the smallest count (>= 16) whose per-band LDS footprint fits k_pyramid_lds; none fits -> global variant.
    auto make_bands = [&](int bands, std::vector<int2>& band_rows) -> size_t {
        band_rows.assign((size_t)bands * num_levels, int2{0, 0});
        size_t worst = 0;
        for (int k = 0; k < bands; ++k) {
            int need_lo = 0, need_hi = 0;
            for (int l = num_levels - 1; l >= 1; --l) {
                const OrbLevel& Lv = C.levels[l];
                int lo = (int)((long long)k * Lv.h / bands), hi = (int)((long long)(k + 1) * Lv.h / bands);
                if (l < num_levels - 1 && need_hi > need_lo) {
                    lo = std::min(lo, need_lo);
                    hi = std::max(hi, need_hi);
                }
                band_rows[(size_t)k * num_levels + l] = int2{lo, hi};
                // rows of level l-1 read by rows [lo, hi) of level l
                need_lo = 1 << 30;
                need_hi = 0;
                for (int dy = lo; dy < hi; ++dy) {
                    const short2 o = yofs[Lv.ytab_off + dy];
                    need_lo = std::min(need_lo, (int)o.x);
                    need_hi = std::max(need_hi, (int)o.y + 1);
                }
            }
            if (need_hi > need_lo) band_rows[(size_t)k * num_levels] = int2{need_lo, need_hi};  // level-0 rows the band reads
            size_t size_a = 0, size_b = 0;  // must mirror the LDS map of k_pyramid_lds: odd / even levels (level 0 included) alternate in two regions
            for (int l = 0; l < num_levels; ++l) {
                const int2 r = band_rows[(size_t)k * num_levels + l];
                const size_t b = (size_t)(r.y - r.x) * (size_t)((C.levels[l].w + 3) & ~3);
                if (l & 1) size_a = std::max(size_a, b);
                else size_b = std::max(size_b, b);
            }
            size_a = (size_a + 15) & ~(size_t)15;
            size_b = (size_b + 15) & ~(size_t)15;
            size_t bytes = size_a + size_b;
            for (int l = 1; l < num_levels; ++l) {
                const int2 r = band_rows[(size_t)k * num_levels + l];
                bytes += (size_t)(r.y - r.x) * 8;
            }
            worst = std::max(worst, bytes);
        }
        return worst;
    };
    std::vector<int2> band_rows;
    // few, tall bands recompute the fewest halo rows; small batches need more bands to fill the 256 CUs.  A footprint of at most half the
    // CU's LDS lets two workgroups share a CU (one computes while the other waits at a level barrier): preferred while <= 32 bands reach it.
    int bands = std::max(8, std::min(32, (512 + max_batch - 1) / std::max(max_batch, 1)));
    bool forced = false;
    if (const char* e = getenv("SVGPU_PYR_BANDS")) {
        bands = std::max(1, atoi(e));
        forced = true;
    }
    xg_ok = xg_ok && (num_levels < 2 || C.levels[1].w <= 4 * 1024);  // one thread per column group of a level: at most 1024 groups
    size_t lds = 0;
    if (!forced && xg_ok) {
        std::vector<int2> trial;
        for (int k = bands; k <= 32; k += 2)
            if (make_bands(k, trial) <= SV_PYR_LDS_HALF) {
                bands = k;
                break;
            }
    }
    for (;; bands += 2) {
        lds = make_bands(bands, band_rows);
        if (lds <= SV_PYR_LDS_MAX && xg_ok) break;
        if (bands >= 256 || !xg_ok) {  // very wide images: chain the levels through global memory instead (k_pyramid)
            bands = 16;
            make_bands(bands, band_rows);
            lds = 0;
            break;
        }
    }
    ctx->pyr_bands = bands;
    ctx->pyr_lds_bytes = lds;
    if (lds) SV_HIP(ctx, sv_pyramid_prepare());
    {
        int rcb;
        if ((rcb = upload(ctx, &ctx->d_band_rows, band_rows))) return rcb;
    }
    std::vector<OrbLevel> lv(C.levels, C.levels + num_levels);
    int rc;
    if ((rc = upload(ctx, &ctx->d_levels, lv))) return rc;
    if ((rc = upload(ctx, &ctx->d_cells, C.cells))) return rc;
    if ((rc = upload(ctx, &ctx->d_xofs, xofs))) return rc;
    if ((rc = upload(ctx, &ctx->d_xa, xa))) return rc;
    if ((rc = upload(ctx, &ctx->d_yofs, yofs))) return rc;
    if ((rc = upload(ctx, &ctx->d_yb, yb))) return rc;
    if ((rc = upload(ctx, &ctx->d_xg, xg))) return rc;
    if ((rc = upload(ctx, &ctx->d_yrow, yrow))) return rc;
    if ((rc = upload(ctx, &ctx->d_gtab, gtab))) return rc;
    if (!C.dbands.empty()) {
        if ((rc = upload(ctx, &ctx->d_dbands, C.dbands))) return rc;
        SV_HIP(ctx, sv_describe_bands_prepare(C.dband_lds_bytes));
    }
    const size_t B = (size_t)max_batch, G = (size_t)(C.total_grid > 0 ? C.total_grid : 1);
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_pyr, B * C.pyr_frame_bytes + 256));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_blur, B * C.blur_frame_bytes + 4096));  // + slack: the band kernel's row pieces run past the last row's end
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_cellpos, B * (G + 1) * sizeof(int32_t)));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_keys, B * G * sizeof(unsigned long long)));
    SV_HIP(ctx, hipMemset(ctx->d_keys, 0, B * G * sizeof(unsigned long long)));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_sel, (B * G + 8) * sizeof(int4)));  // + slack: k_describe reads whole groups of DESC_KPW entries
    // staging for the single-frame host entry point
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_img, (size_t)C.levels[0].pitch * height));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_mask, (size_t)C.levels[0].pitch * height));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_kps, G * sizeof(svgpu_keypoint)));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_desc, G * 32));
    SV_HIP(ctx, hipMalloc((void**)&ctx->d_counts, (1 + SV_MAX_LEVELS) * sizeof(int32_t)));
    C.configured = true;
    return SVGPU_OK;
}

int svgpu_orb_max_keypoints(const svgpu_ctx* ctx) { return (ctx && ctx->orb.configured) ? ctx->orb.total_grid : -1; }

int svgpu_orb_level_size(const svgpu_ctx* ctx, int level, int* width, int* height) {
    if (!ctx || !ctx->orb.configured) return SVGPU_ERR_NOT_CONFIGURED;
    if (level < 0 || level >= ctx->orb.num_levels) return SVGPU_ERR_INVALID;
    if (width) *width = ctx->orb.levels[level].w;
    if (height) *height = ctx->orb.levels[level].h;
    return SVGPU_OK;
}

int svgpu_orb_extract_batch_device(svgpu_ctx* ctx, const uint8_t* imgs_dev, int batch, size_t frame_stride,
                                   int row_stride, const uint8_t* mask_dev, size_t mask_frame_stride,
                                   int mask_row_stride, svgpu_keypoint* kps_dev, uint8_t* desc_dev, int cap,
                                   int32_t* counts_dev, void* stream) {
    return svgpu_orb_extract_batch_device_angles(ctx, imgs_dev, batch, frame_stride, row_stride, mask_dev, mask_frame_stride, mask_row_stride, kps_dev, desc_dev, cap,
                                                 counts_dev, nullptr, stream);
}

int svgpu_orb_extract_batch_device_angles(svgpu_ctx* ctx, const uint8_t* imgs_dev, int batch, size_t frame_stride,
                                          int row_stride, const uint8_t* mask_dev, size_t mask_frame_stride,
                                          int mask_row_stride, svgpu_keypoint* kps_dev, uint8_t* desc_dev, int cap,
                                          int32_t* counts_dev, float* angles_dev, void* stream) {
    if (!ctx) return SVGPU_ERR_INVALID;
    OrbConfig& C = ctx->orb;
    if (!C.configured) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_orb_configure has not been called");
    if (!imgs_dev || !kps_dev || !desc_dev || !counts_dev || batch < 1 || batch > C.max_batch || cap < 1
        || row_stride < C.width)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_orb_extract_batch_device: bad arguments");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ctx->last_extract_n = -1;  // whatever the last svgpu_orb_extract left behind is stale from here on (it sets the count again after its own call)
    const int Lc = C.num_levels;
    // 1. pyramid: chained bilinear resize (level l from level l-1), all levels in one launch (banded, see k_pyramid)
    if (Lc > 1) {
        SvProfScope ps(ctx, s, "k_resize");
        sv_launch_pyramid(s, ctx->d_levels, Lc, ctx->d_band_rows, ctx->pyr_bands, imgs_dev, frame_stride, row_stride, ctx->d_pyr,
                          C.pyr_frame_bytes, ctx->d_xofs, ctx->d_xa, ctx->d_yofs, ctx->d_yb, ctx->d_xg, ctx->d_yrow, batch, ctx->pyr_lds_bytes);
    }
    // 2. blurred copy of every level -- on the auxiliary stream, beside steps 3-4: the blur waits on memory where FAST is bound by
    //    instruction issue, so the two share the CUs well; step 5 joins them
    static const bool fork_blur = getenv("SVGPU_FORK_BLUR") != nullptr;  // opt-in, see DESIGN.md section 6
    hipStream_t sb = (ctx->stream_aux && fork_blur) ? ctx->stream_aux : s;
    if (sb != s) {
        SV_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
        SV_HIP(ctx, hipStreamWaitEvent(sb, ctx->ev_fork, 0));
    }
    {
        SvProfScope ps(ctx, sb, "k_blur");
        // levels the streaming kernel cannot take (caller image not 4-byte aligned, level narrower than 16 px) -> gather kernel
        bool need_gather = (((size_t)imgs_dev | (size_t)frame_stride | (size_t)row_stride) & 3) != 0;
        for (int l = 0; l < Lc; ++l) need_gather = need_gather || C.levels[l].w < 16;
        sv_launch_blur(sb, ctx->d_levels, Lc, C.total_btiles, imgs_dev, frame_stride, row_stride, ctx->d_pyr, C.pyr_frame_bytes,
                       ctx->d_blur, C.blur_frame_bytes, batch, need_gather, C.blur_rows);
    }
    if (sb != s) SV_HIP(ctx, hipEventRecord(ctx->ev_join, sb));
    // 3. FAST per cell + selection-grid arg-max
    SV_HIP(ctx, hipEventRecord(ctx->ev_stage[0], s));
    {
        SvProfScope ps(ctx, s, "k_fast");
        sv_launch_fast(s, ctx->d_levels, Lc, ctx->d_cells, (int)C.cells.size(), imgs_dev, frame_stride, row_stride, ctx->d_pyr,
                   C.pyr_frame_bytes, ctx->d_gtab, ctx->d_keys, C.total_grid, C.ini_thr, C.min_thr, mask_dev,
                       mask_frame_stride, mask_row_stride, C.width, C.height, batch);
    }
    // 4. ordered compaction (+ key reset for the next call)
    {
        SvProfScope ps(ctx, s, "k_select");
        sv_launch_select(s, ctx->d_levels, Lc, ctx->d_keys, C.total_grid, ctx->d_sel, counts_dev, C.dbands.empty() ? nullptr : ctx->d_cellpos, batch);
    }
    // 5. orientation, descriptor, scale correction
    if (sb != s) SV_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    SV_HIP(ctx, hipEventRecord(ctx->ev_stage[1], s));
    ctx->stage_recorded = true;
    SvProfScope ps(ctx, s, "k_describe");
    // bands when the batch alone fills the chip (they are bound by throughput: 0.80 against 1.20 ms per 1 024 frames); for a few frames the
    // per-keypoint kernel's many short workgroups finish sooner (one frame: 11 against 17 us, break-even near 16 frames of 2 400 keypoints)
    const bool bands = !C.dbands.empty() && ((long long)batch * C.total_grid >= 32768 || getenv("SVGPU_DESCRIBE_BANDS") != nullptr);
    if (bands)
        sv_launch_describe_bands(s, ctx->d_levels, Lc, ctx->d_dbands, (int)C.dbands.size(), C.dband_lds_bytes, ctx->d_sel, C.total_grid, ctx->d_cellpos,
                                 counts_dev, imgs_dev, frame_stride, row_stride, ctx->d_pyr, C.pyr_frame_bytes, ctx->d_blur, C.blur_frame_bytes,
                                 kps_dev, desc_dev, cap, batch, angles_dev);
    else
        sv_launch_describe(s, ctx->d_levels, Lc, ctx->d_sel, C.total_grid, counts_dev, imgs_dev, frame_stride, row_stride,
                           ctx->d_pyr, C.pyr_frame_bytes, ctx->d_blur, C.blur_frame_bytes, kps_dev, desc_dev, cap, batch, angles_dev);
    SV_HIP(ctx, hipGetLastError());
    ctx->last_batch = batch;
    ctx->last_imgs = imgs_dev;
    ctx->last_frame_stride = frame_stride;
    ctx->last_row_stride = row_stride;
    return SVGPU_OK;
}

int svgpu_orb_stream_wait_stage(svgpu_ctx* ctx, int stage, void* stream) {
    if (!ctx || stage < 0 || stage > 1 || !stream) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_orb_stream_wait_stage: bad arguments");
    if (!ctx->stage_recorded) return SVGPU_OK;  // no extraction enqueued yet: nothing to wait for
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_stage[stage], 0));
    return SVGPU_OK;
}

int svgpu_orb_extract(svgpu_ctx* ctx, const uint8_t* img, int stride, const uint8_t* mask, int mask_stride,
                      svgpu_keypoint* kps, uint8_t* desc, int cap, int* n_out, int* level_counts) {
    if (!ctx) return SVGPU_ERR_INVALID;
    OrbConfig& C = ctx->orb;
    if (!C.configured) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_orb_configure has not been called");
    if (!img || !n_out || stride < C.width || cap < 0 || (cap > 0 && (!kps || !desc)) || (mask && mask_stride < C.width))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_orb_extract: bad arguments");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int pitch = C.levels[0].pitch;
    SV_HIP(ctx, hipMemcpy2DAsync(ctx->d_img, pitch, img, stride, C.width, C.height, hipMemcpyHostToDevice, s));
    if (mask) SV_HIP(ctx, hipMemcpy2DAsync(ctx->d_mask, pitch, mask, mask_stride, C.width, C.height, hipMemcpyHostToDevice, s));
    const int icap = C.total_grid > 0 ? C.total_grid : 1;
    int rc = svgpu_orb_extract_batch_device(ctx, ctx->d_img, 1, (size_t)pitch * C.height, pitch, mask ? ctx->d_mask : nullptr, 0,
                                            pitch, ctx->d_kps, ctx->d_desc, icap, ctx->d_counts, s);
    if (rc) return rc;
    int32_t counts[1 + SV_MAX_LEVELS];
    SV_HIP(ctx, hipMemcpyAsync(counts, ctx->d_counts, (1 + C.num_levels) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    const int n = counts[0];
    *n_out = n;
    ctx->last_extract_n = n < icap ? n : icap;
    if (level_counts)
        for (int l = 0; l < C.num_levels; ++l) level_counts[l] = counts[1 + l];
    const int m = n < cap ? n : cap;
    if (m > 0) {
        SV_HIP(ctx, hipMemcpyAsync(kps, ctx->d_kps, (size_t)m * sizeof(svgpu_keypoint), hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipMemcpyAsync(desc, ctx->d_desc, (size_t)m * 32, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
    }
    return n > cap ? sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_orb_extract: more keypoints than cap") : SVGPU_OK;
}

static int download_level(svgpu_ctx* ctx, const uint8_t* base, size_t frame_bytes, long long off, int frame, int level,
                          uint8_t* dst, int dst_stride) {
    OrbConfig& C = ctx->orb;
    const OrbLevel& L = C.levels[level];
    if (!dst || dst_stride < L.w) return sv_set_error(ctx, SVGPU_ERR_INVALID, "download: bad destination");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SV_HIP(ctx, hipMemcpy2D(dst, dst_stride, base + (size_t)frame * frame_bytes + off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return SVGPU_OK;
}

int svgpu_orb_pyramid_download(svgpu_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride) {
    if (!ctx) return SVGPU_ERR_INVALID;
    OrbConfig& C = ctx->orb;
    if (!C.configured || ctx->last_batch == 0) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "no extract call yet");
    if (frame < 0 || frame >= ctx->last_batch || level < 0 || level >= C.num_levels)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "pyramid_download: bad frame/level");
    if (level == 0) {
        if (!dst || dst_stride < C.width) return sv_set_error(ctx, SVGPU_ERR_INVALID, "download: bad destination");
        SV_HIP(ctx, hipSetDevice(ctx->device));
        SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SV_HIP(ctx, hipMemcpy2D(dst, dst_stride, ctx->last_imgs + (size_t)frame * ctx->last_frame_stride, ctx->last_row_stride,
                                C.width, C.height, hipMemcpyDeviceToHost));
        return SVGPU_OK;
    }
    return download_level(ctx, ctx->d_pyr, C.pyr_frame_bytes, C.levels[level].pyr_off, frame, level, dst, dst_stride);
}

int svgpu_orb_blurred_download(svgpu_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride) {
    if (!ctx) return SVGPU_ERR_INVALID;
    OrbConfig& C = ctx->orb;
    if (!C.configured || ctx->last_batch == 0) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "no extract call yet");
    if (frame < 0 || frame >= ctx->last_batch || level < 0 || level >= C.num_levels)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "blurred_download: bad frame/level");
    return download_level(ctx, ctx->d_blur, C.blur_frame_bytes, C.levels[level].blur_off, frame, level, dst, dst_stride);
}

}  // extern "C"
