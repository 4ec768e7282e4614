// scan_brute.hip -- the brute-force engine (one lane per needle, brute.hpp): tiling and launch of one pass.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "brute.hpp"

namespace
{
template <int NW>
void launch_brute_nw(const spm_patterns *ps, const brute_params &P, dim3 grid, dim3 block, size_t lds,
                     hipStream_t stream, bool cutoff)
{
    if (cutoff) {
        hipFuncSetAttribute((const void *)myers_cutoff_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_cutoff_kernel<NW>), grid, block, lds, stream, P);
    } else if (ps->algo == SPM_ALGO_MYERS) {
        hipFuncSetAttribute((const void *)myers_brute_kernel<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_brute_kernel<NW, false>), grid, block, lds, stream, P);
    } else if (ps->algo == SPM_ALGO_MYERS_PREFIX) {
        hipFuncSetAttribute((const void *)myers_brute_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((myers_brute_kernel<NW, true>), grid, block, lds, stream, P);
    } else {
        hipFuncSetAttribute((const void *)shiftor_brute_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        hipLaunchKernelGGL((shiftor_brute_kernel<NW>), grid, block, lds, stream, P);
    }
}

void launch_brute(const spm_patterns *ps, const brute_params &P, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                  bool cutoff)
{
    switch (ps->NW) {
    case 1: launch_brute_nw<1>(ps, P, grid, block, lds, s, cutoff); break;
    case 2: launch_brute_nw<2>(ps, P, grid, block, lds, s, cutoff); break;
    case 4: launch_brute_nw<4>(ps, P, grid, block, lds, s, cutoff); break;
    case 8: launch_brute_nw<8>(ps, P, grid, block, lds, s, cutoff); break;
    case 16: launch_brute_nw<16>(ps, P, grid, block, lds, s, cutoff); break;
    case 32: launch_brute_nw<32>(ps, P, grid, block, lds, s, cutoff); break;
    default: launch_brute_nw<64>(ps, P, grid, block, lds, s, cutoff); break;
    }
}

} // namespace

// one brute-force pass; `report` = false suppresses hits (state-only pass)
int run_brute(const scan_args &A, uint64_t begin, uint64_t end, uint64_t ctx_begin, const uint32_t *d_state_in,
              uint32_t *d_state_out, bool report, bool single_tile)
{
    spm_ctx *ctx = A.ctx;
    const spm_patterns *ps = A.ps;
    brute_params P{};
    P.text = A.text->d;
    P.text_alloc = A.text->owned ? A.text->alloc : A.text->n;
    P.scan_begin = begin;
    P.scan_end = end;
    P.ctx_begin = ctx_begin;
    P.pos_offset = A.opts.pos_offset;
    P.n_groups = ps->n_groups;
    P.warm = ps->max_window > 0 ? ps->max_window - 1 : 0;
    P.sigma = ps->sigma;
    P.has_state = d_state_in ? 1 : 0;
    P.peq = ps->d_peq;
    P.hp0 = ps->d_hp0;
    P.m = ps->d_m;
    P.k = ps->d_k;
    P.state_in = d_state_in;
    P.state_out = d_state_out;
    P.hits = A.hits->d_hits;
    P.counters = A.hits->d_count + (report ? 0 : 3); // a state-only pass counts into a dummy slot
    P.hit_cap = report ? A.hits->cap : 0;
    if (A.tiles) { // re-scan of the filter's overflowed spans: report only what its verification has not reported
        P.seen = A.d_seen;
        P.seen_mask = A.seen_mask;
        P.overflow = A.hits->d_count + 2;
    }

    const size_t lds_per_wave = (size_t)(ps->sigma + 1) * ps->NW * 64 * sizeof(uint32_t);
    uint32_t wpw = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / lds_per_wave));
    const size_t lds = lds_per_wave * wpw;
    if (lds > 160 * 1024) {
        SPM_SET_ERR(ctx, "needle set needs %zu bytes of LDS per wave (sigma=%u, %u words); limit 160 KiB", lds,
                    ps->sigma, ps->NW);
        return SPM_E_UNSUPPORTED;
    }
    const uint64_t range = end - begin;
    uint32_t grid = ps->n_groups * std::max(1u, (uint32_t)(ctx->n_cu * 8) / (ps->n_groups * 1));
    grid = std::max(grid, ps->n_groups);
    // grid counts workgroups; keep (grid * wpw) % n_groups == 0 so that a wave keeps its needle group in LDS
    const uint64_t n_waves = (uint64_t)grid * wpw;
    uint64_t tile;
    if (single_tile || ps->algo == SPM_ALGO_MYERS_PREFIX) {
        tile = (range + 3) & ~3ull;
    } else {
        const uint64_t min_tile = std::max<uint64_t>(2048, (((uint64_t)P.warm * 16) + 255) & ~255ull);
        tile = (range * ps->n_groups) / (n_waves * 4) + 1;
        tile = (tile + 255) & ~255ull;
        tile = std::max(tile, min_tile);
        tile = std::min<uint64_t>(tile, 1u << 20);
    }
    if (tile == 0)
        tile = 4;
    P.tile = (uint32_t)std::min<uint64_t>(tile, 0xFFFFFF00u);
    P.n_tiles = (uint32_t)std::max<uint64_t>(1, (range + P.tile - 1) / P.tile);
    if (A.tiles) {
        P.n_tiles = (uint32_t)(A.tiles->size() / 3);
        uint64_t *d_tab = nullptr;
        SPM_HIP_CHECK(ctx, hipMalloc(&d_tab, A.tiles->size() * sizeof(uint64_t)));
        hipFree(A.hits->d_aux[0]);
        A.hits->d_aux[0] = d_tab;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_tab, A.tiles->data(), A.tiles->size() * sizeof(uint64_t),
                                          hipMemcpyHostToDevice, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        P.tile_tab = d_tab;
    } else if (!A.seg_offsets && A.d_seg_offsets) {
        // brute-force run over a device-resident segment table (fallback of the journaled-sequence search)
        scan_args &W = const_cast<scan_args &>(A);
        W.seg_host.resize(A.n_segments + 1);
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(W.seg_host.data(), A.d_seg_offsets, (A.n_segments + 1) * sizeof(uint64_t),
                                          hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        W.seg_offsets = W.seg_host.data();
    }
    if (A.seg_offsets && !A.tiles) {
        // every segment is its own haystack: tiles never cross a segment, warm-up stays inside it
        std::vector<uint64_t> tab;
        for (uint64_t s = 0; s < A.n_segments; ++s) {
            const uint64_t sb = A.seg_offsets[s], se = A.seg_offsets[s + 1];
            for (uint64_t lo = sb; lo < se; lo += P.tile) {
                const uint64_t hi = std::min<uint64_t>(lo + P.tile, se);
                tab.push_back(lo >= sb + P.warm ? lo - P.warm : sb);
                tab.push_back(lo);
                tab.push_back(hi);
            }
        }
        if (tab.empty()) { // only empty segments
            tab = {begin, begin, begin};
        }
        if (tab.size() / 3 > 0xFFFFFFFFull) {
            SPM_SET_ERR(ctx, "segmented scan: too many tiles");
            return SPM_E_UNSUPPORTED;
        }
        P.n_tiles = (uint32_t)(tab.size() / 3);
        uint64_t *d_tab = nullptr;
        SPM_HIP_CHECK(ctx, hipMalloc(&d_tab, tab.size() * sizeof(uint64_t)));
        hipFree(A.hits->d_aux[0]);
        A.hits->d_aux[0] = d_tab;
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice,
                                          ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // `tab` is a host temporary
        P.tile_tab = d_tab;
    }
    if (single_tile && P.n_tiles != 1) {
        SPM_SET_ERR(ctx, "internal: single-tile pass over %llu symbols", (unsigned long long)range);
        return SPM_E_INVALID;
    }
    const uint64_t n_items = (uint64_t)P.n_tiles * P.n_groups;
    const uint32_t need_wg = (uint32_t)std::min<uint64_t>((n_items + wpw - 1) / wpw, grid);
    // shrinking the grid must keep the group<->wave affinity: round up to a multiple of n_groups when possible
    uint32_t launch_grid = need_wg;
    if (launch_grid < grid) {
        const uint32_t q = (launch_grid + ps->n_groups - 1) / ps->n_groups * ps->n_groups;
        launch_grid = std::min(grid, std::max(q, 1u));
    }
    // Ukkonen cut-off kernel for stateless Myers scans (SPM_HIP_BRUTE_CUTOFF=0 selects the full-width kernel)
    const bool cutoff = ps->algo == SPM_ALGO_MYERS && !d_state_in && !d_state_out && ps->d_peq_bot &&
                        A.tune.brute_cutoff != 0;
    if (cutoff)
        P.peq = ps->d_peq_bot;
    launch_brute(ps, P, dim3(launch_grid), dim3(64 * wpw), lds, ctx->stream, cutoff);
    SPM_HIP_CHECK(ctx, hipGetLastError());
    A.hits->stats.main_launches++;
    return SPM_OK;
}


void spm_warm_brute_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)myers_cutoff_kernel<4>);
}
