// bsmm_plan.h -- host-side schedules ("plans") of the grouped / windowed kernels, derived from the reference-format lookup
// tables.  Pure host code (no HIP calls): the caller uploads the resulting int32 array to the device like any other
// lookup table and passes it in bsmm_args.plan.  Segmentation / lock ids of the source table are irrelevant here (every
// plan kernel has one writer per output).  Formats: updat plan 'BSUP', xcol plan 'BSXC' (bsize 32, 16-bit), xcolf plan
// 'BSXF' (bsize 32, fp32), xcol16 plan 'BSX6' (bsize 16); each is described in front of its builder.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <array>
#include <vector>


// =================================================================================================
// updat plan: work items for the windowed weight-gradient kernel (bsmm_updat_win.h).
// The (CB x KB) block grid is cut into UW x UW windows; the nonzero blocks of a window are dealt to the UP_WAVES waves
// of a workgroup, up to UP_MAXB per wave (windows with more blocks are split into several items).  Blocks are sorted
// by (c, k) before dealing so that a wave's consecutive slots often share the X fragment (flag bit 9).
// Item order: workgroup b runs on XCD b % 8 (observed; speed only), so position 8*j + x holds the j-th item of XCD x,
// and XCD x is given a compact PATCH of the window grid: the items of a patch share their X slabs (same window row) and
// DY slabs (same window column) through that XCD's L2.  Shorter lists are padded with empty items.
//
// Layout (int32):  [0] magic 'BSUP'  [1] version  [2] UW  [3] UP_MAXB  [4] nitems (multiple of 8)  [5] nblocks
//                  [6] off_items  [7] UP_WAVES  [8] off_rows: word offset of the 'BSU6' section behind the items (bsize 16 on feature axis 0,
//                  round 5: build_updat16_rows_section below; 0 = none)  [9..11] 0
//   item: UP_ITEM = 4 + UP_WAVES*UP_MAXB*2 words = (c0_block, k0_block, nblocks_in_item | cmask << 16, nslots | kmask << 16)
//         then for wave v, slot j:
//         (nslots = slots every wave of the item walks = max blocks owned by one wave; empty slots have meta = 0;
//          cmask / kmask = window rows / columns that hold a block of the item: the kernels stage only those slab parts)
//         (meta, w)   meta = cidx | kidx << 4 | 1 << 8 (valid) | sameX << 9     (cidx/kidx = block offset inside the window)
// =================================================================================================
namespace bsmm {

constexpr int32_t UPLAN_MAGIC = 0x42535550;
constexpr int32_t UPLAN_VERSION = 6;
constexpr int UW = 8;          // bsize 32: 8x8-block windows, 4 slots per wave
constexpr int UP_WAVES = 8;
constexpr int UP_MAXB = 4;
constexpr int UP_HDR = 12;
constexpr int UP_ITEM = 4 + UP_WAVES * UP_MAXB * 2;
constexpr int UW16 = 16;       // bsize 16: 16x16-block windows (the same 256x256 features), 8 slots per wave
constexpr int UP16_MAXB = 8;
constexpr int UP16_ITEM = 4 + UP_WAVES * UP16_MAXB * 2;

// uw = window side in blocks (256 / bsize), maxb = slots per wave (4 for bsize 32, 8 for bsize 16)
inline long build_updat_plan(const int32_t* updat_lut, int blocks, int CB, int KB, int uw, int maxb, int32_t* out, int waves = 8) {
    if (!updat_lut || blocks <= 0 || CB <= 0 || KB <= 0 || uw < 1 || uw > 16 || maxb < 1 || waves < 1) return -1;
    const int UW = uw, UP_MAXB = maxb, UP_WAVES = waves, UP_ITEM = 4 + waves * maxb * 2;
    const int wc = (CB + UW - 1) / UW, wk = (KB + UW - 1) / UW;
    struct Ent { int c, k, w; };
    std::vector<std::vector<Ent>> win((size_t)wc * wk);
    for (int w = 0; w < blocks; ++w) {
        const int c = updat_lut[2 * w], k = updat_lut[2 * w + 1];
        if (c < 0 || c >= CB || k < 0 || k >= KB) return -1;
        win[(size_t)(c / UW) * wk + (k / UW)].push_back({c, k, w});
    }
    // patch grid pr x pc = 8 XCDs, as square as the window grid allows
    int pr = 4, pc = 2;
    if (wc < 4) { pr = wc >= 2 ? 2 : 1; pc = 8 / pr; }
    if (wk < pc) { pc = wk >= 1 ? std::min(wk, pc) : 1; }
    std::vector<std::vector<int32_t>> per_xcd(8);
    const int cap = UP_WAVES * UP_MAXB;
    for (int wi = 0; wi < wc; ++wi)
        for (int wj = 0; wj < wk; ++wj) {
            auto& v = win[(size_t)wi * wk + wj];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end(), [](const Ent& a, const Ent& b) { return a.c != b.c ? a.c < b.c : a.k < b.k; });
            const int xcd = ((wi * pr / wc) * pc + (wj * pc / wk)) & 7;
            for (size_t beg = 0; beg < v.size(); beg += cap) {
                const int n = (int)std::min<size_t>(cap, v.size() - beg);
                std::vector<int32_t> it(UP_ITEM, 0);
                const int per = (n + UP_WAVES - 1) / UP_WAVES;
                it[0] = wi * UW; it[1] = wj * UW; it[2] = n; it[3] = per;   // contiguous runs per wave keep equal-c blocks together
                uint32_t cmask = 0, kmask = 0;                              // window rows / columns this item touches
                for (int e = 0; e < n; ++e) {
                    const int wave = e / per, slot = e % per;
                    const Ent& en = v[beg + e];
                    cmask |= 1u << (en.c - wi * UW);
                    kmask |= 1u << (en.k - wj * UW);
                    const bool same = slot > 0 && v[beg + e - 1].c == en.c;
                    int32_t* p = &it[4 + (wave * UP_MAXB + slot) * 2];
                    p[0] = (en.c - wi * UW) | ((en.k - wj * UW) << 4) | (1 << 8) | (same ? (1 << 9) : 0);
                    p[1] = en.w;
                }
                it[2] |= (int32_t)(cmask << 16);
                it[3] |= (int32_t)(kmask << 16);
                per_xcd[xcd].insert(per_xcd[xcd].end(), it.begin(), it.end());
            }
        }
    // (round 6) A hub layout puts most of its items into the patches of its first window row / column: the launch is as long as the longest
    // of the eight lists (Barabasi-Albert(256, 14) at bsize 16: 316 us against 130 for a uniform layout of that density).  Lists longer than the
    // mean spill their last items to the shortest lists -- those items lose their patch's sharing in the L2, the launch loses its idle XCDs.
    {
        size_t total_items = 0, longest0 = 0;
        for (auto& l : per_xcd) { total_items += l.size() / UP_ITEM; longest0 = std::max(longest0, l.size() / UP_ITEM); }
        const size_t target = (total_items + 7) / 8;
        if (longest0 > target + target / 4 && target > 0) {
            for (int x = 0; x < 8; ++x)
                while (per_xcd[x].size() / UP_ITEM > target) {
                    int to = 0;
                    for (int y = 1; y < 8; ++y) if (per_xcd[y].size() < per_xcd[to].size()) to = y;
                    if (per_xcd[to].size() / UP_ITEM >= target) break;
                    per_xcd[to].insert(per_xcd[to].end(), per_xcd[x].end() - UP_ITEM, per_xcd[x].end());
                    per_xcd[x].resize(per_xcd[x].size() - UP_ITEM);
                }
        }
    }
    size_t longest = 0;
    for (auto& l : per_xcd) longest = std::max(longest, l.size() / UP_ITEM);
    const long nitems = (long)longest * 8;
    const long total = UP_HDR + nitems * UP_ITEM;
    if (out) {
        const int32_t hdr[UP_HDR] = {UPLAN_MAGIC, UPLAN_VERSION, UW, UP_MAXB, (int32_t)nitems, blocks, UP_HDR, UP_WAVES, 0, 0, 0, 0};
        std::copy(hdr, hdr + UP_HDR, out);
        std::fill(out + UP_HDR, out + total, 0);
        for (int x = 0; x < 8; ++x)
            for (size_t j = 0; j < per_xcd[x].size() / UP_ITEM; ++j)
                std::copy(per_xcd[x].begin() + j * UP_ITEM, per_xcd[x].begin() + (j + 1) * UP_ITEM, out + UP_HDR + (j * 8 + x) * UP_ITEM);
    }
    return total;
}

// -------------------------------------------------------------------------------------------------
// 'BSU6' section: work items of the row-owner weight-gradient kernel for bsize 16 on feature axis 0 (bsmm_updat16_rows.h, round 5).
// A window is U6_WC = 32 block rows (512 features of X) x WK block columns (WK = 32 or 16: 512 / 256 features of DY).  The DY rows of a
// window go through LDS (one slab per 64-wide minibatch chunk, shared by all waves); the X rows do NOT: wave v OWNS up to U6_ROWS block
// rows of the window and loads their fragments straight into registers (on feature axis 0 a fragment is 16 contiguous bytes of a row),
// and every block of those rows is its.  Rows are dealt to the waves longest first onto the least loaded wave (at most U6_ROWS rows and
// U6_MAXB blocks per wave); a wave's blocks are sorted by (row slot, k).  WK = 32 unless some wave of some window would hold more than
// U6_MAXB blocks, then 16; no section when that fails too (dense layouts keep the 256 x 256 windows of the items above).
// Items in window order, row-major: consecutive items share X rows, items WKn apart share DY rows (the launcher keeps runs of consecutive
// items on one XCD).
//   section (int32): [0] magic 'BSU6' [1] version [2] U6_WC [3] WK [4] nitems [5] U6_WAVES | U6_ROWS << 8 | U6_MAXB << 16 [6] U6_ITEM [7] 0
//   item: (c0_block, k0_block, nblocks, 0) then per wave U6_WAVE words:
//         [0] the wave's block rows inside the window, 8 bits each (0xff = none)
//         [1] nb | ii0 << 8 | ni << 16: the wave's blocks; its DMA duty: instructions ii0 .. ii0 + ni - 1 of the 2 WK (1 KiB = 8 rows each) that stage
//             a DY slab -- dealt to the waves with the fewest blocks (a block weighs U6_DMA_PER_BLOCK instructions): a wave stalls at issue while
//             the memory pipeline is full, so the requesting is done by the waves the matrix work leaves idle
//         [2 + j / 2] 16 bits per block j (low half first): column inside the window | row slot << 8
//         [2 + U6_MAXB / 2 + j] weight block id of block j
// -------------------------------------------------------------------------------------------------
constexpr int32_t U6PLAN_MAGIC = 0x42535536;
constexpr int32_t U6PLAN_VERSION = 3;
constexpr int U6_WC = 32, U6_WAVES = 16, U6_ROWS = 2, U6_MAXB = 12, U6_HDR = 8;
constexpr int U6_WAVE = 2 + U6_MAXB / 2 + U6_MAXB;
#ifndef U6_DMA_WEIGHT
#define U6_DMA_WEIGHT 3
#endif
constexpr int U6_DMA_PER_BLOCK = U6_DMA_WEIGHT;
constexpr int U6_ITEM = 4 + U6_WAVES * U6_WAVE;

inline long build_updat16_rows_section(const int32_t* updat_lut, int blocks, int CB, int KB, int32_t* out) {
    if (!updat_lut || blocks <= 0 || CB <= 0 || KB <= 0) return -1;
    struct Ent { int c, k, w; };
    for (int WK = 32; WK >= 16; WK /= 2) {
        const int wc = (CB + U6_WC - 1) / U6_WC, wk = (KB + WK - 1) / WK;
        std::vector<std::vector<Ent>> win((size_t)wc * wk);
        for (int w = 0; w < blocks; ++w) {
            const int c = updat_lut[2 * w], k = updat_lut[2 * w + 1];
            if (c < 0 || c >= CB || k < 0 || k >= KB) return -1;
            win[(size_t)(c / U6_WC) * wk + (k / WK)].push_back({c, k, w});
        }
        std::vector<int32_t> items;
        bool fits = true;
        for (int wi = 0; wi < wc && fits; ++wi)
            for (int wj = 0; wj < wk && fits; ++wj) {
                auto& v = win[(size_t)wi * wk + wj];
                if (v.empty()) continue;
                int cnt[U6_WC] = {0};
                for (const Ent& e : v) ++cnt[e.c - wi * U6_WC];
                int order[U6_WC];
                for (int r = 0; r < U6_WC; ++r) order[r] = r;
                std::stable_sort(order, order + U6_WC, [&](int a, int b) { return cnt[a] > cnt[b]; });
                int load[U6_WAVES] = {0}, nrows[U6_WAVES] = {0}, rows[U6_WAVES][U6_ROWS];
                int owner[U6_WC], slot_of[U6_WC];
                for (int i = 0; i < U6_WC; ++i) {
                    const int r = order[i];
                    owner[r] = -1; slot_of[r] = 0;
                    if (cnt[r] == 0) continue;
                    int best = -1;
                    for (int v2 = 0; v2 < U6_WAVES; ++v2)
                        if (nrows[v2] < U6_ROWS && (best < 0 || load[v2] < load[best])) best = v2;
                    if (best < 0 || load[best] + cnt[r] > U6_MAXB) { fits = false; break; }
                    owner[r] = best; slot_of[r] = nrows[best];
                    rows[best][nrows[best]++] = r;
                    load[best] += cnt[r];
                }
                if (!fits) break;
                std::vector<int32_t> it(U6_ITEM, 0);
                it[0] = wi * U6_WC; it[1] = wj * WK; it[2] = (int32_t)v.size();
                std::sort(v.begin(), v.end(), [&](const Ent& a, const Ent& b) {
                    const int ra = a.c - wi * U6_WC, rb = b.c - wi * U6_WC;
                    if (owner[ra] != owner[rb]) return owner[ra] < owner[rb];
                    if (slot_of[ra] != slot_of[rb]) return slot_of[ra] < slot_of[rb];
                    return a.k < b.k;
                });
                // DMA duties: one instruction at a time to the wave whose blocks + duties weigh least
                int duty[U6_WAVES] = {0}, first[U6_WAVES];
                for (int i = 0; i < 2 * WK; ++i) {
                    int best = 0;
                    for (int v2 = 1; v2 < U6_WAVES; ++v2)
                        if (U6_DMA_PER_BLOCK * load[v2] + duty[v2] < U6_DMA_PER_BLOCK * load[best] + duty[best]) best = v2;
                    ++duty[best];
                }
                for (int v2 = 0, at = 0; v2 < U6_WAVES; ++v2) { first[v2] = at; at += duty[v2]; }
                size_t pos = 0;
                for (int v2 = 0; v2 < U6_WAVES; ++v2) {
                    int32_t* wv = &it[4 + v2 * U6_WAVE];
                    uint32_t rw = 0;
                    for (int sl = 0; sl < U6_ROWS; ++sl) rw |= (uint32_t)(sl < nrows[v2] ? rows[v2][sl] : 0xff) << (8 * sl);
                    wv[0] = (int32_t)rw;
                    int nb = 0;
                    while (pos < v.size() && owner[v[pos].c - wi * U6_WC] == v2) {
                        const Ent& e = v[pos++];
                        const uint32_t m = (uint32_t)(e.k - wj * WK) | ((uint32_t)slot_of[e.c - wi * U6_WC] << 8);
                        wv[2 + nb / 2] |= (int32_t)(m << (16 * (nb & 1)));
                        wv[2 + U6_MAXB / 2 + nb] = e.w;
                        ++nb;
                    }
                    wv[1] = nb | (first[v2] << 8) | (duty[v2] << 16);
                }
                items.insert(items.end(), it.begin(), it.end());
            }
        if (!fits) continue;
        const long nitems = (long)(items.size() / U6_ITEM);
        const long total = U6_HDR + nitems * U6_ITEM;
        if (out) {
            const int32_t hdr[U6_HDR] = {U6PLAN_MAGIC, U6PLAN_VERSION, U6_WC, WK, (int32_t)nitems, U6_WAVES | (U6_ROWS << 8) | (U6_MAXB << 16), U6_ITEM, 0};
            std::copy(hdr, hdr + U6_HDR, out);
            std::copy(items.begin(), items.end(), out + U6_HDR);
        }
        return total;
    }
    return 0;
}

}  // namespace bsmm

// =================================================================================================
// xcol plan: schedule of the "wave owns an output column" xprop kernel (bsmm_xcol.h), feature axis 1.
// XC_G consecutive output blocks form a group; wave v of a workgroup owns output block first+v.  The group walks the
// union of its input-block PAIRS in ascending order; for every step and wave the table says which weight block (if
// any) multiplies the even / odd half of the pair.
// Layout (int32): [0] magic 'BSXC' [1] version [2] XC_G [3] ngroups [4] nsteps_total [5] off_groups [6] off_pairs
//                 [7] off_wtab [8] n_out_blocks
//   groups[ngroups][4] = (step_off, nsteps, first_out_block, n_out_blocks_in_group)
//   pairs [nsteps_total]          pair index p of each step (input blocks 2p, 2p+1)
//   wtab  [group][wave][half][t]  weight block id or -1; group base = off_wtab + 2*XC_G*step_off, index (2*wave+half)*nsteps + t
// =================================================================================================
namespace bsmm {

constexpr int32_t XCPLAN_MAGIC = 0x42535843;
constexpr int32_t XCPLAN_VERSION = 1;
constexpr int XC_G = 8;
constexpr int XC_HDR = 12;

inline long build_xcol_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int32_t* out, int G = XC_G) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0) return -1;
    const int ngroups = (n_out_blocks + G - 1) / G;
    struct E { int p, slot, w; };   // slot = 2*wave + half
    std::vector<std::vector<E>> per_group(ngroups);
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            per_group[ob / G].push_back({c >> 1, 2 * (ob % G) + (c & 1), w});
        }
    }
    std::vector<int32_t> groups, pairs, wtab;
    for (int g = 0; g < ngroups; ++g) {
        auto& v = per_group[g];
        std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.p != b.p ? a.p < b.p : a.slot < b.slot; });
        std::vector<int32_t> gp0;
        for (auto& e : v) if (gp0.empty() || gp0.back() != e.p) gp0.push_back(e.p);
        const int ns = (int)gp0.size();
        const int step_off = (int)pairs.size();
        // (Starting every group at a different pair, to spread the L2 channels, measured slower: 373 vs 399 TF -- it destroys
        //  the L2 reuse between the groups of one row tile.)
        const std::vector<int32_t>& gp = gp0;
        std::vector<int32_t> tab((size_t)2 * G * ns, -1);
        int t0 = -1, cur = -1;
        for (auto& e : v) {
            if (e.p != cur) { cur = e.p; ++t0; }
            tab[(size_t)e.slot * ns + t0] = e.w;
        }
        pairs.insert(pairs.end(), gp.begin(), gp.end());
        wtab.insert(wtab.end(), tab.begin(), tab.end());
        groups.insert(groups.end(), {step_off, ns, g * G, std::min(G, n_out_blocks - g * G)});
    }
    const long total = XC_HDR + (long)groups.size() + (long)pairs.size() + (long)wtab.size();
    if (out) {
        const int off_groups = XC_HDR, off_pairs = off_groups + (int)groups.size(), off_wtab = off_pairs + (int)pairs.size();
        const int32_t hdr[XC_HDR] = {XCPLAN_MAGIC, XCPLAN_VERSION, G, ngroups, (int32_t)pairs.size(), off_groups, off_pairs, off_wtab,
                                     n_out_blocks, 0, 0, 0};
        std::copy(hdr, hdr + XC_HDR, out);
        std::copy(groups.begin(), groups.end(), out + off_groups);
        std::copy(pairs.begin(), pairs.end(), out + off_pairs);
        std::copy(wtab.begin(), wtab.end(), out + off_wtab);
    }
    return total;
}

}  // namespace bsmm


// Which (group, position) owns an output block.  Default: G consecutive output blocks per group.  Where that leaves the groups unbalanced (the busiest
// holds > 1.15 x the mean number of blocks: hub layouts such as the reference's Barabasi-Albert bench layouts, whose oldest nodes are neighbouring
// columns -- the workgroups of that group then take 1.5 - 2.5 x as long as the others and the pass ends when they do), adjacent PAIRS of output
// blocks are dealt to the groups heaviest first, each to the lightest group with room (round 6).  A column still sums its blocks in table order
// on one wave: results are bit-identical.  Returns whether the layout was regrouped; false on a malformed table (grp_of left empty).
namespace bsmm {
inline bool regroup_output_blocks(const int32_t* lut, int segments, int n_out_blocks, int G, bool consecutive, std::vector<int>& grp_of, std::vector<int>& pos_of) {
    const int ngroups = (n_out_blocks + G - 1) / G;
    grp_of.assign(n_out_blocks, 0); pos_of.assign(n_out_blocks, 0);
    for (int ob = 0; ob < n_out_blocks; ++ob) { grp_of[ob] = ob / G; pos_of[ob] = ob % G; }
    if (consecutive || ngroups <= 1) return false;
    std::vector<long> cnt_ob(n_out_blocks, 0), gl(ngroups, 0);
    long total = 0;
    for (int s = 0; s < segments; ++s) {
        const int32_t cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) { grp_of.clear(); return false; }
        cnt_ob[ob] += cnt; gl[ob / G] += cnt; total += cnt;
    }
    const long mx = *std::max_element(gl.begin(), gl.end());
    if (total <= 0 || (double)mx * ngroups <= 1.15 * (double)total) return false;
    const int npairs = (n_out_blocks + 1) / 2;
    std::vector<int> order(npairs);
    for (int i = 0; i < npairs; ++i) order[i] = i;
    auto wt = [&](int i) { return cnt_ob[2 * i] + (2 * i + 1 < n_out_blocks ? cnt_ob[2 * i + 1] : 0); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wt(a) > wt(b); });
    std::vector<long> load(ngroups, 0);
    std::vector<int> used(ngroups, 0);
    for (int i : order) {
        int best = -1;
        for (int g = 0; g < ngroups; ++g)
            if (used[g] + 2 <= G && (best < 0 || load[g] < load[best])) best = g;
        if (best < 0) { grp_of.clear(); return false; }              // (cannot happen: ngroups * G / 2 >= npairs)
        for (int k = 0; k < 2; ++k) {
            const int ob = 2 * i + k;
            if (ob < n_out_blocks) { grp_of[ob] = best; pos_of[ob] = used[best] + k; }
        }
        used[best] += 2; load[best] += wt(i);
    }
    return true;
}
}  // namespace bsmm

// =================================================================================================
// staged xcol plan ('BSX2'): schedule of the kernel that stages the weight blocks through LDS as well (bsmm_xcol_v2.h).
// Groups of X2_G = 16 consecutive output blocks as in the xcol plan, wave v owns output block first + v.  The pair walk of a
// group is cut into PHASES: up to PH steps (pairs) and up to WCAP weight blocks, which is what one half of the LDS ring holds:
// 80 KiB = PH activation slabs of 16 KiB + WCAP + 1 slots of 2 KiB (the last one is the gate table of gated calls).  PH is
// chosen per plan from the mean number of blocks per step: 2 (WCAP 23) for the bench densities, 3 (WCAP 15) / 4 (WCAP 7) for
// sparse layouts, where a phase's cost is the memory round trip it waits for rather than its bytes: fewer, longer phases.
// A step with more blocks than WCAP is split into sub-steps (the same pair twice).  Every weight block of a phase has a SLOT
// in the phase's half of the weight ring; its two 1 KiB halves are fetched by DMA instructions dealt evenly over the 16 waves
// (<= 3 each).
// Layout (int32): [0] magic 'BSX2' [1] version [2] X2_G [3] ngroups [4] nphases_total [5] off_groups [6] off_px
//                 [7] off_tab (multiple of 4) [8] n_out_blocks [9] WCAP [10] max phases of a group [11] PH
//                 [12] off_cols: cols[ngroups][16] = the output block of every (group, wave), -1 = none (version 3)  [13] 1 if regrouped
//   groups[ngroups][4] = (phase_off, nphases, first_out_block, n_out_blocks_in_group)
//   px [nphases_total][2]        pairs of steps 0, 1 | 2, 3: 16 bits each, 0xffff = no such step
//   tab[nphases_total][16][8]    per phase and wave:
//        [0..1] slots this wave multiplies: byte 2*u + half = slot of (step u, half of the pair), 0xff = none
//        [2..4] DMA duties: (2 * weight block + half) | (2 * slot + half) << 26, or -1        [5..7] 0
// =================================================================================================
namespace bsmm {

constexpr int32_t X2PLAN_MAGIC = 0x42535832;
constexpr int32_t X2PLAN_VERSION = 3;   // 3 (round 6): 16 header words, [12] off_cols: the output block of every (group, wave) -- regrouped layouts
constexpr int X2_G = 16;
constexpr int X2_HDR = 16;
constexpr int X2_ROW = 8;                                            // words per (phase, wave)
constexpr int x2_wcap(int ph) { return (81920 - ph * 16384) / 2048 - 1; }   // 23, 15, 7 for PH = 2, 3, 4

inline long build_xcol2_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int32_t* out, int force_ph = 0, bool regroup = false) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0) return -1;
    if (blocks >= (1 << 25)) return 0;                                       // field widths of the tables
    const int G = X2_G, ngroups = (n_out_blocks + G - 1) / G;
    // (round 6: an unbalanced layout is regrouped where the caller allows it -- feature axis 0, whose epilogue stores per output block; the axis-1
    //  epilogue stores whole rows of 16 ADJACENT output blocks and keeps consecutive groups)
    std::vector<int> grp_of, wave_of;
    const bool regrouped = regroup_output_blocks(lut, segments, n_out_blocks, G, !regroup, grp_of, wave_of);
    if (grp_of.empty()) return -1;
    std::vector<std::array<int32_t, X2_G>> gcols(ngroups);
    for (auto& c : gcols) c.fill(-1);
    for (int ob = 0; ob < n_out_blocks; ++ob) gcols[grp_of[ob]][wave_of[ob]] = ob;
    struct E { int p, wave, half, w; };
    std::vector<std::vector<E>> per_group(ngroups);
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            if (c >= 2 * 0xffff) return 0;
            per_group[grp_of[ob]].push_back({c >> 1, wave_of[ob], c & 1, w});
        }
    }
    // steps per phase from the mean number of blocks per (group, pair) step, with 30 % headroom for the spread
    size_t nsteps_all = 0;
    for (int g = 0; g < ngroups; ++g) {
        auto& v = per_group[g];
        std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.p != b.p ? a.p < b.p : (a.wave != b.wave ? a.wave < b.wave : a.half < b.half); });
        for (size_t i = 0; i < v.size(); ++i) nsteps_all += (i == 0 || v[i].p != v[i - 1].p);
    }
    const double mean = nsteps_all ? (double)blocks / (double)nsteps_all : 0.0;
    int PH = force_ph;
    if (PH < 2 || PH > 4) PH = (4 * mean * 1.3 <= x2_wcap(4)) ? 4 : ((3 * mean * 1.3 <= x2_wcap(3)) ? 3 : 2);
    const int WCAP = x2_wcap(PH);
    std::vector<int32_t> groups, px, tab;
    int max_ph = 0;
    for (int g = 0; g < ngroups; ++g) {
        auto& v = per_group[g];
        // steps: runs of equal pair, at most WCAP entries each (a wave's two halves stay in one step)
        struct Step { int p; size_t lo, hi; };
        std::vector<Step> steps;
        for (size_t i = 0; i < v.size();) {
            size_t j = i;
            while (j < v.size() && v[j].p == v[i].p) ++j;
            size_t lo = i;
            while (lo < j) {
                size_t hi = std::min(j, lo + WCAP);
                if (hi < j && hi - lo > 1 && v[hi].wave == v[hi - 1].wave) --hi;   // do not part the halves of one wave
                steps.push_back({v[i].p, lo, hi});
                lo = hi;
            }
            i = j;
        }
        const int phase_off = (int)(px.size() / 2);
        for (size_t s = 0; s < steps.size();) {
            int nst = 1;
            size_t n = steps[s].hi - steps[s].lo;
            while (nst < PH && s + nst < steps.size() && n + (steps[s + nst].hi - steps[s + nst].lo) <= (size_t)WCAP) {
                n += steps[s + nst].hi - steps[s + nst].lo;
                ++nst;
            }
            uint32_t pw[2] = {0xffffffffu, 0xffffffffu};
            for (int u = 0; u < nst; ++u) pw[u >> 1] = (pw[u >> 1] & ~(0xffffu << (16 * (u & 1)))) | ((uint32_t)steps[s + u].p << (16 * (u & 1)));
            px.push_back((int32_t)pw[0]); px.push_back((int32_t)pw[1]);
            std::vector<int32_t> row((size_t)G * X2_ROW, 0);
            for (int wv = 0; wv < G; ++wv)
                for (int k = 0; k < 5; ++k) row[(size_t)wv * X2_ROW + k] = -1;
            int slot = 0, duty = (int)((px.size() / 2) * 5) % G;           // rotate the wave that gets the first duty
            std::vector<int> nduty(G, 0);
            for (int u = 0; u < nst; ++u)
                for (size_t i = steps[s + u].lo; i < steps[s + u].hi; ++i, ++slot) {
                    const E& e = v[i];
                    const int byte = 2 * u + e.half;
                    uint32_t& cw = reinterpret_cast<uint32_t&>(row[(size_t)e.wave * X2_ROW + (byte >> 2)]);
                    cw = (cw & ~(0xffu << (8 * (byte & 3)))) | ((uint32_t)slot << (8 * (byte & 3)));
                    for (int hb = 0; hb < 2; ++hb) {
                        const int wv = duty; duty = (duty + 1) % G;
                        row[(size_t)wv * X2_ROW + 2 + nduty[wv]++] = (int32_t)((uint32_t)(2 * e.w + hb) | ((uint32_t)(2 * slot + hb) << 26));
                    }
                }
            tab.insert(tab.end(), row.begin(), row.end());
            s += nst;
        }
        const int nph = (int)(px.size() / 2) - phase_off;
        max_ph = std::max(max_ph, nph);
        int nob = 0;
        for (int wv = 0; wv < G; ++wv) nob += gcols[g][wv] >= 0 ? 1 : 0;
        groups.insert(groups.end(), {phase_off, nph, gcols[g][0], nob});
    }
    // longest groups first: workgroup (tile, group index i) runs groups[i], and the CUs that finish a short group of one row tile
    // pick up the long groups of the next (skewed layouts: a Barabasi-Albert layout has 1.7x the blocks in its first group)
    std::vector<int32_t> cols;
    {
        std::vector<int> order(ngroups);
        for (int g = 0; g < ngroups; ++g) order[g] = g;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return groups[4 * a + 1] > groups[4 * b + 1]; });
        std::vector<int32_t> sorted;
        for (int g : order) {
            sorted.insert(sorted.end(), groups.begin() + 4 * g, groups.begin() + 4 * g + 4);
            cols.insert(cols.end(), gcols[g].begin(), gcols[g].end());
        }
        groups.swap(sorted);
    }
    const int off_groups = X2_HDR, off_px = off_groups + (int)groups.size();
    const int off_tab = (off_px + (int)px.size() + 3) & ~3;
    const long off_cols = off_tab + (long)tab.size();
    const long total = off_cols + (long)cols.size();
    if (total >= (1L << 31)) return 0;
    if (out) {
        std::fill(out, out + off_tab, 0);
        const int32_t hdr[X2_HDR] = {X2PLAN_MAGIC, X2PLAN_VERSION, G, ngroups, (int32_t)(px.size() / 2), off_groups, off_px, off_tab,
                                     n_out_blocks, WCAP, max_ph, PH, (int32_t)off_cols, regrouped ? 1 : 0, 0, 0};
        std::copy(hdr, hdr + X2_HDR, out);
        std::copy(groups.begin(), groups.end(), out + off_groups);
        std::copy(px.begin(), px.end(), out + off_px);
        std::copy(tab.begin(), tab.end(), out + off_tab);
        std::copy(cols.begin(), cols.end(), out + off_cols);
    }
    return total;
}

}  // namespace bsmm

// =================================================================================================
// flow xcol plan ('BSX4', round 4): schedule of the barrier-free persistent xprop kernel (bsmm_xflow.h).  Groups of X4_G = 16
// consecutive output blocks, wave v owns output block first + v; a group walks the union of its input-block PAIRS in ascending
// order, one pair = one STEP = one 16 KiB activation slab in a ring of X4_D slabs.  The plan is a list of EVENTS per (group, wave),
// in the order the wave executes them; a wave never looks at a step it has nothing to do in:
//   BLOCK(step, half)   multiply my weight block of that step (the block sits in my private slot: fetched two BLOCK events earlier)
//   REQ(step, part)     request part `part` (of X4_PARTS) of the slab of `step` by LDS-DMA.  Placed X4_DX steps ahead of the step
//                       in the wave's order and dealt by the builder to waves WITHOUT blocks around that point (at 20 % density
//                       10 of the 16 waves have none in a given step)
//   ANN(step, part)     one step later in the wave's order: wait for those requests, announce the part
//   NOP                 list head: carries the first two weight fetches
// Every event names the weight block to FETCH once it is over (none: all ones) and, for the wave's progress word, the step of the
// wave's next BLOCK.  The order of a wave's vector-memory operations is therefore fixed by the plan, and BLOCK / ANN events carry the
// exact `vmcnt` they wait with: the number of operations the wave issues between the fetch / request they need and themselves.
// Layout (int32): [0] magic 'BSX4' [1] version [2] X4_G [3] ngroups [4] nsteps_total [5] off_groups [6] off_pairs
//                 [7] off_lists [8] n_out_blocks [9] max list length [10] max steps of a group [11] X4_D | X4_DX << 8 | X4_PARTS << 16
//   groups[ngroups][8] = (step_off, nsteps, output block of wave 0, n_out_blocks_in_group, list_off, lcap, blocks_in_group, 1 if regrouped)
//   pairs [nsteps_total]            pair index p of each step (input blocks 2p, 2p + 1)
//   lists at off_lists + list_off:  counts[16], cols[16] (version 4: the output block each wave owns, -1 = none), then per wave lcap entries of 2 words:
//        word 0: type (2 bits: 0 NOP, 1 BLOCK, 2 REQ, 3 ANN) | half or part << 2 | step << 4 (12 bits) | step of my next BLOCK << 16
//                (12 bits, nsteps if none) | step % X4_D << 28
//        word 1: weight block to fetch after the event (27 bits, all ones = none) | vmcnt to wait with << 27 (capped at 15)
// Groups are sorted longest first (as in the 'BSX2' plan).
// Version 4 (round 6): a group is ANY 16 output blocks.  Where 16 CONSECUTIVE ones would leave the groups unbalanced (the busiest holds > 1.15 x the
// mean: hub layouts such as the reference's Barabasi-Albert bench layout, whose 16 oldest nodes are 16 neighbouring columns -- 64 of the 512 units
// then took 2.5 x as long as the others and the pass ended when they did), adjacent PAIRS of output blocks (one 128-byte line of an output row) are
// dealt to the groups heaviest first, each to the lightest group with room.  A column still sums its blocks in table order on one wave: the
// results are bit-identical; uniform layouts keep consecutive groups (BSMM_PLAN_FLOW_CONSECUTIVE: always).
// =================================================================================================
namespace bsmm {

constexpr int32_t X4PLAN_MAGIC = 0x42535834;
constexpr int32_t X4PLAN_VERSION = 4;
constexpr uint32_t X4_NOFETCH = 0x7ffffffu;
constexpr int X4_G = 16;
constexpr int X4_HDR = 12;
constexpr int X4_GROUP = 8;
#ifndef X4_D_SLABS
#define X4_D_SLABS 5
#endif
#ifndef X4_DX_AHEAD
#define X4_DX_AHEAD 4
#endif
#ifndef X4_DUTY_PARTS
#define X4_DUTY_PARTS 2
#endif
constexpr int X4_D = X4_D_SLABS;          // slabs in the ring
constexpr int X4_DX = X4_DX_AHEAD;        // a slab is requested this many steps ahead
constexpr int X4_PARTS = X4_DUTY_PARTS;   // a slab is requested in this many parts (by different waves): 1, 2 or 4
static_assert(X4_DX >= 1 && X4_DX < X4_D && X4_D <= 7 && (X4_PARTS == 1 || X4_PARTS == 2 || X4_PARTS == 4), "flow plan constants");

inline long build_xflow_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int32_t* out, bool balance = false, bool consecutive = false) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0) return -1;
    if (blocks >= (1 << 21)) return 0;          // the kernel addresses a weight block with a 32-bit byte offset (id << 11): no plan beyond 4 GiB of weights
    const int G = X4_G, ngroups = (n_out_blocks + G - 1) / G;
    // ---- which (group, wave) owns an output block ----
    std::vector<int> grp_of, wave_of;
    const bool regrouped = regroup_output_blocks(lut, segments, n_out_blocks, G, consecutive, grp_of, wave_of);
    if (grp_of.empty()) return -1;
    struct E { int p, wave, half, w; };
    std::vector<std::vector<E>> per_group(ngroups);
    std::vector<std::array<int32_t, X4_G>> gcols(ngroups);
    for (auto& c : gcols) c.fill(-1);
    for (int ob = 0; ob < n_out_blocks; ++ob) gcols[grp_of[ob]][wave_of[ob]] = ob;
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            if (c >= (1 << 24)) return 0;
            per_group[grp_of[ob]].push_back({c >> 1, wave_of[ob], c & 1, w});
        }
    }
    // ORDER of the steps (a sum over input blocks: any order is the same product; fp32 rounding follows the order).  In ascending order a
    // pass is paced by the wave whose blocks happen to cluster inside the ring's window: pace ~ E[max over the 16 waves of their blocks in
    // X4_D consecutive steps] x (cycles per block) / X4_D (profiles/r04_headline_ab.md).  List scheduling against a small timing model of
    // the kernel (a block costs TB cycles on its wave; a slab is usable LAT cycles after every wave has left the slab that held its slot)
    // picks, among the next KC pairs in ascending order, the one after which the groups' clocks stand lowest.  ONE order for all groups:
    // the groups of a row tile run side by side on one XCD and share the activation slabs in its L2 only if they walk them in step
    // (a per-group order measured 6-14 % SLOWER than ascending order for that reason).  MEASURED with one order for all groups: 56.9 / 83.2 /
    // 154.4 us against 56.9 / 83.7 / 157.2 in ascending order (fprop, 10 / 20 / 50 %): nothing -- the pass is not paced by that window
    // (the look-ahead distance does not move it either).  Hence OFF unless BSMM_PLAN_FLOW_SCHEDULED asks for it: ascending order keeps
    // the flow kernel bit-identical to the staged one.
    std::vector<int> rank;                                           // pair -> position (empty: ascending)
    if (balance) {
        constexpr double TB = 1500., LAT = 1900.;
        constexpr size_t KC = 64;
        int max_p = -1;
        for (auto& v : per_group) for (auto& e : v) max_p = std::max(max_p, e.p);
        const int np = max_p + 1;
        if (np > 2 && (double)np * ngroups <= 4e6) {
            // load[g][p][wave]: sparse per group (pairs it touches)
            std::vector<std::vector<std::pair<int, std::array<uint8_t, X4_G>>>> at(np);   // per pair: (group, loads)
            for (int g = 0; g < ngroups; ++g) {
                std::vector<int> idx(np, -1);
                for (auto& e : per_group[g]) {
                    if (idx[e.p] < 0) { idx[e.p] = (int)at[e.p].size(); at[e.p].push_back({g, {}}); at[e.p].back().second.fill(0); }
                    at[e.p][idx[e.p]].second[e.wave]++;
                }
            }
            std::vector<int> left;
            for (int p = 0; p < np; ++p) if (!at[p].empty()) left.push_back(p);
            std::vector<std::array<double, X4_G>> wt(ngroups);
            for (auto& w : wt) w.fill(0.);
            std::vector<std::vector<double>> done(ngroups);             // per group: its clock after each of ITS steps
            std::vector<double> gclock(ngroups, 0.);
            rank.assign(np, 0);
            int pos = 0;
            while (!left.empty()) {
                size_t best = 0; double best_sum = 0., best_sq = 0.;
                const size_t nc = std::min(left.size(), KC);
                for (size_t k = 0; k < nc; ++k) {
                    double sum = 0., sq = 0.;
                    for (auto& gl : at[left[k]]) {
                        const int g = gl.first;
                        const size_t ps = done[g].size();
                        const double usable = (ps < (size_t)X4_D ? 0. : done[g][ps - X4_D]) + LAT;
                        double mx = gclock[g];
                        for (int wv = 0; wv < G; ++wv)
                            if (gl.second[wv]) { const double v = std::max(wt[g][wv], usable) + gl.second[wv] * TB; mx = std::max(mx, v); sq += v * v - wt[g][wv] * wt[g][wv]; }
                        sum += mx - gclock[g];
                    }
                    if (k == 0 || sum < best_sum || (sum == best_sum && sq < best_sq)) { best = k; best_sum = sum; best_sq = sq; }
                }
                const int p = left[best];
                left.erase(left.begin() + best);
                for (auto& gl : at[p]) {
                    const int g = gl.first;
                    const size_t ps = done[g].size();
                    const double usable = (ps < (size_t)X4_D ? 0. : done[g][ps - X4_D]) + LAT;
                    for (int wv = 0; wv < G; ++wv)
                        if (gl.second[wv]) { wt[g][wv] = std::max(wt[g][wv], usable) + gl.second[wv] * TB; gclock[g] = std::max(gclock[g], wt[g][wv]); }
                    done[g].push_back(gclock[g]);
                }
                rank[p] = pos++;
            }
        }
    }
    std::vector<int32_t> groups, pairs, lists;
    int max_l = 0, max_s = 0;
    for (int g = 0; g < ngroups; ++g) {
        auto& v = per_group[g];
        std::stable_sort(v.begin(), v.end(), [&](const E& a, const E& b) {     // (stable: entries that name the same input block twice -- the doubled
            const int ra = rank.empty() ? a.p : rank[a.p], rb = rank.empty() ? b.p : rank[b.p];   //  tables of gated calls -- keep their table order)
            return ra != rb ? ra < rb : (a.wave != b.wave ? a.wave < b.wave : a.half < b.half);
        });
        const int step_off = (int)pairs.size();
        struct B { int step, half, w; };
        std::vector<std::vector<B>> wb(G);
        int t = -1, cur = -1;
        for (auto& e : v) {
            if (e.p != cur) { cur = e.p; ++t; pairs.push_back(e.p); }
            wb[e.wave].push_back({t, e.half, e.w});
        }
        const int nsteps = t + 1;
        if (nsteps >= 4096) return 0;                               // 12-bit step fields
        // blocks per (wave, step), for the dealing of the duties
        std::vector<std::vector<int>> busy(G, std::vector<int>(nsteps + 1, 0));
        for (int wv = 0; wv < G; ++wv) for (auto& b : wb[wv]) busy[wv][b.step]++;
        // events: (trigger, kind order, payload).  REQ(step s) has trigger s - X4_DX, its ANN one step later; at equal trigger a wave
        // announces first, then requests, then multiplies
        struct Ev { int trig, kind, a, b; };                        // kind 0 ANN / 1 REQ (a = step, b = part), 2 BLOCK (a = index into wb[wave])
        std::vector<std::vector<Ev>> evs(G);
        for (int wv = 0; wv < G; ++wv) for (size_t i = 0; i < wb[wv].size(); ++i) evs[wv].push_back({wb[wv][i].step, 2, (int)i, 0});
        std::vector<int> nduty(G, 0), last_trig(G, -1000);
        int rot = g * 5;
        for (int s = 0; s < nsteps; ++s)
            for (int q = 0; q < X4_PARTS; ++q) {
                const int trig = s - X4_DX;
                int best = -1; long best_cost = 0;
                for (int k = 0; k < G; ++k) {
                    const int wv = (rot + k) % G;
                    long cost = 0;
                    for (int d = -1; d <= 2; ++d) { const int st = trig + d; if (st >= 0 && st < nsteps) cost += busy[wv][st] * (d == 0 || d == 1 ? 100 : 40); }
                    if (trig - last_trig[wv] < 2) cost += 60;        // it is still waiting for its previous requests
                    cost += nduty[wv];
                    if (best < 0 || cost < best_cost) { best = wv; best_cost = cost; }
                }
                rot = (best + 1) % G;
                evs[best].push_back({trig, 1, s, q});
                evs[best].push_back({trig + 1, 0, s, q});
                nduty[best]++; last_trig[best] = trig;
            }
        int lcap = 1;
        std::vector<std::vector<int32_t>> wl(G);
        for (int wv = 0; wv < G; ++wv) {
            auto& ev = evs[wv];
            std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.trig != b.trig ? a.trig < b.trig : a.kind < b.kind; });
            const auto& bl = wb[wv];
            if (bl.size() >= (size_t)X4_NOFETCH) return 0;
            auto next_block_step = [&](size_t from_block) { return from_block < bl.size() ? bl[from_block].step : nsteps; };
            auto word0 = [&](int type, int hp, int step, int nxt) {
                return (int32_t)((uint32_t)type | ((uint32_t)hp << 2) | ((uint32_t)step << 4) | ((uint32_t)nxt << 16) | ((uint32_t)(step % X4_D) << 28));
            };
            auto word1 = [&](long fetch, long wait) {
                return (int32_t)((fetch < 0 ? X4_NOFETCH : (uint32_t)fetch) | ((uint32_t)std::min<long>(std::max<long>(wait, 0), 15) << 27));
            };
            // the wave's vector-memory operations in program order: `ops` = issued so far; a fetch is 2, a request 16 / X4_PARTS
            long ops = 0;
            std::vector<long> fetch_seq(bl.size() + 2, 0);          // ops right after the fetch of block j
            std::vector<long> req_seq((size_t)nsteps * X4_PARTS, 0);
            size_t nfetch = 0;
            auto fetch_next = [&]() -> long { if (nfetch < bl.size()) { ops += 2; fetch_seq[nfetch] = ops; return bl[nfetch++].w; } return -1; };
            // two NOPs carry the first two fetches
            for (int k = 0; k < 2; ++k) { wl[wv].push_back(word0(0, 0, 0, next_block_step(0))); const long f = fetch_next(); wl[wv].push_back(word1(f, 0)); }
            size_t nb = 0;                                          // blocks seen so far in event order
            for (auto& e : ev) {
                if (e.kind == 1) {
                    wl[wv].push_back(word0(2, e.b, e.a, next_block_step(nb))); wl[wv].push_back(word1(-1, 0));
                    ops += 16 / X4_PARTS; req_seq[(size_t)e.a * X4_PARTS + e.b] = ops;
                } else if (e.kind == 0) {
                    wl[wv].push_back(word0(3, e.b, e.a, next_block_step(nb))); wl[wv].push_back(word1(-1, ops - req_seq[(size_t)e.a * X4_PARTS + e.b]));
                } else {
                    const B& b = bl[e.a];
                    const long wait = ops - fetch_seq[nb];          // operations issued after this block's fetch
                    wl[wv].push_back(word0(1, b.half, b.step, next_block_step(nb + 1)));
                    const long f = fetch_next();                    // block nb + 2 goes into the slot this one frees
                    wl[wv].push_back(word1(f, wait));
                    ++nb;
                }
            }
            lcap = std::max(lcap, (int)wl[wv].size() / 2);
        }
        max_l = std::max(max_l, lcap); max_s = std::max(max_s, nsteps);
        const int list_off = (int)lists.size();
        for (int wv = 0; wv < G; ++wv) lists.push_back((int32_t)wl[wv].size() / 2);
        for (int wv = 0; wv < G; ++wv) lists.push_back(gcols[g][wv]);
        for (int wv = 0; wv < G; ++wv) {
            lists.insert(lists.end(), wl[wv].begin(), wl[wv].end());
            lists.insert(lists.end(), (size_t)2 * lcap - wl[wv].size(), 0);
        }
        int nob = 0;
        for (int wv = 0; wv < G; ++wv) nob += gcols[g][wv] >= 0 ? 1 : 0;
        groups.insert(groups.end(), {step_off, nsteps, gcols[g][0], nob, list_off, lcap, (int32_t)v.size(), regrouped ? 1 : 0});
    }
    {
        std::vector<int> order(ngroups);
        for (int g = 0; g < ngroups; ++g) order[g] = g;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return groups[X4_GROUP * a + 6] > groups[X4_GROUP * b + 6]; });
        std::vector<int32_t> sorted;
        for (int g : order) sorted.insert(sorted.end(), groups.begin() + X4_GROUP * g, groups.begin() + X4_GROUP * (g + 1));
        groups.swap(sorted);
    }
    const int off_groups = X4_HDR, off_pairs = off_groups + (int)groups.size(), off_lists = off_pairs + (int)pairs.size();
    const long total = off_lists + (long)lists.size();
    if (out) {
        const int32_t hdr[X4_HDR] = {X4PLAN_MAGIC, X4PLAN_VERSION, G, ngroups, (int32_t)pairs.size(), off_groups, off_pairs, off_lists,
                                     n_out_blocks, max_l, max_s, X4_D | (X4_DX << 8) | (X4_PARTS << 16)};
        std::copy(hdr, hdr + X4_HDR, out);
        std::copy(groups.begin(), groups.end(), out + off_groups);
        std::copy(pairs.begin(), pairs.end(), out + off_pairs);
        std::copy(lists.begin(), lists.end(), out + off_lists);
    }
    return total;
}

}  // namespace bsmm

// =================================================================================================
// staged xcol16 plan ('BSX7', bsize 16): the staged scheme of the 'BSX2' plan for 16x16 blocks (bsmm_xcol16_v2.h).  Groups of
// X7_G = 32 consecutive output blocks, wave v of 16 owns blocks 2v and 2v+1; a step is a QUAD of input blocks (64 features);
// a phase = up to two steps and up to X7_WCAP weight blocks = one half of the LDS ring (2 activation slabs of 16 KiB + X7_WCAP
// slots of 512 B).  Weight blocks are fetched two at a time (one 1 KiB DMA instruction = slots 2j and 2j+1), the instructions
// dealt evenly over the 16 waves (<= 3 each).
// Layout (int32): [0] magic 'BSX7' [1] version [2] X7_G [3] ngroups [4] nphases_total [5] off_groups [6] off_px
//                 [7] off_tab (multiple of 4) [8] n_out_blocks [9] X7_WCAP [10] max phases of a group [11] off_lists
//                 [12] off_cols: cols[ngroups][32] = the output block of every (group, column position), -1 = none (version 3)  [13] 1 if regrouped
//   groups[ngroups][4] = (phase_off, nphases, first_out_block, n_out_blocks_in_group)
//   px [nphases_total]           quad of step 0 | quad of step 1 << 16   (0xffff = no such step)
//   tab[nphases_total][16][12]   per phase and wave:
//        [0..3]  16 slot bytes: byte 8 * u + 4 * c + sub = slot of (step u, my output block c, input block sub of the quad), 0xff = none
//        [4..9]  three DMA duties (A, B): A = first weight block | slot pair << 26 (or -1: no duty), B = second weight block
//        [10]    blocks of my column 0 in this phase | blocks of my column 1 << 8 | (i + 1) << 16 if I am issuer i (0..3) of the NEXT
//                phase's requests during this phase  (version 2)
//        [11]    0
//   list section (version 2, at [11] off_lists), X7_PHW = 768 words per phase:
//     [16][40] per wave: word 38 = tab word [10] of this phase, word 39 = tab word [10] of the next phase (0 behind the group's last), 32..37 zero; words 0..31 the blocks of column 0, then of column 1, in (step, input block) order, two words each:
//        activation position bits = sub << 5 | sub << 12 | step << 14 (the kernel keeps the field of its feature axis: byte offset bits
//        of the block's 16 features inside the slab pair), weight slot * 512
//     [64][2]  the phase's requests as ONE lane-indexed table (xcol16_list_kernel: four waves issue all requests of a phase):
//        entry k < npairs: byte offsets of the weight blocks of slots 2k and 2k+1 inside W;  entry 48: (px of the phase, npairs)
// =================================================================================================
namespace bsmm {

constexpr int32_t X7PLAN_MAGIC = 0x42535837;
constexpr int32_t X7PLAN_VERSION = 3;   // 3 (round 6): 16 header words, [12] off_cols: the output block of every (group, column position) -- regrouped layouts
constexpr int X7_G = 32;
constexpr int X7_HDR = 16;
constexpr int X7_WCAP = 94;            // even; slot X7_WCAP of each ring half stays zero (the fragment of an absent block)
constexpr int X7_ROW = 12;             // words per (phase, wave)
constexpr int X7_LIST = 40;            // words of a (phase, wave) block list: 16 entries of two words, words 0..31 entries, 38 = counts | role, 39 = the same of the next phase
constexpr int X7_PHW = 16 * X7_LIST + 128;   // words of a phase in the list section: 16 block lists, then the request table

inline long build_xcol16s_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int32_t* out, bool regroup = true) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0) return -1;
    if (blocks >= (1 << 23)) return 0;                                       // 32-bit byte offsets into W
    const int G = X7_G, ngroups = (n_out_blocks + G - 1) / G;
    std::vector<int> grp_of, pos_of;                      // (round 6: unbalanced layouts are regrouped, regroup_output_blocks)
    const bool regrouped = regroup_output_blocks(lut, segments, n_out_blocks, G, !regroup, grp_of, pos_of);
    if (grp_of.empty()) return -1;
    std::vector<std::array<int32_t, X7_G>> gcols(ngroups);
    for (auto& c : gcols) c.fill(-1);
    for (int ob = 0; ob < n_out_blocks; ++ob) gcols[grp_of[ob]][pos_of[ob]] = ob;
    struct E { int p, col, sub, w; };
    std::vector<std::vector<E>> per_group(ngroups);
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            if (c >= 4 * 0xffff) return 0;
            per_group[grp_of[ob]].push_back({c >> 2, pos_of[ob], c & 3, w});
        }
    }
    std::vector<int32_t> groups, px, tab, lists;
    int max_ph = 0;
    for (int g = 0; g < ngroups; ++g) {
        auto& v = per_group[g];
        std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.p != b.p ? a.p < b.p : (a.col != b.col ? a.col < b.col : a.sub < b.sub); });
        struct Step { int p; size_t lo, hi; };      // runs of equal quad, at most X7_WCAP entries each
        std::vector<Step> steps;
        for (size_t i = 0; i < v.size();) {
            size_t j = i;
            while (j < v.size() && v[j].p == v[i].p) ++j;
            for (size_t lo = i; lo < j; lo += X7_WCAP) steps.push_back({v[i].p, lo, std::min(j, lo + (size_t)X7_WCAP)});
            i = j;
        }
        const int phase_off = (int)px.size();
        for (size_t s = 0; s < steps.size();) {
            const size_t n0 = steps[s].hi - steps[s].lo;
            const bool two = s + 1 < steps.size() && n0 + (steps[s + 1].hi - steps[s + 1].lo) <= (size_t)X7_WCAP;
            const int nst = two ? 2 : 1;
            px.push_back(steps[s].p | ((two ? steps[s + 1].p : 0xffff) << 16));
            std::vector<int32_t> row((size_t)16 * X7_ROW, 0);
            for (int wv = 0; wv < 16; ++wv)
                for (int k = 0; k < 10; ++k) row[(size_t)wv * X7_ROW + k] = -1;
            std::vector<int32_t> ws;                 // weight block of every slot, in slot order
            std::vector<int32_t> lrow((size_t)16 * X7_LIST, 0);
            std::vector<int32_t> pend[32];           // list words per column of the group
            for (int u = 0; u < nst; ++u)
                for (size_t i = steps[s + u].lo; i < steps[s + u].hi; ++i) {
                    const E& e = v[i];
                    const int slot = (int)ws.size();
                    ws.push_back(e.w);
                    const int byte = 8 * u + 4 * (e.col & 1) + e.sub;
                    uint32_t& word = reinterpret_cast<uint32_t&>(row[(size_t)(e.col >> 1) * X7_ROW + (byte >> 2)]);
                    word = (word & ~(0xffu << (8 * (byte & 3)))) | ((uint32_t)slot << (8 * (byte & 3)));
                    pend[e.col].push_back((e.sub << 5) | (e.sub << 12) | (u << 14));
                    pend[e.col].push_back(slot * 512);
                }
            for (int wv = 0; wv < 16; ++wv) {
                const std::vector<int32_t>&c0 = pend[2 * wv], &c1 = pend[2 * wv + 1];       // <= 8 entries each (2 steps x 4 input blocks)
                std::copy(c0.begin(), c0.end(), lrow.begin() + (size_t)wv * X7_LIST);
                std::copy(c1.begin(), c1.end(), lrow.begin() + (size_t)wv * X7_LIST + c0.size());
                row[(size_t)wv * X7_ROW + 10] = (int32_t)(c0.size() / 2) | ((int32_t)(c1.size() / 2) << 8);
                row[(size_t)wv * X7_ROW + 11] = 0;
            }
            {   // the four waves that issue the requests of the NEXT phase during this one: a wave that issues is held for as long as the
                // memory system takes to deliver the phase (~1500 cycles), so the job goes to the waves with the fewest blocks here
                // (ties: rotate with the phase, so that no SIMD's waves are picked every time)
                int order[16];
                const int rot = (int)(px.size() * 5) % 16;
                for (int i = 0; i < 16; ++i) order[i] = (i + rot) % 16;
                std::stable_sort(order, order + 16, [&](int a, int b) {
                    return pend[2 * a].size() + pend[2 * a + 1].size() < pend[2 * b].size() + pend[2 * b + 1].size(); });
                for (int i = 0; i < 4; ++i) row[(size_t)order[i] * X7_ROW + 10] |= (i + 1) << 16;
            }
            {   // the same words inside the list rows (what xcol16_list_kernel reads): mine into word 38, and into word 39 of the previous phase
                int32_t* mine = lrow.data();
                for (int wv = 0; wv < 16; ++wv) mine[(size_t)wv * X7_LIST + 38] = row[(size_t)wv * X7_ROW + 10];
                if ((int)px.size() - 1 > phase_off)
                    for (int wv = 0; wv < 16; ++wv) (lists.data() + lists.size() - X7_PHW)[(size_t)wv * X7_LIST + 39] = row[(size_t)wv * X7_ROW + 10];
            }
            lists.insert(lists.end(), lrow.begin(), lrow.end());
            {
                std::vector<int32_t> req(128, 0);
                const int npairs = (int)(ws.size() + 1) / 2;                     // <= X7_WCAP / 2 = 47
                for (int k = 0; k < npairs; ++k) {
                    req[2 * k] = (int32_t)((uint32_t)ws[2 * k] * 512u);
                    req[2 * k + 1] = (int32_t)((uint32_t)((size_t)(2 * k + 1) < ws.size() ? ws[2 * k + 1] : ws[2 * k]) * 512u);
                }
                req[96] = px.back();
                req[97] = npairs;
                lists.insert(lists.end(), req.begin(), req.end());
            }
            int duty = (int)(px.size() * 5) % 16;
            std::vector<int> nduty(16, 0);
            for (size_t pair = 0; 2 * pair < ws.size(); ++pair) {
                const int32_t a = ws[2 * pair], b = (2 * pair + 1 < ws.size()) ? ws[2 * pair + 1] : ws[2 * pair];
                const int wv = duty; duty = (duty + 1) % 16;
                row[(size_t)wv * X7_ROW + 4 + 2 * nduty[wv]] = (int32_t)((uint32_t)a | ((uint32_t)pair << 26));
                row[(size_t)wv * X7_ROW + 5 + 2 * nduty[wv]] = b;
                ++nduty[wv];
            }
            tab.insert(tab.end(), row.begin(), row.end());
            s += nst;
        }
        const int nph = (int)px.size() - phase_off;
        max_ph = std::max(max_ph, nph);
        int nob = 0;
        for (int c = 0; c < G; ++c) nob += gcols[g][c] >= 0 ? 1 : 0;
        groups.insert(groups.end(), {phase_off, nph, gcols[g][0], nob});
    }
    const int off_groups = X7_HDR, off_px = off_groups + (int)groups.size();
    const int off_tab = (off_px + (int)px.size() + 3) & ~3;
    const long off_lists = off_tab + (long)tab.size();
    const long off_cols = off_lists + (long)lists.size();
    const long total = off_cols + (long)ngroups * G;
    if (total >= (1L << 31)) return 0;
    if (out) {
        std::fill(out, out + off_tab, 0);
        const int32_t hdr[X7_HDR] = {X7PLAN_MAGIC, X7PLAN_VERSION, G, ngroups, (int32_t)px.size(), off_groups, off_px, off_tab,
                                     n_out_blocks, X7_WCAP, max_ph, (int32_t)off_lists, (int32_t)off_cols, regrouped ? 1 : 0, 0, 0};
        std::copy(hdr, hdr + X7_HDR, out);
        std::copy(groups.begin(), groups.end(), out + off_groups);
        std::copy(px.begin(), px.end(), out + off_px);
        std::copy(tab.begin(), tab.end(), out + off_tab);
        std::copy(lists.begin(), lists.end(), out + off_lists);
        for (int g = 0; g < ngroups; ++g) std::copy(gcols[g].begin(), gcols[g].end(), out + off_cols + (long)g * G);
    }
    return total;
}

}  // namespace bsmm



// =================================================================================================
// super8 plans (bsize 8, 16-bit types): the 8x8 blocks are grouped into the 32x32 SUPER-blocks of the block grid that
// hold at least one of them; the bsize-32 matrix-core kernels then run on the super layout (absent 8x8 sub-blocks are
// zeros: xprop multiplies an expanded copy of W, updat computes whole super-blocks and keeps the present parts).
// Even at 10 % density 81 % of the super-blocks are populated, but the matrix cores are > 10x faster than the V_FMA
// kernels that walk the 8x8 blocks one by one, and every activation byte is read once per 256-feature group instead of
// once per 8x8 block.  (The reference concatenates 8-wide blocks along K for its tensor-core kernels in the same
// spirit: src/blocksparse_hgemm_cn_64_op_gpu.cu:541-624.)
// Layout (int32): [0] magic 'BSS8' [1] version [2] nsuper [3] off_sub [4] off_lut32 [5] off_nested [6] total words [7] kind
//   sub  [nsuper][16]   8x8 block id or -1 of sub-block (a, b): index 4a + b;  xprop: a = in block & 3, b = out block & 3;
//                       updat: a = c & 3, b = k & 3
//   lut32[nsuper][2]    updat: (c32, k32) of every super-block (the updat_lut of the super layout); xprop: (in32, out32)
//   nested              xprop: 'BSXC' plan of the super layout;  updat: 'BSUP' plan of the super layout
// The host passes nsuper in bsmm_args.plan_aux (and the nested updat plan's item count in plan_items).
// =================================================================================================
namespace bsmm {

constexpr int32_t S8PLAN_MAGIC = 0x42535338;
constexpr int32_t S8PLAN_VERSION = 1;
constexpr int S8_HDR = 8;

inline int s8_off_lut32(int nsuper) { return S8_HDR + 16 * nsuper; }
inline int s8_off_nested(int nsuper) { return (S8_HDR + 18 * nsuper + 3) & ~3; }

struct S8Super { int a32, b32; int32_t sub[16]; };

// triples (a8, b8, w) -> super-blocks sorted by (b32, a32) [xprop: by output block] or (a32, b32) [updat]
inline bool s8_collect(const std::vector<int32_t>& trip, bool sort_by_b, std::vector<S8Super>& supers) {
    struct T3 { int a, b, w; };
    std::vector<T3> v(trip.size() / 3);
    for (size_t i = 0; i < v.size(); ++i) v[i] = {trip[3 * i], trip[3 * i + 1], trip[3 * i + 2]};
    std::sort(v.begin(), v.end(), [sort_by_b](const T3& x, const T3& y) {
        const int xa = x.a >> 2, xb = x.b >> 2, ya = y.a >> 2, yb = y.b >> 2;
        if (sort_by_b) return xb != yb ? xb < yb : (xa != ya ? xa < ya : x.w < y.w);
        return xa != ya ? xa < ya : (xb != yb ? xb < yb : x.w < y.w);
    });
    for (auto& t : v) {
        if (supers.empty() || supers.back().a32 != (t.a >> 2) || supers.back().b32 != (t.b >> 2)) {
            S8Super s;
            s.a32 = t.a >> 2; s.b32 = t.b >> 2;
            std::fill(s.sub, s.sub + 16, -1);
            supers.push_back(s);
        }
        int32_t& slot = supers.back().sub[4 * (t.a & 3) + (t.b & 3)];
        if (slot >= 0) return false;        // the same 8x8 position listed twice
        slot = t.w;
    }
    return true;
}

inline long s8_emit(const std::vector<S8Super>& supers, const std::vector<int32_t>& nested, int kind, int32_t* out) {
    const int ns = (int)supers.size();
    const int off_nested = s8_off_nested(ns);
    const long total = off_nested + (long)nested.size();
    if (out) {
        std::fill(out, out + off_nested, 0);
        const int32_t hdr[S8_HDR] = {S8PLAN_MAGIC, S8PLAN_VERSION, ns, S8_HDR, s8_off_lut32(ns), off_nested, (int32_t)total, kind};
        std::copy(hdr, hdr + S8_HDR, out);
        for (int s = 0; s < ns; ++s) {
            std::copy(supers[s].sub, supers[s].sub + 16, out + S8_HDR + 16 * s);
            out[s8_off_lut32(ns) + 2 * s] = supers[s].a32;
            out[s8_off_lut32(ns) + 2 * s + 1] = supers[s].b32;
        }
        std::copy(nested.begin(), nested.end(), out + off_nested);
    }
    return total;
}

// xprop: lut = the bsize-8 segment table of the pass (headers (offset, count, out block, lock), then (in block, w) pairs)
inline long build_super8_xprop_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int32_t* out, int G = XC_G, bool staged = false) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0) return -1;
    if (n_out_blocks % 4 != 0) return 0;                       // the super grid needs whole 32-feature blocks
    std::vector<int32_t> trip;
    trip.reserve((size_t)blocks * 3);
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            trip.insert(trip.end(), {c, ob, w});
        }
    }
    std::vector<S8Super> supers;
    if (!s8_collect(trip, true, supers)) return -1;
    // segment table of the super layout: one segment per 32-wide output block (empty ones included), entries (in32, s)
    const int n_out32 = n_out_blocks / 4, ns = (int)supers.size();
    std::vector<int32_t> lut32((size_t)4 * n_out32 + 2 * ns);
    int pos = 0;
    for (int ob = 0; ob < n_out32; ++ob) {
        const int first = pos;
        while (pos < ns && supers[pos].b32 == ob) {
            lut32[(size_t)4 * n_out32 + 2 * pos] = supers[pos].a32;
            lut32[(size_t)4 * n_out32 + 2 * pos + 1] = pos;
            ++pos;
        }
        lut32[4 * ob] = 2 * n_out32 + first; lut32[4 * ob + 1] = pos - first; lut32[4 * ob + 2] = ob; lut32[4 * ob + 3] = -1;
    }
    // nested plan: the round-1 'BSXC' plan.  `staged`: the staged kernel's 'BSX2' plan instead -- measured SLOWER here (4096^2 bsize 8 10 %:
    // 293 / 274 us against 220 / 200): the super layout of a 10 % bsize-8 layout is ~80 % dense, a pair step then holds more blocks than
    // a ring half has weight slots and is cut into sub-steps that fetch the same activation slab twice
    if (staged) {
        const long n2 = build_xcol2_plan(lut32.data(), n_out32, ns, n_out32, nullptr, 0);
        if (n2 > 0) {
            std::vector<int32_t> nested((size_t)n2);
            build_xcol2_plan(lut32.data(), n_out32, ns, n_out32, nested.data(), 0);
            return s8_emit(supers, nested, 0, out);
        }
    }
    const long nw = build_xcol_plan(lut32.data(), n_out32, ns, n_out32, nullptr, G);
    if (nw <= 0) return -1;
    std::vector<int32_t> nested((size_t)nw);
    build_xcol_plan(lut32.data(), n_out32, ns, n_out32, nested.data(), G);
    return s8_emit(supers, nested, 0, out);
}

inline long build_updat2_plan(const int32_t* updat_lut, int blocks, int CB, int KB, int ws, int32_t* out, int force_sets, int direct_max = 0);   // ('BSU2', below)

// stream: nest the streaming kernel's 'BSU2' plan (round 3; the super layout of a 10 % bsize-8 layout is ~80 % dense: 8x8 windows) instead of the
// round-1 windowed 'BSUP' plan; feature axis 1 and 0 alike (the streaming kernel serves both since round 3)
inline long build_super8_updat_plan(const int32_t* updat_lut, int blocks, int CB, int KB, int32_t* out, bool stream = true) {
    if (!updat_lut || blocks <= 0 || CB <= 0 || KB <= 0) return -1;
    if (CB % 4 != 0 || KB % 4 != 0) return 0;
    std::vector<int32_t> trip;
    trip.reserve((size_t)blocks * 3);
    for (int w = 0; w < blocks; ++w) {
        const int c = updat_lut[2 * w], k = updat_lut[2 * w + 1];
        if (c < 0 || c >= CB || k < 0 || k >= KB) return -1;
        trip.insert(trip.end(), {c, k, w});
    }
    std::vector<S8Super> supers;
    if (!s8_collect(trip, false, supers)) return -1;
    const int ns = (int)supers.size();
    std::vector<int32_t> lut32((size_t)2 * ns);
    for (int s = 0; s < ns; ++s) { lut32[2 * s] = supers[s].a32; lut32[2 * s + 1] = supers[s].b32; }
    if (stream) {
        const double windows = (double)((CB / 4 + 15) / 16) * ((KB / 4 + 15) / 16);
        const int ws = ns <= 56.0 * windows ? 16 : 8;            // as bsmm_api.hip chooses for a bsize-32 layout
        const long n2 = build_updat2_plan(lut32.data(), ns, CB / 4, KB / 4, ws, nullptr, 0);
        if (n2 > 0) {
            std::vector<int32_t> nested((size_t)n2);
            build_updat2_plan(lut32.data(), ns, CB / 4, KB / 4, ws, nested.data(), 0);
            return s8_emit(supers, nested, 1, out);
        }
    }
    return 0;      // (the windowed bsize-32 kernel that once took what the streaming plan cannot hold was retired in round 4: no plan, per-block kernels)
}

}  // namespace bsmm

// =================================================================================================
// bsize 64 (feature axis 1 -- the reference's second axis-1 block size, blocksparse/matmul.py:84-89): a 64x64 block is four 32x32
// QUADRANTS of the layout kron(layout, ones(2, 2)); a quadrant of weight block w is weight block 4 w + 2 i + j of the quadrant view,
// i = which half of the CALL's input block, j = which half of its output block (fprop: (row half, column half) of W's block; bprop the
// other way round -- the library stores the quadrant copy of W accordingly, bsmm_b64.h; updat: (row half, column half)).  The composite plan 'BS64' carries the lookup table
// of the quadrant view (what the kernels without a plan read, device side) and the nested bsize-32 plan built from it; the library
// repacks W into quadrant order (once per weights version when the caller keeps the result: bsmm_prepare_weights) and runs the
// bsize-32 path.  Layout (int32): [0] magic 'BS64' [1] version [2] blocks (64x64) [3] kind: 0 = xprop, 1 = updat
//   [4] off_lut32 (= B64_HDR) [5] off_nested (multiple of 4) [6] total words [7] segments of the 64-block lut (xprop) / 0
//   lut32: xprop: 2 x segments headers (offset / 2, entries, output block, lock) + 8 x blocks entry words, the two halves of an
//          output block column are consecutive segments (2 s, 2 s + 1), lock ids 2 l - 1 and 2 l;   updat: [4 x blocks][2] = (c, k)
// =================================================================================================
namespace bsmm {

constexpr int32_t B64PLAN_MAGIC = 0x42533634;
constexpr int32_t B64PLAN_VERSION = 1;
constexpr int B64_HDR = 8;
inline long b64_lut32_words(int kind, int segments64, int blocks64) { return kind == 0 ? 8L * segments64 + 8L * blocks64 : 8L * blocks64; }
inline long b64_off_nested(int kind, int segments64, int blocks64) { return (B64_HDR + b64_lut32_words(kind, segments64, blocks64) + 3) & ~3L; }

// quadrant view of an xprop lut: false = malformed
inline bool b64_expand_xprop_lut(const int32_t* lut, int segments, int blocks, std::vector<int32_t>& out) {
    out.assign((size_t)(8L * segments + 8L * blocks), 0);
    long pos = 4L * segments;                   // entry pairs written so far (the 2 x segments headers take 4 x segments pairs)
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2], lock = lut[4 * s + 3];
        if (cnt < 0 || off < 0) return false;
        for (int j = 0; j < 2; ++j) {
            int32_t* h = &out[(size_t)4 * (2 * s + j)];
            h[0] = (int32_t)pos; h[1] = 2 * cnt; h[2] = 2 * ob + j; h[3] = lock > 0 ? 2 * lock - 1 + j : lock;
            for (int e = 0; e < cnt; ++e) {
                const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
                if (w < 0 || w >= blocks || c < 0) return false;
                for (int i = 0; i < 2; ++i) {
                    out[(size_t)2 * pos] = 2 * c + i;
                    out[(size_t)2 * pos + 1] = 4 * w + 2 * i + j;
                    ++pos;
                }
            }
        }
    }
    return 2 * pos == (long)out.size();
}

inline long b64_emit(int kind, int blocks64, int segments64, const std::vector<int32_t>& lut32, const std::vector<int32_t>& nested, int32_t* out) {
    const long off_nested = b64_off_nested(kind, segments64, blocks64);
    const long total = off_nested + (long)nested.size();
    if (total >= (1L << 31)) return 0;
    if (out) {
        std::fill(out, out + off_nested, 0);
        const int32_t hdr[B64_HDR] = {B64PLAN_MAGIC, B64PLAN_VERSION, blocks64, kind, B64_HDR, (int32_t)off_nested, (int32_t)total, kind == 0 ? segments64 : 0};
        std::copy(hdr, hdr + B64_HDR, out);
        std::copy(lut32.begin(), lut32.end(), out + B64_HDR);
        std::copy(nested.begin(), nested.end(), out + off_nested);
    }
    return total;
}

}  // namespace bsmm

// =================================================================================================
// updat v2 plan ('BSU2'): work items of the streaming weight-gradient kernel (bsmm_updat_v2.h), bsize 32, feature axis 1,
// 16-bit types.  The block grid is cut into WS x WS windows (WS = 16 for layouts up to ~22 % density: a window then holds
// up to 64 blocks = 16 waves x 4 accumulator slots; WS = 8 above).  Inside an item every wave owns up to U2_SLOTS blocks
// taken from at most TWO block rows of the window: the X^T fragment of a row is read from LDS once and reused by all of the
// wave's blocks in that row (row pieces are packed first-fit-decreasing so that most waves hold a single row).  A window
// with more blocks than slots, or whose rows cannot be packed that way, is cut into several items.
// Schedule: workgroup u runs on XCD u % 8 (observed; speed only).  The items are stored in NSETS lists ("sets": compact
// patches of the window grid) and the minibatch is cut into 8 / NSETS parts; XCD x owns set x / (8 / NSETS) and part
// x % (8 / NSETS), and its workgroups walk that set's items in lockstep through that part of the minibatch, so the slabs
// of a window row / column are fetched into the XCD's L2 once and shared.  Few items (<= ~200: 16x16 windows of a 4096^2
// layout) -> 2 sets (upper / lower half of the block rows) x 4 minibatch quarters: an XCD reads a quarter of HALF of X and
// of all of DY (1.5x the compulsory traffic; a window patch per XCD reads 3x), partial sums meet in an fp32 scratch.
// Many items -> 8 sets x the whole minibatch: every item is one workgroup's, stored directly.
// Layout (int32): [0] magic 'BSU2' [1] version [2] WS [3] U2_SLOTS [4] nitems [5] nblocks [6] off_items [7] U2_WAVES
//                 [8] NSETS (1, 2, 4 or 8) [9 + 2 s], [10 + 2 s] first item / item count of set s
//                 [25] the item count of every set if they are all equal, else 0  [26] off_bmap  [27] longest set
//                 [28] n_direct  [29] off_direct  [30] U2_DIRECT_PARTS  [31] 0
//   bmap[nblocks] (behind the items): item << 8 | wave * U2_SLOTS + slot  of every block; -(2 + d) for DIRECT block d
//   direct[n_direct][4] = (block, c, k, 0)  (round 6, version 3): the blocks a window cannot hold in its 16 waves -- a 65th block, or rows that do
//         not pack -- used to form small OVERFLOW items that every set walked in a last, sliced round: at the bench layout (4096^2, 20 %: two
//         windows of 65 blocks) 64 workgroups streamed the whole 32 KiB window per chunk again for THREE blocks, nothing shared in the L2, and
//         the kernel ended 8.5 us later on those XCDs (profiles/r06_updat_loop.md).  With `direct_max` > 0 (feature axis 1, partial-sum schedules,
//         at most that many blocks) such a block gets U2_DIRECT_PARTS extra workgroups BEHIND the schedule's: each multiplies one quarter of the
//         minibatch from the block's own 64-byte row pieces (2 KiB per 16 rows instead of 32) and leaves one partial sum for the summing pass.
//         They start as the first workgroups of the schedule retire and finish long before the last.
//   item: U2_ITEM = 4 + U2_WAVES * 5 words = (c0_block, k0_block, nblocks_in_item, 0) then per wave
//         word 0 = n0 | n1 << 4 | cidx0 << 8 | cidx1 << 12 | kidx[0] << 16 | kidx[1] << 20 | kidx[2] << 24 | kidx[3] << 28
//                  slots [0, n0) are blocks (cidx0, kidx[j]), slots [n0, n0 + n1) blocks (cidx1, kidx[j]) of the window
//         words 1..4 = weight block id of slot j (-1 = empty)
// =================================================================================================
namespace bsmm {

constexpr int32_t U2PLAN_MAGIC = 0x42535532;
constexpr int32_t U2PLAN_VERSION = 3;   // 2: block map behind the items (header words 26, 27); 3: direct blocks (header words 28 .. 31)
constexpr int U2_WAVES = 16;
constexpr int U2_SLOTS = 4;
constexpr int U2_WWORDS = 5;
constexpr int U2_ITEM = 4 + U2_WAVES * U2_WWORDS;
constexpr int U2_HDR = 32;   // 9 + 2 * 8 set descriptors, block map, longest set, direct blocks
constexpr int U2_DIRECT_PARTS = 4;       // workgroups (= quarters of the minibatch) per direct block
constexpr int U2_DIRECT_MAX = 64;        // more overflow blocks than this: overflow items as before (the call descriptor has 10 bits for the count)

inline long build_updat2_plan(const int32_t* updat_lut, int blocks, int CB, int KB, int ws, int32_t* out, int force_sets, int direct_max) {
    if (!updat_lut || blocks <= 0 || CB <= 0 || KB <= 0 || (ws != 8 && ws != 16 && ws != 32) || blocks >= (1 << 29)) return -1;
    const int WS = ws;
    const int wc = (CB + WS - 1) / WS, wk = (KB + WS - 1) / WS;
    struct Ent { int c, k, w; };
    std::vector<std::vector<Ent>> win((size_t)wc * wk);
    for (int w = 0; w < blocks; ++w) {
        const int c = updat_lut[2 * w], k = updat_lut[2 * w + 1];
        if (c < 0 || c >= CB || k < 0 || k >= KB) return -1;
        win[(size_t)(c / WS) * wk + (k / WS)].push_back({c, k, w});
    }
    struct Piece { int row; std::vector<Ent> e; };            // <= U2_SLOTS blocks of one window row
    struct Wave { std::vector<Piece> p; int n = 0; };
    // sets: 2 -> upper / lower half of the window rows; 8 -> a 4 x 2 grid of window patches (as square as the grid allows)
    long nwin = 0;
    for (auto& v : win) nwin += !v.empty();
    int nsets = (nwin >= 192 && wc >= 4 && wk >= 2) ? 8 : (wc >= 2 ? 2 : 1);
    if (force_sets == 1 || (force_sets == 2 && wc >= 2) || (force_sets == 4 && wc >= 2 && wk >= 2) || (force_sets == 8 && wc >= 4 && wk >= 2))
        nsets = force_sets;
    const int pr = nsets == 8 ? 4 : (nsets == 4 ? 2 : nsets), pc = nsets >= 4 ? 2 : 1;       // patch grid over the windows
    // items are collected per window ROW first: with two sets the split row is chosen afterwards so that both sets hold about
    // as many items (a Barabasi-Albert layout has three times the blocks in its first window rows: cutting the grid in the
    // middle gave 50 + 33 items, i.e. a whole extra round for half of the XCDs)
    std::vector<std::vector<std::vector<int32_t>>> row_items(wc), row_overflow(wc);
    std::vector<std::vector<int>> row_items_wj(wc), row_overflow_wj(wc);
    bool overflow_item = false;
    const bool direct_ok = direct_max > 0 && nsets <= 2;      // (8 sets: every item may be one workgroup's and stored directly -- no summing pass to meet in)
    std::vector<int32_t> direct;                              // (block, c, k, 0) per direct block
    // (Round 6 also tried LIGHT windows -- a window with <= 12 blocks, a quarter of the mean load, and the small overflow pieces of dense windows as
    //  direct blocks, up to 768 of them -- for unbalanced layouts: the reference's Barabasi-Albert bench layout 130.3 -> 133.4 us, power-law columns
    //  121.3 -> 144.6 us: a direct block costs four workgroups, more than a light item costs its window's stream.  profiles/r06_updat_light_windows.txt)
    auto emit = [&](int wi, int wj, const std::vector<Wave>& waves) {
        std::vector<int32_t> it(U2_ITEM, 0);
        int n = 0;
        for (int v = 0; v < U2_WAVES; ++v) {
            int32_t* wd = &it[4 + v * U2_WWORDS];
            wd[1] = wd[2] = wd[3] = wd[4] = -1;
            if (v >= (int)waves.size() || waves[v].p.empty()) continue;
            const Wave& W = waves[v];
            // window-local indices: 4 bits each in word 0; with 32 x 32 windows their fifth bit rides in the id words: bit 30 of slot j's word
            // = bit 4 of kidx[j], bit 29 of slot 0's / slot n0's word = bit 4 of cidx0 / cidx1
            const int n0 = (int)W.p[0].e.size(), n1 = W.p.size() > 1 ? (int)W.p[1].e.size() : 0;
            const int c0i = W.p[0].row - wi * WS, c1i = n1 ? W.p[1].row - wi * WS : 0;
            uint32_t m = (uint32_t)n0 | ((uint32_t)n1 << 4) | ((uint32_t)(c0i & 15) << 8) | ((uint32_t)(c1i & 15) << 12);
            int j = 0;
            for (const Piece& pc : W.p)
                for (const Ent& e : pc.e) {
                    const int ki = e.k - wj * WS;
                    m |= (uint32_t)(ki & 15) << (16 + 4 * j);
                    wd[1 + j] = e.w | ((ki >> 4) << 30);
                    ++j; ++n;
                }
            wd[1] |= (c0i >> 4) << 29;
            if (n1) wd[1 + n0] |= (c1i >> 4) << 29;
            wd[0] = (int32_t)m;
        }
        it[0] = wi * WS; it[1] = wj * WS; it[2] = n;
        (overflow_item ? row_overflow : row_items)[wi].push_back(std::move(it));
        (overflow_item ? row_overflow_wj : row_items_wj)[wi].push_back(wj);
    };
    for (int wj = 0; wj < wk; ++wj)                 // column-major over the windows: consecutive items share their DY panel
        for (int wi = 0; wi < wc; ++wi) {
            auto& v = win[(size_t)wi * wk + wj];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end(), [](const Ent& a, const Ent& b) { return a.c != b.c ? a.c < b.c : a.k < b.k; });
            // row pieces: whole groups of U2_SLOTS first, then the remainder of each row
            std::vector<Piece> full, part;
            for (size_t i = 0; i < v.size();) {
                size_t j = i;
                while (j < v.size() && v[j].c == v[i].c) ++j;
                for (size_t b = i; b < j; b += U2_SLOTS) {
                    Piece pc{v[i].c, std::vector<Ent>(v.begin() + b, v.begin() + std::min(j, b + U2_SLOTS))};
                    ((int)pc.e.size() == U2_SLOTS ? full : part).push_back(std::move(pc));
                }
                i = j;
            }
            std::sort(part.begin(), part.end(), [](const Piece& a, const Piece& b) { return a.e.size() > b.e.size(); });
            // pack: a wave holds one full piece, or one / two partial pieces with <= U2_SLOTS blocks.  Loose first (every
            // piece its own wave while the item has waves left: one row per wave, even matrix-pipe load), tight (best fit
            // decreasing) when that needs more than U2_WAVES waves.
            auto pack = [&](bool tight) {
                std::vector<Wave> ws;
                for (auto& pc : full) { Wave w; w.n = U2_SLOTS; w.p.push_back(pc); ws.push_back(std::move(w)); }
                for (auto& pc : part) {
                    const int sz = (int)pc.e.size();
                    Wave* best = nullptr;
                    for (auto& w : ws)
                        if (w.p.size() == 1 && w.n + sz <= U2_SLOTS && (!best || w.n > best->n)) best = &w;
                    if (best && (tight || ws.size() >= (size_t)U2_WAVES)) { best->p.push_back(pc); best->n += sz; }
                    else { Wave w; w.n = sz; w.p.push_back(pc); ws.push_back(std::move(w)); }
                }
                return ws;
            };
            std::vector<Wave> waves = pack(false);
            if (waves.size() > (size_t)U2_WAVES) waves = pack(true);
            // more than U2_WAVES waves: the window's main item takes the U2_WAVES most loaded waves, the few that remain
            // become small OVERFLOW items at the end of the set's list -- the kernel cuts the last, incomplete round of a set
            // into minibatch slices over all workgroups, so a small item there costs little (each slice adds its partial sums
            // to the scratch: a big item in that position doubled the atomic traffic of the bench layout)
            const size_t per_item = U2_WAVES;
            // spread the load over the four SIMDs: waves v, v+4, v+8, v+12 share a matrix pipe -> deal waves sorted by load
            std::sort(waves.begin(), waves.end(), [](const Wave& a, const Wave& b) { return a.n > b.n; });
            // (Round 6 also tried to CAP an item at 60 / 56 / 52 blocks and give the rest to the direct list -- items of 57 - 64 blocks are compute-paced,
            //  63 - 66 us against the 57.5 us of the stream-paced ones: 94.5 - 96.8 / 101.5 - 102.7 / 97.5 - 99.5 us per call against 92.5 - 93.1:
            //  a few dozen direct blocks x 4 workgroups cost more than the dense items lose, profiles/r06_updat_item_cap.txt)
            for (size_t beg = 0; beg < waves.size(); beg += per_item) {
                const size_t cnt = std::min<size_t>(per_item, waves.size() - beg);
                std::vector<Wave> dealt(U2_WAVES);
                int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
                // (waves 0..7 are wave set A of the kernel, 8..15 set B: dealing in decreasing load puts the two heaviest waves of every SIMD into set
                //  A.  Round 6 tried the balanced deal -- {heaviest, lightest} to A, the middle two to B: every item with more than 20 blocks in
                //  set B got SLOWER, 50-block items 57.5 -> 63.8 us per range: the item's time follows set B's load, profiles/r06_updat_loop.md)
                static const int kPos[4] = {0, 4, 8, 12};
                for (size_t i = 0; i < cnt; ++i) {
                    int s = -1;
                    for (int q = 0; q < 4; ++q)
                        if (used[q] < 4 && (s < 0 || load[q] < load[s])) s = q;
                    dealt[s + kPos[used[s]]] = waves[beg + i];
                    load[s] += waves[beg + i].n; ++used[s];
                }
                overflow_item = beg > 0;
                if (overflow_item && direct_ok) {
                    for (size_t i = 0; i < cnt; ++i)
                        for (const Piece& pcs : waves[beg + i].p)
                            for (const Ent& e : pcs.e) direct.insert(direct.end(), {e.w, e.c, e.k, 0});
                    continue;
                }
                emit(wi, wj, dealt);
            }
        }
    if ((long)direct.size() / 4 > direct_max) return build_updat2_plan(updat_lut, blocks, CB, KB, ws, out, force_sets, 0);   // too many: overflow items
    // window row -> row band of the set grid: equal thirds / halves of the grid, except for two sets: balanced item counts
    std::vector<int> band(wc);
    for (int wi = 0; wi < wc; ++wi) band[wi] = wi * pr / wc;
    if (nsets == 2) {
        long total_items = 0, acc_items = 0, best = -1;
        for (int wi = 0; wi < wc; ++wi) total_items += (long)(row_items[wi].size() + row_overflow[wi].size());
        int split = wc / 2;
        for (int r = 1; r < wc; ++r) {                      // rows [0, r) -> set 0
            acc_items += (long)(row_items[r - 1].size() + row_overflow[r - 1].size());
            const long diff = std::labs(2 * acc_items - total_items);
            if (best < 0 || diff < best) { best = diff; split = r; }
        }
        for (int wi = 0; wi < wc; ++wi) band[wi] = wi < split ? 0 : 1;
    }
    std::vector<std::vector<std::vector<int32_t>>> set_items(8), set_overflow(8);
    // (column-major over the windows inside a set, as the items were produced: consecutive items share their DY panel)
    for (int wj = 0; wj < wk; ++wj)
        for (int wi = 0; wi < wc; ++wi) {
            for (int o = 0; o < 2; ++o) {
                auto& src = o ? row_overflow[wi] : row_items[wi];
                auto& swj = o ? row_overflow_wj[wi] : row_items_wj[wi];
                for (size_t i = 0; i < src.size(); ++i) {
                    if (swj[i] != wj) continue;
                    const int set = (nsets == 1) ? 0 : ((band[wi] * pc + (wj * pc / wk)) % nsets);
                    (o ? set_overflow : set_items)[set].push_back(src[i]);
                }
            }
        }
    std::vector<int32_t> items;
    int32_t set_first[8] = {0}, set_count[8] = {0};
    for (int st = 0; st < 8; ++st) {
        set_first[st] = (int32_t)(items.size() / U2_ITEM);
        set_count[st] = (int32_t)(set_items[st].size() + set_overflow[st].size());
        for (auto& it : set_items[st]) items.insert(items.end(), it.begin(), it.end());
        for (auto& it : set_overflow[st]) items.insert(items.end(), it.begin(), it.end());
    }
    const long nitems = (long)(items.size() / U2_ITEM);
    const long off_bmap = U2_HDR + (long)items.size();
    const long n_direct = (long)direct.size() / 4;
    const long off_direct = (off_bmap + blocks + 3) & ~3L;
    const long total = n_direct > 0 ? off_direct + 4 * n_direct : off_bmap + blocks;
    if (out) {
        int32_t hdr[U2_HDR] = {U2PLAN_MAGIC, U2PLAN_VERSION, WS, U2_SLOTS, (int32_t)nitems, blocks, U2_HDR, U2_WAVES, nsets};
        for (int st = 0; st < 8; ++st) { hdr[9 + 2 * st] = set_first[st]; hdr[10 + 2 * st] = set_count[st]; }
        bool equal = true;
        int32_t longest = 0;
        for (int st = 0; st < nsets; ++st) { equal = equal && set_count[st] == set_count[0]; longest = std::max(longest, set_count[st]); }
        hdr[25] = equal ? set_count[0] : 0;
        hdr[26] = (int32_t)off_bmap;
        hdr[27] = longest;
        hdr[28] = (int32_t)n_direct; hdr[29] = n_direct > 0 ? (int32_t)off_direct : 0; hdr[30] = U2_DIRECT_PARTS; hdr[31] = 0;
        std::copy(hdr, hdr + U2_HDR, out);
        std::copy(items.begin(), items.end(), out + U2_HDR);
        // block -> (item, accumulator slot wave * U2_SLOTS + j) for the summing pass over the per-workgroup partial sums
        int32_t* bmap = out + off_bmap;
        std::fill(bmap, bmap + blocks, -1);
        for (long it = 0; it < nitems; ++it) {
            const int32_t* ip = items.data() + it * U2_ITEM;
            for (int wv = 0; wv < U2_WAVES; ++wv) {
                const int32_t* wd = ip + 4 + wv * U2_WWORDS;
                const int n = (wd[0] & 15) + ((wd[0] >> 4) & 15);
                for (int j = 0; j < n; ++j) {
                    const int32_t wid = wd[1 + j] & 0x1fffffff;                 // (bits 29 / 30: fifth index bits of 32 x 32 windows)
                    if (wd[1 + j] < 0 || wid >= blocks || bmap[wid] != -1) return -1;
                    bmap[wid] = (int32_t)((it << 8) | (wv * U2_SLOTS + j));
                }
            }
        }
        for (long d = 0; d < n_direct; ++d) {
            const int32_t wid = direct[4 * d];
            if (wid < 0 || wid >= blocks || bmap[wid] != -1) return -1;
            bmap[wid] = (int32_t)(-2 - d);
        }
        for (int w = 0; w < blocks; ++w) if (bmap[w] == -1) return -1;
        if (n_direct > 0) {
            std::fill(out + off_bmap + blocks, out + off_direct, 0);
            std::copy(direct.begin(), direct.end(), out + off_direct);
        }
    }
    return total;
}

}  // namespace bsmm
