// bsmm_plan.h -- host-side schedule ("plan") for the grouped xprop kernel, derived from a reference-format
// xprop lookup table.  Pure host code (no HIP calls): the caller uploads the resulting int32 array to the
// device like any other lookup table and passes it in bsmm_args.plan.
//
// Idea: G consecutive output feature blocks form a group.  A workgroup that owns one group and one
// minibatch tile walks the UNION of the group's input blocks once, in ascending order; each step names one
// input block c, a bit mask of the group members that have a nonzero block (c, ob) and the weight-block ids
// of those members in ascending member order.  Steps are batched into stages whose weight blocks fit the
// LDS staging buffer.  Segmentation/lock ids of the source table are irrelevant here (one writer per output).
//
// Two step flavours (header word [14]):
//   pair = 0: one step per input block c                         member_mask bit = member
//   pair = 1: one step per PAIR of input blocks (2p, 2p+1)       member_mask bit = 2*member + (c & 1); steps[].in_block = p
//             (axis-1 activations keep the two blocks of a pair in one 128-byte line per minibatch row, so a pair
//              step costs the L2 the same number of requests as a single block)
//
// Layout (int32):
//   [0] magic  [1] version  [2] G  [3] SB (max weight blocks per stage)  [4] ngroups  [5] nstages  [6] nsteps
//   [7] nblocks  [8] off_groups  [9] off_stages  [10] off_steps  [11] off_wlist  [12] n_out_blocks  [13] off_meta  [14] pair
//   groups[ngroups][4] = (stage_beg, nstages, first_out_block, n_out_blocks_in_group)
//   stages[nstages][4] = (step_beg, nsteps, w_beg, nw)
//   steps [nsteps][2]  = (in_block, member_mask)
//   wlist [nblocks]    = weight block ids, in step order then ascending member order
//   meta  [nblocks]    = member | (256 if this block is the first of its step), same order as wlist
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace bsmm {

constexpr int32_t PLAN_MAGIC = 0x42534d50;
constexpr int32_t PLAN_VERSION = 3;
constexpr int PLAN_HDR = 16;

struct PlanEntry {
    int32_t c, member, w;   // c = step key (input block, or pair index), member = mask bit
};

// Builds the plan into `out` (may be null: size query).  Returns the number of int32 words, or -1 on bad input.
inline long build_xprop_plan(const int32_t* lut, int segments, int blocks, int n_out_blocks, int G, int SB, int pair, int32_t* out) {
    if (!lut || segments <= 0 || blocks <= 0 || n_out_blocks <= 0 || G < 1 || G > (pair ? 16 : 32) || SB < 1) return -1;
    const int ngroups = (n_out_blocks + G - 1) / G;
    std::vector<std::vector<PlanEntry>> per_group(ngroups);
    for (int s = 0; s < segments; ++s) {
        const int32_t off = lut[4 * s], cnt = lut[4 * s + 1], ob = lut[4 * s + 2];
        if (ob < 0 || ob >= n_out_blocks || cnt < 0) return -1;
        for (int e = 0; e < cnt; ++e) {
            const int32_t c = lut[2 * (off + e)], w = lut[2 * (off + e) + 1];
            if (w < 0 || w >= blocks || c < 0) return -1;
            if (pair) per_group[ob / G].push_back({c >> 1, 2 * (ob % G) + (c & 1), w});
            else      per_group[ob / G].push_back({c, ob % G, w});
        }
    }
    std::vector<int32_t> groups, stages, steps, wlist, meta;
    groups.reserve(4 * ngroups);
    wlist.reserve(blocks);
    for (int g = 0; g < ngroups; ++g) {
        auto& ents = per_group[g];
        std::sort(ents.begin(), ents.end(), [](const PlanEntry& a, const PlanEntry& b) {
            return a.c != b.c ? a.c < b.c : a.member < b.member;
        });
        const int stage_beg = (int)(stages.size() / 4);
        int st_step_beg = (int)(steps.size() / 2), st_w_beg = (int)wlist.size();
        size_t i = 0;
        while (i < ents.size()) {
            size_t j = i;
            uint32_t mask = 0;
            while (j < ents.size() && ents[j].c == ents[i].c) {
                mask |= 1u << ents[j].member;
                ++j;
            }
            if (j - i > (size_t)SB) {   // a step that cannot fit one stage is split (same key, disjoint masks)
                j = i + SB;
                mask = 0;
                for (size_t t = i; t < j; ++t) mask |= 1u << ents[t].member;
            }
            const int nb = (int)(j - i);
            if ((int)wlist.size() - st_w_beg + nb > SB) {   // close the current stage
                stages.insert(stages.end(), {st_step_beg, (int)(steps.size() / 2) - st_step_beg, st_w_beg, (int)wlist.size() - st_w_beg});
                st_step_beg = (int)(steps.size() / 2);
                st_w_beg = (int)wlist.size();
            }
            steps.push_back(ents[i].c);
            steps.push_back((int32_t)mask);
            for (size_t t = i; t < j; ++t) {
                wlist.push_back(ents[t].w);
                meta.push_back(ents[t].member | (t == i ? 256 : 0));
            }
            i = j;
        }
        if ((int)(steps.size() / 2) > st_step_beg)
            stages.insert(stages.end(), {st_step_beg, (int)(steps.size() / 2) - st_step_beg, st_w_beg, (int)wlist.size() - st_w_beg});
        const int nob = std::min(G, n_out_blocks - g * G);
        groups.insert(groups.end(), {stage_beg, (int)(stages.size() / 4) - stage_beg, g * G, nob});
    }
    const long total = PLAN_HDR + (long)groups.size() + (long)stages.size() + (long)steps.size() + 2 * (long)wlist.size();
    if (out) {
        int32_t* h = out;
        const int off_groups = PLAN_HDR;
        const int off_stages = off_groups + (int)groups.size();
        const int off_steps = off_stages + (int)stages.size();
        const int off_wlist = off_steps + (int)steps.size();
        const int off_meta = off_wlist + (int)wlist.size();
        const int32_t hdr[PLAN_HDR] = {PLAN_MAGIC, PLAN_VERSION, G, SB, ngroups, (int32_t)(stages.size() / 4), (int32_t)(steps.size() / 2),
                                       (int32_t)wlist.size(), off_groups, off_stages, off_steps, off_wlist, n_out_blocks, off_meta, pair, 0};
        std::copy(hdr, hdr + PLAN_HDR, h);
        std::copy(groups.begin(), groups.end(), h + off_groups);
        std::copy(stages.begin(), stages.end(), h + off_stages);
        std::copy(steps.begin(), steps.end(), h + off_steps);
        std::copy(wlist.begin(), wlist.end(), h + off_wlist);
        std::copy(meta.begin(), meta.end(), h + off_meta);
    }
    return total;
}

}  // namespace bsmm
