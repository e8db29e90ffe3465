// Stream-K schedule of the persistent conv GEMM (k_conv_dma_sk, icn_kernels.hip): which (tile, k-chunk range) segments a
// block runs, in which order, and where the pieces of a split tile come from.  Shared by the kernel and by the host
// (icn_table_stream_k: the schedule as a table, checked on the CPU by tests/test_stream_k_plan.py), so it is plain arithmetic.
#pragma once
#if defined(__HIPCC__)
#define ICN_HD __host__ __device__
#else
#define ICN_HD
#endif

namespace icn {

// Plan of one XCD's share of a launch (q_x tiles for GL persistent blocks, nku units of k-chunks per tile).
// Whole tiles are dealt round-robin as before (local tile indices < dp_l); the last 1 + frac rounds (nsk tiles) are cut into
// GL equal ranges of units, so a block's tail is never a mostly idle round.
struct SkPlan { int dp_l, nsk; };
ICN_HD inline SkPlan sk_plan(int q_x, int GL, int nku) {
    const int R = q_x / GL, frac = q_x - R * GL;
    SkPlan p{q_x, 0};
    if (frac == 0) return p;                              // whole rounds: nothing to balance
    if (R >= 1) {                                         // ranges of 1 .. 2 tiles: a tile is shared by at most 3 blocks
        p.dp_l = (R - 1) * GL;
        p.nsk = GL + frac;
    } else if ((long)frac * nku >= 2L * GL && 4 * frac >= GL) {
        p.dp_l = 0;                                       // fewer tiles than blocks: >= 2 units per block, <= 5 blocks per tile
        p.nsk = frac;
    }
    return p;
}

// A block's walk: whole tiles bl, bl + GL, ... of its XCD residue class first, then its range [u0, u1) of the class's split
// tiles BACK TO FRONT.  Tile ids are global (local index * 8 + residue); k ranges are in k-chunks (unit = ku k-chunks).
//   piece with k1 < nk : does not reach its tile's end -- parked in the block's slot (at most one per block, and it is the
//                        first split-phase segment of the walk)
//   piece with k0 > 0, k1 == nk : finishes its tile -- adds the pieces parked by blocks bl - 1, bl - 2, ... (same residue
//                        class, LOWER ids) until the tile's first unit is covered: range_start(nb) gives where block nb's
//                        range begins
struct SkWalk {
    int round, u0, pos, dp_l, GL, x, bl, nku, ku;
    long U;                                               // units of this residue class's split tiles
    ICN_HD void init(int block, int grid, int ntiles, int nk, int ku_) {
        x = block % 8; bl = block / 8; GL = grid / 8; ku = ku_; nku = nk / ku_;
        const SkPlan pl = sk_plan(ntiles / 8 + (x < ntiles % 8 ? 1 : 0), GL, nku);
        dp_l = pl.dp_l;
        U = (long)pl.nsk * nku;
        u0 = (int)(U * bl / GL);
        pos = (int)(U * (bl + 1) / GL);
        round = 0;
    }
    ICN_HD int range_start(int nb) const { return (int)(U * nb / GL); }
    ICN_HD int next(int& tile, int& k0, int& k1) {
        const int li = bl + round * GL;
        if (li < dp_l) {
            ++round;
            tile = li * 8 + x; k0 = 0; k1 = nku * ku;
            return 1;
        }
        if (pos > u0) {
            const int lt = (pos - 1) / nku;
            const int e1 = pos - lt * nku, len = e1 < pos - u0 ? e1 : pos - u0, e0 = e1 - len;
            pos -= len;
            k0 = e0 * ku; k1 = e1 * ku;
            tile = (dp_l + lt) * 8 + x;
            return 1;
        }
        return 0;
    }
};

}  // namespace icn
