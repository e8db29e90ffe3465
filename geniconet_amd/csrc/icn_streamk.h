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

// Plan of one XCD's share of a launch (q_x tiles for GL persistent blocks, nku = the most pieces a tile may be cut into).
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

// Boundaries of the GL ranges a residue class's U = nsk * S split-phase K-steps are cut into: bnd[b] .. bnd[b + 1] is block
// b's range (host only: the launcher builds the table once per launch shape and the kernel reads it with scalar loads).
//   * Targets: block b's WHOLE share (whole tiles + range) is proportional to a speed factor fac[slot] / 1000, slot =
//     b * nslots / GL.  The grid is nslots = blocks-per-CU rounds of one block per CU, dispatched in order, so the blocks of
//     "slot" k are the k-th arrivals on their CUs -- and the earlier arrival wins the issue arbitration for as long as it is
//     resident (measured with icn_debug_trace on equal shares: two blocks per CU, the first exited first on 256 of 256 CUs,
//     9 us ahead on a 210 us launch; three per CU: 185 / 205 / 225 us).  Equal finishing times keep every CU at its full
//     occupancy to the end.  fac = null: equal shares.  (The dispatch order is an observation, not a contract: if it does
//     not hold the shares are off by the factors' few per cent, nothing else.)
//   * A boundary may not fall within `mp` steps of a tile edge without lying on it, or a piece shorter than mp steps would
//     result (the kernel's ring fill and metadata look-ahead need 4).  Boundaries are chosen in sequence: among the allowed
//     positions within 3 steps of the target, the one that brings the PREVIOUS block's load closest to its share -- snapping
//     every boundary to the nearest allowed position independently lets a block gain at both ends (+6 steps of ~80; with
//     whole k-chunks as units, rounds 1-2, every fourth block of a typical launch carried +5: 14 us of a 220 us launch).
inline void sk_boundaries(int q, int GL, int S, int mp, int nslots, const int* fac, int* bnd) {
    const SkPlan pl = sk_plan(q, GL, S / mp > 0 ? S / mp : 1);
    const long U = (long)pl.nsk * S;
    const int whole = GL > 0 ? pl.dp_l / GL : 0;          // whole tiles per block ahead of the split phase
    if (nslots < 1 || fac == nullptr) nslots = 1;
    auto F = [&](int b) {                                 // sum of the factors of blocks < b
        long f = 0;
        for (int k = 0; k < nslots; ++k) {
            const int lo = (int)((long)GL * k / nslots), hi = (int)((long)GL * (k + 1) / nslots);
            const int n = b <= lo ? 0 : (b >= hi ? hi - lo : b - lo);
            f += (long)n * (fac ? fac[k] : 1000);
        }
        return f;
    };
    auto target = [&](int b) {                            // cumulative split-phase steps of blocks < b
        if (U == 0) return 0L;
        long c = (long)q * S * F(b) / F(GL) - (long)whole * S * b;
        return c < 0 ? 0L : (c > U ? U : c);
    };
    auto allowed = [&](long p) {
        const long r = p % S;
        return r == 0 || (r >= mp && S - r >= mp);
    };
    bnd[0] = 0;
    for (int b = 1; b <= GL; ++b) {
        const long x = b == GL ? U : target(b), want = x - target(b - 1);
        long best = -1, best_err = 0;
        for (long p = x - 3; p <= x + 3; ++p) {
            if (p < bnd[b - 1] || p > U || !allowed(p)) continue;
            if (p != bnd[b - 1] && p - bnd[b - 1] < mp && p / S == bnd[b - 1] / S && p % S != 0) continue;   // piece too short
            const long len = p - bnd[b - 1];
            const long err = 8 * (len > want ? len - want : want - len) + (p > x ? p - x : x - p);
            if (best < 0 || err < best_err) { best = p; best_err = err; }
        }
        if (b == GL) best = U;
        if (best < 0) best = bnd[b - 1];                  // nothing allowed in reach: an empty range
        bnd[b] = (int)best;
    }
}

// A block's walk: whole tiles bl, bl + GL, ... of its XCD residue class first, then its range [bnd[bl], bnd[bl + 1]) of the
// class's split tiles BACK TO FRONT.  Tile ids are global (local index * 8 + residue).  Ranges are in K-STEPS (S per tile;
// up to round 2 they were whole k-chunks of 7 steps).
//   piece with s1 < S : does not reach its tile's end -- parked in the block's slot (at most one per block, and it is the
//                       first split-phase segment of the walk)
//   piece with s0 > 0, s1 == S : finishes its tile -- adds the pieces parked by blocks bl - 1, bl - 2, ... (same residue
//                       class, LOWER ids) until the tile's first step is covered: range_start(nb) gives where block nb's
//                       range begins
struct SkWalk {
    int round, u0, pos, dp_l, GL, x, bl, S;
    const int* bnd;                                       // [GL + 1] boundaries of this block's residue class (sk_boundaries)
    ICN_HD int range_start(int nb) const { return bnd[nb]; }
    ICN_HD void init(int block, int grid, int ntiles, int steps_per_tile, int min_piece, const int* tables) {
        x = block % 8; bl = block / 8; GL = grid / 8; S = steps_per_tile;
        const int mp = min_piece < 1 ? 1 : min_piece, q = ntiles / 8 + (x < ntiles % 8 ? 1 : 0);
        const SkPlan pl = sk_plan(q, GL, S / mp > 0 ? S / mp : 1);
        dp_l = pl.dp_l;
        bnd = tables + (x < ntiles % 8 ? 0 : GL + 1);     // two tables: classes with ntiles / 8 + 1 tiles, then with ntiles / 8
        u0 = bnd[bl];
        pos = bnd[bl + 1];
        round = 0;
    }
    ICN_HD int next(int& tile, int& s0, int& s1) {
        const int li = bl + round * GL;
        if (li < dp_l) {
            ++round;
            tile = li * 8 + x; s0 = 0; s1 = S;
            return 1;
        }
        if (pos > u0) {
            const int lt = (pos - 1) / S;
            const int e1 = pos - lt * S, len = e1 < pos - u0 ? e1 : pos - u0, e0 = e1 - len;
            pos -= len;
            s0 = e0; s1 = e1;
            tile = (dp_l + lt) * 8 + x;
            return 1;
        }
        return 0;
    }
};
// the two boundary tables of a launch, [2][GL + 1]: residue classes with ntiles / 8 + 1 tiles first (empty when ntiles % 8 == 0)
inline void sk_tables(int ntiles, int grid, int S, int mp, int nslots, const int* fac, int* out) {
    const int GL = grid / 8;
    sk_boundaries(ntiles / 8 + 1, GL, S, mp, nslots, fac, out);
    sk_boundaries(ntiles / 8, GL, S, mp, nslots, fac, out + GL + 1);
}

}  // namespace icn
