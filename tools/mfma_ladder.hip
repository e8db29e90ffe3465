// Diagnostic ladder: start from a bare MFMA loop and add the conv kernel's ingredients one at a time.
//   L0 registers only | L1 + swizzled ds_read_b128 fragments (TM+TN per 4*TM*TN MFMAs) | L2 + one barrier per K-step
//   L3 + LDS-DMA of (BM+BN) x 128 B per K-step from an L2-resident buffer into a 3-stage ring (counted vmcnt)
//   L4 the conv's real gather traffic | L5 = L4 with the A rows fetched by global_load_lds (64-bit per-lane addresses)
//   NST (template): ring stages; the DMA pointer runs NST - 1 stages ahead, the counted wait leaves (NST - 2) * NDMA in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;
using glb_ptr_t = __attribute__((address_space(1))) void*;
constexpr int BK = 32;
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int BM, int BN, int LEVEL, int NST = 3>
__global__ __launch_bounds__(256) void k_ladder(const float* src, unsigned src_bytes, float* out, int steps, int rows_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = BM / 64, TN = BN / 64, RA = BM / 32, RB = BN / 32, NDMA = RA + RB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + NST * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5, rsub = lane >> 3, pc = lane & 7;
    for (int i = tid; i < NST * (BM + BN) * BK; i += 256) As[i] = 1e-3f * (float)((i * 37) % 101 - 50);
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    unsigned aoff[RA], boff[RB];
    for (int i = 0; i < RA; ++i) aoff[i] = (unsigned)(((blockIdx.x * BM + 8 * (wave + 4 * i) + rsub) % rows_total) * 1024 + 16 * pc);
    for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((8 * (wave + 4 * i) + rsub) % rows_total) * 1024 + 16 * pc + 512);
    // LEVEL 4: the conv's real traffic -- A = 92160 rows x 1 KiB (K = 256 fp32), 7 taps = the tile's rows shifted by
    // {0, +64, +1, -63, -64, -1, +63} (a 64-wide chart), 8 channel chunks of 128 B; B = weight panel [7][128][1 KiB]
    // behind the activations; 56 K-steps per 64-row tile, tiles b, b + grid, ...
    const int a_rows = 92160;
    const unsigned b_base = (unsigned)a_rows * 1024u;
    const int shift[7] = {0, 64, 1, -63, -64, -1, 63};
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fl = swz(l31);
    f32x4 fa[2][TM], fb[2][TN];
    for (int i = 0; i < TM; ++i) fa[0][i] = fa[1][i] = f32x4{1e-3f * lane, 2e-3f, -1e-3f, 5e-4f};
    for (int j = 0; j < TN; ++j) fb[0][j] = fb[1][j] = f32x4{-1e-3f, 1e-3f * (lane & 7), 3e-3f, 1e-3f};
    int ring = 0, iring = NST - 1;             // the DMA pointer runs NST - 1 stages ahead
    auto dma = [&](int slot, int step) {
        if (LEVEL >= 4) {
            const int tile = (blockIdx.x + (step / 56) * gridDim.x) % (a_rows / BM);
            const int t = step % 7, kc = (step / 7) % 8;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                int row = tile * BM + 8 * (wave + 4 * i) + rsub + shift[t];
                row = row < 0 ? row + a_rows : (row >= a_rows ? row - a_rows : row);
                if (LEVEL >= 5)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(reinterpret_cast<const char*>(src) + (size_t)row * 1024u + 16u * pc + kc * 128),
                                                     (lds_ptr_t)(As + slot * BM * BK + 8 * (wave + 4 * i) * BK), 16, 0, 0);
                else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(As + slot * BM * BK + 8 * (wave + 4 * i) * BK), 16,
                                                         (unsigned)row * 1024u + 16u * pc, kc * 128, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(Bs + slot * BN * BK + 8 * (wave + 4 * i) * BK), 16,
                                                         b_base + (unsigned)(t * BN + 8 * (wave + 4 * i) + rsub) * 1024u + 16u * pc,
                                                         kc * 128, 0, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(As + slot * BM * BK + 8 * (wave + 4 * i) * BK), 16, aoff[i],
                                                     (step & 7) * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(Bs + slot * BN * BK + 8 * (wave + 4 * i) * BK), 16, boff[i],
                                                     (step & 7) * 128, 0, 0);
    };
    if (LEVEL >= 3) {
#pragma unroll
        for (int q = 0; q < NST - 1; ++q) dma(q, q);
    }
    for (int step = 0; step < steps; ++step) {
        const float* a_base = As + ring * BM * BK + (wr * (BM / 2) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / 2) + l31) * BK;
        if (LEVEL >= 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + 4 * (h ^ fl));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + 4 * (h ^ fl));
        }
        if (LEVEL >= 3) dma(iring, step + NST - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (LEVEL >= 1 && kk < 3) {
                const int off = 4 * ((2 * (kk + 1) + h) ^ fl);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LEVEL >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NDMA) : "memory");   // stage step+1 landed
        if (LEVEL >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        ring = ring == NST - 1 ? 0 : ring + 1;
        iring = iring == NST - 1 ? 0 : iring + 1;
    }
    float sum = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
#endif
}

// Waves per block: the production kernels run 4 waves (2 x 2) per block; NW = 8 arranges them WR x WC over the tile, each wave a
// (BM / WR) x (BN / WC) sub-tile of 32 x 32 MFMA tiles -- same LDS per block, twice the waves per SIMD to cover a block's barrier.
// Level-4 traffic only (the conv's real gather), 3-stage ring.
template <int BM, int BN, int WR, int WC, int NST = 3>
__global__ __launch_bounds__(64 * WR * WC) void k_ladder_w(const float* src, unsigned src_bytes, float* out, int steps, int rows_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WR * WC;
    constexpr int TM = BM / WR / 32, TN = BN / WC / 32, RA = BM / 8 / NW, RB = BN / 8 / NW, NDMA = RA + RB;
    static_assert(TM >= 1 && TN >= 1 && RA >= 1 && RB >= 1, "tile too small for this wave grid");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + NST * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC, l31 = lane & 31, h = lane >> 5, rsub = lane >> 3, pc = lane & 7;
    for (int i = tid; i < NST * (BM + BN) * BK; i += 64 * NW) As[i] = 1e-3f * (float)((i * 37) % 101 - 50);
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const int a_rows = 92160;
    const unsigned b_base = (unsigned)a_rows * 1024u;
    const int shift[7] = {0, 64, 1, -63, -64, -1, 63};
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fl = swz(l31);
    f32x4 fa[2][TM], fb[2][TN];
    int ring = 0, iring = NST - 1;
    auto dma = [&](int slot, int step) {
        const int tile = (blockIdx.x + (step / 56) * gridDim.x) % (a_rows / BM);
        const int t = step % 7, kc = (step / 7) % 8;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int row = tile * BM + 8 * (wave + NW * i) + rsub + shift[t];
            row = row < 0 ? row + a_rows : (row >= a_rows ? row - a_rows : row);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(As + slot * BM * BK + 8 * (wave + NW * i) * BK), 16,
                                                     (unsigned)row * 1024u + 16u * (pc ^ swz(8 * (wave + NW * i) + rsub)), kc * 128, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(Bs + slot * BN * BK + 8 * (wave + NW * i) * BK), 16,
                                                     b_base + (unsigned)(t * BN + 8 * (wave + NW * i) + rsub) * 1024u +
                                                         16u * (pc ^ swz(8 * (wave + NW * i) + rsub)), kc * 128, 0, 0);
    };
#pragma unroll
    for (int q = 0; q < NST - 1; ++q) dma(q, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int step = 0; step < steps; ++step) {
        const float* a_base = As + ring * BM * BK + (wr * (BM / WR) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / WC) + l31) * BK;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + 4 * (h ^ fl));
        dma(iring, step + NST - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int off = 4 * ((2 * (kk + 1) + h) ^ fl);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NDMA) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ring = ring == NST - 1 ? 0 : ring + 1;
        iring = iring == NST - 1 ? 0 : iring + 1;
    }
    float sum = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(blockIdx.x * 64 * NW + tid) % (1024 * 256)] = sum;
#endif
}

static int g_steps_div = 1;      // mode 's': launches of ~200 us instead of ~4 ms (the chip holds a higher clock on short launches)
template <int BM, int BN, int WR, int WC, int NST = 3>
void run_w(int blocks_per_cu, const float* src, unsigned src_bytes, float* out) {
    const int blocks = 256 * blocks_per_cu, steps = 4000 / blocks_per_cu * 8192 / (BM * BN) / g_steps_div;
    const size_t lds = (size_t)NST * (BM + BN) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ladder_w<BM, BN, WR, WC, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < (g_steps_div > 1 ? 12 : 3); ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_ladder_w<BM, BN, WR, WC, NST>), dim3(blocks), dim3(64 * WR * WC), lds, 0, src, src_bytes, out, steps, (int)(src_bytes / 1024));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * steps * 2.0 * BM * BN * BK;
    printf("tile %3dx%-3d  %d waves (%d x %d)  %d-stage ring  %d block(s)/CU  %.3f ms  %.1f TFLOP/s\n", BM, BN, WR * WC, WR, WC, NST, blocks_per_cu, best, flops / best / 1e9);
}

// Producer / consumer waves ("warp specialisation"): WR x WC consumer waves do nothing but fragment reads + MFMAs + the step
// barrier; ONE extra wave issues every LDS-DMA of a stage (all (BM + BN) / 8 instructions), waits for the previous stage with
// the counted vmcnt and joins the same barrier.  The consumers' instruction streams then contain no DMA issue, no address
// arithmetic and no vmcnt wait -- the 13 % "stage issue" of the production K-step (profiles/r03_kstep_decomposition.txt) moves
// to a wave that does not use the MFMA pipe.  Level-4 traffic, 3-stage ring.
template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(64 * (WR * WC + 1)) void k_ladder_p(const float* src, unsigned src_bytes, float* out, int steps, int rows_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WR * WC, NST = 3;
    constexpr int TM = BM / WR / 32, TN = BN / WC / 32, NA = BM / 8, NB = BN / 8, NDMA = NA + NB;   // DMA instructions per stage (one wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + NST * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, rsub = lane >> 3, pc = lane & 7;
    for (int i = tid; i < NST * (BM + BN) * BK; i += 64 * (NW + 1)) As[i] = 1e-3f * (float)((i * 37) % 101 - 50);
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const int a_rows = 92160;
    const unsigned b_base = (unsigned)a_rows * 1024u;
    const int shift[7] = {0, 64, 1, -63, -64, -1, 63};
    if (wave == NW) {
        // ---- producer
        auto dma = [&](int slot, int step) {
            const int tile = (blockIdx.x + (step / 56) * gridDim.x) % (a_rows / BM);
            const int t = step % 7, kc = (step / 7) % 8;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int row = tile * BM + 8 * i + rsub + shift[t];
                row = row < 0 ? row + a_rows : (row >= a_rows ? row - a_rows : row);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(As + slot * BM * BK + 8 * i * BK), 16,
                                                         (unsigned)row * 1024u + 16u * (pc ^ swz(8 * i + rsub)), kc * 128, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(Bs + slot * BN * BK + 8 * i * BK), 16,
                                                         b_base + (unsigned)(t * BN + 8 * i + rsub) * 1024u + 16u * (pc ^ swz(8 * i + rsub)),
                                                         kc * 128, 0, 0);
        };
        dma(0, 0);
        dma(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int iring = NST - 1;
        for (int step = 0; step < steps; ++step) {
            dma(iring, step + NST - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA > 63 ? 63 : NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            iring = iring == NST - 1 ? 0 : iring + 1;
        }
        return;
    }
    // ---- consumers
    const int wr = wave / WC, wc = wave % WC;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fl = swz(l31);
    f32x4 fa[2][TM], fb[2][TN];
    int ring = 0;
    __builtin_amdgcn_s_barrier();
    for (int step = 0; step < steps; ++step) {
        const float* a_base = As + ring * BM * BK + (wr * (BM / WR) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / WC) + l31) * BK;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int off = 4 * ((2 * (kk + 1) + h) ^ fl);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ring = ring == NST - 1 ? 0 : ring + 1;
    }
    float sum = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(blockIdx.x * 64 * NW + tid) % (1024 * 256)] = sum;
#endif
}

template <int BM, int BN, int WR, int WC>
void run_p(int blocks_per_cu, const float* src, unsigned src_bytes, float* out) {
    const int blocks = 256 * blocks_per_cu, steps = 4000 / blocks_per_cu * 8192 / (BM * BN);
    const size_t lds = (size_t)3 * (BM + BN) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ladder_p<BM, BN, WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_ladder_p<BM, BN, WR, WC>), dim3(blocks), dim3(64 * (WR * WC + 1)), lds, 0, src, src_bytes, out, steps, (int)(src_bytes / 1024));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * steps * 2.0 * BM * BN * BK;
    printf("tile %3dx%-3d  %d consumer waves (%d x %d) + 1 producer  %d block(s)/CU  %.3f ms  %.1f TFLOP/s\n", BM, BN, WR * WC, WR, WC, blocks_per_cu, best, flops / best / 1e9);
}

// "BK = 64 as two stages": 4 ring slots = 2 double-stages; at the top of a double-step the two stages of the NEXT double-step
// are issued, both stages are computed, then vmcnt(0) + one barrier -- half the barriers per FLOP, 2 blocks per CU for 64x64.
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_ladder2(const float* src, unsigned src_bytes, float* out, int steps, int rows_total) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = BM / 64, TN = BN / 64, RA = BM / 32, RB = BN / 32, NST = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + NST * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5, rsub = lane >> 3, pc = lane & 7;
    for (int i = tid; i < NST * (BM + BN) * BK; i += 256) As[i] = 1e-3f * (float)((i * 37) % 101 - 50);
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const int a_rows = 92160;
    const unsigned b_base = (unsigned)a_rows * 1024u;
    const int shift[7] = {0, 64, 1, -63, -64, -1, 63};
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fl = swz(l31);
    f32x4 fa[2][TM], fb[2][TN];
    auto dma = [&](int slot, int step) {
        const int tile = (blockIdx.x + (step / 56) * gridDim.x) % (a_rows / BM);
        const int t = step % 7, kc = (step / 7) % 8;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int row = tile * BM + 8 * (wave + 4 * i) + rsub + shift[t];
            row = row < 0 ? row + a_rows : (row >= a_rows ? row - a_rows : row);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(As + slot * BM * BK + 8 * (wave + 4 * i) * BK), 16,
                                                     (unsigned)row * 1024u + 16u * pc, kc * 128, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(Bs + slot * BN * BK + 8 * (wave + 4 * i) * BK), 16,
                                                     b_base + (unsigned)(t * BN + 8 * (wave + 4 * i) + rsub) * 1024u + 16u * pc,
                                                     kc * 128, 0, 0);
    };
    auto compute = [&](int ring) {
        const float* a_base = As + ring * BM * BK + (wr * (BM / 2) + l31) * BK;
        const float* b_base = Bs + ring * BN * BK + (wc * (BN / 2) + l31) * BK;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + 4 * (h ^ fl));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int off = 4 * ((2 * (kk + 1) + h) ^ fl);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * BK + off);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * BK + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    dma(0, 0);
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int half = 0;                                   // slots {0,1} or {2,3} hold the double-stage being computed
    for (int step = 0; step < steps; step += 2) {
        dma(2 * (half ^ 1), step + 2);
        dma(2 * (half ^ 1) + 1, step + 3);
        compute(2 * half);
        compute(2 * half + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        half ^= 1;
    }
    float sum = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
#endif
}

template <int BM, int BN>
void run2(int blocks_per_cu, const float* src, unsigned src_bytes, float* out) {
    const int blocks = 256 * blocks_per_cu, steps = 4000 / blocks_per_cu;
    const size_t lds = (size_t)4 * (BM + BN) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ladder2<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_ladder2<BM, BN>), dim3(blocks), dim3(256), lds, 0, src, src_bytes, out, steps, (int)(src_bytes / 1024));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * steps * 2.0 * BM * BN * BK;
    printf("tile %3dx%-3d level 4  double-step (barrier every 2 K-steps, 4 slots)  %d block(s)/CU  %.3f ms  %.1f TFLOP/s\n", BM, BN, blocks_per_cu, best, flops / best / 1e9);
}

template <int BM, int BN, int LEVEL, int NST = 3>
void run(int blocks_per_cu, const float* src, unsigned src_bytes, float* out) {
    const int blocks = 256 * blocks_per_cu, steps = 4000 / blocks_per_cu;
    const size_t lds = (size_t)NST * (BM + BN) * BK * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ladder<BM, BN, LEVEL, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_ladder<BM, BN, LEVEL, NST>), dim3(blocks), dim3(256), lds, 0, src, src_bytes, out, steps, (int)(src_bytes / 1024));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * steps * 2.0 * BM * BN * BK;
    printf("tile %3dx%-3d level %d  %d-stage ring  %d block(s)/CU  %.3f ms  %.1f TFLOP/s\n", BM, BN, LEVEL, NST, blocks_per_cu, best, flops / best / 1e9);
}

int main(int argc, char** argv) {
    const unsigned src_bytes = 96u << 20;   // >= 92160 KiB of activations + 0.9 MB of weights
    float *src, *out;
    hipMalloc(&src, src_bytes); hipMalloc(&out, 1024 * 256 * 4);
    {   // random normal-ish operands: zero / trivial data lets the chip hold a higher clock than real activations do
        std::vector<float> h(src_bytes / 4);
        unsigned long long st = 88172645463325252ull;
        for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = ((int)(st & 0xffff) - 32768) / 16384.0f * ((st >> 20) & 1 ? 1.f : 0.37f); }
        if (argc > 1 && argv[1][0] == 'z') std::fill(h.begin(), h.end(), 0.f);
        hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice);
        printf("source data: %s\n", (argc > 1 && argv[1][0] == 'z') ? "zeros" : "random");
    }
#define LADDER(BM, BN, OCC) run<BM, BN, 0>(OCC, src, src_bytes, out); run<BM, BN, 1>(OCC, src, src_bytes, out); \
                            run<BM, BN, 2>(OCC, src, src_bytes, out); run<BM, BN, 3>(OCC, src, src_bytes, out); \
                            run<BM, BN, 4>(OCC, src, src_bytes, out);
    if (argc > 1 && argv[1][0] == 'r') {   // ring depth vs occupancy at the conv's real traffic (level 4), each measured twice
        for (int rep = 0; rep < 2; ++rep) {
            run<64, 64, 4, 3>(3, src, src_bytes, out);      // production: 48 KB, 3 blocks / CU
            run<64, 64, 4, 4>(2, src, src_bytes, out);      // 64 KB, 2 blocks / CU
            run<64, 64, 4, 3>(2, src, src_bytes, out);      // control: 3 stages at 2 blocks / CU
            run<64, 128, 4, 3>(2, src, src_bytes, out);     // production: 72 KB, 2 blocks / CU
            run<64, 128, 4, 4>(1, src, src_bytes, out);     // 96 KB, 1 block / CU
            run<64, 128, 4, 2>(2, src, src_bytes, out);     // 48 KB: would fit 3 blocks / CU
            run<64, 128, 4, 2>(3, src, src_bytes, out);
            run<64, 64, 4, 2>(3, src, src_bytes, out);      // 32 KB: control at production occupancy
            run<64, 64, 4, 2>(4, src, src_bytes, out);      // ... and with the freed LDS spent on a 4th block
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {   // barrier every second K-step (BK = 64 as two stages) against the production configs
        for (int rep = 0; rep < 2; ++rep) {
            run<64, 64, 4, 3>(3, src, src_bytes, out);      // production: 48 KB, 3 blocks / CU
            run2<64, 64>(2, src, src_bytes, out);           // 64 KB, 2 blocks / CU
            run2<64, 64>(1, src, src_bytes, out);
            run<64, 128, 4, 3>(2, src, src_bytes, out);     // production: 72 KB, 2 blocks / CU
            run2<64, 128>(1, src, src_bytes, out);          // 96 KB, 1 block / CU
            run<128, 64, 4, 3>(2, src, src_bytes, out);
            run2<128, 64>(1, src, src_bytes, out);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'w') {   // waves per block at the conv's real traffic
        for (int rep = 0; rep < 2; ++rep) {
            run_w<64, 128, 2, 2>(2, src, src_bytes, out);     // production shape
            run_w<64, 128, 2, 4>(2, src, src_bytes, out);     // 8 waves of 32 x 32
            run_w<64, 64, 2, 2>(3, src, src_bytes, out);      // production shape
            run_w<128, 64, 2, 2>(2, src, src_bytes, out);
            run_w<128, 64, 4, 2>(2, src, src_bytes, out);     // 8 waves of 32 x 32
            run_w<128, 128, 2, 2>(1, src, src_bytes, out);
            run_w<128, 128, 2, 4>(1, src, src_bytes, out);    // 8 waves of 64 x 32
            run_w<128, 128, 4, 4>(1, src, src_bytes, out);    // 16 waves of 32 x 32
            run_w<128, 256, 2, 4>(1, src, src_bytes, out);    // 144 KB: 8 waves of 64 x 64
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 's') {   // SHORT launches (~200 us, as in the training step): wave count at the clock short launches hold
        g_steps_div = 20;
        for (int rep = 0; rep < 3; ++rep) {
            run_w<64, 128, 2, 2>(2, src, src_bytes, out);        // production shape
            run_w<64, 128, 2, 4>(2, src, src_bytes, out);        // 8 waves of 32 x 32
            run_w<128, 128, 2, 4>(1, src, src_bytes, out);       // 8 waves of 64 x 32
            run_w<64, 64, 2, 2>(3, src, src_bytes, out);
            run_w<128, 64, 2, 2>(2, src, src_bytes, out);
            run_w<128, 64, 4, 2>(2, src, src_bytes, out);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'n') {   // ring depth for the eight-wave blocks (round 5): does a 4-stage ring (prefetch distance 3) pay at one block per CU?
        for (int rep = 0; rep < 2; ++rep) {
            run_w<64, 128, 2, 2>(2, src, src_bytes, out);        // production shape
            run_w<64, 128, 2, 4>(2, src, src_bytes, out);        // 8 waves of 32 x 32, 72 KB
            run_w<128, 128, 2, 4>(1, src, src_bytes, out);       // 8 waves of 64 x 32, 96 KB
            run_w<128, 128, 2, 4, 4>(1, src, src_bytes, out);    // ... 4 stages, 128 KB
            run_w<128, 128, 4, 2, 4>(1, src, src_bytes, out);    // 8 waves of 32 x 64
            run_w<128, 128, 4, 4, 4>(1, src, src_bytes, out);    // 16 waves of 32 x 32, 4 stages
            run_w<128, 64, 4, 2, 4>(1, src, src_bytes, out);     // 128 x 64, 8 waves, 4 stages at ONE block (96 KB): control
            run_w<128, 64, 4, 2, 3>(2, src, src_bytes, out);
            run_w<64, 128, 2, 4, 4>(1, src, src_bytes, out);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {   // producer / consumer waves against the production structure
        for (int rep = 0; rep < 2; ++rep) {
            run_w<64, 128, 2, 2>(2, src, src_bytes, out);     // production shape
            run_p<64, 128, 2, 2>(2, src, src_bytes, out);
            run_w<64, 64, 2, 2>(3, src, src_bytes, out);      // production shape
            run_p<64, 64, 2, 2>(3, src, src_bytes, out);
            run_w<128, 64, 2, 2>(2, src, src_bytes, out);
            run_p<128, 64, 2, 2>(2, src, src_bytes, out);
            run_w<128, 128, 2, 2>(1, src, src_bytes, out);
            run_p<128, 128, 2, 2>(1, src, src_bytes, out);
            run_p<128, 128, 2, 4>(1, src, src_bytes, out);    // 8 consumers of 64 x 32 + 1 producer
            run_p<64, 128, 2, 4>(2, src, src_bytes, out);     // 8 consumers of 32 x 32 + 1 producer
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'g') {   // buffer vs global LDS-DMA addressing of the gathered rows
        run<64, 128, 4>(2, src, src_bytes, out); run<64, 128, 5>(2, src, src_bytes, out);
        run<64, 64, 4>(3, src, src_bytes, out); run<64, 64, 5>(3, src, src_bytes, out);
        run<128, 64, 4>(2, src, src_bytes, out); run<128, 64, 5>(2, src, src_bytes, out);
        return 0;
    }
    LADDER(128, 128, 1)
    LADDER(128, 64, 2)
    LADDER(64, 128, 2)
    LADDER(64, 64, 3)
    LADDER(128, 64, 1)
    return 0;
}
