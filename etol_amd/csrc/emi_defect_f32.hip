// emi_defect_f32.hip -- K4 for fp32 contexts (BASELINE config 5): defect rows += D.X on the f32
// matrix cores (v_mfma_f32_32x32x2_f32), in a form that survives single precision.
//
// A plain f32 GEMM of D.X is useless at M = 4096: |D_ij| reaches N(N+1)/4 ~ 4e6 and the terms
// cancel to O(1), so rounding X and D to f32 alone costs ~6e-8 * sum|D_ij x_j| ~ O(1) absolute.
// Because every row of D sums to zero, any constant may be subtracted from x first:
//        (D x)_i = sum_j D_ij (x_j - s)            for every s.
// Each 32-column output tile uses s = x at the tile's centre node: near the diagonal, where D is
// large, x_j - s is small (and exact in f32 by Sterbenz), so the products are O(x') instead of
// O(N^2 x).  The shift is applied when the A fragment is read from LDS (one v_sub per MFMA), so
// the contraction is still an ordinary MFMA GEMM.  The f32 copy of D is built on the host with
// an EXACTLY zero row sum in f32 (diagonal = -(f64 sum of the rounded off-diagonals)), which is
// what makes the shift free of bias (emi_api.hip, emi_set_mesh).
//
// Workgroup tile 64 rows x 128 columns, K tile 32, 4 waves as 2 x 2 (each 32 rows x 64 columns =
// two 32x32 accumulators), double-buffered LDS with register prefetch; LDS rows padded to 33 floats
// (conflict-free ds_read_b32 of the [row = lane & 31][k = lane >> 5] fragments).
// Requires M % 128 == 0; other shapes take the f64-accumulating fallback in emi_kernels.hip.
#include <type_traits>
#include <hip/hip_runtime.h>

#include "emi_kernels.hpp"
#include "emi_models.hpp"
#include "emi_node_kernels.hpp"

namespace emi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef EMI_F32_WGS_PER_CU
#define EMI_F32_WGS_PER_CU 2
#endif
// bid: tile id in XCD-local runs (an XCD gets a contiguous range, i.e. whole D column panels); nwg: tiles of the launch.
// ATOMIC: the epilogue adds its sums with no-return float atomics (the one-launch pass below) instead of a plain read-modify-write.
template <bool ATOMIC>
__device__ __forceinline__ void emi_defect_f32_body(const DefectArgsF32& a, const int bid, const int nwg) {
    constexpr int TM = 64, TN = 128, BK = 32, LDK = BK + 1;
    __shared__ float As[2][TM][LDK];
    __shared__ float Bs[2][TN][LDK];

    const int R = a.R, M = a.M;
    const int ntiles = M / TN;
    const int mtiles = nwg / ntiles;
    const int ntile = bid / mtiles, mtile = bid - ntile * mtiles;
    const int m0 = mtile * TM, n0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l32 = lane & 31, lk = lane >> 5;

    // per-lane shift of each of the wave's two column tiles: x[row][centre node of the tile]
    const int row = m0 + wr * 32 + l32;
    float shift[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
        shift[c] = row < R ? a.X[(size_t)row * M + n0 + wc * 64 + c * 32 + 16] : 0.f;

    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    float4 pa[2], pb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + 256 * p, r = idx >> 3, c4 = idx & 7;
            const float4 v = (m0 + r < R) ? *reinterpret_cast<const float4*>(a.X + (size_t)(m0 + r) * M + k0 + 4 * c4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            pa[p] = make_float4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + 256 * p, r = idx >> 3, c4 = idx & 7;
            const float4 v = *reinterpret_cast<const float4*>(a.D + (size_t)(n0 + r) * M + k0 + 4 * c4);
            pb[p] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + 256 * p, r = idx >> 3, c4 = idx & 7;
            float* d = &As[buf][r][4 * c4];
            d[0] = pa[p].x; d[1] = pa[p].y; d[2] = pa[p].z; d[3] = pa[p].w;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + 256 * p, r = idx >> 3, c4 = idx & 7;
            float* d = &Bs[buf][r][4 * c4];
            d[0] = pb[p].x; d[1] = pb[p].y; d[2] = pb[p].z; d[3] = pb[p].w;
        }
    };

    const int nkt = M / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0) the compiler's wait counting sees: no scalar load stays pending into the loop
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK);
        const float* Ar = &As[cur][wr * 32 + l32][lk];
        const float* B0 = &Bs[cur][wc * 64 + l32][lk];
        const float* B1 = &Bs[cur][wc * 64 + 32 + l32][lk];
        // Fragments one group (two k-steps, four MFMAs) ahead of the matrix pipe, in registers: read just in time, every
        // group waited for its LDS reads with the pipe idle (the ISA had s_waitcnt lgkmcnt(0) in front of two of every four
        // MFMAs).  sched_barrier pins the reads ahead of the MFMAs (the scheduler sinks them otherwise).
        struct Frag { float a[2], b0[2], b1[2]; };
        auto read_frag = [&](Frag& f, int g) {
            f.a[0] = Ar[4 * g];  f.a[1] = Ar[4 * g + 2];
            f.b0[0] = B0[4 * g]; f.b0[1] = B0[4 * g + 2];
            f.b1[0] = B1[4 * g]; f.b1[1] = B1[4 * g + 2];
        };
        Frag fr[2];
        read_frag(fr[0], 0);
#pragma unroll
        for (int g = 0; g < BK / 4; ++g) {
            if (g + 1 < BK / 4) read_frag(fr[(g + 1) & 1], g + 1);
            __builtin_amdgcn_sched_barrier(0);
            const Frag& f = fr[g & 1];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q] - shift[0], f.b0[q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q] - shift[1], f.b1[q], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < nkt) {
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // C/D map of the 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = n0 + wc * 64 + c * 32 + l32;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = m0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
            if (r < R) {
                const int inst = r / a.ns, st = r - inst * a.ns;
                float* o = a.RES + ((size_t)inst * a.nres + st) * M + n;
                if constexpr (ATOMIC) unsafeAtomicAdd(o, acc[c][i]);
                else *o += acc[c][i];
            }
        }
    }
}

__global__ __launch_bounds__(256, EMI_F32_WGS_PER_CU) void emi_defect_f32_mfma_kernel(DefectArgsF32 a) {
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: workgroups that share a D column panel share blockIdx % 8
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    emi_defect_f32_body<false>(a, bid, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// The fp32 evaluation pass (config 5) as ONE launch, like emi_pass_f64_kernel: MFMA-role workgroups (the body above) and
// node-role workgroups (emi_nodes_body) dealt per XCD by pass_role_of, COST finished in-kernel by ticket.  The two roles meet in the
// defect rows: the caller zeroes them, the node role ADDS -h f and the MFMA role ADDS D.X, both with no-return float atomics
// (global_atomic_add_f32: executed at the memory side, 50 MB of them per role at 256 instances against a ~1.3 TB/s rate).  Two
// contributions per element: 0 + a + b = 0 + b + a exactly, so the rows are bitwise what the sequential pair of launches leaves
// (-h f stored, then D.X added to it), whichever role gets to an element first.
struct PassArgsF32 {
    DefectArgsF32 d;
    NodeArgs<float> n;
    int nm8, nn8, nbx, order;
};
template <class Model>
__global__ __launch_bounds__(256, 3) void emi_pass_f32_kernel(PassArgsF32 a) {      // 3 workgroups per CU (50.7 KB of LDS each): <= 168 registers
    const int g = blockIdx.x, xcd = g & 7, j = g >> 3;
    const PassRole role = pass_role_of(j, a.nm8, a.nn8, a.order);
    if (role.mfma) {
        __builtin_amdgcn_s_setprio(3);
        emi_defect_f32_body<true>(a.d, xcd * a.nm8 + role.index, 8 * a.nm8);
    } else {
        const int nid = xcd * a.nn8 + role.index;
        emi_nodes_body<float, Model, 2, true, true, 0, true>(a.n, nid % a.nbx, nid / a.nbx, a.nbx);
    }
}

bool pass_f32_supported(int model, int R, int M, int B) {
    if (model != EMI_MODEL_FIXEDWING12 || M < 128 || M % 128 != 0) return false;
    const int nm = ((R + 63) / 64) * (M / 128), nn = ((M + 2 * EMI_NODE_THREADS - 1) / (2 * EMI_NODE_THREADS)) * B;
    return nm % 8 == 0 && nn % 8 == 0 && R % 64 == 0;
}

hipError_t launch_pass_f32(int model, const DefectArgsF32& d, const NodeArgs<float>& n, int order, hipStream_t s) {
    if (model != EMI_MODEL_FIXEDWING12) return hipErrorInvalidValue;
    PassArgsF32 a;
    a.d = d;
    a.n = n;
    const int nm = ((d.R + 63) / 64) * (d.M / 128);
    a.nbx = (n.M + 2 * EMI_NODE_THREADS - 1) / (2 * EMI_NODE_THREADS);
    const int nn = a.nbx * n.B;
    if (nm % 8 || nn % 8) return hipErrorInvalidConfiguration;
    a.nm8 = nm / 8;
    a.nn8 = nn / 8;
    a.order = order;
    hipLaunchKernelGGL((emi_pass_f32_kernel<FixedWing12<float>>), dim3(nm + nn), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ring form of the kernel above (round 3): the same tile (64 rows x 128 columns, K tile 32, 4 waves as 2 x 2, shifted-difference
// contraction), but the operands go global -> LDS by LDS-DMA into a 3-stage ring two K tiles ahead, issued from inline assembly with
// counted vmcnt waits, and the K loop is software-pipelined: the fragment reads of tile kt+1 and the DMA of tile kt+3 are dealt
// into the gaps between the MFMAs of tile kt -- the structure that took the f64 kernel from 0.172 to 0.133 ms
// (emi_symdefect_kernels.hpp, profiles/r02_notes.md section 9).  The register-staged form spends ~450 of every 2048 cycles of a K tile
// between two tiles' MFMAs (LDS stores, barrier, first fragment reads; 75 % of the matrix pipe busy whatever the occupancy:
// profiles/r03_notes.md section 6).
//   * A stage is 192 rows of 128 B (64 of X, 128 of D), 24 DMA wave instructions of 1 KB, six per wave.  A DMA instruction writes
//     8 rows linearly, so rows cannot be padded: 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) (applied to the
//     per-lane SOURCE address and to the fragment read address).
//   * The MFMA sums over k, so which k a lane holds is free as long as A and B agree: lane half lk takes the 16-byte chunks of its
//     parity, c = 2q + lk (k = 4c + j in MFMA j of the chunk) -- one ds_read_b128 per operand row and chunk pair, 12 per wave and K
//     tile (the first ring form read 8 bytes per chunk: 24 reads and 16 address additions per tile; every non-MFMA vector
//     instruction of a wave costs the matrix pipe 5 - 7 cycles here, profiles/r03_notes.md section 6).
//   * The K loop is unrolled over the ring (6 steps: 3 stages x 2 fragment sets), so stage offsets are immediates and the 8 per-lane
//     fragment addresses never change; the shifted operands are formed as packed additions of -shift (v_pk_add_f32).
typedef __attribute__((address_space(3))) void* emi_lds_ptr32_t;
__device__ __forceinline__ constexpr int f32ring_swz(int r) { return (r >> 1) & 7; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void emi_defect_f32_ring_kernel(DefectArgsF32 a) {
    constexpr int TM = 64, TN = 128, BK = 32, NST = 3, LOOK = NST - 1;
    constexpr int ROWS = TM + TN;                 // 192 rows of BK floats
    constexpr int STAGE = ROWS * BK;              // floats per stage (24 KB)
    constexpr int L = ROWS * 8 / 64 / 4;          // DMA instructions per wave and stage: 192 rows x 8 chunks / 64 lanes / 4 waves = 6
    extern __shared__ __attribute__((aligned(16))) float smf[];      // [NST][STAGE]

    const int R = a.R, M = a.M;
    const int ntiles = M / TN;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: workgroups that share a D column panel share blockIdx % 8
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int mtiles = gridDim.x / ntiles;
    const int ntile = bid / mtiles, mtile = bid - ntile * mtiles;
    const int m0 = mtile * TM, n0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int l32 = lane & 31, lk = lane >> 5;

    const int row = m0 + wr * 32 + l32;
    f32x2 nshift[2];                              // minus the shift of the wave's two column tiles, both halves (packed adds)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float sh = row < R ? a.X[(size_t)row * M + n0 + wc * 64 + c * 32 + 16] : 0.f;
        nshift[c] = f32x2{-sh, -sh};
    }

    // DMA addressing: instruction t of wave wid moves stage rows 8 (wid + 4t) .. +7, all of X or all of D: wave-uniform 64-bit base,
    // constant per-lane 32-bit offset, +128 bytes per K tile
    unsigned voff[L];
    unsigned long long gnext[L];
#pragma unroll
    for (int t = 0; t < L; ++t) {
        const int r0 = (wid + 4 * t) * 8;                 // wave-uniform first stage row of the instruction
        const int r = r0 + (lane >> 3), p = lane & 7;     // stage row, chunk position
        const int c = p ^ f32ring_swz(r);
        if (r0 < TM) {
            int gr = m0 + r;
            gr = gr < R ? gr : R - 1;                     // rows past the batch are never written out
            voff[t] = (unsigned)(((size_t)gr * M + 4 * c) * sizeof(float));
            gnext[t] = (unsigned long long)a.X;
        } else {
            voff[t] = (unsigned)(((size_t)(n0 + r - TM) * M + 4 * c) * sizeof(float));
            gnext[t] = (unsigned long long)a.D;
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long)(emi_lds_ptr32_t)smf;
    const int nkt = M / BK;
    int nd = 0;
    // DS = stage the tile goes to (compile-time in the K loop: the loop is unrolled over the ring)
    auto issue_one = [&](int t, int DS) {
        const unsigned dst = lds0 + (unsigned)(DS * STAGE * 4) + (unsigned)(wid + 4 * t) * 1024u;   // wave-uniform
        asm volatile("s_mov_b32 m0, %2\n\t" EMI_M0_NOP "global_load_lds_dwordx4 %0, %1" ::"v"(voff[t]), "s"(gnext[t]), "s"(dst) : "memory", "m0");
        gnext[t] += (unsigned long long)(BK * sizeof(float));
        if (t == L - 1) ++nd;
    };
    auto issue = [&](int DS) {
#pragma unroll
        for (int t = 0; t < L; ++t) issue_one(t, DS);
    };

    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

#pragma unroll
    for (int t = 0; t < LOOK; ++t)
        if (t < nkt) issue(t);

    // Fragment addresses.  The lane's A row ra and its B rows rb0, rb1 = rb0 + 32 have the SAME swizzle (l32 >> 1) & 7 (their row numbers
    // differ by multiples of 16), and lane half lk takes the chunks of its parity, c = 2q + lk: four ds_read_b128 per row and K tile,
    // at per-lane byte offsets that never change (offA[q], offB[q]); stage and the +32 rows of rb1 are immediate offsets.
    const int ra = wr * 32 + l32, rb0 = TM + wc * 64 + l32;
    const int sw = f32ring_swz(ra);
    const float* pA[4];
    const float* pB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pA[q] = smf + ra * BK + (((2 * q + lk) ^ sw) << 2);
        pB[q] = smf + rb0 * BK + (((2 * q + lk) ^ sw) << 2);
    }
    struct Frag { f32x4 b0[4], b1[4]; f32x2 s0[4][2], s1[4][2]; };      // s0 / s1: a - shift of the wave's two column tiles
    constexpr int NR = 12, NM = 32;                       // fragment reads / MFMAs per wave and K tile
    // MFMA i of a tile: q = i >> 3, element j = (i >> 1) & 3 of the lane's chunk 2q + lk (k = 4 (2q + lk) + j), column tile i & 1
    auto mfma_one = [&](const Frag& f, int i) {
        const int q = i >> 3, j = (i >> 1) & 3;
        if ((i & 1) == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.s0[q][j >> 1][j & 1], f.b0[q][j], acc[0], 0, 0, 0);
        else              acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.s1[q][j >> 1][j & 1], f.b1[q][j], acc[1], 0, 0, 0);
    };
    // K-loop step with everything about the ring known at compile time: the fragments of tile kt+1 come from stage RS, the DMA of
    // tile kt+3 goes to stage (RS + 2) % 3 (the stage of tile kt, whose fragments were read during the previous step).  One fragment
    // read in each of the first 12 gaps (A chunk first: its shifted copies are formed four gaps later, as packed adds), then nothing,
    // then one DMA instruction in each of gaps 24 .. 29.
    f32x4 araw[4];
    auto step = [&](const Frag& cur, Frag& nxt, int kt, auto rs_tag) {
        constexpr int RS = decltype(rs_tag)::value, DS = (RS + 2) % 3;
        if (kt + LOOK < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (LOOK - 1)) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0), visible to the compiler's wait counting
        asm volatile("s_barrier" ::: "memory");           // tile kt+1 has landed for every wave; the stage of tile kt is free
        const bool more = nd < nkt;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma_one(cur, i);
            if (i < NR) {
                const int q = i / 3, w = i % 3;
                if (w == 0) araw[q] = *reinterpret_cast<const f32x4*>(pA[q] + RS * STAGE);
                else if (w == 1) nxt.b0[q] = *reinterpret_cast<const f32x4*>(pB[q] + RS * STAGE);
                else nxt.b1[q] = *reinterpret_cast<const f32x4*>(pB[q] + RS * STAGE + 32 * BK);
            }
            if (i >= 4 && i < 4 + NR && (i - 4) % 3 == 0) {
                const int q = (i - 4) / 3;
                nxt.s0[q][0] = araw[q].xy + nshift[0];
                nxt.s0[q][1] = araw[q].zw + nshift[0];
                nxt.s1[q][0] = araw[q].xy + nshift[1];
                nxt.s1[q][1] = araw[q].zw + nshift[1];
            }
            if (i >= 24 && i < 24 + L) { if (more) issue_one(i - 24, DS); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto multiply = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < NM; ++i) mfma_one(f, i);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    Frag f0, f1;
    if (nkt > LOOK - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (LOOK - 1)) : "memory");
    else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (LOOK < nkt) issue(LOOK % NST);
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                         // tile 0 from stage 0
        const f32x4 av = *reinterpret_cast<const f32x4*>(pA[q]);
        f0.b0[q] = *reinterpret_cast<const f32x4*>(pB[q]);
        f0.b1[q] = *reinterpret_cast<const f32x4*>(pB[q] + 32 * BK);
        f0.s0[q][0] = av.xy + nshift[0];
        f0.s0[q][1] = av.zw + nshift[0];
        f0.s1[q][0] = av.xy + nshift[1];
        f0.s1[q][1] = av.zw + nshift[1];
    }
    // tile t lives in stage t % 3 and, by turns, in f0 / f1: six steps bring both back to where they started
    int kt = 0;
    for (; kt + 6 < nkt; kt += 6) {
        step(f0, f1, kt, S1{});
        step(f1, f0, kt + 1, S2{});
        step(f0, f1, kt + 2, S0{});
        step(f1, f0, kt + 3, S1{});
        step(f0, f1, kt + 4, S2{});
        step(f1, f0, kt + 5, S0{});
    }
    const int rem = nkt - kt;                             // 1 .. 6 tiles left, tile kt in f0 / stage 0
    if (rem > 1) step(f0, f1, kt, S1{});
    if (rem > 2) step(f1, f0, kt + 1, S2{});
    if (rem > 3) step(f0, f1, kt + 2, S0{});
    if (rem > 4) step(f1, f0, kt + 3, S1{});
    if (rem > 5) step(f0, f1, kt + 4, S2{});
    if (rem & 1) multiply(f0);
    else         multiply(f1);

    // C/D map of the 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = n0 + wc * 64 + c * 32 + l32;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = m0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
            if (r < R) {
                const int inst = r / a.ns, st = r - inst * a.ns;
                float* o = a.RES + ((size_t)inst * a.nres + st) * M + n;
                *o += acc[c][i];
            }
        }
    }
}

bool defect_f32_mfma_supported(int M) { return M >= 128 && M % 128 == 0; }

hipError_t launch_defect_f32_mfma(const DefectArgsF32& a, hipStream_t s, int ring, int wgs_per_cu) {
    const int mtiles = (a.R + 63) / 64, ntiles = a.M / 128;
    dim3 grid(mtiles * ntiles), block(256);
    if (ring) {
        // 72 KB: two workgroups per CU.  wgs_per_cu == 1: the launch ASKS for 100 KB, so that one workgroup takes a CU -- the kernel is as
        // fast that way (0.8879 against 0.8908 ms, profiles/r03_notes.md section 6) and leaves 300 registers per SIMD to the waves of a
        // node kernel running beside it on a second stream (with two ring workgroups per CU at 212 registers each none fits)
        const size_t lds_min = (size_t)3 * (64 + 128) * 32 * sizeof(float), lds_one = (size_t)100 * 1024;
        const size_t lds = wgs_per_cu == 1 ? lds_one : lds_min;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)emi_defect_f32_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_one);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(emi_defect_f32_ring_kernel, grid, block, lds, s, a);
    } else {
        hipLaunchKernelGGL(emi_defect_f32_mfma_kernel, grid, block, 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace emi
