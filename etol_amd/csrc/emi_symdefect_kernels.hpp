// emi_symdefect_kernels.hpp -- the even/odd fp64 MFMA defect kernels (see emi_symdefect.hip for the
// derivation and the measurements that shaped them).  Model-templated: instantiated for the built-in
// models in emi_symdefect.hip and, through hiprtc, for generated model structs (emi_rtc.hip).
#pragma once
#include "emi_args.hpp"
#include "emi_node_kernels.hpp"

namespace emi {

typedef double d4 __attribute__((ext_vector_type(4)));

// CT = 16-column tiles per wave (4 waves side by side): tile = 16 instances x 64*CT half-indices.
// CT = 1 keeps a wave at ~216 registers, so three 96-register waves of the node kernel fit on
// the same SIMD; CT = 2 halves the re-reads of X but takes the whole register file.
template <class Model, int CT>
__global__ __launch_bounds__(256, CT == 1 ? 2 : 1) void emi_symdefect_f64_kernel(SymDefectArgs a) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    constexpr int TI = FUSED_TI, TM = NS * TI, TN = 64 * CT, BK = FUSED_BK, LDK = BK + 2;
    constexpr int A_PASS = TM * (BK / 2) / 256;   // double2 pieces per thread for E/O
    constexpr int B_PASS = TN * (BK / 2) / 256;   // and for De / Do
    static_assert(TM * (BK / 2) % 256 == 0 && TN * (BK / 2) % 256 == 0, "staging shape");
    static_assert(TN % 64 == 0 && TI == 16, "wave layout below assumes 16 instances x 64*CT columns");
    constexpr int STAGE = (2 * TM + 2 * TN) * LDK;   // doubles per LDS buffer

    extern __shared__ __attribute__((aligned(16))) double smem[];   // [2][STAGE]: E, O, De, Do

    const int M = a.M, Hh = M >> 1, B = a.B;
    const int mtiles = (B + TI - 1) / TI;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: workgroups that share a De/Do panel share blockIdx % 8
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    // a.order 0: an XCD walks a De/Do panel over all instance groups (panel stays in its L2, X is
    // re-read once per panel from the Infinity Cache); 1: the ntiles workgroups of one instance
    // group run together on one XCD (X tiles are shared in L2, De/Do cycles through it)
    const int ntiles_ = gridDim.x / mtiles;
    const int ntile = a.order ? bid % ntiles_ : bid / mtiles;
    const int mtile = a.order ? bid / ntiles_ : bid - ntile * mtiles;
    const int inst0 = mtile * TI, i0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    d4 acc_a[NS][CT], acc_b[NS][CT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            acc_a[s][c] = d4{0.0, 0.0, 0.0, 0.0};
            acc_b[s][c] = d4{0.0, 0.0, 0.0, 0.0};
        }

    // ---- staging: next K tile global -> registers (e/o formed on the fly) -> other LDS buffer
    double2 pe[A_PASS], po[A_PASS], pde[B_PASS], pdo[B_PASS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx >> 3, c2 = idx & 7;
            const int inst = inst0 + (row & 15), st = row >> 4;
            double2 v = make_double2(0.0, 0.0), m = v;
            if (inst < B) {
                const double* xr = a.X + ((size_t)inst * NS + st) * M;
                const int j = k0 + 2 * c2;
                v = *reinterpret_cast<const double2*>(xr + j);            // x_j, x_{j+1}
                m = *reinterpret_cast<const double2*>(xr + (M - 2 - j));  // x_{N-j-1}, x_{N-j}
            }
            pe[p] = make_double2(v.x + m.y, v.y + m.x);
            po[p] = make_double2(v.x - m.y, v.y - m.x);
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx >> 3, c2 = idx & 7;
            const size_t off = (size_t)(i0 + row) * Hh + k0 + 2 * c2;
            const double2 te = *reinterpret_cast<const double2*>(a.De + off);
            const double2 to = *reinterpret_cast<const double2*>(a.Do + off);
            pde[p] = make_double2(te.x, te.y);
            pdo[p] = make_double2(to.x, to.y);
        }
    };
    auto lstore = [&](int buf) {
        double* Es = smem + (size_t)buf * STAGE;
        double* Os = Es + TM * LDK;
        double* Des = Os + TM * LDK;
        double* Dos = Des + TN * LDK;
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx >> 3, c2 = idx & 7;
            *reinterpret_cast<double2*>(Es + row * LDK + 2 * c2) = pe[p];
            *reinterpret_cast<double2*>(Os + row * LDK + 2 * c2) = po[p];
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx >> 3, c2 = idx & 7;
            *reinterpret_cast<double2*>(Des + row * LDK + 2 * c2) = pde[p];
            *reinterpret_cast<double2*>(Dos + row * LDK + 2 * c2) = pdo[p];
        }
    };

    const int nkt = Hh / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK);   // in flight under this tile's MFMAs
        const double* Es = smem + (size_t)cur * STAGE;
        const double* Os = Es + TM * LDK;
        const double* Db = Os + TM * LDK + (wid * (16 * CT) + r16) * LDK + kq;
        const double* Ob = Db + TN * LDK;
#pragma unroll 2
        for (int ks = 0; ks < BK / 4; ++ks) {
            double bfe[CT], bfo[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                bfe[c] = Db[c * 16 * LDK + ks * 4];
                bfo[c] = Ob[c * 16 * LDK + ks * 4];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double afe = Es[(s * 16 + r16) * LDK + ks * 4 + kq];
                const double afo = Os[(s * 16 + r16) * LDK + ks * 4 + kq];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc_a[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afe, bfe[c], acc_a[s][c], 0, 0, 0);
                    acc_b[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afo, bfo[c], acc_b[s][c], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nkt) {
            lstore(cur ^ 1);               // the other buffer: nobody reads it during this tile
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: defect = D.X - h f, forward node i and mirrored node N-i ---------------
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = wid * (16 * CT) + c * 16 + r16;
        const int node_f = i0 + col, node_m = M - 1 - node_f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = inst0 + kq + 4 * i;
            if (inst >= B) continue;
            const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
            const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
            double* __restrict__ Rb = a.RES + (size_t)inst * a.nres * M;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int node = side == 0 ? node_f : node_m;
                double z[NV], f[NS];
#pragma unroll
                for (int v = 0; v < NS; ++v) z[v] = Xb[(size_t)v * M + node];
#pragma unroll
                for (int v = 0; v < NC; ++v) z[NS + v] = Ub[(size_t)v * M + node];
                Model::f(a.P, z, a.node_t[node], f);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double dx = side == 0 ? acc_a[s][c][i] + acc_b[s][c][i] : acc_b[s][c][i] - acc_a[s][c][i];
                    Rb[(size_t)s * M + node] = dx - a.h * f[s];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ring variant (CT = 1): operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no
// staging registers) into a 3-stage ring, two K tiles ahead of the MFMAs, with a COUNTED vmcnt so
// that the loads stay in flight across the one barrier per K tile.  Raw X tiles are stored (the
// forward range and the mirrored range); e = x_j + x_{N-j}, o = x_j - x_{N-j} are formed when the
// fragments are read.  An LDS-DMA wave instruction writes 1 KB linearly (8 rows of 128 B), so the
// rows cannot be padded: instead 16-byte chunk c of row r is stored at chunk position
// c ^ ((r >> 1) & 7) (the swizzle is applied to the per-lane SOURCE address and again to the read
// address), which keeps the ds_read_b64 fragment reads conflict-free.
// With no staging registers a wave needs ~150 VGPRs, and two K tiles of look-ahead tolerate the
// operand latency seen while the node kernel saturates HBM on the same CUs.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* emi_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* emi_glb_ptr_t;

template <class Model>
__global__ __launch_bounds__(256, 2) void emi_symdefect_ring_f64_kernel(SymDefectArgs a) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    constexpr int TI = FUSED_TI, TM = NS * TI, TN = 64, BK = 16, NST = 3;
    constexpr int STAGE = (2 * TM + 2 * TN) * BK;        // doubles per ring stage
    constexpr int NINSTR = (2 * TM + 2 * TN) * 8 / 64;   // LDS-DMA wave instructions per stage
    constexpr int L = NINSTR / 4;                        // per wave (4 waves)
    static_assert(NINSTR % 4 == 0, "stage does not split evenly over 4 waves");

    extern __shared__ __attribute__((aligned(16))) double smem[];   // [NST][STAGE]: XF, XM, De, Do

    const int M = a.M, Hh = M >> 1, B = a.B;
    const int mtiles = (B + TI - 1) / TI;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    // a.order 0: an XCD walks a De/Do panel over all instance groups (panel stays in its L2, X is
    // re-read once per panel from the Infinity Cache); 1: the ntiles workgroups of one instance
    // group run together on one XCD (X tiles are shared in L2, De/Do cycles through it)
    const int ntiles_ = gridDim.x / mtiles;
    const int ntile = a.order ? bid % ntiles_ : bid / mtiles;
    const int mtile = a.order ? bid / ntiles_ : bid - ntile * mtiles;
    const int inst0 = mtile * TI, i0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- per-lane source pointers of this wave's L DMA instructions (tile 0), fixed per launch
    const double* src[L];
    int kdir[L];                                       // +1: advances with k0, -1: mirrored range
#pragma unroll
    for (int t = 0; t < L; ++t) {
        const int q = (wid + 4 * t) * 64 + lane;       // 16-byte chunk id within the stage
        int row, p;
        if (q < 2 * TM * 8) {
            const bool mir = q >= TM * 8;
            const int qr = mir ? q - TM * 8 : q;
            row = qr >> 3; p = qr & 7;
            const int c = p ^ ((row >> 1) & 7);
            int inst = inst0 + (row & 15);
            inst = inst < B ? inst : B - 1;            // rows past the batch are never written out
            const double* xr = a.X + ((size_t)inst * NS + (row >> 4)) * M;
            src[t] = mir ? xr + (M - BK) + 2 * c : xr + 2 * c;
            kdir[t] = mir ? -1 : 1;
        } else {
            const bool od = q >= (2 * TM + TN) * 8;
            const int qr = q - (2 * TM + (od ? TN : 0)) * 8;
            row = qr >> 3; p = qr & 7;
            const int c = p ^ ((row >> 1) & 7);
            src[t] = (od ? a.Do : a.De) + (size_t)(i0 + row) * Hh + 2 * c;
            kdir[t] = 1;
        }
    }
    // The DMA is issued from inline assembly: an LDS-DMA instruction the COMPILER knows about makes it put
    // "s_waitcnt vmcnt(0)" in front of every later LDS read (it cannot tell which ring stage the DMA writes), i.e. right
    // after the DMA of tile kt+LOOK is issued the wave waited for that very tile -- no tile in flight at all, every K tile
    // paid a full DMA round trip (the ISA of rounds 1-2; profiles/r02_notes.md section 9).  The counted vmcnt waits of
    // the K loop are the only synchronisation with the DMA the ring needs.  m0 = LDS address of the 1 KB the wave writes.
    // (s_nop 0 between the SALU write of m0 and the DMA: gfx9-family parts need one wait state there, and the hazard
    // recogniser does not look inside inline assembly -- the builtin form gets the same s_nop from the compiler.)
    const unsigned lds0 = (unsigned)(unsigned long)(emi_lds_ptr_t)smem;          // LDS byte address of the ring
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int t = 0; t < L; ++t) {
            const unsigned dst = lds0 + (unsigned)stage * (unsigned)(STAGE * 8) + (unsigned)(wid + 4 * t) * 1024u;   // wave-uniform
            const double* g = src[t] + (ptrdiff_t)kdir[t] * kt * BK;
            asm volatile("s_mov_b32 m0, %1\n\t" EMI_M0_NOP "global_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
        }
    };

    d4 acc_a[NS], acc_b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc_a[s] = d4{0.0, 0.0, 0.0, 0.0};
        acc_b[s] = d4{0.0, 0.0, 0.0, 0.0};
    }

    const int nkt = Hh / BK;
    issue(0, 0);
    if (nkt > 1) issue(1, 1);
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed once all but this wave's L youngest DMA instructions are done
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");     // every wave's part landed; stage (kt-1)%3 is free
        if (kt + 2 < nkt && !(a.ablate & 2)) issue((kt + 2) % NST, kt + 2);
        if (a.ablate & 1) continue;
        const double* XF = smem + (size_t)(kt % NST) * STAGE;
        const double* XM = XF + TM * BK;
        const double* DE = XM + TM * BK;
        const double* DO = DE + TN * BK;
        // Fragment reads are software-pipelined by hand: with one wave per SIMD nothing else
        // hides the LDS latency, so the raw values of k-step ks+1 are requested before the
        // MFMAs of k-step ks are issued (the compiler then waits with a counted lgkmcnt).
        const int rb = wid * 16 + r16, swb = (rb >> 1) & 7;
        double xf_c[NS], xm_c[NS], be_c, bo_c;
        auto frag = [&](int ks, double (&xf)[NS], double (&xm)[NS], double& be, double& bo) {
            const int kk = ks * 4 + kq, km = BK - 1 - kk;          // forward / mirrored position
            const int pb = ((kk >> 1) ^ swb) * 2 + (kk & 1);
            be = DE[rb * BK + pb];
            bo = DO[rb * BK + pb];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int r = s * 16 + r16, sw = (r >> 1) & 7;
                xf[s] = XF[r * BK + ((kk >> 1) ^ sw) * 2 + (kk & 1)];
                xm[s] = XM[r * BK + ((km >> 1) ^ sw) * 2 + (km & 1)];
            }
        };
        frag(0, xf_c, xm_c, be_c, bo_c);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double xf_n[NS], xm_n[NS], be_n = 0.0, bo_n = 0.0;
            if (ks + 1 < BK / 4) frag(ks + 1, xf_n, xm_n, be_n, bo_n);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc_a[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf_c[s] + xm_c[s], be_c, acc_a[s], 0, 0, 0);
                acc_b[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf_c[s] - xm_c[s], bo_c, acc_b[s], 0, 0, 0);
            }
            if (ks + 1 < BK / 4) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { xf_c[s] = xf_n[s]; xm_c[s] = xm_n[s]; }
                be_c = be_n;
                bo_c = bo_n;
            }
        }
    }

    // ---- epilogue: defect = D.X - h f, forward node i and mirrored node N-i ---------------
    const int col = wid * 16 + r16;
    const int node_f = i0 + col, node_m = M - 1 - node_f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = inst0 + kq + 4 * i;
        if (inst >= B || (a.ablate & 4)) continue;
        const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
        const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
        double* __restrict__ Rb = a.RES + (size_t)inst * a.nres * M;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int node = side == 0 ? node_f : node_m;
            double z[NV], f[NS];
#pragma unroll
            for (int v = 0; v < NS; ++v) z[v] = Xb[(size_t)v * M + node];
#pragma unroll
            for (int v = 0; v < NC; ++v) z[NS + v] = Ub[(size_t)v * M + node];
            Model::f(a.P, z, a.node_t[node], f);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double dx = side == 0 ? acc_a[s][i] + acc_b[s][i] : acc_b[s][i] - acc_a[s][i];
                Rb[(size_t)s * M + node] = dx - a.h * f[s];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ring kernel, second form: SW states per workgroup, 8-deep K tiles, 16-byte fragment reads.
//   * A ring stage is 20 KB (SW = 6; three stages 60 KB), so TWO workgroups share a CU: the prologue (two DMA
//     round trips), the epilogue (f at the output nodes, stores) and the barrier skew of one workgroup are
//     covered by the other's MFMAs.  Measured with operands in registers (tools/diag/clock_probe.hip):
//     v_mfma_f64_16x16x4_f64 issues every 64-68 cycles per SIMD at 2.39 GHz with one wave per SIMD and with
//     two, i.e. 74-76 TFLOP/s: the matrix pipe is not what held the first ring kernel at 37.
//   * One ds_read_b128 per operand row and K tile.  The MFMA sums over k, so WHICH k a lane holds in which
//     k-step is free as long as A and B agree: lane group kq takes k = 2kq in the first step of a tile and
//     k = 2kq+1 in the second, i.e. one 16-byte chunk per row.  (With k = 4 ks + kq the compiler pairs the
//     8-byte reads into ds_read2st64_b64, which moves half the bytes per LDS cycle and, with 32 banks in
//     its lane groups, was 2-way conflicted on this image: 224 LDS cycles per wave and tile against 56 now.)
//   * SW < NS splits the states of an instance group over NS/SW workgroups (each re-reads the De/Do panel,
//     which is L2-resident): a shard of config 4 (128 instances) then launches 384 workgroups instead of 64.
// Tile: 16 instances x SW states (rows of XF / XM, 64 B each) against 64 half-indices (rows of De / Do); 4 waves,
// wave w owns half-indices 16w .. 16w+15.  Row r keeps its 16-byte chunk c at position c ^ swz(r), swz(r) =
// (-(r >> 2)) & 3, applied to the DMA source address and to the fragment read address: each of the four
// 16-lane groups of a ds_read_b128 then covers a 256-byte bank row exactly once (MI355X_MICROARCH.md, LDS).
// ---------------------------------------------------------------------------------------------
EMI_DEV constexpr int ring_swz(int r) { return (0 - (r >> 2)) & 3; }
#ifndef EMI_EPILOGUE_PREFETCH
#define EMI_EPILOGUE_PREFETCH 0         // build-time A/B switch (tools/ab_build.sh): the epilogue's first loads ahead of the K loop.
#endif                                  // Measured (profiles/r04_notes.md section 13): no gain at any batch size, off.
// The same for K tiles of depth BK (rows of BK doubles = BK / 2 sixteen-byte chunks): the sixteen rows r .. r + 15 a lane group of a
// ds_read_b128 touches at one logical chunk must land on sixteen different 16-byte bank slots of a 256-byte bank row.  BK = 8 (64-byte
// rows, four to a bank row): the chunk position changes every four rows, ring_swz above.  BK = 16 (128-byte rows, two to a bank row):
// it changes every two rows, eight positions.
template <int BK> EMI_DEV constexpr int ring_swz_k(int r) { return BK == 8 ? ring_swz(r) : ((r >> 1) & 7); }

// defect = D.X - h f at the tile's output nodes: forward node i (a + b) and mirrored node N-i (b - a), states
// S0 .. S0+SW-1; lane (r16, kq) of wave wid holds half-index column 16 wid + r16 and instances kq + 4 i.
// S0 is a template parameter: with the state group known at compile time only the components of f this workgroup
// writes are computed and only the variables they depend on are loaded (6-state quadrotor, SW = 2: the group (x, y)
// needs v_x, v_y and no sine / cosine at all; measured with the epilogue switched off, it was 13 % of the pass at 1024
// instances when every group evaluated all of f: tools/diag/x_traffic_probe.py).
// the node variables of instance `inst` at a lane's two output nodes (forward node_f and its mirror): what f is evaluated on
template <class Model>
EMI_DEV void ring_epilogue_load(const SymDefectArgs& a, double (&z)[2][Model::NV], int inst, int node_f) {
    constexpr int NS = Model::NS, NC = Model::NC;
    const int M = a.M, node_m = M - 1 - node_f;
    inst = inst < a.B ? inst : a.B - 1;                    // (rows past the batch: loaded from the last instance, never stored)
    const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
    const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int node = side == 0 ? node_f : node_m;
#pragma unroll
        for (int v = 0; v < NS; ++v) z[side][v] = Xb[(size_t)v * M + node];
#pragma unroll
        for (int v = 0; v < NC; ++v) z[side][NS + v] = Ub[(size_t)v * M + node];
    }
}
// zpre (SW > 1, PRE): the variables of the lane's FIRST instance, already requested by the caller (emi_ring2_body asks for them
// ahead of its K loop: the epilogue's first memory round trip is then over when the last MFMA retires); otherwise loaded here.
// (A compile-time choice and an array reference: with a pointer that may be null, or a run-time flag, the array went to scratch.)
template <class Model, int SW, int S0, bool PRE>
EMI_DEV void ring_epilogue_s0(const SymDefectArgs& a, const d4 (&acc_a)[SW], const d4 (&acc_b)[SW], int inst0, int i0,
                              int wid, int r16, int kq, const double (&zpre)[2][Model::NV]) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    const int M = a.M, B = a.B;
    const int col = wid * 16 + r16;
    const int node_f = i0 + col, node_m = M - 1 - node_f;
    const double t_f = a.node_t[node_f], t_m = a.node_t[node_m];
    // one instance (two nodes: forward and mirrored) at a time: with all eight (instance, node) pairs of a lane in
    // flight at once the kernel needs 184 registers instead of 90 (SW = 2) and loses more to occupancy than the single
    // memory round trip gains (0.2475 against 0.2394 ms per pass at 1024 instances)
    // (keeping this loop rolled -- a quarter of the code, 41 KB for the pass kernel otherwise -- changes nothing: 0.2314
    // against 0.2318 ms, tools/ab_build.sh)
    if constexpr (SW == 1) {
        // One state per workgroup is the form of SMALL batches (fewer than 128 sixteen-instance tiles), where a pass waits for
        // the MFMA role's latency chain and occupancy is no concern: all eight (instance, node) pairs of a lane in one memory
        // round trip instead of four (measured with the epilogue switched off: 4.2 of the role's 27 us at 128 instances).
        if (a.ablate & 4) return;
        double zz[4][2][NV];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int inst = inst0 + kq + 4 * i;
            inst = inst < B ? inst : B - 1;                // (rows past the batch: loaded from the last instance, not stored)
            const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
            const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int node = side == 0 ? node_f : node_m;
#pragma unroll
                for (int v = 0; v < NS; ++v) zz[i][side][v] = Xb[(size_t)v * M + node];
#pragma unroll
                for (int v = 0; v < NC; ++v) zz[i][side][NS + v] = Ub[(size_t)v * M + node];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = inst0 + kq + 4 * i;
            if (inst >= B) continue;
            double* __restrict__ Rb = a.RES + (size_t)inst * a.nres * M;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                double f[NS];
                Model::f(a.P, zz[i][side], side == 0 ? t_f : t_m, f);
                const double dx = side == 0 ? acc_a[0][i] + acc_b[0][i] : acc_b[0][i] - acc_a[0][i];
                Rb[(size_t)S0 * M + (side == 0 ? node_f : node_m)] = dx - a.h * f[S0];
            }
        }
        return;
    }
    // ... so the four instances go through a two-deep pipeline instead: the X / U values of instance i+1 are requested before f of
    // instance i is evaluated and stored (one exposed memory round trip per workgroup instead of four; the epilogue is 8 % of the
    // pass at 1024 instances, measured by switching it off: profiles/r03_notes.md section 8)
    if (a.ablate & 4) return;
    auto load_z = [&](double (&z)[2][NV], int i) {
        int inst = inst0 + kq + 4 * i;
        inst = inst < B ? inst : B - 1;                    // (rows past the batch: loaded from the last instance, never stored)
        const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
        const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int node = side == 0 ? node_f : node_m;
#pragma unroll
            for (int v = 0; v < NS; ++v) z[side][v] = Xb[(size_t)v * M + node];
#pragma unroll
            for (int v = 0; v < NC; ++v) z[side][NS + v] = Ub[(size_t)v * M + node];
        }
    };
    double zc[2][NV], zn[2][NV];
    if constexpr (PRE) {
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int v = 0; v < NV; ++v) zc[side][v] = zpre[side][v];
    } else {
        load_z(zc, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < 3) load_z(zn, i + 1);
        const int inst = inst0 + kq + 4 * i;
        if (inst < B) {
            double* __restrict__ Rb = a.RES + (size_t)inst * a.nres * M;
            double hf[2][SW];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                double f[NS];
                Model::f(a.P, zc[side], side == 0 ? t_f : t_m, f);
#pragma unroll
                for (int s = 0; s < SW; ++s) hf[side][s] = a.h * f[S0 + s];
            }
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int node = side == 0 ? node_f : node_m;
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const double dx = side == 0 ? acc_a[s][i] + acc_b[s][i] : acc_b[s][i] - acc_a[s][i];
                    Rb[(size_t)(S0 + s) * M + node] = dx - hf[side][s];
                }
            }
        }
        if (i < 3) {
#pragma unroll
            for (int side = 0; side < 2; ++side)
#pragma unroll
                for (int v = 0; v < NV; ++v) zc[side][v] = zn[side][v];
        }
    }
}
template <class Model, int SW, bool PRE, int SG = 0>
EMI_DEV void ring_epilogue_pre(const SymDefectArgs& a, const d4 (&acc_a)[SW], const d4 (&acc_b)[SW], int inst0, int i0,
                               int s0, int wid, int r16, int kq, const double (&zpre)[2][Model::NV]) {
    if constexpr (SG * SW < Model::NS) {
        if (s0 == SG * SW) ring_epilogue_s0<Model, SW, SG * SW, PRE>(a, acc_a, acc_b, inst0, i0, wid, r16, kq, zpre);
        else ring_epilogue_pre<Model, SW, PRE, SG + 1>(a, acc_a, acc_b, inst0, i0, s0, wid, r16, kq, zpre);
    }
}
template <class Model, int SW>
EMI_DEV void ring_epilogue(const SymDefectArgs& a, const d4 (&acc_a)[SW], const d4 (&acc_b)[SW], int inst0, int i0,
                           int s0, int wid, int r16, int kq) {
    const double none[2][Model::NV] = {};
    ring_epilogue_pre<Model, SW, false>(a, acc_a, acc_b, inst0, i0, s0, wid, r16, kq, none);
}

// a.ksplit > 1: the K range of a tile is cut into ksplit slices, one workgroup each (a shard of config 4 has 64
// tiles for 256 CUs); a slice leaves its partial sums in a.slab[tile][slice][2 SW][4][256 threads] and
// emi_symdefect_combine_kernel adds the slices IN SLICE ORDER (bitwise reproducible) and runs the epilogue.
// BK = 16 (round 4): K tiles twice as deep -- half as many barriers, counted waits and ring bookkeeping per flop, twice the bytes in
// flight per stage.  A pass of a small batch (the 128-instance shard of config 4) is the MFMA role's dependency chain of K tiles, and
// with one state per workgroup a tile holds only four MFMAs (256 matrix-pipe cycles) against several hundred cycles of per-tile
// work (profiles/r03_notes.md section 7: ~0.3 us per tile); the deep form halves the number of tiles.
// CT = 2 (round 4): two 64-column sub-tiles per workgroup (wave w owns half-indices 16 w .. 16 w + 15 of each): an X tile is then read
// by M / 256 column tiles instead of M / 128 -- the X operand tiles are the largest part of what the MFMA role fetches (X is read
// ntiles times per pass: 384 of ~1550 HBM bytes per node-eval once the inputs of a large batch no longer sit in the Infinity Cache,
// profiles/r03_notes.md section 2) -- and the A fragments (sums / differences of x) are shared by the MFMAs of both sub-tiles: 16
// MFMAs per wave and K tile against 8 fragment reads and 5 DMA instructions (CT = 1, SW = 2: 8 against 6 and 3).
// HS = 2 (round 4): the K range of a tile cut in two INSIDE the workgroup -- 512 threads, waves 0 - 3 take the first half of the K tiles,
// waves 4 - 7 the second, each half with its own operand ring; the halves meet once, at the end, where the second leaves its partial sums
// in LDS (in the ring it no longer needs) and the first adds them and runs the epilogue.  What K slices through the slab do for small
// batches (a shorter dependency chain, two MFMA waves per SIMD filling each other's gaps) without a second workgroup, global partial
// sums or a ticket.  The sums of the two halves are added once: not the bits of the unsplit form, the same values to rounding.
template <class Model, int SW, int NST = 3, int BK = 8, int CT = 1, int HS = 1>
EMI_DEV void emi_ring2_body(const SymDefectArgs& a, const int bid /* tile-and-slice id, XCD-local runs */) {
    constexpr int NS = Model::NS;
    constexpr int TI = FUSED_TI, TM = SW * TI, TN = 64 * CT, CH = BK / 2, NSG = NS / SW;
    static_assert(CT == 1 || CT == 2, "one or two column sub-tiles per wave");
    constexpr int KH = BK / 8;                           // 16-byte fragment reads per operand row and K tile (each: two k-steps)
    constexpr int RPI = 64 / CH;                         // rows one DMA wave instruction moves (1 KB)
    static_assert(BK == 8 || BK == 16, "K tiles of 8 or 16");
    constexpr int LOOK = NST - 1;                        // K tiles in flight ahead of the one being multiplied
    constexpr int ROWS = 2 * TM + 2 * TN;
    constexpr int ROWS_PAD = (ROWS + 63) / 64 * 64;      // a DMA wave instruction moves RPI rows: whole instructions for 4 waves
    constexpr int STAGE = ROWS_PAD * BK;                 // doubles per ring stage
    constexpr int L = ROWS_PAD * CH / 256;               // LDS-DMA instructions per wave and stage (1 KB each)
    static_assert(NS % SW == 0, "states split evenly over workgroups");

    extern __shared__ __attribute__((aligned(16))) double smem_all[];   // [HS][NST][STAGE]: XF, XM, De, Do (, padding)
    static_assert(HS == 1 || HS == 2, "the K range whole or in two halves");
    const int half = HS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    double* const smem = smem_all + (size_t)half * NST * STAGE;          // this half's ring

    const int M = a.M, Hh = M >> 1, B = a.B;
    const int ntiles = Hh / TN;
    const int KS = (HS == 1 && a.ksplit > 1) ? a.ksplit : 1;            // (slices through the slab: the undivided workgroup only)
    const int kslice = bid % KS;
    // Which tile.  bid / KS counts tiles in XCD-local runs (an XCD gets a contiguous range).  Plain order: the ntiles
    // column tiles of an X tile (same instance group, same states) are neighbours, so an XCD's L2 sees each X tile once --
    // but every XCD then walks ALL of De / Do (4 MB at 1024 nodes = the whole L2 of an XCD), and beside the node role's
    // store stream the panels are fetched again and again: 12 % of the pass at 1024 instances (tools/diag/x_traffic_probe.py:
    // 0.2408 ms against 0.2107 with all tiles reading the same 64 rows).  Partitioned order (a.cpart > 0): XCD x works on
    // column partition x % cpart only (its share of De / Do stays in L2) and on group partition x / cpart; X tiles are then
    // read by cpart XCDs instead of one, which costs little (they come from the Infinity Cache).
    const RingTile rt = ring_tile_of(bid / KS, ntiles, (B + TI - 1) / TI * NSG, a.cpart, a.cx, NSG);
    const int ntile = rt.ntile, grp = rt.grp;
    const int tile = grp * ntiles + ntile;              // slab / ticket index
    const int sg = grp % NSG, mtile = grp / NSG;
    const int inst0 = mtile * TI, i0 = ntile * TN, s0 = sg * SW;

    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;

    // DMA addressing: an instruction of wave wid moves rows 16 (wid + 4t) .. +15 of the stage, which are all forward x,
    // all mirrored x, all De or all Do -- so the 64-bit base and its advance per K tile are wave-uniform (scalar
    // registers, SALU) and a lane contributes a 32-bit byte offset that never changes (saddr form of global_load_lds:
    // no vector address arithmetic per tile; the 64-bit per-lane form cost two v_mul_lo_u32, a v_mad_u64_u32 and a
    // v_add3_u32 per instruction and tile).
    unsigned voff[L];                  // per lane: byte offset from the instruction's base
    unsigned long long gbase[L];       // wave-uniform: base address at K tile 0
    int gstep[L];                      // wave-uniform: bytes per K tile (+64 forward, -64 mirrored x, 0 padding)
#pragma unroll
    for (int t = 0; t < L; ++t) {
        const int row0 = (wid + 4 * t) * RPI;          // wave-uniform
        const int row = row0 + lane / CH, p = lane % CH;
        if (row0 < 2 * TM) {
            const bool mir = row0 >= TM;
            const int rr = row - (mir ? TM : 0);
            const int c = p ^ ring_swz_k<BK>(rr);
            int inst = inst0 + (rr & 15);
            inst = inst < B ? inst : B - 1;            // rows past the batch are never written out
            if (a.ablate & 8) inst &= 15;              // diagnostics: every tile reads the first 16 instances (X traffic ~ 0)
            voff[t] = (unsigned)((((size_t)inst * NS + s0 + (rr >> 4)) * M + 2 * c) * sizeof(double));
            gbase[t] = (unsigned long long)(a.X + (mir ? M - BK : 0));
            gstep[t] = mir ? -(int)(BK * sizeof(double)) : (int)(BK * sizeof(double));
        } else if (row0 < ROWS) {
            const bool od = row0 >= 2 * TM + TN;
            const int rr = row - 2 * TM - (od ? TN : 0);
            const int c = p ^ ring_swz_k<BK>(rr);
            voff[t] = (unsigned)(((size_t)((a.ablate & 64) ? rr : i0 + rr) * Hh + 2 * c) * sizeof(double));   // (64: diagnostics, all tiles read the first 64 rows)
            gbase[t] = (unsigned long long)(od ? a.Do : a.De);
            gstep[t] = (int)(BK * sizeof(double));
        } else {                                       // padding rows of the last DMA instruction: never read
            voff[t] = (unsigned)(2 * p * sizeof(double));
            gbase[t] = (unsigned long long)a.De;
            gstep[t] = 0;
        }
    }
    // The DMA is issued from inline assembly: an LDS-DMA instruction the COMPILER knows about makes it put
    // "s_waitcnt vmcnt(0)" in front of every later LDS read (it cannot tell which ring stage the DMA writes), i.e. right
    // after the DMA of tile kt+LOOK is issued the wave waited for that very tile -- no tile in flight at all, every K tile
    // paid a full DMA round trip (the ISA of rounds 1-2; profiles/r02_notes.md section 9).  The counted vmcnt waits of
    // the K loop are the only synchronisation with the DMA the ring needs.  m0 = LDS address of the 1 KB the wave writes.
    // (s_nop 0 between the SALU write of m0 and the DMA: gfx9-family parts need one wait state there, and the hazard
    // recogniser does not look inside inline assembly -- the builtin form gets the same s_nop from the compiler.)
    const unsigned lds0 = (unsigned)(unsigned long)(emi_lds_ptr_t)smem;          // LDS byte address of the ring
    // running state of the ring, all wave-uniform (SALU): next tile to request, where it goes, where the next fragment
    // reads come from -- an add per DMA instruction instead of a 64-bit multiply by the tile index and a modulo
    const int nkt = (Hh / BK) / KS / HS, kt0 = (kslice * HS + half) * nkt;       // this slice's (half's) K tiles: kt0 .. kt0 + nkt - 1
    unsigned long long gnext[L];
#pragma unroll
    for (int t = 0; t < L; ++t) gnext[t] = gbase[t] + (long long)gstep[t] * kt0;
    unsigned st_dma = 0, st_rd = 0;     // byte offsets of the stage the next DMA tile / the next fragment reads use
    int nd = 0;                         // tiles requested so far
    auto issue_one = [&](int t) {       // instruction t of tile nd; the last one moves the ring on
        const unsigned dst = lds0 + st_dma + (unsigned)(wid + 4 * t) * 1024u;   // wave-uniform
        asm volatile("s_mov_b32 m0, %2\n\t" EMI_M0_NOP "global_load_lds_dwordx4 %0, %1" ::"v"(voff[t]), "s"(gnext[t]), "s"(dst) : "memory", "m0");
        gnext[t] += (long long)gstep[t];
        if (t == L - 1) {
            st_dma = st_dma == (unsigned)((NST - 1) * STAGE * 8) ? 0u : st_dma + (unsigned)(STAGE * 8);
            ++nd;
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int t = 0; t < L; ++t) issue_one(t);
    };

    d4 acc_a[SW][CT], acc_b[SW][CT];
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            acc_a[s][c] = d4{0.0, 0.0, 0.0, 0.0};
            acc_b[s][c] = d4{0.0, 0.0, 0.0, 0.0};
        }

    // Build-time option (EMI_EPILOGUE_PREFETCH, off): the epilogue's first memory round trip started HERE -- the node variables of the
    // lane's first instance at its two output nodes requested before the ring's first DMA instruction (older than every DMA instruction,
    // so the counted vmcnt waits of the K loop mean what they meant), held in 2 NV registers through the loop.  It takes one exposed round
    // trip off a workgroup's dependency chain on paper; measured on one box against the same build without it: 128 instances 0.0296
    // against 0.0293 ms, 1024: 0.2121 - 0.2135 against 0.2110 - 0.2112 -- the epilogue's loads are not what a chain waits for.
    constexpr bool EPI_PRE = EMI_EPILOGUE_PREFETCH && SW >= 2 && SW <= 3 && HS == 1 && CT == 1;   // (the wide and the all-states forms have no registers to spare)
    double zpre[2][Model::NV] = {};
    if constexpr (EPI_PRE) ring_epilogue_load<Model>(a, zpre, inst0 + kq, i0 + wid * 16 + r16);

#pragma unroll
    for (int t = 0; t < LOOK; ++t)
        if (t < nkt) issue();
    // fragment addresses (doubles within a stage): B rows of this wave, A rows of every state
    // (half h of a deep tile: chunk kq + 4 h forward, k = 8 h + 2 kq, + 1; its mirror sits at position BK - 1 - k of the mirrored
    // tile, i.e. in chunk CH - 1 - kq - 4 h)
    int off_b[CT][KH], off_f[SW][KH], off_m[SW][KH];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int rb = c * 64 + wid * 16 + r16;        // row of De / Do within the tile: sub-tile c, this wave's sixteen half-indices
#pragma unroll
        for (int h = 0; h < KH; ++h) off_b[c][h] = rb * BK + (((kq + 4 * h) ^ ring_swz_k<BK>(rb)) << 1);
    }
#pragma unroll
    for (int s = 0; s < SW; ++s) {
        const int r = s * 16 + r16;
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            off_f[s][h] = r * BK + (((kq + 4 * h) ^ ring_swz_k<BK>(r)) << 1);                  // chunk kq + 4h: x_(8h + 2kq), x_(8h + 2kq + 1)
            off_m[s][h] = (TM + r) * BK + (((CH - 1 - kq - 4 * h) ^ ring_swz_k<BK>(r)) << 1);  // their mirrors
        }
    }
    // fragments of one K tile: De / Do rows of this wave (B operands) and, per state, the A operands of the even / odd products:
    // sp = forward + mirrored x, dm = forward - mirrored x for the tile's two k-steps.  They are formed when the x fragments have
    // arrived (in the gaps of the PREVIOUS tile's MFMAs), not in front of the MFMA that consumes them: a v_add_f64 directly ahead of
    // its MFMA holds the matrix pipe for the result (the same finding as the shift subtractions of the fp32 kernel, r03_notes.md).
    struct Frag {
        double2 be[CT][KH], bo[CT][KH], sp[SW][KH], dm[SW][KH];
    };
    double2 txf[SW][KH], txm[SW][KH];                   // x fragments between their read and their sums
    constexpr int NRH = 2 * CT + 2 * SW;                // fragment reads per half of a tile
    constexpr int NR = NRH * KH, NM = 4 * SW * CT * KH; // fragment reads / MFMAs per wave and K tile
    // read r of a tile: half h = r / NRH, then (De, Do) of every sub-tile and (forward, mirrored) x of every state
    auto read_one = [&](Frag& f, const double* S, int r) {
        const int h = r / NRH, q = r % NRH;
        if (q < 2 * CT) {
            if (q & 1) f.bo[q >> 1][h] = *reinterpret_cast<const double2*>(S + (2 * TM + TN) * BK + off_b[q >> 1][h]);
            else f.be[q >> 1][h] = *reinterpret_cast<const double2*>(S + 2 * TM * BK + off_b[q >> 1][h]);
        } else if (q & 1) txm[(q - 2 * CT) >> 1][h] = *reinterpret_cast<const double2*>(S + off_m[(q - 2 * CT) >> 1][h]);
        else txf[(q - 2 * CT) >> 1][h] = *reinterpret_cast<const double2*>(S + off_f[(q - 2 * CT) >> 1][h]);
    };
    // k = 8h + 2kq (+1): forward x_k in xf.x (.y), its mirror x_(N-k) at position BK-1-k of the mirrored tile: xm.y (.x)
    auto sums_one = [&](Frag& f, int s) {
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            f.sp[s][h] = double2{txf[s][h].x + txm[s][h].y, txf[s][h].y + txm[s][h].x};
            f.dm[s][h] = double2{txf[s][h].x - txm[s][h].y, txf[s][h].y - txm[s][h].x};
        }
    };
    auto rd_stage = [&]() -> const double* {            // stage of the next fragment set; moves on
        const double* S = smem + (st_rd >> 3);
        st_rd = st_rd == (unsigned)((NST - 1) * STAGE * 8) ? 0u : st_rd + (unsigned)(STAGE * 8);
        return S;
    };
    auto read_frag = [&](Frag& f) {
        const double* S = rd_stage();
#pragma unroll
        for (int r = 0; r < NR; ++r) read_one(f, S, r);
#pragma unroll
        for (int s = 0; s < SW; ++s) sums_one(f, s);
    };
    // MFMA i of a tile: the 2 SW MFMAs of the tile's first k-step, then those of the second, ... (2 KH k-steps; k-step j uses half
    // j / 2 of the fragments, element j % 2)
    // (within a k-step: state, then sub-tile, then even / odd product)
    auto mfma_one = [&](const Frag& f, int i) {
        const int per = 2 * SW * CT, q = i % per, j = i / per, h = j >> 1;
        const int s = q / (2 * CT), c = (q >> 1) % CT;
        const bool second = j & 1, odd = q & 1;
        if (!odd) acc_a[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(second ? f.sp[s][h].y : f.sp[s][h].x, second ? f.be[c][h].y : f.be[c][h].x, acc_a[s][c], 0, 0, 0);
        else      acc_b[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(second ? f.dm[s][h].y : f.dm[s][h].x, second ? f.bo[c][h].y : f.bo[c][h].x, acc_b[s][c], 0, 0, 0);
    };
    auto multiply = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < NM; ++i) mfma_one(f, i);
    };
    if constexpr (SW <= 3) {
        // Software-pipelined K loop.  A wave issues in order and an MFMA waits for the matrix pipe (64 cycles per
        // v_mfma_f64_16x16x4_f64), so whatever stands between two tiles' MFMAs in program order runs with the pipe idle:
        // measured on a 128-instance shard (one wave per SIMD, nothing else to hide behind) skeleton 11 us + DMA issue 14
        // + fragment reads and MFMAs 19 + epilogue 6, all serial (tools/small_batch_anatomy.py).  Here the DMA of tile
        // kt+1+LOOK and the fragment reads of tile kt+1 are dealt into the gaps BETWEEN the MFMAs of tile kt, one DMA
        // instruction or a few reads per gap; sched_barrier pins that order (the scheduler sinks the reads behind the
        // MFMAs otherwise).  Stage (kt+1+LOOK) % NST is the stage of tile kt (NST = LOOK + 1), whose fragment reads every
        // wave has completed before it arrives at the barrier (lgkmcnt(0) ahead of it).
        constexpr int RPG = (NR + (NM - L) - 1) / (NM - L);     // fragment reads per gap
        constexpr int GR = (NR + RPG - 1) / RPG;                // gaps that take reads: the first GR; then L gaps with a DMA each
        static_assert(NM > L && GR + L <= NM, "a gap for every DMA instruction after the reads");
        auto step = [&](const Frag& cur, Frag& nxt, int kt) {   // kt + 1 < nkt: MFMAs of tile kt, tile kt+1 made ready
            if (kt + LOOK < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (LOOK - 1)) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), as an instruction the compiler's own wait counting sees
            asm volatile("s_barrier" ::: "memory");     // tile kt+1 has landed for every wave; the stage of tile kt is free
            const bool more = nd < nkt;                 // (nd == kt + 1 + LOOK while there are tiles left)
            const double* S = rd_stage();
            // reads first: the next tile's first MFMA needs them, the DMA has LOOK tiles of slack
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mfma_one(cur, i);
                if (i < GR) {
#pragma unroll
                    for (int r = i * RPG; r < (i + 1) * RPG && r < NR; ++r) read_one(nxt, S, r);
                } else if (i < GR + L) {
                    if (more) issue_one(i - GR);
                }
                if (i >= NM - SW) sums_one(nxt, i - (NM - SW));     // the A operands of tile kt+1: its x fragments were read >= 2 gaps ago
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        Frag f0, f1;
        if (nkt > LOOK - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (LOOK - 1)) : "memory");
        else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (LOOK < nkt) issue();
        // scalar loads of arguments the epilogue needs are still pending in the compiler's model here; with them pending it
        // turns every fragment wait of the loop into lgkmcnt(0), i.e. waits for the reads it has just issued as well
        __builtin_amdgcn_s_waitcnt(0xC07F);
        read_frag(f0);
        int kt = 0;                                     // nkt is even (M % 128 == 0, ksplit a power of two <= 8)
        for (; kt + 2 < nkt; kt += 2) {
            step(f0, f1, kt);
            step(f1, f0, kt + 1);
        }
        step(f0, f1, kt);
        multiply(f1);
    } else {
        // all states in one workgroup: two fragment sets (2 x 56 registers) beside 96 accumulator registers would spill;
        // 24 MFMAs per wave and tile and two workgroups per CU hide the reads here
        for (int kt = 0; kt < nkt; ++kt) {
            // tile kt has landed once all but the DMA instructions of the (up to LOOK - 1) younger tiles are done
            if (kt + LOOK - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (LOOK - 1)) : "memory");
            else                     asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail: at most LOOK - 1 tiles early
            asm volatile("s_barrier" ::: "memory");     // every wave's part landed; stage (kt-1) % NST is free
            if (kt + LOOK < nkt) issue();
            Frag f;
            read_frag(f);
            multiply(f);
        }
    }

    if (CT == 1 && KS > 1) {       // (K slices: single sub-tile form only) partial sums of this K slice -> slab (coalesced: one 2 KB row per register)
        double* sl = a.slab + ((size_t)tile * KS + kslice) * (2 * SW * 4) * 256 + tid;
        if (a.tile_ticket == nullptr) {                 // emi_symdefect_combine_kernel adds the slices
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sl[(size_t)((2 * s) * 4 + i) * 256] = acc_a[s][0][i];
                    sl[(size_t)((2 * s + 1) * 4 + i) * 256] = acc_b[s][0][i];
                }
            return;
        }
        // In-kernel combine: the workgroup that draws the last ticket of its tile adds the slices IN SLICE ORDER (its own
        // from the slab as well: bitwise the result of the combine kernel) and runs the epilogue.  Partial sums travel
        // write-through / cache-bypassing (agent-scope atomics: store, drain, then the ticket -- as the COST partials of
        // the node kernel do); the ticket word resets itself.
#pragma unroll
        for (int s = 0; s < SW; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __hip_atomic_store(sl + (size_t)((2 * s) * 4 + i) * 256, acc_a[s][0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sl + (size_t)((2 * s + 1) * 4 + i) * 256, acc_b[s][0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                // every thread's partial sums are out (and the ring is no longer read)
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.tile_ticket + tile, 1u, EMI_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)KS - 1u) return;
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            acc_a[s][0] = d4{0.0, 0.0, 0.0, 0.0};
            acc_b[s][0] = d4{0.0, 0.0, 0.0, 0.0};
        }
        for (int k = 0; k < KS; ++k) {                  // fixed order
            const double* sk = a.slab + ((size_t)tile * KS + k) * (2 * SW * 4) * 256 + tid;
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc_a[s][0][i] += __hip_atomic_load(sk + (size_t)((2 * s) * 4 + i) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc_b[s][0][i] += __hip_atomic_load(sk + (size_t)((2 * s + 1) * 4 + i) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        if (tid == 0) __hip_atomic_store(a.tile_ticket + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_setprio(0);      // the epilogue gives the instruction arbiter back to the waves still in their K loops (0.5 - 1 % of the pass)
    if constexpr (HS == 2) {
        // the halves meet: the second half's partial sums through LDS (every wave is past its last fragment read: barrier), [register][thread]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        double* cb = smem_all + tid;
        if (half == 1) {
#pragma unroll
            for (int s2 = 0; s2 < SW; ++s2)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cb[(size_t)(((2 * s2) * CT + c) * 4 + i) * 256] = acc_a[s2][c][i];
                        cb[(size_t)(((2 * s2 + 1) * CT + c) * 4 + i) * 256] = acc_b[s2][c][i];
                    }
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int s2 = 0; s2 < SW; ++s2)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc_a[s2][c][i] += cb[(size_t)(((2 * s2) * CT + c) * 4 + i) * 256];
                    acc_b[s2][c][i] += cb[(size_t)(((2 * s2 + 1) * CT + c) * 4 + i) * 256];
                }
    }
    // one sub-tile after the other: its accumulators, its 64 output half-indices
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        d4 ea[SW], eb[SW];
#pragma unroll
        for (int s2 = 0; s2 < SW; ++s2) { ea[s2] = acc_a[s2][c]; eb[s2] = acc_b[s2][c]; }
        if (c == 0) ring_epilogue_pre<Model, SW, EPI_PRE>(a, ea, eb, inst0, i0, s0, wid, r16, kq, zpre);
        else ring_epilogue<Model, SW>(a, ea, eb, inst0, i0 + 64 * c, s0, wid, r16, kq);
    }
}

template <class Model, int SW>
__global__ __launch_bounds__(256, 2) void emi_symdefect_ring2_f64_kernel(SymDefectArgs a) {
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: an XCD gets a contiguous run of tiles
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    emi_ring2_body<Model, SW>(a, bid);
}

// second half of a split-K launch: one workgroup per tile, thread layout of the ring kernel
template <class Model, int SW>
__global__ __launch_bounds__(256) void emi_symdefect_combine_kernel(SymDefectArgs a) {
    constexpr int NS = Model::NS, TI = FUSED_TI, TN = 64, NSG = NS / SW;
    const int ntiles = (a.M >> 1) / TN, KS = a.ksplit;
    const int tile = blockIdx.x;
    const int ntile = tile % ntiles, grp = tile / ntiles;
    const int sg = grp % NSG, mtile = grp / NSG;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    d4 acc_a[SW], acc_b[SW];
#pragma unroll
    for (int s = 0; s < SW; ++s) {
        acc_a[s] = d4{0.0, 0.0, 0.0, 0.0};
        acc_b[s] = d4{0.0, 0.0, 0.0, 0.0};
    }
    for (int k = 0; k < KS; ++k) {                      // fixed order
        const double* sl = a.slab + ((size_t)tile * KS + k) * (2 * SW * 4) * 256 + tid;
#pragma unroll
        for (int s = 0; s < SW; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc_a[s][i] += sl[(size_t)((2 * s) * 4 + i) * 256];
                acc_b[s][i] += sl[(size_t)((2 * s + 1) * 4 + i) * 256];
            }
    }
    ring_epilogue<Model, SW>(a, acc_a, acc_b, mtile * TI, ntile * TN, sg * SW, wid, lane & 15, lane >> 4);
}

// ---------------------------------------------------------------------------------------------
// The whole evaluation pass in ONE launch: some workgroups take the MFMA defect role (emi_ring2_body), the others
// the streaming node role (emi_nodes_body, which also finishes COST through its ticket).  Two launches on two
// streams need a fork and a join through events at every pass: measured 21 us of idle chip between consecutive
// 250 us passes at 1024 instances, 22 us inside a 75 us pass at 128 (profiles/r02_notes.md); one launch has none.
// (The single-launch form of round 1 lost because its MFMA role needed 219 registers, which capped the streaming
// waves at two per SIMD; the state-split ring role needs 78, less than the node role itself.)
// Roles are dealt per XCD (blocks b, b+8, ... share one): block j of an XCD is an MFMA block when
// floor((j+1) nm / t) > floor(j nm / t), nm of the XCD's t blocks being MFMA blocks -- evenly interleaved, so both
// roles are resident on every CU throughout, and the MFMA blocks of an XCD are a contiguous run of tiles (they
// share X and De/Do panels in that XCD's L2).  Counts that are no multiple of 8 are rounded up: the surplus workgroups return at once.
// ---------------------------------------------------------------------------------------------

#define EMI_STR2(x) #x
#define EMI_STR(x) EMI_STR2(x)
#ifndef EMI_PASS_WAVES_PER_EU
#define EMI_PASS_WAVES_PER_EU 0         // build-time experiment switch (tools/ab_build.sh): register cap of the pass kernel as waves per SIMD
#endif
#if EMI_PASS_WAVES_PER_EU > 0
#define EMI_PASS_OCC __attribute__((amdgpu_waves_per_eu(EMI_PASS_WAVES_PER_EU, EMI_PASS_WAVES_PER_EU)))
#elif EMI_PASS_WAVES_PER_EU < 0
#define EMI_PASS_OCC                    // (A/B: whatever the compiler arrives at)
#else
// Two waves per SIMD, stated: the ring's LDS admits two workgroups per CU and both roles want to be resident, but the register
// count of this kernel swings with details of the source (the 128-column form came out at 224 or at 280 registers for the same
// arithmetic) -- beyond 256 only ONE workgroup fits a CU and the roles no longer overlap.
// (The all-states form, SW > 3, holds 96 accumulator registers and has always run one workgroup per CU.)
#define EMI_PASS_OCC __attribute__((amdgpu_waves_per_eu(SW > 3 ? 1 : 2, 2)))
#endif
template <class Model, int SW, int VEC, int ST, int NST = 3, int BK = 8, int CT = 1, int HS = 1>
__global__ __launch_bounds__(256 * HS) EMI_PASS_OCC void emi_pass_f64_kernel(PassArgs a) {
#ifdef EMI_ENTRY_PAD_NOPS      // build-time experiment (tools/ab_build.sh): shift the whole instruction stream by 4-byte steps
    asm volatile(".rept " EMI_STR(EMI_ENTRY_PAD_NOPS) "\n\ts_nop 0\n\t.endr" ::: "memory");
#endif
    const int g = blockIdx.x, xcd = g & 7, j = g >> 3;
    const PassRole role = pass_role_of(j, a.nm8, a.nn8, a.s.mfma_first);
    if (a.s.ablate & (role.mfma ? 16 : 32)) return;    // diagnostics: one of the two roles does nothing
    if (role.mfma) {
        const int tid = xcd * a.nm8 + role.index;
        if (tid >= a.nm) return;                        // (the role's count rounded up to a multiple of 8: whole workgroup, before any barrier)
        // the MFMA role is the latency chain of a small pass (64 dependent K tiles); the streaming role beside it on the
        // same SIMDs waits on memory most of the time: instruction arbitration goes to the MFMA waves first
        __builtin_amdgcn_s_setprio(3);
        emi_ring2_body<Model, SW, NST, BK, CT, HS>(a.s, tid);
    } else {
        // a.nn counts node chunks; a workgroup of HS x 256 threads takes HS neighbouring ones (a half that has none leaves:
        // barriers only count the waves still running)
        const int nid = (xcd * a.nn8 + role.index) * HS + (HS > 1 ? (int)(threadIdx.x >> 8) : 0);
        if (nid >= a.nn) return;
        emi_nodes_body<double, Model, VEC, true, false, ST>(a.n, nid % a.nbx, nid / a.nbx, a.nbx);
    }
}

}  // namespace emi
