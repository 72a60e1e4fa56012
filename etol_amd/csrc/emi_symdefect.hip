// emi_symdefect.hip -- K4 in its fast form: the defect rows  D.X - h f  in one gfx950 kernel (fp64).
//
//   * D.X on v_mfma_f64_16x16x4_f64, with the flops HALVED by the centro-antisymmetry of the LGL
//     differentiation matrix, D[N-i][N-j] = -D[i][j]:
//          e_j = x_j + x_{N-j},  o_j = x_j - x_{N-j}                        (j < M/2)
//          a_i = sum_j De[i][j] e_j,   b_i = sum_j Do[i][j] o_j             (i < M/2)
//          (D x)_i = a_i + b_i,        (D x)_{N-i} = -a_i + b_i
//     De = (D[i][j] + D[i][N-j])/2, Do = (D[i][j] - D[i][N-j])/2 are built once on the host.
//   * Epilogue: accumulator rows are ordered state-major inside a workgroup (row = state*16 +
//     instance), so one lane holds all NS components of D.X for its (instance, node) pairs: it
//     evaluates f there and writes  defect = D.X - h f  directly.  The kernel therefore depends
//     only on X and U and runs CONCURRENTLY with the streaming node kernel (emi_kernels.hip) on a
//     second HIP stream: the matrix pipe works under the ~1 KB/node of HBM stores.
//     (Two single-kernel fusions were built and measured first -- node work inside this kernel's
//     K loop, and MFMA / streaming workgroup roles in one launch.  Both lost to the two-stream
//     form: all workgroups follow one schedule, so the chip alternates between all-MFMA and
//     all-store phases; vmcnt retires in order, so an operand wait also waits for every older
//     store; and a kernel that holds 96 accumulator registers caps the streaming waves at 2 per
//     SIMD, which is too few to keep HBM busy.  profiles/r01_notes.md has the numbers.)
//
// Workgroup = 256 threads = 4 waves; tile = 16 instances x 64*CT half-indices i (nodes i and N-i);
// each wave owns NS x CT x 2 accumulator tiles (96*CT registers for 6 states), one workgroup per CU.  LDS: two buffers (register prefetch of the next K tile) of E,O [NS*16][18], De,Do
// [64*CT][18] doubles; rows padded to 18 doubles: 16-byte aligned for ds_write_b128, conflict-free for the
// ds_read_b64 fragment reads.  blockIdx -> tile map is XCD-aware: workgroups that share a De/Do
// panel share blockIdx % 8, i.e. one XCD's L2.
//
// Requires M % (128*CT) == 0 and an exactly centro-antisymmetric D (emi_lgl guarantees it; emi_set_mesh
// checks); every other shape takes the general defect kernel of emi_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "emi_kernels.hpp"
#include "emi_models.hpp"
#include "emi_symdefect_kernels.hpp"

namespace emi {

template <class Model>
static hipError_t launch_symdefect_ring_model(const SymDefectArgs& a, hipStream_t s, bool set_attr) {
    constexpr int NS = Model::NS;
    const int mtiles = (a.B + FUSED_TI - 1) / FUSED_TI, ntiles = (a.M / 2) / 64;
    const size_t lds = (size_t)3 * (2 * NS * FUSED_TI + 2 * 64) * 16 * sizeof(double);
    if (set_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)emi_symdefect_ring_f64_kernel<Model>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(mtiles * ntiles), block(256);
    hipLaunchKernelGGL((emi_symdefect_ring_f64_kernel<Model>), grid, block, lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
template <class Model, int CT>
static hipError_t launch_symdefect_model(const SymDefectArgs& a, hipStream_t s, bool set_attr) {
    constexpr int NS = Model::NS, TN = 64 * CT;
    const int mtiles = (a.B + FUSED_TI - 1) / FUSED_TI, ntiles = (a.M / 2) / TN;
    // two LDS buffers of E, O, De, Do tiles: 92 KB (CT=1) / 129 KB (CT=2) for the 6-state model,
    // i.e. one workgroup per CU, which leaves the streaming node kernel's waves room beside it
    const size_t lds = (size_t)2 * (2 * NS * FUSED_TI + 2 * TN) * (FUSED_BK + 2) * sizeof(double);
    if (set_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)emi_symdefect_f64_kernel<Model, CT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(mtiles * ntiles), block(256);
    hipLaunchKernelGGL((emi_symdefect_f64_kernel<Model, CT>), grid, block, lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// second ring form: SW states per workgroup, 8-deep stages, 16-byte fragment reads, two (or more) workgroups per CU,
// optionally split-K (partial sums through a slab, combined in slice order by a second launch)
template <class Model, int SW>
static hipError_t launch_ring2_model(const SymDefectArgs& a, hipStream_t s) {
    constexpr int NS = Model::NS;
    const int mtiles = (a.B + FUSED_TI - 1) / FUSED_TI, ntiles = (a.M / 2) / 64;
    const size_t lds = (size_t)3 * ((2 * SW * FUSED_TI + 2 * 64 + 63) / 64 * 64) * 8 * sizeof(double);
    static bool attr_done = false;          // per instantiation; a repeated call is harmless
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)emi_symdefect_ring2_f64_kernel<Model, SW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = mtiles * ntiles * (NS / SW), ks = a.ksplit > 1 ? a.ksplit : 1;
    hipLaunchKernelGGL((emi_symdefect_ring2_f64_kernel<Model, SW>), dim3(tiles * ks), dim3(256), lds, s, a);
    if (ks > 1 && a.tile_ticket == nullptr) hipLaunchKernelGGL((emi_symdefect_combine_kernel<Model, SW>), dim3(tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

// The variant for a batch, by what was measured on MI355X with the node kernel running beside it
// (profiles/r02_pass_variants.json; 6-state model, 1024 nodes; "tiles" = 16-instance x 64-half-index tiles):
//   >= 448 tiles (B = 1024: 512): SW = 2 -- 78 registers and 36 KB of LDS per workgroup leave the streaming
//      kernel's waves the room they need on every CU (pass 0.24-0.25 ms against 0.26-0.27 with 60 KB workgroup pairs);
//   320 .. 447 tiles (B = 768): SW = NS, all workgroups resident at once, two per CU (0.172 ms against 0.198);
//   192 .. 319 tiles (B = 512): the one-workgroup-per-CU ring kernel, whose grid is then a single round;
//   fewer (the shard of config 4: 128 instances = 64 tiles for 256 CUs): SW = 1 (<= 96 tiles) or 2, i.e. more workgroups
//      than CUs (0.055 ms at 128 instances; SW = NS with 4 K slices, the choice before the K loop was software-pipelined: 0.089).
// ct 5 / 6 / 7 / 8 force SW = NS / 2 / 1 / 3; ksplit_opt > 0 forces the slice count.
SymPlan plan_symdefect(int ns, int B, int M, int ct, int ksplit_opt, int cpart_opt, int gblk_opt, int cx_opt, int bk_opt, int ct_cols) {
    SymPlan p;
    p.bk = bk_opt == 16 ? 16 : 8;
    const int tiles = ((B + FUSED_TI - 1) / FUSED_TI) * ((M / 2) / 64);    // in 64-column tiles: what the thresholds below are written in
    const int nkt = (M / 2) / p.bk;
    if (ct == 0 || ct == 4) {
        if (tiles >= 448) p.sw = (ns % 2 == 0 && ns > 2) ? 2 : ns;
        else if (tiles >= 320) p.sw = ns;
        else if (tiles >= 192) { p.ring1 = true; return p; }
        else p.sw = (tiles <= 96 || ns % 2 != 0 || ns <= 2) ? 1 : 2;
    } else if (ct >= 5 && ct <= 8) {
        p.sw = ct == 5 ? ns : (ct == 6 ? 2 : (ct == 8 ? 3 : 1));
        if (ns % p.sw != 0) p.sw = 1;
    } else {
        p.ring1 = true;
        return p;
    }
    if (ksplit_opt > 0) p.ks = ksplit_opt;
    while (p.ks > 1 && (nkt % p.ks != 0 || nkt / p.ks < 2 || (nkt / p.ks) % 2 != 0)) p.ks >>= 1;
    p.ct = (ct_cols == 2 && p.sw == 2 && p.ks == 1 && M % 256 == 0) ? 2 : 1;      // column sub-tiles per workgroup (one-launch pass, SW = 2, unsplit)
    const int TNp = 64 * p.ct;
    p.tiles = (tiles / p.ct) * (ns / p.sw);
    {   // tile order: the share of De / Do an XCD works on should stay in its 4 MB L2 beside everything else (<= 2 MB)
        const int ntiles = (M / 2) / TNp, ngrp = ((B + FUSED_TI - 1) / FUSED_TI) * (ns / p.sw);
        const int target_cols = std::max(1, 4096 / M / p.ct);   // column tiles whose two panels (512 M bytes each per 64 columns) make 2 MB
        auto valid = [&](int cp) { return cp >= 1 && cp <= 8 && ntiles % cp == 0 && ngrp % (8 / cp) == 0 && (ngrp * ntiles) % 8 == 0; };
        int cp = 0;
        if (cpart_opt > 0) cp = valid(cpart_opt) ? cpart_opt : 0;
        else if (cpart_opt == 0 && tiles >= 192)    // (small batches: an XCD's few tiles do not walk the panels often enough to matter)
            for (int c = 1; c <= 8 && !cp; c *= 2)
                if (valid(c) && (ntiles / c <= target_cols || c == 8 || !valid(2 * c))) cp = c;
        const int mtiles = (B + FUSED_TI - 1) / FUSED_TI;
        if (gblk_opt > 0 && mtiles % (8 * gblk_opt) == 0 && (ngrp * ntiles) % 8 == 0) {
            // grouped order ("sym_gblk"): super-blocks of gblk instance groups per XCD, column blocks of cx tiles
            int cxg = cx_opt > 0 ? cx_opt : 2;
            while (cxg > 1 && ntiles % cxg != 0) --cxg;
            p.cpart = -gblk_opt;
            p.cx = cxg;
        } else if (cp > 0) {
            p.cpart = cp;
            p.cx = 1;                                            // largest divisor of the partition's column count within the target
            for (int d = 1; d <= std::min(ntiles / cp, target_cols); ++d)
                if ((ntiles / cp) % d == 0) p.cx = d;
            if (cp == 1 && p.cx == ntiles) p.cpart = p.cx = 0;   // that is the plain order
        }
    }
    if (p.ks > 1) p.slab_bytes = (size_t)p.tiles * p.ks * (2 * p.sw * 4) * 256 * sizeof(double);
    return p;
}

template <class Model>
static hipError_t launch_ring2_planned(const SymDefectArgs& a, hipStream_t s, const SymPlan& p) {
    constexpr int NS = Model::NS;
    if (p.sw == NS) return launch_ring2_model<Model, NS>(a, s);
    if constexpr (NS > 2 && NS % 2 == 0) {
        if (p.sw == 2) return launch_ring2_model<Model, 2>(a, s);
    }
    if constexpr (NS > 3 && NS % 3 == 0) {
        if (p.sw == 3) return launch_ring2_model<Model, 3>(a, s);
    }
    return launch_ring2_model<Model, 1>(a, s);
}

// ---------------------------------------------------------------------------------------------
// the pass as one launch (emi_pass_f64_kernel): MFMA-role and node-role workgroups interleaved per XCD
template <class Model, int SW, int NST, int BK = 8, int CT = 1, int HS = 1>
static hipError_t launch_pass_model(const SymDefectArgs& sa, const NodeArgs<double>& na, hipStream_t s) {
    constexpr int NS = Model::NS;
    PassArgs a;
    a.s = sa;
    a.n = na;
    const int mtiles = (sa.B + FUSED_TI - 1) / FUSED_TI, ntiles = (sa.M / 2) / (64 * CT);
    const int nm = mtiles * ntiles * (NS / SW) * (sa.ksplit > 1 ? sa.ksplit : 1);
    a.nbx = (na.M + 2 * EMI_NODE_THREADS - 1) / (2 * EMI_NODE_THREADS);
    const int nn = a.nbx * na.B;
    a.nm = nm;
    a.nn = nn;
    a.nm8 = (nm + 7) / 8;
    a.nn8 = ((nn + HS - 1) / HS + 7) / 8;          // node-role WORKGROUPS per XCD: one takes HS chunks
    const size_t lds = (size_t)HS * NST * ((2 * SW * FUSED_TI + 2 * 64 * CT + 63) / 64 * 64) * BK * sizeof(double);
    static bool attr_done[4] = {false, false, false, false};
    const int st = (na.store_mode >= 0 && na.store_mode <= 3) ? na.store_mode : 0;   // result stores: plain / sc1 (write-through) / non-temporal / nt sc1
    auto kern = st == 2 ? emi_pass_f64_kernel<Model, SW, 2, 2, NST, BK, CT, HS>
              : (st == 1 ? emi_pass_f64_kernel<Model, SW, 2, 1, NST, BK, CT, HS>
                         : (st == 3 ? emi_pass_f64_kernel<Model, SW, 2, 3, NST, BK, CT, HS> : emi_pass_f64_kernel<Model, SW, 2, 0, NST, BK, CT, HS>));
    if (!attr_done[st]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[st] = true;
    }
    hipLaunchKernelGGL(kern, dim3(8 * (a.nm8 + a.nn8)), dim3(256 * HS), lds, s, a);
    return hipGetLastError();
}

template <class Model>
static hipError_t launch_pass_planned(const SymDefectArgs& sa, const NodeArgs<double>& na, hipStream_t s, const SymPlan& p) {
    constexpr int NS = Model::NS;
    if (p.sw == NS) return launch_pass_model<Model, NS, 3>(sa, na, s);
    if constexpr (NS > 2 && NS % 2 == 0) {
        if (p.sw == 2 && p.hs == 2 && p.ct == 1) return p.bk == 16 ? launch_pass_model<Model, 2, 3, 16, 1, 2>(sa, na, s) : launch_pass_model<Model, 2, 3, 8, 1, 2>(sa, na, s);
        if (p.sw == 2 && p.ct == 2) return p.bk == 16 ? launch_pass_model<Model, 2, 3, 16, 2>(sa, na, s) : launch_pass_model<Model, 2, 3, 8, 2>(sa, na, s);
        if (p.sw == 2 && p.bk == 16) return launch_pass_model<Model, 2, 3, 16>(sa, na, s);
        if (p.sw == 2) return p.nst > 3 ? launch_pass_model<Model, 2, 4>(sa, na, s) : launch_pass_model<Model, 2, 3>(sa, na, s);
    }
    if constexpr (NS > 3 && NS % 3 == 0) {
        if (p.sw == 3) return launch_pass_model<Model, 3, 3>(sa, na, s);
    }
    if (p.hs == 2) return p.bk == 16 ? launch_pass_model<Model, 1, 3, 16, 1, 2>(sa, na, s) : launch_pass_model<Model, 1, 3, 8, 1, 2>(sa, na, s);
    if (p.bk == 16) return launch_pass_model<Model, 1, 3, 16>(sa, na, s);
    return p.nst > 3 ? launch_pass_model<Model, 1, 4>(sa, na, s) : launch_pass_model<Model, 1, 3>(sa, na, s);
}

// true if the pass can go out as one launch with this plan (state-split ring role, K slices combined in-kernel by
// ticket, whole XCD shares)
bool pass_supported(int model, int ns, int B, int M, const SymPlan& p) {
    if (p.ring1 || p.sw < 1 || (model != EMI_MODEL_POINTMASS2D && model != EMI_MODEL_QUADROTOR2D)) return false;
    if (M % 128 != 0 || ns % p.sw != 0) return false;
    return B >= 1;
}

hipError_t launch_pass(int model, const SymDefectArgs& sa, const NodeArgs<double>& na, hipStream_t s, const SymPlan& p) {
    if (model == EMI_MODEL_POINTMASS2D) return launch_pass_planned<PointMass2D<double>>(sa, na, s, p);
    if (model == EMI_MODEL_QUADROTOR2D) return launch_pass_planned<Quadrotor2D<double>>(sa, na, s, p);
    return hipErrorInvalidValue;
}

bool fused_supported(int model, int M, int ct) {
    if (ct == 0 || (ct >= 4 && ct <= 8))
        return (model == EMI_MODEL_POINTMASS2D || model == EMI_MODEL_QUADROTOR2D) && M >= 128 && M % 128 == 0;
    const int w = ct == 2 ? 2 : 1;     // ct == 3 is the ring variant of the 64-column tiling
    return (model == EMI_MODEL_POINTMASS2D || model == EMI_MODEL_QUADROTOR2D) && ct >= 1 && ct <= 3 &&
           M >= 128 * w && M % (128 * w) == 0;
}

hipError_t launch_symdefect(int model, const SymDefectArgs& a, hipStream_t s, bool set_attr, int ct, const SymPlan& plan) {
    if (!plan.ring1) {
        if (model == EMI_MODEL_POINTMASS2D) return launch_ring2_planned<PointMass2D<double>>(a, s, plan);
        if (model == EMI_MODEL_QUADROTOR2D) return launch_ring2_planned<Quadrotor2D<double>>(a, s, plan);
        return hipErrorInvalidValue;
    }
    if (ct == 0 || ct >= 4) {
        ct = 3;
        set_attr = true;    // cheap; the attribute bookkeeping of the caller is per requested ct
    }
    if (ct == 3) {   // LDS-DMA ring variant (column tiling as ct == 1)
        if (model == EMI_MODEL_POINTMASS2D) return launch_symdefect_ring_model<PointMass2D<double>>(a, s, set_attr);
        if (model == EMI_MODEL_QUADROTOR2D) return launch_symdefect_ring_model<Quadrotor2D<double>>(a, s, set_attr);
        return hipErrorInvalidValue;
    }
    if (model == EMI_MODEL_POINTMASS2D)
        return ct == 1 ? launch_symdefect_model<PointMass2D<double>, 1>(a, s, set_attr)
                       : launch_symdefect_model<PointMass2D<double>, 2>(a, s, set_attr);
    if (model == EMI_MODEL_QUADROTOR2D)
        return ct == 1 ? launch_symdefect_model<Quadrotor2D<double>, 1>(a, s, set_attr)
                       : launch_symdefect_model<Quadrotor2D<double>, 2>(a, s, set_attr);
    return hipErrorInvalidValue;
}

}  // namespace emi
