// emi_args.hpp -- tile constants and kernel argument blocks.  Shared by the kernels built into
// libemi355x.so and by the model programs compiled at run time (emi_rtc.hip), so it stays free of
// host-only includes.
#pragma once
#ifdef __HIPCC_RTC__
// values of include/emi355x.h (checked against it on the static side, emi_kernels.hpp)
#define EMI_PATH_ELLIPSE 0
#define EMI_PATH_DISC 1
#define EMI_PATH_TRACK 2
#define EMI_PATH_REC 8
#else
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "emi355x.h"
#endif

#define EMI_NODE_THREADS 256

// Memory order of the arrival tickets (COST finish of the node role, split-K combine of the MFMA role).  The partial sums
// travel as agent-scope (sc1: write-through, L1- and L2-line-dropping) stores, every storing wave drains vmcnt, a workgroup
// barrier, then ONE lane's agent-scope atomic add on the ticket; the workgroup whose add came last reads them back with
// agent-scope (sc1) loads -- row 1 of the hand-off table of MI355X_MICROARCH.md ("Valid forms"), measured valid on gfx950.
// The ticket itself stays RELAXED: as ACQ_REL its release half is `buffer_wbl2 sc1`, a write-back of the XCD's whole L2,
// issued once per node workgroup under the pass's 1 GB store stream -- measured on one box, same process: 0.2848 ms per
// pass against 0.2243 at 1024 instances, 0.0433 against 0.0347 at 128 (tools/ab_build.sh relaxed; profiles/r03_notes.md).
// -DEMI_TICKET_ACQ_REL builds the fenced form (ordering by the HIP memory model rather than by the ISA's cache behaviour).
#ifdef EMI_TICKET_ACQ_REL
#define EMI_TICKET_ORDER __ATOMIC_ACQ_REL
#else
#define EMI_TICKET_ORDER __ATOMIC_RELAXED
#endif
// one wait state between an SALU write of m0 and an LDS-DMA instruction (-DEMI_NO_M0_NOP: the round-2 form, A/B only)
#ifdef EMI_NO_M0_NOP
#define EMI_M0_NOP ""
#else
#define EMI_M0_NOP "s_nop 0\n\t"
#endif

// defect GEMM tile (fp64 MFMA path)
#define DEF_TM 96
#define DEF_TN 128
#define DEF_BK 16

// even/odd defect kernel (emi_symdefect.hip): 16 instances x 64*CT half-indices per workgroup, K tile 16
#define FUSED_TI 16
#define FUSED_BK 16

#include "emi_models.hpp"

namespace emi {

template <typename T> struct NodeArgs {
    const T* X;         // [B][ns][M]
    const T* U;         // [B][nc][M]
    T* RES;             // [B][nres][M]
    T* VALS;            // [B][nvals][M]
    T* cost_part;       // [B][nchunks]
    T* cost;            // [B]
    unsigned* cost_ticket;  // [B] zeroed once; non-null: the node kernel finishes COST itself (no emi_cost_finish_kernel)
    const T* w;         // [M]  LGL weights
    const T* node_t;    // [M]  node times t0 + h (tau+1)
    const T* Ddiag;     // [M]  D_kk
    const T* path;      // [path_sets][np][EMI_PATH_REC]
    const T* track_x;   // [track_sets][ntracks][M]
    const T* track_y;
    int M, B, np, nres, nvals;
    int path_sets, track_sets, ntracks;
    int px, py;
    int store_mode;     // launcher only: cache policy of the result stores (0 plain, 1 write-through sc1, 2 non-temporal)
    T h, sgn;
    ModelParams<T> P;
};

template <typename T> struct HessArgs {
    const T* X;
    const T* U;
    const T* lamF;      // [B][ns][M]
    const T* lamC;      // [B][np][M]
    T* H;               // [B][nhess][M]
    const T* w;
    const T* node_t;
    const T* path;
    int M, B, np, path_sets, px, py;
    T h, sgn, sigma;
    ModelParams<T> P;
};

struct SymDefectArgs {
    const double* X;
    const double* U;
    double* RES;
    const double* node_t;
    const double* De;       // [M/2][M/2]  (D[i][j] + D[i][N-j]) / 2
    const double* Do;       // [M/2][M/2]  (D[i][j] - D[i][N-j]) / 2
    int M, B, nres;
    int order;              // block -> tile order within an XCD (see emi_symdefect.hip)
    int ablate;             // diagnostics (results invalid): 4 skip epilogue, 8 all tiles read the X rows of the first 16 instances,
                            // 16 / 32 the MFMA / node role of the one-launch pass returns at once
                            // (1 skip MFMAs, 2 skip operand DMA: the round-1 ring kernel only)
    int ksplit;             // state-split ring kernel: K slices per tile (1 = none); > 1 goes through `slab`
    double* slab;           // [tiles][ksplit][2 SW][4][256] partial sums of a split-K launch
    unsigned* tile_ticket;  // [tiles] zero before the first launch, self-resetting: the slices of a tile are combined by the
                            // workgroup that draws the tile's last ticket (nullptr: emi_symdefect_combine_kernel does it)
    int mfma_first;         // one-launch pass: 1 the MFMA workgroups of an XCD come first in its share of the grid, 0 evenly interleaved
                            // with the node workgroups
    int cpart, cx;          // tile order of the state-split ring kernel (plan_symdefect): the column tiles are cut into cpart
                            // partitions, one per group of 8 / cpart XCDs, and cx column tiles of an X tile run as neighbours
                            // (cpart = 0: plain order, the column tiles of an X tile as neighbours, X tiles dealt over the XCDs;
                            //  cpart = -G: grouped order, super-blocks of G instance groups per XCD, column blocks of cx: ring_tile_of)
    double h;
    ModelParams<double> P;
};

struct DefectArgs {
    const double* X;    // [R][M], R = B*ns
    const double* D;    // [M][M] row-major
    double* RES;        // [B][nres][M]; defect rows are accumulated into
    int R, M, ns, nres;
};
struct DefectArgsF32 {
    const float* X;
    const float* D;
    float* RES;
    int R, M, ns, nres;
};
// Which tile a workgroup of the state-split ring kernel works on.  tl counts tiles in XCD-local runs (an XCD gets a contiguous
// range of the launch); ntiles column tiles, ngrp (instance group x state group) rows of tiles.  cpart = 0: plain order, the
// column tiles of a group are neighbours.  cpart > 0: XCD x works on column partition x % cpart (ntiles / cpart columns, cx of
// them interleaved at a time: cx divides that count) and on group partition x / cpart (ngrp cpart / 8 groups).  cpart < 0: the
// grouped order described in the function (G = -cpart instance groups per super-block, nsg state groups per instance group;
// needs (ngrp / nsg) % (8 G) == 0 and ntiles % cx == 0: plan_symdefect checks).  One function for
// the kernel and for the host-side check that every tile is visited exactly once (emi_debug_tile_order, tests/test_abi.py).
struct RingTile { int ntile, grp; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline RingTile ring_tile_of(int tl, int ntiles, int ngrp, int cpart, int cx, int nsg = 1) {
    RingTile t;
    if (cpart > 0) {
        const int per_xcd = ngrp * ntiles / 8;
        const int x = tl / per_xcd, m = tl - x * per_xcd;
        const int pc = x % cpart, pg = x / cpart;
        const int ncol = ntiles / cpart, ng = ngrp / (8 / cpart);
        const int cb = m / (ng * cx), r = m - cb * ng * cx;
        t.grp = pg * ng + r / cx;
        t.ntile = pc * ncol + cb * cx + r % cx;
    } else if (cpart < 0) {
        // Grouped order (cpart = -G): an XCD keeps its own contiguous range of instance groups (the range whose node-role
        // workgroups it also runs in the one-launch pass) and walks it in super-blocks of G groups; inside a super-block the
        // column tiles go in blocks of cx, and for one column block all G groups x nsg state groups run before the next
        // block.  X and U of a super-block are then fetched into this XCD's L2 once, by whichever role gets there first,
        // and a column block's share of De / Do is walked once per super-block instead of once per (group, state group).
        const int G = -cpart, per_xcd = ngrp * ntiles / 8;
        const int x = tl / per_xcd, m = tl - x * per_xcd;
        const int per_sb = G * nsg * ntiles, sb = m / per_sb, r = m - sb * per_sb;
        const int per_cb = G * nsg * cx, cb = r / per_cb, r2 = r - cb * per_cb;
        const int g = r2 / (nsg * cx), r3 = r2 - g * (nsg * cx);
        const int mt = x * (ngrp / nsg / 8) + sb * G + g;
        t.grp = mt * nsg + r3 / cx;
        t.ntile = cb * cx + r3 % cx;
    } else {
        t.ntile = tl % ntiles;
        t.grp = tl / ntiles;
    }
    return t;
}
// Role of block j of an XCD's share of the one-launch pass grid (nm MFMA-role and nn node-role blocks): {is_mfma, index within
// the role}.  order 0: evenly interleaved; 1: all MFMA blocks first; >= 100: interleaved with the MFMA blocks at order / 100 times
// the even density until they are used up, node blocks at the tail (a density beyond one MFMA block per block saturates HERE at
// order 1, which keeps the map a bijection for every caller: the first form of this map, unclamped, sent node workgroups out of range).  One function for the kernel and for the host-side check (emi_debug_pass_roles).
struct PassRole { int mfma, index; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline PassRole pass_role_of(int j, int nm, int nn, int order) {
    const long long t = (long long)nm + nn;
    long long m0, m1;
    const long long d = order >= 100 ? order : 100;
    // a density beyond one MFMA block per block (d nm > 100 t) would deal two MFMA indices to one block -- the map would no
    // longer be a bijection: it saturates HERE, for every caller, at "all MFMA blocks first"
    if (order == 1 || d * nm > 100 * t) {
        m0 = j < nm ? j : nm;
        m1 = j < nm ? j + 1 : nm;
    } else {
        m0 = ((long long)j * nm * d) / (100 * t);
        m1 = ((long long)(j + 1) * nm * d) / (100 * t);
        if (m0 > nm) m0 = nm;
        if (m1 > nm) m1 = nm;
    }
    PassRole r;
    r.mfma = m1 > m0;
    r.index = (int)(r.mfma ? m0 : j - m0);
    return r;
}
// the whole pass as one launch (emi_pass_f64_kernel): both roles' arguments and how the grid is dealt between them
struct PassArgs {
    SymDefectArgs s;
    NodeArgs<double> n;
    int nm8, nn8;           // MFMA / node workgroups per XCD (counts rounded up to a multiple of 8 ...
    int nm, nn;             // ... the workgroups past these real counts do nothing)
    int nbx;                // node chunks per instance
};

}  // namespace emi
