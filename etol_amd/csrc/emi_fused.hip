// emi_fused.hip -- the whole evaluation pass in ONE gfx950 kernel (fp64).
//
//   K4  defect product D.X on v_mfma_f64_16x16x4_f64, with the flops halved by the
//       centro-antisymmetry of the LGL differentiation matrix, D[N-i][N-j] = -D[i][j]:
//          e_j = x_j + x_{N-j},  o_j = x_j - x_{N-j}                        (j < M/2)
//          a_i = sum_j De[i][j] e_j,   b_i = sum_j Do[i][j] o_j             (i < M/2)
//          (D x)_i = a_i + b_i,        (D x)_{N-i} = -a_i + b_i
//       De = (D[i][j] + D[i][N-j])/2, Do = (D[i][j] - D[i][N-j])/2 are built once on the host.
//   K1/K2/K3/K5  node Jacobian blocks, path rows, cost gradient and cost quadrature are
//       produced by STREAM workgroups of the same launch: workgroups alternate (in groups of 8,
//       one per XCD) between the MFMA role and the streaming role, so every CU holds both kinds
//       and the ~1 KB/node of HBM stores runs under the matrix pipe.  (Doing the node work
//       inside the MFMA workgroups' K loop was measured first: every workgroup then follows the
//       same schedule, the chip alternates between an all-MFMA and an all-store phase, and
//       vmcnt's in-order retirement makes each K tile's operand wait also wait for the stores.)
//   Epilogue: accumulator rows are ordered state-major inside a workgroup (row = state*16 +
//       instance), so one lane holds all NS components of D.X for its (instance, node) pairs:
//       it evaluates f there and writes  defect = D.X - h f  directly, with no F round trip.
//
// Workgroup = 256 threads = 4 waves; tile = 16 instances x 64 half-indices i (= 128 nodes: i and
// N-i).  LDS (one buffer, register prefetch of the next K tile): E,O [NS*16][18], De,Do [64][18]
// doubles; rows padded to 18 doubles: 16-byte aligned for ds_write_b128, conflict-free for the
// ds_read_b64 fragment reads.  blockIdx -> tile map is XCD-aware: workgroups that share a De/Do
// panel share blockIdx % 8, i.e. one XCD's L2.
//
// Requires M % 128 == 0 and an exactly centro-antisymmetric D (emi_lgl guarantees it; emi_set_mesh
// checks); every other shape takes the general two-kernel path of emi_kernels.hip.
#include <hip/hip_runtime.h>

#include "emi_kernels.hpp"
#include "emi_models.hpp"

namespace emi {

typedef double d4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T fused_wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- Jacobian / path / cost work of VEC adjacent nodes (everything except the defect rows) ----
// VEC = 2: 16-byte loads and stores per lane, 1 KB per wave store instruction; an 8-byte-per-lane
// version of this item was store-ISSUE bound (~2x the wave instructions for the same bytes).
template <int VEC> struct DPack;
template <> struct DPack<1> { using type = double; };
template <> struct DPack<2> { using type = double2; };
template <int VEC> __device__ __forceinline__ void ldv(const double* __restrict__ p, double (&r)[VEC]) {
    using P = typename DPack<VEC>::type;
    const P v = *reinterpret_cast<const P*>(p);
    const double* e = reinterpret_cast<const double*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = e[i];
}
template <int VEC> __device__ __forceinline__ void stv(double* __restrict__ p, const double (&r)[VEC]) {
    using P = typename DPack<VEC>::type;
    P v;
    double* e = reinterpret_cast<double*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) e[i] = r[i];
    *reinterpret_cast<P*>(p) = v;
}

template <class Model, bool JAC, int VEC>
__device__ __forceinline__ double fused_node_item(const FusedArgs& a, int inst, int node, bool active) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    double lterm = 0.0;
    if (active) {
        const int M = a.M;
        const double* __restrict__ Xb = a.X + (size_t)inst * NS * M;
        const double* __restrict__ Ub = a.U + (size_t)inst * NC * M;
        double* __restrict__ Rb = a.RES + (size_t)inst * a.nres * M;
        double* __restrict__ Vb = a.VALS + (size_t)inst * a.nvals * M;
        double z[NV][VEC];
#pragma unroll
        for (int v = 0; v < NS; ++v) ldv<VEC>(Xb + (size_t)v * M + node, z[v]);
#pragma unroll
        for (int v = 0; v < NC; ++v) ldv<VEC>(Ub + (size_t)v * M + node, z[NS + v]);
        double wk[VEC], tk[VEC];
        ldv<VEC>(a.w + node, wk);
        ldv<VEC>(a.node_t + node, tk);
        const double h = a.h;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            double ze[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) ze[v] = z[v][e];
            lterm += wk[e] * Model::cost(a.P, ze, tk[e]);
        }
        if (JAC) {
            double dkk[VEC];
            ldv<VEC>(a.Ddiag + node, dkk);
            double J[NS][NV][VEC], g[NV][VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double ze[NV], Je[NS][NV], ge[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) ze[v] = z[v][e];
                Model::jac(a.P, ze, tk[e], Je);
                Model::grad(a.P, ze, tk[e], ge);
                const double cw = a.sgn * h * wk[e];
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int v = 0; v < NV; ++v) J[i][v][e] = -h * Je[i][v] + (v == i ? dkk[e] : 0.0);
#pragma unroll
                for (int v = 0; v < NV; ++v) g[v][e] = cw * ge[v];
            }
            // entries are M doubles apart: walk one pointer instead of forming 56 addresses
            double* __restrict__ p = Vb + node;
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    stv<VEC>(p, J[i][v]);
                    p += M;
                }
            p += (size_t)(2 * a.np) * M;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                stv<VEC>(p, g[v]);
                p += M;
            }
        }
        const int np = a.np;
        if (np > 0) {
            const int set = a.path_sets > 1 ? inst : 0;
            // the table is read-only for the whole launch: address it as constant memory so that the
            // wave-uniform record reads become scalar loads (lgkmcnt), not vector loads whose
            // vmcnt waits would also wait for the streaming stores
            typedef const __attribute__((address_space(4))) double* cptr_t;
            cptr_t rec = (cptr_t)(a.path + (size_t)set * np * EMI_PATH_REC);
            double* __restrict__ pc = Rb + (size_t)NS * M + node;
            double* __restrict__ pj = Vb + (size_t)(NS * NV) * M + node;
            double px[VEC], py[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                px[e] = z[0][e];
                py[e] = z[0][e];
#pragma unroll
                for (int v = 1; v < NS; ++v) {
                    px[e] = (v == a.px) ? z[v][e] : px[e];
                    py[e] = (v == a.py) ? z[v][e] : py[e];
                }
            }
            for (int j = 0; j < np; ++j) {
                cptr_t r = rec + j * EMI_PATH_REC;
                const int kind = (int)r[0];
                double c[VEC], cx[VEC], cy[VEC];
                if (kind == EMI_PATH_DISC) {
                    const double xc = r[1], yc = r[2], rsq = r[3];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const double dx = px[e] - xc, dy = py[e] - yc;
                        c[e] = (dx * dx + dy * dy) * -1.0 + rsq;
                        cx[e] = -2.0 * dx;
                        cy[e] = -2.0 * dy;
                    }
                } else if (kind == EMI_PATH_ELLIPSE) {
                    const double xc = r[1], yc = r[2], ct = r[3], st = r[4], asq = r[5], bsq = r[6];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const double dx = px[e] - xc, dy = py[e] - yc;
                        const double delx = ct * dx - st * dy, dely = st * dx + ct * dy;
                        c[e] = asq * bsq - (bsq * (delx * delx) + asq * (dely * dely));
                        cx[e] = -2.0 * (bsq * delx * ct + asq * dely * st);
                        cy[e] = -2.0 * (-bsq * delx * st + asq * dely * ct);
                    }
                } else {
                    const int trk = (int)r[1];
                    const double rsq = r[2];
                    const int tset = a.track_sets > 1 ? inst : 0;
                    const size_t off = ((size_t)tset * a.ntracks + trk) * M + node;
                    double xc[VEC], yc[VEC];
                    ldv<VEC>(a.track_x + off, xc);
                    ldv<VEC>(a.track_y + off, yc);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const double dx = px[e] - xc[e], dy = py[e] - yc[e];
                        c[e] = (dx * dx + dy * dy) * -1.0 + rsq;
                        cx[e] = -2.0 * dx;
                        cy[e] = -2.0 * dy;
                    }
                }
                stv<VEC>(pc, c);
                pc += M;
                if (JAC) {
                    stv<VEC>(pj, cx);
                    stv<VEC>(pj + M, cy);
                    pj += 2 * (size_t)M;
                }
            }
        }
    }
    return lterm;
}

// ---- streaming role: 2048 contiguous (instance, node) pairs, 4 items of 2 nodes per thread ----
template <class Model, bool JAC>
__device__ __forceinline__ void fused_stream_role(const FusedArgs& a, int g, double* csum) {
    constexpr int IV = 2, RUN = FUSED_TI * 2 * FUSED_TN, NITEMS = RUN / (256 * IV);
    const int M = a.M, B = a.B;
    const int mtiles = (B + FUSED_TI - 1) / FUSED_TI;
    const int ntile = g / mtiles, mtile = g - ntile * mtiles;      // same tile numbering as the MFMA role
    const int inst0 = mtile * FUSED_TI;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll 1
    for (int item = 0; item < NITEMS; ++item) {
        // The ntiles stream workgroups of one 16-instance group split its 16*M (instance, node)
        // pairs into contiguous runs of 2048; the 4 waves of one item write 4 adjacent 1 KB pieces
        // (4 KB contiguous) of every output row.  A wave's 128 pairs belong to ONE instance: the
        // instance index is made provably uniform so that the keep-out records are scalar loads.
        const int flat0 = __builtin_amdgcn_readfirstlane(ntile * RUN + item * (256 * IV) + wid * (64 * IV));
        const int il = flat0 / M;
        const int node = flat0 - il * M + IV * lane;
        const int inst = inst0 + il;
        const double ls = fused_node_item<Model, JAC, IV>(a, inst, node, inst < B);
        const double ws = fused_wave_sum(ls);
        if (lane == 0) csum[item * 4 + wid] = ws;   // one slot per wave-item, summed in order below
    }
    __syncthreads();
    // cost partial of every instance this workgroup held: fixed-order sum of its slots
    const int per_inst = M < RUN ? M : RUN;
    const int ninst = RUN / per_inst, slots = per_inst / (64 * IV);
    if (tid < ninst) {
        const int flat = ntile * RUN + tid * per_inst;
        const int il = flat / M, chunk = (flat - il * M) / RUN;
        double sacc = 0.0;
        for (int q = 0; q < slots; ++q) sacc += csum[tid * slots + q];
        if (inst0 + il < B) a.cost_part[(size_t)(inst0 + il) * a.cost_chunks + chunk] = sacc;
    }
}

template <class Model, bool JAC>
__global__ __launch_bounds__(256, 2) void emi_fused_f64_kernel(FusedArgs a) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    constexpr int TI = FUSED_TI, TM = NS * TI, TN = FUSED_TN, BK = FUSED_BK, LDK = BK + 2;
    constexpr int A_PASS = TM * (BK / 2) / 256;   // double2 pieces per thread for E/O
    constexpr int B_PASS = TN * (BK / 2) / 256;   // and for De / Do
    static_assert(TM * (BK / 2) % 256 == 0 && TN * (BK / 2) % 256 == 0, "staging shape");
    static_assert(TN == 64 && TI == 16, "wave layout below assumes a 16 x 64 tile");

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Es = smem;                    // [TM][LDK]
    double* Os = Es + TM * LDK;           // [TM][LDK]
    double* Des = Os + TM * LDK;          // [TN][LDK]
    double* Dos = Des + TN * LDK;         // [TN][LDK]
    double* csum = Dos + TN * LDK;        // [16] cost partials of a stream workgroup

    // roles alternate in groups of 8 consecutive block ids (= one workgroup per XCD under the
    // observed round-robin placement): ids 0-7 MFMA, 8-15 stream, 16-23 MFMA, ...
    const int role = (blockIdx.x >> 3) & 1;
    const int g = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7);   // index within the role
    const int M = a.M, Hh = M >> 1, B = a.B;
    const int mtiles = (B + TI - 1) / TI;
    const int ntl = mtiles * (Hh / TN);                          // tiles per role; the grid is padded to 8
    if (role == 1) {
        if (g < ntl && !(a.ablate & 2)) fused_stream_role<Model, JAC>(a, g, csum);
        return;
    }

    int bid = g;
    {   // XCD-aware bijective remap: MFMA workgroups that share a De/Do panel share g % 8
        const int nwg = gridDim.x >> 1, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    if (bid >= ntl) return;
    const int ntile = bid / mtiles, mtile = bid - ntile * mtiles;
    const int inst0 = mtile * TI, i0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    d4 acc_a[NS], acc_b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc_a[s] = d4{0.0, 0.0, 0.0, 0.0};
        acc_b[s] = d4{0.0, 0.0, 0.0, 0.0};
    }

    // ---- staging: next K tile global -> registers (e/o formed on the fly) -> LDS ----------
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
    auto lstore = [&]() {
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
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK);
        const double* Db = Des + (wid * 16 + r16) * LDK + kq;
        const double* Ob = Dos + (wid * 16 + r16) * LDK + kq;
#pragma unroll 2
        for (int ks = (a.ablate & 1) ? BK / 4 : 0; ks < BK / 4; ++ks) {
            const double bfe = Db[ks * 4], bfo = Ob[ks * 4];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double afe = Es[(s * 16 + r16) * LDK + ks * 4 + kq];
                const double afo = Os[(s * 16 + r16) * LDK + ks * 4 + kq];
                acc_a[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(afe, bfe, acc_a[s], 0, 0, 0);
                acc_b[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(afo, bfo, acc_b[s], 0, 0, 0);
            }
        }
        __syncthreads();                   // every wave is done reading this K tile
        if (kt + 1 < nkt) {
            lstore();
            __syncthreads();               // refill visible
        }
    }

    // ---- epilogue: defect = D.X - h f, forward node i and mirrored node N-i ---------------
    const int col = wid * 16 + r16;
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
                const double dx = side == 0 ? acc_a[s][i] + acc_b[s][i] : acc_b[s][i] - acc_a[s][i];
                Rb[(size_t)s * M + node] = dx - a.h * f[s];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <class Model>
static hipError_t launch_fused_model(const FusedArgs& a, bool jac, hipStream_t s, bool set_attr) {
    constexpr int NS = Model::NS;
    const int mtiles = (a.B + FUSED_TI - 1) / FUSED_TI, ntiles = (a.M / 2) / FUSED_TN;
    const size_t lds = ((size_t)2 * NS * FUSED_TI + 2 * FUSED_TN) * (FUSED_BK + 2) * sizeof(double) +
                       2 * FUSED_TI * sizeof(double);
    if (set_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)emi_fused_f64_kernel<Model, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)emi_fused_f64_kernel<Model, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // one MFMA and one stream workgroup per tile, alternating in groups of 8 block ids
    const int ntl = mtiles * ntiles;
    dim3 grid(2 * ((ntl + 7) / 8) * 8), block(256);
    if (jac) hipLaunchKernelGGL((emi_fused_f64_kernel<Model, true>), grid, block, lds, s, a);
    else     hipLaunchKernelGGL((emi_fused_f64_kernel<Model, false>), grid, block, lds, s, a);
    return hipGetLastError();
}

bool fused_supported(int model, int M) {
    return (model == EMI_MODEL_POINTMASS2D || model == EMI_MODEL_QUADROTOR2D) && M >= 128 && M % 128 == 0;
}

int fused_cost_chunks(int M) { return M <= 2048 ? 1 : M / 2048; }

hipError_t launch_fused(int model, const FusedArgs& a, bool jac, hipStream_t s, bool set_attr) {
    switch (model) {
        case EMI_MODEL_POINTMASS2D: return launch_fused_model<PointMass2D<double>>(a, jac, s, set_attr);
        case EMI_MODEL_QUADROTOR2D: return launch_fused_model<Quadrotor2D<double>>(a, jac, s, set_attr);
    }
    return hipErrorInvalidValue;
}

}  // namespace emi
