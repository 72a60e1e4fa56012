// emi_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the collocation hot path.
//
//   K1  per-node dynamics + Jacobian          \
//   K2  per-node path constraints + Jacobian   |  emi_nodes_kernel  (one launch)
//   K3  integrand cost + gradient, reduced     |
//   K5  Jacobian values into the NLP array    /
//   K3' emi_cost_finish_kernel   fixed-order sum of the per-block cost partials
//   K4  emi_defect_f64_kernel    defect rows += X . D^T   (v_mfma_f64_16x16x4_f64)
//   KH  emi_hess_kernel          Lagrangian Hessian node blocks
//
// The reference has no device code: the functional spec is the CPU arithmetic
// of ePSOPT::dae / integrand_cost (reference src/ePSOPT/ePSOPT.cpp:186-276),
// the example node functions (src/Examples/PSOPT/etol_psopt_example1.cpp:
// 101-258) and PSOPT's Legendre pseudospectral transcription selected at
// ePSOPT.cpp:62-72.  Array layouts are documented in include/emi355x.h.
//
// Written for gfx950 only: 64-lane wavefronts, DPP/shuffle reductions across
// the wave, MFMA for the dense D.X product, coalesced 16-byte-per-lane HBM
// accesses along the node axis.
#include <hip/hip_runtime.h>
#include "emi_kernels.hpp"
#include "emi_models.hpp"

namespace emi {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
template <typename T, int VEC> struct Pack;
template <> struct Pack<double, 1> { using type = double; };
template <> struct Pack<double, 2> { using type = double2; };
template <> struct Pack<float, 1> { using type = float; };
template <> struct Pack<float, 2> { using type = float2; };
template <> struct Pack<float, 4> { using type = float4; };

template <typename T, int VEC>
EMI_DEV void load_vec(const T* __restrict__ p, T (&r)[VEC]) {
    using P = typename Pack<T, VEC>::type;
    const P v = *reinterpret_cast<const P*>(p);
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = e[i];
}
template <typename T, int VEC>
EMI_DEV void store_vec(T* __restrict__ p, const T (&r)[VEC]) {
    using P = typename Pack<T, VEC>::type;
    P v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) e[i] = r[i];
    *reinterpret_cast<P*>(p) = v;
}

// wave64 sum (all lanes end with lane 0 holding the total)
template <typename T> EMI_DEV T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// K1+K2+K3+K5: one thread = VEC consecutive nodes of one instance.
//   grid.x = node chunks (EMI_NODE_THREADS*VEC nodes each), grid.y = instance
// Reads  z = X[b][:][k], U[b][:][k]           (coalesced along k)
// Writes RES defect rows  = -h f_i            (K4 adds D.X on top)
//        RES path rows    = c_j
//        VALS             = Jacobian values, cost gradient
//        cost_part[b][chunk] = sum_k w_k L_k  over this block (wave DPP + LDS)
// DEFROWS = false leaves the defect rows alone: the even/odd defect kernel
// (emi_symdefect.hip) then produces them, concurrently, on another stream.
// A thread issues all its loads first and only stores afterwards: vmcnt
// retires in order, so a load behind the ~116 streaming stores would wait for
// every one of them.  For the same reason the wave-uniform keep-out records
// are read through the constant address space (scalar loads, lgkmcnt).
// ---------------------------------------------------------------------------
template <typename T, class Model, int VEC, bool JAC, bool DEFROWS>
__global__ __launch_bounds__(EMI_NODE_THREADS) void emi_nodes_kernel(NodeArgs<T> a) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    const int b = blockIdx.y;
    const int M = a.M;
    const int k0 = (blockIdx.x * EMI_NODE_THREADS + threadIdx.x) * VEC;
    const bool active = k0 < M;  // M % VEC == 0 by dispatch, so the pack is whole

    T lsum = T(0);
    if (active) {
        const T* __restrict__ Xb = a.X + (size_t)b * NS * M;
        const T* __restrict__ Ub = a.U + (size_t)b * NC * M;
        T* __restrict__ Rb = a.RES + (size_t)b * a.nres * M;
        T* __restrict__ Vb = JAC ? a.VALS + (size_t)b * a.nvals * M : nullptr;

        T z[NV][VEC];
#pragma unroll
        for (int v = 0; v < NS; ++v) load_vec<T, VEC>(Xb + (size_t)v * M + k0, z[v]);
#pragma unroll
        for (int v = 0; v < NC; ++v) load_vec<T, VEC>(Ub + (size_t)v * M + k0, z[NS + v]);
        T wk[VEC], tk[VEC], dkk[VEC];
        load_vec<T, VEC>(a.w + k0, wk);
        load_vec<T, VEC>(a.node_t + k0, tk);
        if (JAC) load_vec<T, VEC>(a.Ddiag + k0, dkk);

        const T h = a.h;
        // ---- K1 dynamics ------------------------------------------------
        {
            T fo[NS][VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T ze[NV], fe[NS];
#pragma unroll
                for (int v = 0; v < NV; ++v) ze[v] = z[v][e];
                if (DEFROWS) {
                    Model::f(a.P, ze, tk[e], fe);
#pragma unroll
                    for (int i = 0; i < NS; ++i) fo[i][e] = -h * fe[i];
                }
                lsum += wk[e] * Model::cost(a.P, ze, tk[e]);
            }
            if (DEFROWS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) store_vec<T, VEC>(Rb + (size_t)i * M + k0, fo[i]);
            }
        }
        if (JAC) {
            // ---- K1' dynamics Jacobian block + K5 placement ---------------
            T Jv[NS][NV][VEC];
            T gv[NV][VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T ze[NV], Je[NS][NV], ge[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) ze[v] = z[v][e];
                Model::jac(a.P, ze, tk[e], Je);
                Model::grad(a.P, ze, tk[e], ge);
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                        Jv[i][v][e] = -h * Je[i][v] + (v == i ? dkk[e] : T(0));
                const T cw = a.sgn * h * wk[e];
#pragma unroll
                for (int v = 0; v < NV; ++v) gv[v][e] = cw * ge[v];
            }
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    store_vec<T, VEC>(Vb + (size_t)(i * NV + v) * M + k0, Jv[i][v]);
            T* __restrict__ Gb = Vb + (size_t)(NS * NV + 2 * a.np) * M;
#pragma unroll
            for (int v = 0; v < NV; ++v) store_vec<T, VEC>(Gb + (size_t)v * M + k0, gv[v]);
        }
        // ---- K2 path constraints (records are wave-uniform: scalar loads) --
        const int np = a.np;
        if (np > 0) {
            const int set = a.path_sets > 1 ? b : 0;
            typedef const __attribute__((address_space(4))) T* cptr_t;   // read-only table: scalar loads
            cptr_t rec = (cptr_t)(a.path + (size_t)set * np * EMI_PATH_REC);
            T* __restrict__ Cb = Rb + (size_t)NS * M;
            T* __restrict__ JCb = JAC ? Vb + (size_t)(NS * NV) * M : nullptr;
            // the keep-outs act on two runtime-chosen states: select with
            // compares, a runtime register index would go to scratch
            T px[VEC], py[VEC];
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
                T c[VEC], cx[VEC], cy[VEC];
                if (kind == EMI_PATH_DISC) {
                    // r^2 - ((x-xc)^2 + (y-yc)^2): etol_psopt_example1.cpp:243-247
                    const T xc = r[1], yc = r[2], rsq = r[3];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T dx = px[e] - xc, dy = py[e] - yc;
                        c[e] = (dx * dx + dy * dy) * T(-1) + rsq;
                        cx[e] = T(-2) * dx;
                        cy[e] = T(-2) * dy;
                    }
                } else if (kind == EMI_PATH_ELLIPSE) {
                    // etol_psopt_example1.cpp:174-182 (constants precomputed on host)
                    const T xc = r[1], yc = r[2], ct = r[3], st = r[4], asq = r[5], bsq = r[6];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T dx = px[e] - xc, dy = py[e] - yc;
                        const T delx = ct * dx - st * dy;
                        const T dely = st * dx + ct * dy;
                        c[e] = asq * bsq - (bsq * (delx * delx) + asq * (dely * dely));
                        cx[e] = T(-2) * (bsq * delx * ct + asq * dely * st);
                        cy[e] = T(-2) * (-bsq * delx * st + asq * dely * ct);
                    }
                } else {  // EMI_PATH_TRACK: centre tabulated at the node times
                    const int trk = (int)r[1];
                    const T rsq = r[2];
                    const int tset = a.track_sets > 1 ? b : 0;
                    const size_t off = ((size_t)tset * a.ntracks + trk) * M + k0;
                    T xc[VEC], yc[VEC];
                    load_vec<T, VEC>(a.track_x + off, xc);
                    load_vec<T, VEC>(a.track_y + off, yc);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T dx = px[e] - xc[e], dy = py[e] - yc[e];
                        c[e] = (dx * dx + dy * dy) * T(-1) + rsq;
                        cx[e] = T(-2) * dx;
                        cy[e] = T(-2) * dy;
                    }
                }
                store_vec<T, VEC>(Cb + (size_t)j * M + k0, c);
                if (JAC) {
                    store_vec<T, VEC>(JCb + (size_t)(2 * j) * M + k0, cx);
                    store_vec<T, VEC>(JCb + (size_t)(2 * j + 1) * M + k0, cy);
                }
            }
        }
    }
    // ---- K3 cost quadrature: wave reduction, then across the block's waves --
    __shared__ T wsum[EMI_NODE_THREADS / 64];
    const T ws = wave_sum(lsum);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wsum[wid] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        T s = T(0);
#pragma unroll
        for (int i = 0; i < EMI_NODE_THREADS / 64; ++i) s += wsum[i];
        a.cost_part[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
}

template <typename T>
__global__ void emi_cost_finish_kernel(const T* __restrict__ part, T* __restrict__ cost, int B,
                                       int nchunks, T scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    T s = T(0);
    for (int c = 0; c < nchunks; ++c) s += part[(size_t)b * nchunks + c];
    cost[b] = scale * s;
}

// ---------------------------------------------------------------------------
// KH: Lagrangian Hessian node blocks (packed lower triangle, (NV)(NV+1)/2).
//   H = sigma*sgn*h*w_k L_zz  - h sum_i lamF[i][k] f_i,zz  + sum_j lamC[j][k] c_j,zz
// ---------------------------------------------------------------------------
template <typename T, class Model>
__global__ __launch_bounds__(EMI_NODE_THREADS) void emi_hess_kernel(HessArgs<T> a) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV, NH = NV * (NV + 1) / 2;
    const int b = blockIdx.y;
    const int M = a.M;
    const int k = blockIdx.x * EMI_NODE_THREADS + threadIdx.x;
    if (k >= M) return;
    T z[NV], cf[NS], H[NH];
#pragma unroll
    for (int v = 0; v < NS; ++v) z[v] = a.X[((size_t)b * NS + v) * M + k];
#pragma unroll
    for (int v = 0; v < NC; ++v) z[NS + v] = a.U[((size_t)b * NC + v) * M + k];
#pragma unroll
    for (int i = 0; i < NS; ++i) cf[i] = -a.h * a.lamF[((size_t)b * NS + i) * M + k];
#pragma unroll
    for (int q = 0; q < NH; ++q) H[q] = T(0);
    const T cL = a.sigma * a.sgn * a.h * a.w[k];
    Model::hess(a.P, z, a.node_t[k], cL, cf, H);
    const int np = a.np;
    if (np > 0) {
        const int set = a.path_sets > 1 ? b : 0;
        const T* __restrict__ rec = a.path + (size_t)set * np * EMI_PATH_REC;
        T hxx = T(0), hxy = T(0), hyy = T(0);
        for (int j = 0; j < np; ++j) {
            const T* __restrict__ r = rec + j * EMI_PATH_REC;
            const int kind = (int)r[0];
            const T mu = a.lamC[((size_t)b * np + j) * M + k];
            if (kind == EMI_PATH_ELLIPSE) {
                const T ct = r[3], st = r[4], asq = r[5], bsq = r[6];
                hxx += mu * T(-2) * (bsq * ct * ct + asq * st * st);
                hyy += mu * T(-2) * (bsq * st * st + asq * ct * ct);
                hxy += mu * T(-2) * ct * st * (asq - bsq);
            } else {  // disc / track: -2 I
                hxx += mu * T(-2);
                hyy += mu * T(-2);
            }
        }
        const int lo = a.px < a.py ? a.px : a.py, hi = a.px < a.py ? a.py : a.px;
        const int qxx = a.px * (a.px + 1) / 2 + a.px, qyy = a.py * (a.py + 1) / 2 + a.py;
        const int qxy = hi * (hi + 1) / 2 + lo;
#pragma unroll
        for (int q = 0; q < NH; ++q)
            H[q] += (q == qxx ? hxx : T(0)) + (q == qyy ? hyy : T(0)) + (q == qxy ? hxy : T(0));
    }
    T* __restrict__ Hb = a.H + (size_t)b * NH * M;
#pragma unroll
    for (int q = 0; q < NH; ++q) Hb[(size_t)q * M + k] = H[q];
}

// ---------------------------------------------------------------------------
// K4 (fp64): defect rows += X . D^T      out[r][n] += sum_j X[r][j] * D[n][j]
//   r = (instance, state) = row of X viewed as [R = B*ns][M]
//   v_mfma_f64_16x16x4_f64:  lane l supplies A[l&15][l>>4], B[l>>4][l&15];
//   result reg i of lane l is out[(l>>4) + 4i][l&15]   (CDNA4 f64 C/D map).
// Block tile DEF_TM x DEF_TN, K tile DEF_BK, 4 waves side by side along n,
// double-buffered LDS with register prefetch of the next K tile.
// LDS rows are padded to DEF_BK+2 doubles: 16-byte aligned for ds_write_b128
// and conflict-free for the ds_read_b64 fragment reads (36 r + 2 kk mod 64
// covers every bank once per 32-lane half).
// Block -> tile map is XCD-aware: blocks that share a D column panel share
// blockIdx % 8, i.e. one XCD's L2.
// ---------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void emi_defect_f64_kernel(DefectArgs a) {
    constexpr int TM = DEF_TM, TN = DEF_TN, BK = DEF_BK, LDK = BK + 2;
    constexpr int RT = TM / 16;             // row tiles per wave
    constexpr int CT = TN / 64;             // col tiles per wave (4 waves along n)
    constexpr int A_PASS = TM * BK / 2 / 256;  // double2 loads per thread
    constexpr int B_PASS = TN * BK / 2 / 256;
    static_assert(TM * BK / 2 % 256 == 0 && TN * BK / 2 % 256 == 0, "staging shape");

    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                        // [2][TM][LDK]
    double* Bs = smem + 2 * TM * LDK;         // [2][TN][LDK]

    const int R = a.R, M = a.M;
    // XCD-aware bijective remap of the linear block id
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int mtiles = (R + TM - 1) / TM;
    const int ntile = bid / mtiles, mtile = bid - ntile * mtiles;
    const int m0 = mtile * TM, n0 = ntile * TN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- accumulators start from the rows already in RES (-h f) -----------
    d4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n = n0 + wid * (16 * CT) + ct * 16 + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = m0 + rt * 16 + kq + 4 * i;
                double v = 0.0;
                if (r < R && n < M) {
                    const int inst = r / a.ns, st = r - inst * a.ns;
                    v = a.RES[((size_t)inst * a.nres + st) * M + n];
                }
                acc[rt][ct][i] = v;
            }
        }

    double2 pa[A_PASS], pb[B_PASS];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx / (BK / 2), c2 = idx % (BK / 2);
            const int r = m0 + row, k = k0 + 2 * c2;
            double2 v = make_double2(0.0, 0.0);
            if (r < R) {
                const double* src = a.X + (size_t)r * M + k;
                if (ALIGNED) {
                    if (k < M) v = *reinterpret_cast<const double2*>(src);
                } else {
                    if (k < M) v.x = src[0];
                    if (k + 1 < M) v.y = src[1];
                }
            }
            pa[p] = v;
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx / (BK / 2), c2 = idx % (BK / 2);
            const int n = n0 + row, k = k0 + 2 * c2;
            double2 v = make_double2(0.0, 0.0);
            if (n < M) {
                const double* src = a.D + (size_t)n * M + k;
                if (ALIGNED) {
                    if (k < M) v = *reinterpret_cast<const double2*>(src);
                } else {
                    if (k < M) v.x = src[0];
                    if (k + 1 < M) v.y = src[1];
                }
            }
            pb[p] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx / (BK / 2), c2 = idx % (BK / 2);
            *reinterpret_cast<double2*>(As + ((size_t)buf * TM + row) * LDK + 2 * c2) = pa[p];
        }
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int idx = tid + 256 * p, row = idx / (BK / 2), c2 = idx % (BK / 2);
            *reinterpret_cast<double2*>(Bs + ((size_t)buf * TN + row) * LDK + 2 * c2) = pb[p];
        }
    };

    const int nkt = (M + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload((kt + 1) * BK);
        const double* Ab = As + (size_t)cur * TM * LDK;
        const double* Bb = Bs + ((size_t)cur * TN + wid * (16 * CT)) * LDK;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double af[RT], bf[CT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) af[rt] = Ab[(rt * 16 + r16) * LDK + ks * 4 + kq];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bf[ct] = Bb[(ct * 16 + r16) * LDK + ks * 4 + kq];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
        }
        if (kt + 1 < nkt) {
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: defect rows out ------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n = n0 + wid * (16 * CT) + ct * 16 + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = m0 + rt * 16 + kq + 4 * i;
                if (r < R && n < M) {
                    const int inst = r / a.ns, st = r - inst * a.ns;
                    a.RES[((size_t)inst * a.nres + st) * M + n] = acc[rt][ct][i];
                }
            }
        }
}

// ---------------------------------------------------------------------------
// K4 (fp32 arithmetic type): same contraction with plain f32 FMAs and the
// product accumulated in f64 (|D_ij| reaches N(N+1)/4 ~ 4e6 at M=4096, so an
// f32 accumulator loses every digit; SURVEY.md section 7 "hard parts").
// First correct version: one thread per output element, D row and X row
// streamed through LDS tiles.  The MFMA split-precision kernel replaces it.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emi_defect_f32_kernel(DefectArgsF32 a) {
    constexpr int TR = 16, TC = 16, BK = 64;
    __shared__ float As[TR][BK + 1];
    __shared__ float Bs[TC][BK + 1];
    const int R = a.R, M = a.M;
    const int n0 = blockIdx.x * TC, m0 = blockIdx.y * TR;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc = 0.0;
    for (int k0 = 0; k0 < M; k0 += BK) {
        for (int idx = threadIdx.x; idx < TR * BK; idx += 256) {
            const int row = idx / BK, kk = idx % BK;
            const int r = m0 + row, k = k0 + kk;
            As[row][kk] = (r < R && k < M) ? a.X[(size_t)r * M + k] : 0.f;
        }
        for (int idx = threadIdx.x; idx < TC * BK; idx += 256) {
            const int row = idx / BK, kk = idx % BK;
            const int n = n0 + row, k = k0 + kk;
            Bs[row][kk] = (n < M && k < M) ? a.D[(size_t)n * M + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int kk = 0; kk < BK; ++kk) acc += (double)As[ty][kk] * (double)Bs[tx][kk];
        __syncthreads();
    }
    const int r = m0 + ty, n = n0 + tx;
    if (r < R && n < M) {
        const int inst = r / a.ns, st = r - inst * a.ns;
        float* o = a.RES + ((size_t)inst * a.nres + st) * M + n;
        *o = (float)((double)*o + acc);
    }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
template <typename T, class Model, bool DEFROWS>
static hipError_t launch_nodes_model(const NodeArgs<T>& a, bool jac, hipStream_t s) {
    const int M = a.M;
    const bool vec2 = (M % 2 == 0);
    const int per_block = EMI_NODE_THREADS * (vec2 ? 2 : 1);
    dim3 grid((M + per_block - 1) / per_block, a.B), block(EMI_NODE_THREADS);
    if (vec2) {
        if (jac) hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 2, true, DEFROWS>), grid, block, 0, s, a);
        else     hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 2, false, DEFROWS>), grid, block, 0, s, a);
    } else {
        if (jac) hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 1, true, DEFROWS>), grid, block, 0, s, a);
        else     hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 1, false, DEFROWS>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

template <typename T>
hipError_t launch_cost_finish(const T* part, T* cost, int B, int nchunks, T scale, hipStream_t s) {
    hipLaunchKernelGGL((emi_cost_finish_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, s, part, cost, B, nchunks,
                       scale);
    return hipGetLastError();
}
template hipError_t launch_cost_finish<double>(const double*, double*, int, int, double, hipStream_t);
template hipError_t launch_cost_finish<float>(const float*, float*, int, int, float, hipStream_t);

int node_chunks(int M) {
    const int per_block = EMI_NODE_THREADS * ((M % 2 == 0) ? 2 : 1);
    return (M + per_block - 1) / per_block;
}

// node kernel only; the caller launches launch_cost_finish(node_chunks(M)) behind it
template <typename T>
hipError_t launch_nodes(int model, const NodeArgs<T>& a, bool jac, bool defect_rows, hipStream_t s) {
    if (defect_rows) {
        switch (model) {
            case 0: return launch_nodes_model<T, PointMass2D<T>, true>(a, jac, s);
            case 1: return launch_nodes_model<T, Quadrotor2D<T>, true>(a, jac, s);
            case 2: return launch_nodes_model<T, FixedWing12<T>, true>(a, jac, s);
        }
    } else {
        switch (model) {
            case 0: return launch_nodes_model<T, PointMass2D<T>, false>(a, jac, s);
            case 1: return launch_nodes_model<T, Quadrotor2D<T>, false>(a, jac, s);
        }
    }
    return hipErrorInvalidValue;
}
template hipError_t launch_nodes<double>(int, const NodeArgs<double>&, bool, bool, hipStream_t);
template hipError_t launch_nodes<float>(int, const NodeArgs<float>&, bool, bool, hipStream_t);

template <typename T>
hipError_t launch_hess(int model, const HessArgs<T>& a, hipStream_t s) {
    dim3 grid((a.M + EMI_NODE_THREADS - 1) / EMI_NODE_THREADS, a.B), block(EMI_NODE_THREADS);
    switch (model) {
        case 0: hipLaunchKernelGGL((emi_hess_kernel<T, PointMass2D<T>>), grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((emi_hess_kernel<T, Quadrotor2D<T>>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template hipError_t launch_hess<double>(int, const HessArgs<double>&, hipStream_t);
template hipError_t launch_hess<float>(int, const HessArgs<float>&, hipStream_t);

hipError_t launch_defect_f64(const DefectArgs& a, hipStream_t s) {
    const int mtiles = (a.R + DEF_TM - 1) / DEF_TM, ntiles = (a.M + DEF_TN - 1) / DEF_TN;
    const size_t lds = (size_t)2 * (DEF_TM + DEF_TN) * (DEF_BK + 2) * sizeof(double);
    dim3 grid(mtiles * ntiles), block(256);
    if (a.M % 2 == 0) hipLaunchKernelGGL((emi_defect_f64_kernel<true>), grid, block, lds, s, a);
    else              hipLaunchKernelGGL((emi_defect_f64_kernel<false>), grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_defect_f32(const DefectArgsF32& a, hipStream_t s) {
    dim3 grid((a.M + 15) / 16, (a.R + 15) / 16), block(256);
    hipLaunchKernelGGL(emi_defect_f32_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t defect_f64_set_attr() {
    const int lds = 2 * (DEF_TM + DEF_TN) * (DEF_BK + 2) * (int)sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)emi_defect_f64_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)emi_defect_f64_kernel<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace emi
