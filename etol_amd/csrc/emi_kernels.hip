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
#include "emi_node_kernels.hpp"

namespace emi {

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
// K4 (fp64, few instances): defect rows += X . D^T as a skinny product.
// With one or a handful of instances (the single solve of eMI355X::solve, B*ns <= 96 rows) a tiled MFMA kernel
// has 8 workgroups to run and walks K serially; here the 8 MB of D are simply streamed once by every CU:
// one wave per output node n reads row n of D coalesced (64 lanes x 8 B per step), multiplies it with the same
// stretch of every X row and wave-reduces the R sums.  The X rows go through LDS one K tile at a time (48 KB,
// shared by the 8 nodes of a workgroup): read straight from L2 they cost 18 us per launch at one instance of 1024 nodes.
// ---------------------------------------------------------------------------
template <int RC, int KT, int NW>                        // NW nodes per wave, 4 NW per workgroup
__global__ __launch_bounds__(256) void emi_defect_small_f64_kernel(DefectArgs a) {
    __shared__ double xs[RC][KT];                        // the X rows of this chunk, one K tile (48 KB)
    const int M = a.M, R = a.R;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nbase = blockIdx.x * (4 * NW) + wid * NW;
    for (int r0 = 0; r0 < R; r0 += RC) {                // row chunks (one chunk for B*ns <= RC)
        double acc[NW][RC];
#pragma unroll
        for (int nn = 0; nn < NW; ++nn)
#pragma unroll
            for (int r = 0; r < RC; ++r) acc[nn][r] = 0.0;
        for (int k0 = 0; k0 < M; k0 += KT) {
            __syncthreads();                             // the previous tile has been consumed
            for (int idx = threadIdx.x; idx < RC * KT; idx += 256) {
                const int r = idx / KT, j = idx - r * KT;
                xs[r][j] = (r0 + r < R && k0 + j < M) ? a.X[(size_t)(r0 + r) * M + k0 + j] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int nn = 0; nn < NW; ++nn) {
                const int n = nbase + nn;
                if (n >= M) continue;
                const double* __restrict__ Dn = a.D + (size_t)n * M + k0;
                constexpr int NQ = KT / 64 < 8 ? KT / 64 : 8;      // loads of D in flight per lane
#pragma unroll
                for (int s0 = 0; s0 < KT / 64; s0 += NQ) {
                    double d[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int j = (s0 + q) * 64 + lane;
                        d[q] = k0 + j < M ? Dn[j] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int r = 0; r < RC; ++r) acc[nn][r] += d[q] * xs[r][(s0 + q) * 64 + lane];
                }
            }
        }
#pragma unroll
        for (int nn = 0; nn < NW; ++nn) {
            const int n = nbase + nn;
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                double v = acc[nn][r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0 && n < M && r0 + r < R) {
                    const int row = r0 + r, inst = row / a.ns, st = row - inst * a.ns;
                    double* o = a.RES + ((size_t)inst * a.nres + st) * M + n;
                    *o += v;
                }
            }
        }
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
    if constexpr (sizeof(T) == 8 && !DEFROWS) {
        // the kernel that runs beside the MFMA defect kernel: its stores may bypass L2 retention
        if (vec2 && jac && a.store_mode == 1) { hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 2, true, DEFROWS, 1>), grid, block, 0, s, a); return hipGetLastError(); }
        if (vec2 && jac && a.store_mode == 2) { hipLaunchKernelGGL((emi_nodes_kernel<T, Model, 2, true, DEFROWS, 2>), grid, block, 0, s, a); return hipGetLastError(); }
    }
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
            case 2: return launch_nodes_model<T, FixedWing12<T>, false>(a, jac, s);
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
        case 2: hipLaunchKernelGGL((emi_hess_kernel<T, FixedWing12<T>>), grid, block, 0, s, a); break;
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

bool defect_small_supported(int R) { return R <= 96; }
hipError_t launch_defect_small_f64(const DefectArgs& a, hipStream_t s) {
    dim3 g1((a.M + 3) / 4), g2((a.M + 7) / 8), block(256);
    if (a.R <= 6) hipLaunchKernelGGL((emi_defect_small_f64_kernel<6, 1024, 1>), g1, block, 0, s, a);
    else if (a.R <= 12) hipLaunchKernelGGL((emi_defect_small_f64_kernel<12, 512, 1>), g1, block, 0, s, a);
    else hipLaunchKernelGGL((emi_defect_small_f64_kernel<24, 256, 2>), g2, block, 0, s, a);
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
