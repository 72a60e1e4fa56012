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
    const int ntile = bid / mtiles, mtile = bid - ntile * mtiles;
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

bool fused_supported(int model, int M, int ct) {
    return (model == EMI_MODEL_POINTMASS2D || model == EMI_MODEL_QUADROTOR2D) && (ct == 1 || ct == 2) &&
           M >= 128 * ct && M % (128 * ct) == 0;
}

hipError_t launch_symdefect(int model, const SymDefectArgs& a, hipStream_t s, bool set_attr, int ct) {
    if (model == EMI_MODEL_POINTMASS2D)
        return ct == 1 ? launch_symdefect_model<PointMass2D<double>, 1>(a, s, set_attr)
                       : launch_symdefect_model<PointMass2D<double>, 2>(a, s, set_attr);
    if (model == EMI_MODEL_QUADROTOR2D)
        return ct == 1 ? launch_symdefect_model<Quadrotor2D<double>, 1>(a, s, set_attr)
                       : launch_symdefect_model<Quadrotor2D<double>, 2>(a, s, set_attr);
    return hipErrorInvalidValue;
}

}  // namespace emi
