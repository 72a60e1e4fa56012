// emi_node_kernels.hpp -- the model-templated streaming kernels (K1+K2+K3+K5 node kernel, K3' cost
// finish, KH Hessian blocks).  Instantiated for the built-in models in emi_kernels.hip and, through
// hiprtc, for model structs generated from traced user callbacks (emi_rtc.hip).
#pragma once
#include "emi_args.hpp"

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
// ST: cache policy of the streaming stores.  0 plain; 1 write-through (sc1): the line is not kept in the XCD's L2, so
// the ~1 GB result stream does not evict the operands of the MFMA defect kernel that runs beside it
// (MI355X_MICROARCH.md, stores of each flavour); 2 non-temporal.  1 and 2 exist for 16-byte packs only.
typedef int emi_v4i __attribute__((ext_vector_type(4)));
template <typename T, int VEC, int ST = 0>
EMI_DEV void store_vec(T* __restrict__ p, const T (&r)[VEC]) {
    using P = typename Pack<T, VEC>::type;
    P v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) e[i] = r[i];
    if constexpr (ST == 1 && sizeof(P) == 16) {
        emi_v4i q;
        __builtin_memcpy(&q, &v, 16);
        // the trailing s_nop keeps the compiler's next instruction off the data registers until the store has read them
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
    } else if constexpr (ST == 2 && sizeof(P) == 16) {
        emi_v4i q;
        __builtin_memcpy(&q, &v, 16);
        __builtin_nontemporal_store(q, reinterpret_cast<emi_v4i*>(p));
    } else {
        *reinterpret_cast<P*>(p) = v;
    }
}

// Result rows of one instance (RES or VALS block: wave-uniform base, per-lane element offset).  ST 0 plain, 2 non-temporal: global
// stores through store_vec.  ST 1 (sc1: write-through, the line is DROPPED from the XCD's L2) and 3 (nt sc1): buffer stores issued by
// the compiler (`buffer_store_dwordx4 ... sc1` from __builtin_amdgcn_raw_buffer_store_b128, cache-policy bits sc1 = 16, nt = 2) -- the
// inline-assembly sc1 form above pins every store in program order behind a wait state and was 50 % slower in the one-launch pass.
template <typename T, int VEC, int ST> struct RowStore {
    T* base;
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs;
#endif
    EMI_DEV explicit RowStore(T* b) : base(b) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (ST == 1 || ST == 3) rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, 0x7fffffff, 0x00027000);
#endif
    }
    EMI_DEV void operator()(size_t off, const T (&r)[VEC]) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((ST == 1 || ST == 3) && sizeof(T) * VEC == 16) {
            using P = typename Pack<T, VEC>::type;
            P v;
            T* e = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) e[i] = r[i];
            emi_v4i q;
            __builtin_memcpy(&q, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)(off * sizeof(T)), 0, ST == 1 ? 16 : 18);
            return;
        }
#endif
        store_vec<T, VEC, (ST == 3 ? 2 : ST)>(base + off, r);
    }
};

// VALS entries of the rows traced from constraint callbacks: NPATH * PW (0 for the hand-written models)
template <class Model> EMI_DEV constexpr int emi_traced_partials() {
    if constexpr (Model::NPATH > 0) return Model::NPATH * Model::PW;
    else return 0;
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
// bx / nbx: node chunk of this workgroup and chunks per instance; b: instance.  (A device function so that the
// one-launch pass kernel of emi_symdefect_kernels.hpp can give some of its workgroups this role.)
// DEFATOMIC (with DEFROWS): -h f is ADDED to the defect rows with no-return float atomics instead of stored -- the one-launch fp32 pass
// (emi_defect_f32.hip), where the MFMA role adds D.X to the same, zeroed, rows in no particular order: two contributions per element,
// and 0 + a + b = 0 + b + a exactly, so the result does not depend on which role comes first.
template <typename T, class Model, int VEC, bool JAC, bool DEFROWS, int ST, bool DEFATOMIC = false>
EMI_DEV void emi_nodes_body(const NodeArgs<T>& a, const int bx, const int b, const int nbx) {
    constexpr int NS = Model::NS, NC = Model::NC, NV = Model::NV;
    const int M = a.M;
    // a workgroup of 2 x EMI_NODE_THREADS threads (the one-launch pass with its K range in two halves) runs two of these
    // bodies side by side, one per half, each on its own (bx, b): tix is the thread's index within its half
    const int tix = threadIdx.x % EMI_NODE_THREADS, half = threadIdx.x / EMI_NODE_THREADS;
    const int k0 = (bx * EMI_NODE_THREADS + tix) * VEC;
    const bool active = k0 < M;  // M % VEC == 0 by dispatch, so the pack is whole

    T lsum = T(0);
    if (active) {
        const T* __restrict__ Xb = a.X + (size_t)b * NS * M;
        const T* __restrict__ Ub = a.U + (size_t)b * NC * M;
        T* __restrict__ Rb = a.RES + (size_t)b * a.nres * M;
        T* __restrict__ Vb = JAC ? a.VALS + (size_t)b * a.nvals * M : nullptr;
        const RowStore<T, VEC, ST> stR(Rb), stV(Vb);      // every result row of this instance goes through these

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
            if constexpr (DEFROWS && DEFATOMIC) {
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) unsafeAtomicAdd(Rb + (size_t)i * M + k0 + e, fo[i][e]);
            } else if (DEFROWS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) stR((size_t)i * M + k0, fo[i]);
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
                    stV((size_t)(i * NV + v) * M + k0, Jv[i][v]);
            // cost gradient: behind the dynamics block, two partials per table row and PW per traced row
            const size_t goff = (size_t)(NS * NV + 2 * (a.np - Model::NPATH) + emi_traced_partials<Model>()) * M;
#pragma unroll
            for (int v = 0; v < NV; ++v) stV(goff + (size_t)v * M + k0, gv[v]);
        }
        // ---- K2 path constraints (records are wave-uniform: scalar loads) --
        const int np = a.np - Model::NPATH;      // rows of the record table; the model's own rows follow them
        if (a.np > 0) {
            const int set = a.path_sets > 1 ? b : 0;
            typedef const __attribute__((address_space(4))) T* cptr_t;   // read-only table: scalar loads
            cptr_t rec = (cptr_t)(a.path + (size_t)set * np * EMI_PATH_REC);
            const size_t coff = (size_t)NS * M, jcoff = (size_t)(NS * NV) * M;      // path rows in RES, their partials in VALS
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
                stR(coff + (size_t)j * M + k0, c);
                if (JAC) {
                    stV(jcoff + (size_t)(2 * j) * M + k0, cx);
                    stV(jcoff + (size_t)(2 * j + 1) * M + k0, cy);
                }
            }
            if constexpr (Model::NPATH > 0) {
                // rows traced from the user's constraint callbacks (generated straight-line code): values, then PW
                // partials per row (w.r.t. the variables Model::pvar(0..PW-1)) behind the table rows' pairs
                constexpr int NPM = Model::NPATH, PW = Model::PW;
                T cm[NPM][VEC], cdm[NPM * PW][VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    T ze[NV], ce[NPM], cde[NPM * PW];
#pragma unroll
                    for (int v = 0; v < NV; ++v) ze[v] = z[v][e];
                    Model::path(a.P, ze, tk[e], ce, cde);
#pragma unroll
                    for (int j = 0; j < NPM; ++j) cm[j][e] = ce[j];
#pragma unroll
                    for (int j = 0; j < NPM * PW; ++j) cdm[j][e] = cde[j];
                }
#pragma unroll
                for (int j = 0; j < NPM; ++j) stR(coff + (size_t)(np + j) * M + k0, cm[j]);
                if (JAC) {
#pragma unroll
                    for (int j = 0; j < NPM * PW; ++j) stV(jcoff + (size_t)(2 * np + j) * M + k0, cdm[j]);
                }
            }
        }
    }
    // ---- K3 cost quadrature: wave reduction, then across the block's waves --
    __shared__ T wsum[2][EMI_NODE_THREADS / 64];
    const T ws = wave_sum(lsum);
    const int lane = tix & 63, wid = tix >> 6;
    if (lane == 0) wsum[half][wid] = ws;
    __syncthreads();
    if (tix == 0) {
        T s = T(0);
#pragma unroll
        for (int i = 0; i < EMI_NODE_THREADS / 64; ++i) s += wsum[half][i];
        T* part = a.cost_part + (size_t)b * nbx;
        if (a.cost_ticket == nullptr) {
            part[bx] = s;                                  // emi_cost_finish_kernel sums the partials
        } else {
            // In-kernel finish: the workgroup that draws the last ticket of its instance adds the partials IN CHUNK
            // ORDER (bitwise reproducible).  Partials travel write-through / L1-bypassing (agent-scope atomics,
            // HIP guide Guideline 16 R1: store, drain, then the ticket); the ticket word resets itself.
            __hip_atomic_store(part + bx, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(a.cost_ticket + b, 1u, EMI_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)nbx - 1u) {
                T tot = T(0);
                for (int c = 0; c < nbx; ++c) tot += __hip_atomic_load(part + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.cost[b] = a.sgn * a.h * tot;
                __hip_atomic_store(a.cost_ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <typename T, class Model, int VEC, bool JAC, bool DEFROWS, int ST = 0>
__global__ __launch_bounds__(EMI_NODE_THREADS) void emi_nodes_kernel(NodeArgs<T> a) {
    emi_nodes_body<T, Model, VEC, JAC, DEFROWS, ST>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
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
    const int np = a.np - Model::NPATH;          // rows of the record table; the model's own rows follow them
    if (a.np > 0) {
        const int set = a.path_sets > 1 ? b : 0;
        const T* __restrict__ rec = a.path + (size_t)set * np * EMI_PATH_REC;
        T hxx = T(0), hxy = T(0), hyy = T(0);
        for (int j = 0; j < np; ++j) {
            const T* __restrict__ r = rec + j * EMI_PATH_REC;
            const int kind = (int)r[0];
            const T mu = a.lamC[((size_t)b * a.np + j) * M + k];
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
        if constexpr (Model::NPATH > 0) {      // traced rows: sum_j mu_j c_j,zz straight into the packed triangle
            T mu[Model::NPATH];
#pragma unroll
            for (int j = 0; j < Model::NPATH; ++j) mu[j] = a.lamC[((size_t)b * a.np + np + j) * M + k];
            Model::path_hess(a.P, z, a.node_t[k], mu, H);
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

}  // namespace emi
