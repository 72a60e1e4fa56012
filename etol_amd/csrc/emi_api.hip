// emi_api.hip -- device side of the C ABI declared in include/emi355x.h.
//
// A context owns: the HIP stream, the mesh constants on the device (w, node
// times, diag(D), D), the path/track tables, and scratch for the cost
// partials.  Trajectory and result arrays belong to the caller (device
// pointers), except in the *_host forms which stage through context-owned
// buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "emi355x.h"
#include "emi_kernels.hpp"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ProfEvents {
    hipEvent_t e[3];        // sequential path: e0 | node | e1 | defect | e2 ; overlapped: e0 fork, e1 join
    hipEvent_t k[4];        // overlapped path: MFMA kernel k0..k1 (main stream), node kernel k2..k3 (stream 2)
    bool has_node, has_defect, fused;
    int level;              // 1: every bracket; 2: the defect (MFMA) kernel only; 3: the node kernel only
};

}  // namespace

struct emi_ctx_s {
    int device = 0;
    bool f32 = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;          // the node kernel runs here while the MFMA defect kernel runs on `stream`
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    // "cu_split" option: the two kernels of the overlapped pass on disjoint CU sets (CU-masked streams)
    int cu_split = 0;                       // CUs given to the MFMA defect kernel; 0 = both kernels share every CU
    hipStream_t s_mfma = nullptr, s_node = nullptr;
    std::string err;
    std::string last_defect_kernel;

    // mesh
    int M = 0;
    double t0 = 0, tf = 0;
    DevBuf d_w, d_t, d_Ddiag, d_D, d_De, d_Do;
    bool symmetric = false;   // D is exactly centro-antisymmetric and M is even: De/Do are valid
    bool points_only = false; // emi_set_mesh(D = NULL): abscissae without a differentiation matrix
    bool allow_fused = true;      // "overlap" option: even/odd MFMA defect kernel || node kernel on two streams
    int sym_ct = 0;               // MFMA kernel variant (emi_symdefect.hip): 0 = chosen from the batch, 3 = LDS-DMA ring, 5..8 state-split ring, 1/2 = register-staged
    int sym_order = 1;
    int sym_ablate = 0;
    int small_rows = 24;          // "small_rows": up to this many rows B*ns the skinny defect kernel replaces the MFMA ones
                                  // (measured at 1024 nodes, 6 states: B = 1 / 2 / 4: 21 / 29 / 53 us per pass against 83 us)
    int overlap_mode = 0;         // 0: by batch size (3 below 192 tiles, else 2); 1: one stream, back to back; 2: two streams; 3: one launch
    int node_store = -1;          // cache policy of the node kernel's stores on the overlapped path: 0 plain, 1 sc1, 2 nt, -1 by size
    unsigned fused_attr_mask = 0;
    std::vector<double> h_tau, h_w;
    // model
    int model = -1, ns = 0, nc = 0, maximize = 0;
    double params[EMI_MAX_PARAMS] = {0};
    emi::KktWorkspace* kkt = nullptr;   // Newton-step workspace (emi_kkt_factor)
    int kkt_method = 1;                 // 1: Schur complement + Cholesky (falls back to 0 if not quasi-definite); 0: LU of K
    emi::RtcModel* rtc = nullptr;   // model == EMI_MODEL_SOURCE: code object compiled at emi_set_model_source
    // batch / path
    int B = 0;
    int np = 0, path_sets = 0, px = 0, py = 1;
    int np_model = 0;           // path rows computed by the model itself (emi_set_model_source npath)
    std::vector<int> pvars;     // node variables those rows depend on (PW of them): PW partials per traced row in VALS
    DevBuf d_path;
    int ntracks = 0, track_sets = 0;
    DevBuf d_trkx, d_trky;
    DevBuf d_cost_part;
    DevBuf d_slab;              // partial sums of a split-K defect launch
    DevBuf d_tile_ticket;       // ... and the tickets of its in-kernel combine (zero between launches)
    DevBuf d_cost_part2;        // cost partials of the values-only pre-kernel of the overlapped f32 pass (discarded)
    DevBuf d_ticket;            // [B] arrival counters of the in-kernel COST finish (zeroed once, self-resetting)
    bool cost_in_kernel = true; // "cost_in_kernel": the node kernel of the overlapped pass finishes COST itself (ticket), no emi_cost_finish_kernel
    int sym_nst = 3;            // "sym_nst": ring stages of the one-launch pass (3 or 4)
    int sym_hs = 0;             // "sym_hs": 2: K range of a tile in two halves inside the workgroup (512 threads); 1: undivided; 0: by batch size
    int sym_ctc = 0;            // "sym_ctc": 64-column sub-tiles per MFMA workgroup of the one-launch pass (1 or 2; 0: by batch size, plan_pass)
    int sym_bk = 0;             // "sym_bk": depth of a K tile of the one-launch pass (8 or 16; 0: by batch size, plan_pass)
    // delayed values (emi_set_delays): x_horizon - 1 delayed copies of every state and u_horizon of every control, appended to the
    // controls the node functions see: nc = nc_free + nch; W[d] = interpolation matrix of delay (d + 1) dt on this mesh
    int xh = 0, uh = 0, nch = 0;
    double delay_dt = 0.0;
    bool delay_dirty = true;    // W must be rebuilt (mesh or delays changed)
    DevBuf d_W;                 // [max(xh - 1, uh)][M][M]
    DevBuf d_uext;              // [B][nc][M]: the caller's controls, then the delayed values
    int f32_ring_wgs = 2;       // "f32_ring_wgs": workgroups of the fp32 ring kernel per CU (1: room for a node kernel's waves beside it, overlap_mode 2)
    int f32_ring = 1;           // "f32_ring": the fp32 MFMA defect kernel in its LDS-DMA ring form (0: register-staged operands, the round-2 form)
    int f32_one_launch = 0;     // "f32_one_launch": fp32 contexts take the one-launch pass (emi_pass_f32_kernel) by themselves where it applies.
                                // Off: measured at config 5 (256 instances, 4096 nodes) 1.12 - 1.26 ms per pass in every block order against
                                // 1.04 ms for the node kernel followed by the MFMA kernel (profiles/r03_notes.md section 6)
    int slice = 0;              // "slice" option: > 0: batches above 2 * slice instances are evaluated in pieces of this many; 0: one launch (see emi_eval_dev)
    int slice_first = 0;        // first instance of the slice emi_eval_dev is working on (per-instance tables are offset by it)
    int sym_ksplit = 0;         // "sym_ksplit" option: K slices per tile of the state-split ring kernel (0: by batch size)
    int sym_cpart = 0;          // "sym_cpart" option: column partitions of the tile order (0: by mesh size, -1: plain order, 1/2/4/8)
    int sym_gblk = 0, sym_cx = 0;   // "sym_gblk" / "sym_cx" options: grouped tile order, instance groups per super-block (0: off) and column tiles per block (0: 2)
    int pass_order = -1;        // "pass_order" option: one-launch pass, MFMA workgroups of an XCD first (1), interleaved with the node
                                // workgroups (0), or by batch size (-1: first for small batches)
    int sym_combine = 1;        // "sym_combine" option: 1 slices combined in-kernel by ticket, 0 by emi_symdefect_combine_kernel
    // host-form staging
    DevBuf s_X, s_U, s_RES, s_VALS, s_COST, s_LF, s_LC, s_H;
    // measurement
    hipEvent_t t_start = nullptr, t_stop = nullptr;
    int profile = 0;          // emi_profile_enable level (0 off)
    std::vector<ProfEvents> prof;
    size_t prof_used = 0;
    bool attr_set = false;
};

namespace {

int fail(emi_ctx_t c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                              \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail((c), EMI_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                        hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

int ensure(emi_ctx_t c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return EMI_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) return EMI_OK;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return EMI_OK;
}

// upload a host double array in the context's real type
int upload_real(emi_ctx_t c, DevBuf& b, const double* src, size_t n) {
    const size_t rb = c->f32 ? 4 : 8;
    int st = ensure(c, b, n * rb);
    if (st) return st;
    if (n == 0) return EMI_OK;
    if (c->f32) {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
        HIP_TRY(c, hipMemcpyAsync(b.p, tmp.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(b.p, src, n * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return EMI_OK;
}

int download_real(emi_ctx_t c, double* dst, const void* dsrc, size_t n) {
    if (!dst || n == 0) return EMI_OK;
    if (c->f32) {
        std::vector<float> tmp(n);
        HIP_TRY(c, hipMemcpyAsync(tmp.data(), dsrc, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; ++i) dst[i] = tmp[i];
    } else {
        HIP_TRY(c, hipMemcpyAsync(dst, dsrc, n * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return EMI_OK;
}

// the even/odd MFMA defect kernel beside the node kernel: needs an exactly centro-antisymmetric D
bool overlapped_path(emi_ctx_t c) {
    if (c->f32 || !c->allow_fused || !c->symmetric || c->M <= 0 || c->model < 0) return false;
    if (c->model == EMI_MODEL_SOURCE) return emi::rtc_has_symdefect(c->rtc) && c->M % 128 == 0;
    return emi::fused_supported(c->model, c->M, c->sym_ct);
}

// the one-launch pass in its small-batch form (SW = 1, plain stores) is available to this context by the default dispatch
bool pass_takes_small_batches(emi_ctx_t c) {
    if (!overlapped_path(c) || (c->overlap_mode != 0 && c->overlap_mode != 3) || c->M % 128 != 0) return false;
    if (c->rtc) return emi::rtc_pass_supported(c->rtc, c->B, c->M, 1, 1, 0);
    const emi::SymPlan p = emi::plan_symdefect(c->ns, c->B, c->M, 7, 1, c->sym_cpart, c->sym_gblk, c->sym_cx);
    return emi::pass_supported(c->model, c->ns, c->B, c->M, p);
}

int np_total(emi_ctx_t c) { return c->np + c->np_model; }     // rows of the record table, then the model's own (traced) rows
int nvals_of(emi_ctx_t c) { return c->ns * (c->ns + c->nc) + 2 * c->np + c->np_model * (int)c->pvars.size() + (c->ns + c->nc); }
int nres_of(emi_ctx_t c) { return c->ns + np_total(c); }
int nhess_of(emi_ctx_t c) { const int nv = c->ns + c->nc; return nv * (nv + 1) / 2; }

int ready(emi_ctx_t c) {
    if (!c) return EMI_ERR_ARG;
    if (c->M <= 0) return fail(c, EMI_ERR_STATE, "emi_set_mesh has not been called");
    if (c->model < 0) return fail(c, EMI_ERR_STATE, "emi_set_model has not been called");
    if (c->B <= 0) return fail(c, EMI_ERR_STATE, "emi_set_batch has not been called");
    if (c->np > 0 && c->path_sets != 1 && c->path_sets != c->B)
        return fail(c, EMI_ERR_STATE, "path table has %d sets, batch is %d", c->path_sets, c->B);
    if (c->ntracks > 0 && c->track_sets != 1 && c->track_sets != c->B)
        return fail(c, EMI_ERR_STATE, "track table has %d sets, batch is %d", c->track_sets, c->B);
    return EMI_OK;
}

// Result-store flavour of the node role for a launch of B instances ("node_store" forces it).  Non-temporal once a pass writes
// about what the Infinity Cache holds (256 MB; RES + VALS above 230 MiB): measured on the one-launch pass, M = 1024
// (profiles/r03_mid_sweep.json): 256 instances (244 MiB) 0.0581 ms against 0.0720 with plain stores, 320: 0.0749 / 0.0973,
// 384: 0.0879 / 0.1050; 224 instances (214 MiB): plain 0.0557 / nt 0.0606, 128: 0.0338 / 0.0354.  (Round 2 switched at 300 MB of
// VALS, i.e. above 384 instances: the 256 .. 384 band ran 20 % slow.)
// Below that: write-through (sc1) stores for the built-in fp64 models since the end of round 4 -- plain stores leave a small pass's results
// dirty in L2 for the write-back at the end of the kernel, write-through streams them out while the kernel runs.  One box, ms per pass plain /
// sc1 (profiles/r04_mid_sweep_sc1_stores.jsonl): 1 instance 0.0125 / 0.0105, 16: 0.0149 / 0.0128, 32: 0.0176 / 0.0172, 64: 0.0209 / 0.0215,
// 80: 0.0233 / 0.0221, 112: 0.0290 / 0.0283, 128: 0.0291 / 0.0286, 144: 0.0305 / 0.0296, 192: 0.0434 / 0.0430, 224: 0.0532 / 0.0505.
// (Run-time compiled models hold a plain and a non-temporal instantiation only.)
int store_mode_for(emi_ctx_t c, int B) {
    if (c->node_store >= 0) return c->node_store;
    if ((size_t)B * (nvals_of(c) + nres_of(c)) * c->M * (c->f32 ? 4 : 8) > ((size_t)230 << 20)) return 2;
    // (17 .. 32 tiles -- 33 .. 64 instances at 1024 nodes, the two-slice band -- are the one place where plain stores stay ahead: 48 instances
    // 0.0197 / 0.0203, 64: 0.0209 / 0.0215)
    const int tiles16 = ((B + 15) / 16) * (c->M / 128);
    if (tiles16 > 16 && tiles16 <= 32) return 0;
    return (!c->rtc && !c->f32) ? 1 : 0;
}

// Everything the default dispatch decides about ONE launch of the evaluation pass as emi_pass_f64_kernel, in one place:
// plan_pass() is what eval_dev_slice launches by and what emi_plan_pass reports (tests and tools read the policy from the
// library instead of restating it).
struct PassPlan {
    bool one_launch = false;    // the pass goes out as ONE launch (MFMA-role + node-role workgroups)
    emi::SymPlan sym;           // MFMA role: states per workgroup, K slices per tile, ring stages, tile order
    int tiles16 = 0;            // 16-instance x 128-node tiles of the launch (what the thresholds below are written in)
    int store_mode = 0;         // node role: 0 plain, 1 sc1, 2 non-temporal, 3 nt sc1
    int mfma_first = 0;         // block order (pass_role_of): 1 MFMA workgroups first, 0 evenly interleaved, >= 100: at that % of the even density
};

// The pass as ONE launch: MFMA-role and node-role workgroups in one grid, COST finished in-kernel -- since round 3 at EVERY
// batch size (round 2: two streams between 384 and 767 instances, which ran 20 - 25 % under the rest).  One box, interleaved
// rounds, M = 1024, ms per pass, best one-launch form against the round-2 choice (profiles/r03_mid_sweep.json): 256: 0.0581 /
// 0.0731, 320: 0.0749 / 0.0927, 384: 0.0879 / 0.1177 (two streams), 448: 0.1035 / 0.1038, 512: 0.1188 / 0.1162, 576: 0.1332 /
// 0.1777, 640: 0.146 / 0.155, 704: 0.160 / 0.181.
//   * SW (states per MFMA workgroup): 1 below 128 sixteen-instance x 128-node tiles (more workgroups than CUs), else 2;
//   * K slices per tile ("sym_ksplit"; partial sums combined in-kernel by ticket, in slice order).  By itself only where the MFMA
//     role has fewer workgroups than the chip has places for them, i.e. where a pass waits for one 64-tile dependency chain per
//     workgroup: 4 slices while that keeps the role within 256 workgroups, 2 within 512.  One box, M = 1024, ms per pass
//     unsplit / 2 / 4 slices (tools/mid_sweep.py, profiles/r03_notes.md section 7): B = 8: 0.0204 / 0.0148 / 0.0140,
//     16: 0.0235 / 0.0179 / 0.0155, 32: 0.0242 / 0.0190 / 0.0192, 64: 0.0257 / 0.0218 / 0.0261, 80: 0.0263 / 0.0249 / 0.0312,
//     96: 0.0288 / 0.0290 / 0.0407, 128: 0.0333 / 0.0387 / 0.0496 (from ~500 workgroups the split loses: three and more MFMA
//     waves per SIMD share the matrix pipe and the node role starts behind them);
//   * block order (pass_role_of): MFMA workgroups first below 208 tiles (their 64-tile dependency chains start at once, the
//     streaming workgroups fill in behind; since the end of round 4 only below 144 tiles: 1.5 x the even density from there, see
//     deep_band below), at 1.25 x the even density up to 384 tiles, at 1.1 x up to 768, evenly interleaved
//     from there (B >= 768, where "first" would hold the node role back: 0.288 against 0.222 at 1024).  (Round 2 measured
//     "first" against "interleaved" WITH PLAIN STORES at 256 instances and found interleaved ahead, 0.0727 / 0.0748; with
//     non-temporal stores "first" wins up to 448 instances: 256: 0.0581 against 0.0756 interleaved.  End of round 3, one box,
//     e9 node-evals/s at first / 1.25 x / 1.5 x / even: 448 instances 4.21 / 4.41 / 4.22 / 4.31, 512: 3.60 / 4.32 / 4.38 / 4.22,
//     640: 3.76 / 4.48 / 4.37 / 4.37, 704: 3.89 / 4.63 / 4.45 / 4.50; ms per pass at even / 1.1 x / 1.25 x: 768: 0.1747 /
//     0.1696 / 0.1760, 896: 0.1964 / 0.1959 / 0.2022, 1024: 0.2230 / 0.2200 / 0.2304, 1536: 0.3215 / 0.3219 / 0.3411,
//     2048: 0.4239 / 0.4227 / 0.4415);
//   * tile order: grouped (an XCD's MFMA tiles and node workgroups walk the same instance groups together) for launches of more
//     than 2048 instances in whole super-blocks, else column partitions by mesh size (plan_symdefect);
//   * stores: store_mode_for (non-temporal from about 256 instances).
// Every choice can be forced through emi_set_option (sym_ct, sym_ksplit, sym_cpart, sym_gblk, sym_cx, sym_nst, pass_order,
// node_store); a run-time compiled model holds two instantiations of the pass kernel -- SW = 1 with plain stores (small batches)
// and SW = 2 (1 for an odd number of states) with non-temporal stores (large ones) -- and is planned within those.
PassPlan plan_pass(emi_ctx_t c, int B, bool jac) {
    PassPlan p;
    p.tiles16 = ((B + 15) / 16) * (c->M / 128);
    p.store_mode = store_mode_for(c, B);
    const bool auto_mode = c->overlap_mode == 0;
    if (!((c->overlap_mode == 3 || auto_mode) && jac) || c->M % 128 != 0) return p;
    const bool auto_ct = auto_mode && (c->sym_ct == 0 || c->sym_ct == 4);
    const int gblk = (c->sym_gblk == 0 && c->sym_cpart == 0 && B > 2048 && B % 256 == 0) ? 2 : c->sym_gblk;
    const int gblk_first = (auto_ct || c->rtc) ? gblk : c->sym_gblk;
    int ct = c->sym_ct;                                  // 5 / 6 / 7 / 8 = SW NS / 2 / 1 / 3 (plan_symdefect)
    // Round 4: between 64 and 127 tiles (128 .. 255 instances at 1024 nodes: the shard of config 4) two states per workgroup with K
    // tiles of 16 -- half the barriers and counted waits of the MFMA role's dependency chain, which is what such a pass waits for.
    // One box, ms per pass, SW = 1 / 8-deep (the round-3 choice) against SW = 2 / 16-deep (profiles/r04_mid_sweep.jsonl): 128 instances
    // 0.0320 / 0.0300, 192: 0.0482 / 0.0442; at 64 instances the sliced SW = 1 form stays ahead (0.0210 / 0.0269), from 256 the 8-deep
    // SW = 2 form (0.0548 / 0.0617).
    // End of round 4 (profiles/r04_mid_sweep_small_72_120.jsonl, one box, ms per pass): the rule "2 K slices while the MFMA role stays within
    // 512 workgroups" held up to 80 instances, where the finer sweep found 0.0324 ms against 0.0208 at 64 and 0.0277 at 96.  SW = 1 with 2
    // slices / SW = 1 unsplit 8-deep / SW = 1 unsplit 16-deep / SW = 2 unsplit 16-deep: 72 instances 0.0323 / 0.0246 / 0.0231 / 0.0261, 80:
    // 0.0324 / 0.0248 / 0.0234 / 0.0267, 96: 0.0363 / 0.0278 / 0.0266 / 0.0278, 112: 0.0411 / 0.0338 / 0.0335 / 0.0291.  So: above 32 tiles
    // (64 instances) no slices any more; 33 .. 48 tiles one state per workgroup with 16-deep K tiles (deep_small), from 49 tiles two states
    // (deep_mid, which began at 64 tiles).
    const bool deep_base = auto_ct && !c->rtc && c->sym_bk == 0 && c->sym_ksplit == 0 && c->sym_nst == 3;
    // (both for an even number of states above two, where they were measured: the 6-state quadrotor)
    const bool deep_mid = deep_base && c->ns % 2 == 0 && c->ns > 2 && p.tiles16 >= 49 && p.tiles16 < 128;
    const bool deep_small = deep_base && c->ns % 2 == 0 && c->ns > 2 && p.tiles16 >= 33 && p.tiles16 < 49;
    if (c->rtc) ct = (p.store_mode == 2 && emi::rtc_pass_sw_large(c->rtc) == 2) ? 6 : 7;
    else if (auto_ct) ct = (p.tiles16 < 128 && !deep_mid) ? 7 : 6;
    emi::SymPlan plan = emi::plan_symdefect(c->ns, B, c->M, ct, 1, c->sym_cpart, gblk_first, c->sym_cx);
    if (plan.ring1) plan = emi::plan_symdefect(c->ns, B, c->M, 5, 1, c->sym_cpart, c->sym_gblk, c->sym_cx);
    int ks_want = c->sym_ksplit;
    if (ks_want == 0 && auto_ct && !deep_mid && !deep_small) ks_want = plan.tiles * 4 <= 256 ? 4 : (plan.tiles * 2 <= 512 ? 2 : 1);
    if (ks_want > 1) {
        const int ct_now = plan.sw == c->ns ? 5 : (plan.sw == 2 ? 6 : (plan.sw == 3 ? 8 : 7));
        plan = emi::plan_symdefect(c->ns, B, c->M, ct_now, ks_want, c->sym_cpart, c->rtc ? gblk : c->sym_gblk, c->sym_cx);
    } else {
        plan.ks = 1;
    }
    plan.nst = c->rtc ? 3 : c->sym_nst;
    // K tiles of 16 (built-in models, SW 1 or 2, three stages, unsplit): "sym_bk" 16 forces them
    // ... and between 208 and 767 tiles (416 .. 1535 instances, SW = 2 at 1.25 x / 1.1 x the even MFMA density): one box, ms per pass 8- /
    // 16-deep, 448 instances 0.1112 / 0.1018, 512: 0.1226 / 0.1151, 576: 0.1351 / 0.1277, 640: 0.1492 / 0.1431, 768: 0.1675 / 0.1649,
    // 896: 0.1921 / 0.1896, 1024: 0.2164 / 0.2143; not at 256 .. 384 instances (MFMA workgroups first: 320: 0.0752 / 0.0924) nor from 2048
    // (0.4147 / 0.4205; 4096 in the grouped order 1.081 / 1.175)
    // (up to 1024 tiles -- 2048 instances -- since the end of round 4: with the pass kernel's register allocation stated, 8- / 16-deep at 1536
    // instances 0.3204 / 0.3117, 1792: 0.3725 / 0.3611, 2048: 0.4441 / 0.4322, profiles/r04_mid_sweep_1280_2048.jsonl)
    const bool deep_large = auto_ct && !c->rtc && c->sym_bk == 0 && c->sym_nst == 3 && c->pass_order < 0 && p.tiles16 >= 208 && p.tiles16 <= 1024;
    // ... and between 144 and 207 tiles (288 .. 415 instances) TOGETHER with the MFMA workgroups at 1.5 x the even density instead of all
    // of them first: from ~300 instances the role's 3 x tiles workgroups no longer fit beside the node role (64 places per XCD), which then
    // starts a workgroup generation late.  End of round 4, one box, ms per pass, first + 8-deep (the choice until then) / 1.5 x + 16-deep:
    // 288 instances 0.0730 / 0.0699, 320: 0.0859 / 0.0754, 352: 0.1004 / 0.0820, 384: 0.1022 / 0.0875; 272: 0.0644 / 0.0698 (stays),
    // 416 (1.25 x + 16-deep already): 0.0946 / 0.0951 (profiles/r04_mid_sweep_band_288_416.jsonl)
    const bool deep_band = auto_ct && !c->rtc && c->sym_bk == 0 && c->sym_nst == 3 && c->pass_order < 0 && c->ns % 2 == 0 && c->ns > 2 &&
                           p.tiles16 >= 144 && p.tiles16 < 208;
    const int bk_want = c->sym_bk ? c->sym_bk : ((deep_mid || deep_small || deep_large || deep_band) ? 16 : 8);
    if (bk_want == 16 && !c->rtc && plan.ks == 1 && (plan.sw == 1 || plan.sw == 2) && plan.nst == 3) plan.bk = 16;
    // two column sub-tiles per MFMA workgroup ("sym_ctc" 2; built-in models, SW = 2, unsplit, three stages): the plan is made again with
    // the wider tiles (tile counts and tile order change with the column width)
    // By itself for launches of more than 2048 instances (the grouped tile order; inputs beyond the Infinity Cache, where every operand
    // read is an HBM read): one box, ms per pass one / two sub-tiles, 4096 instances 1.181 / 1.101, 16384: 4.407 / 4.044 (3.81e9 -> 4.15e9
    // node-evals/s) with column blocks of one 128-column tile; at 2048 instances 0.4736 / 0.4636, at 1024 and below the narrow form is ahead
    // (0.2222 / 0.2425: the wide workgroups need 60 KB of LDS and 160 registers) (profiles/r04_mid_sweep.jsonl)
    const bool wide_large = auto_ct && c->sym_ctc == 0 && B > 2048 && B % 256 == 0 && c->sym_cpart == 0;
    const int ctc_want = c->sym_ctc ? c->sym_ctc : (wide_large ? 2 : 1);
    if (ctc_want == 2 && !c->rtc && plan.ks == 1 && plan.sw == 2 && plan.nst == 3 && c->M % 256 == 0) {
        const int bk_keep = plan.bk;
        plan = emi::plan_symdefect(c->ns, B, c->M, 6, 1, c->sym_cpart, gblk_first, (wide_large && c->sym_cx == 0) ? 1 : c->sym_cx, bk_keep, 2);
        plan.ks = 1;
        plan.nst = 3;
        plan.bk = bk_keep;
    }
    // the K range in two halves inside the workgroup ("sym_hs" 2; built-in models, SW 1 or 2, unsplit, one sub-tile, three stages)
    const int hs_want = c->sym_hs ? c->sym_hs : 1;
    if (hs_want == 2 && !c->rtc && plan.ks == 1 && plan.ct == 1 && (plan.sw == 1 || plan.sw == 2) && plan.nst == 3 &&
        ((c->M / 2) / plan.bk) % 4 == 0)
        plan.hs = 2;
    p.sym = plan;
    p.mfma_first = c->pass_order >= 0 ? c->pass_order
                                      : (deep_band ? 150 : (p.tiles16 < 208 ? 1 : (p.tiles16 < 384 ? 125 : (p.tiles16 < 768 ? 110 : 0))));
    p.one_launch = c->rtc ? emi::rtc_pass_supported(c->rtc, B, c->M, plan.sw, plan.ks, p.store_mode)
                          : emi::pass_supported(c->model, c->ns, B, c->M, plan);
    return p;
}

// Large batches: the instances one launch of emi_eval_dev's default dispatch takes (0: the whole batch in one).  Round 2 cut
// everything above 2048 instances into 1024-instance launches (the two-stream form drifted apart on long launches); with the pass
// as ONE launch that buys nothing, and inputs of more than ~256 MB no longer stay in the Infinity Cache from one pass to the next,
// which is what really slows a large batch (B = 16384: 3.47e9 node-evals/s sliced or not, profiles/r03_notes.md).  Now: one launch
// over the whole batch in the GROUPED tile order: 4.10e9 /s at 16384 instances, 4.13e9 at 4096.  Pieces remain only where
// something forces them: the "slice" option (> 0: pieces of that many instances once B > 2 slice), the 32-bit operand offsets of
// the MFMA role (X of a launch below 4 GB), and a remainder that is not a multiple of 256 instances (the grouped order wants whole
// super-blocks on every XCD) as a second launch.
int plan_piece(emi_ctx_t c) {
    const long long cap = ((0xFFFFFFFFLL / ((long long)c->ns * c->M * 8)) / 256) * 256;     // instances whose X stays below 4 GB
    int piece = 0;
    if (c->slice > 0) { if (c->B > 2 * c->slice) piece = c->slice; }
    else if (c->B > 2048) piece = (int)std::min<long long>(cap > 0 ? cap : 256, c->B - c->B % 256);
    return piece >= c->B ? 0 : piece;
}

template <typename T>
void fill_node_args(emi_ctx_t c, emi::NodeArgs<T>& a, const void* dX, const void* dU, void* dRES,
                    void* dVALS, void* dCOST) {
    a.X = (const T*)dX;
    a.U = (const T*)dU;
    a.RES = (T*)dRES;
    a.VALS = (T*)dVALS;
    a.cost_part = (T*)c->d_cost_part.p + (size_t)c->slice_first * emi::node_chunks(c->M);
    a.cost = (T*)dCOST;
    a.cost_ticket = nullptr;
    a.w = (const T*)c->d_w.p;
    a.node_t = (const T*)c->d_t.p;
    a.Ddiag = (const T*)c->d_Ddiag.p;
    a.path = (const T*)c->d_path.p + (c->path_sets > 1 ? (size_t)c->slice_first * c->np * EMI_PATH_REC : 0);
    a.track_x = (const T*)c->d_trkx.p + (c->track_sets > 1 ? (size_t)c->slice_first * c->ntracks * c->M : 0);
    a.track_y = (const T*)c->d_trky.p + (c->track_sets > 1 ? (size_t)c->slice_first * c->ntracks * c->M : 0);
    a.M = c->M;
    a.B = c->B;
    a.np = np_total(c);
    a.nres = nres_of(c);
    a.nvals = nvals_of(c);
    a.path_sets = c->path_sets;
    a.track_sets = c->track_sets;
    a.ntracks = c->ntracks;
    a.px = c->px;
    a.py = c->py;
    a.store_mode = store_mode_for(c, c->B);
    a.h = (T)((c->tf - c->t0) / 2.0);
    a.sgn = c->maximize ? T(-1) : T(1);
    for (int i = 0; i < EMI_MAX_PARAMS; ++i) a.P.p[i] = (T)c->params[i];
}

// Delayed values.  Row k of W(delay) holds the Lagrange basis of the LGL nodes at the node coordinate of max(t_k - delay, t0):
// what PSOPT's get_delayed_state / get_delayed_control hand ePSOPT::dae (reference src/ePSOPT/ePSOPT.cpp:231-248) -- the value
// at t - delay of the polynomial that interpolates the variable's node values ("Legendre" collocation: Lagrange interpolation).
// PSOPT 5.0.0 is not in the reference tree; times before t0 are CLAMPED to t0 here (the history of a delayed variable is its
// initial value), which is an assumption of this build, stated in include/emi355x.h and DESIGN.md section 5.
// Barycentric form with the LGL weights lambda_j ~ (-1)^j sqrt(w_j) (w_j = 2 / (N (N+1) P_N(tau_j)^2)).
void delay_matrix(const std::vector<double>& tau, const std::vector<double>& w, double t0, double tf, double delay, double* W) {
    const int M = (int)tau.size();
    std::vector<double> lam(M);
    for (int j = 0; j < M; ++j) lam[j] = ((j & 1) ? -1.0 : 1.0) * std::sqrt(w[j]);
    const double hh = (tf - t0) / 2.0;
    for (int k = 0; k < M; ++k) {
        double ts = t0 + hh * (tau[k] + 1.0) - delay;
        if (ts < t0) ts = t0;
        const double x = (ts - t0) / hh - 1.0;
        double* row = W + (size_t)k * M;
        int hit = -1;
        for (int j = 0; j < M; ++j)
            if (x == tau[j]) hit = j;
        if (ts <= t0) hit = 0;
        if (hit >= 0) {
            for (int j = 0; j < M; ++j) row[j] = j == hit ? 1.0 : 0.0;
            continue;
        }
        double den = 0.0;
        for (int j = 0; j < M; ++j) {
            row[j] = lam[j] / (x - tau[j]);
            den += row[j];
        }
        for (int j = 0; j < M; ++j) row[j] /= den;
    }
}

// dU_free [B][nc - nch][M] -> *dU_ext [B][nc][M] = [U | x(t - dt) .. x(t - (xh-1) dt) | u(t - dt) .. u(t - uh dt)], the delayed
// rows as products with W on the general MFMA defect kernel (rows += Z . W^T onto zeroed rows)
int extend_controls(emi_ctx_t c, const void* dX, const void* dU_free, const void** dU_ext) {
    if (c->f32) return fail(c, EMI_ERR_UNSUPPORTED, "delayed values: f64 contexts only");
    if (c->points_only) return fail(c, EMI_ERR_STATE, "delayed values need a collocation mesh (this context holds a points-only mesh)");
    const int M = c->M, ncf = c->nc - c->nch, nd = std::max(c->xh - 1, c->uh);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->attr_set) {
        HIP_TRY(c, emi::defect_f64_set_attr());
        c->attr_set = true;
    }
    int st;
    if (c->delay_dirty) {
        std::vector<double> W((size_t)nd * M * M);
        for (int d = 0; d < nd; ++d) delay_matrix(c->h_tau, c->h_w, c->t0, c->tf, (d + 1) * c->delay_dt, W.data() + (size_t)d * M * M);
        if ((st = upload_real(c, c->d_W, W.data(), W.size()))) return st;
        HIP_TRY(c, hipStreamSynchronize(c->stream));        // W is a local: the copy must be done before it goes
        c->delay_dirty = false;
    }
    const size_t row = (size_t)M * 8;
    if ((st = ensure(c, c->d_uext, (size_t)c->B * c->nc * row))) return st;
    HIP_TRY(c, hipMemsetAsync(c->d_uext.p, 0, (size_t)c->B * c->nc * row, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(c->d_uext.p, (size_t)c->nc * row, dU_free, (size_t)ncf * row, (size_t)ncf * row, c->B, hipMemcpyDeviceToDevice, c->stream));
    double* base = (double*)c->d_uext.p + (size_t)ncf * M;
    for (int i = 1; i < c->xh; ++i) {       // x(t - i dt): all states against W[i-1]
        emi::DefectArgs a{(const double*)dX, (const double*)c->d_W.p + (size_t)(i - 1) * M * M, base + (size_t)(i - 1) * c->ns * M,
                          c->B * c->ns, M, c->ns, c->nc};
        HIP_TRY(c, emi::launch_defect_f64(a, c->stream));
    }
    base += (size_t)std::max(c->xh - 1, 0) * c->ns * M;
    for (int i = 1; i <= c->uh; ++i) {      // u(t - i dt): the caller's controls against W[i-1]
        emi::DefectArgs a{(const double*)dU_free, (const double*)c->d_W.p + (size_t)(i - 1) * M * M, base + (size_t)(i - 1) * ncf * M,
                          c->B * ncf, M, ncf, c->nc};
        HIP_TRY(c, emi::launch_defect_f64(a, c->stream));
    }
    *dU_ext = c->d_uext.p;
    return EMI_OK;
}

}  // namespace

extern "C" {

int emi_device_count(int* count) {
    if (!count) return EMI_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return EMI_OK;
}

static int create_impl(int device_id, bool f32, emi_ctx_t* out) {
    if (!out) return EMI_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return EMI_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return EMI_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return EMI_ERR_HIP;
    // gfx950 only: the code object holds no other ISA
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return EMI_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return EMI_ERR_HIP;
    emi_ctx_t c = new emi_ctx_s();
    c->device = device_id;
    c->f32 = f32;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return EMI_ERR_HIP;
    }
    c->own_stream = true;
    // (stream2 is created when a two-stream form first asks for it, need_stream2: every stream takes a share of one of the runtime's few
    // hardware queues, and a Monte-Carlo run has a context per host thread)
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return EMI_ERR_HIP;
    }
    if (hipEventCreate(&c->t_start) != hipSuccess || hipEventCreate(&c->t_stop) != hipSuccess) {
        delete c;
        return EMI_ERR_HIP;
    }
    *out = c;
    return EMI_OK;
}

int emi_create(int device_id, emi_ctx_t* out) { return create_impl(device_id, false, out); }
int emi_create_f32(int device_id, emi_ctx_t* out) { return create_impl(device_id, true, out); }

// the second stream of the two-stream forms (node kernel beside the MFMA defect kernel), created on first use
static int need_stream2(emi_ctx_t c) {
    if (c->stream2) return EMI_OK;
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) {
        c->stream2 = nullptr;
        return fail(c, EMI_ERR_HIP, "cannot create the second stream of the two-stream pass");
    }
    return EMI_OK;
}

int emi_destroy(emi_ctx_t c) {
    if (!c) return EMI_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_w, &c->d_t, &c->d_Ddiag, &c->d_D, &c->d_De, &c->d_Do, &c->d_path, &c->d_trkx, &c->d_trky,
                      &c->d_cost_part, &c->d_slab, &c->d_tile_ticket, &c->d_cost_part2, &c->d_ticket, &c->s_X, &c->s_U, &c->s_RES, &c->s_VALS, &c->s_COST,
                      &c->s_LF, &c->s_LC, &c->s_H};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& pe : c->prof) {
        for (int i = 0; i < 3; ++i) (void)hipEventDestroy(pe.e[i]);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(pe.k[i]);
    }
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->s_mfma) { (void)hipStreamSynchronize(c->s_mfma); (void)hipStreamDestroy(c->s_mfma); }
    if (c->s_node) { (void)hipStreamSynchronize(c->s_node); (void)hipStreamDestroy(c->s_node); }
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->t_start) (void)hipEventDestroy(c->t_start);
    if (c->t_stop) (void)hipEventDestroy(c->t_stop);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    emi::rtc_destroy(c->rtc);
    emi::kkt_destroy(c->kkt);
    delete c;
    return EMI_OK;
}

const char* emi_last_error(emi_ctx_t c) { return c ? c->err.c_str() : "null context"; }

int emi_set_stream(emi_ctx_t c, void* s) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (s == nullptr) {
        if (!c->own_stream) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        return EMI_OK;
    }
    if (c->own_stream) { HIP_TRY(c, hipStreamDestroy(c->stream)); c->own_stream = false; }
    c->stream = (hipStream_t)s;
    return EMI_OK;
}

int emi_get_stream(emi_ctx_t c, void** s) {
    if (!c || !s) return EMI_ERR_ARG;
    *s = (void*)c->stream;
    return EMI_OK;
}

int emi_synchronize(emi_ctx_t c) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMI_OK;
}

int emi_set_mesh(emi_ctx_t c, int M, const double* tau, const double* w, const double* D, double t0,
                 double tf) {
    if (!c || M < 2 || !tau || !w) return fail(c, EMI_ERR_ARG, "emi_set_mesh: bad argument");
    if (!D) {
        // points-only mesh: the node functions are evaluated at arbitrary abscissae (the ODE-error estimate between
        // the collocation nodes); there is no differentiation matrix, so EMI_EVAL_DEFECT and the KKT entry points refuse
        HIP_TRY(c, hipSetDevice(c->device));
        const double hh = (tf - t0) / 2.0;
        if (!(tf > t0)) return fail(c, EMI_ERR_ARG, "emi_set_mesh: tf must exceed t0");
        std::vector<double> nt(M), zero(M, 0.0);
        for (int k = 0; k < M; ++k) nt[k] = t0 + hh * (tau[k] + 1.0);
        int st;
        if ((st = upload_real(c, c->d_w, w, M))) return st;
        if ((st = upload_real(c, c->d_t, nt.data(), M))) return st;
        if ((st = upload_real(c, c->d_Ddiag, zero.data(), M))) return st;
        c->symmetric = false;
        c->points_only = true;
        c->delay_dirty = true;
        emi::kkt_mesh_changed(c->kkt);
        c->h_tau.assign(tau, tau + M);
        c->h_w.assign(w, w + M);
        c->M = M; c->t0 = t0; c->tf = tf;
        c->ntracks = 0; c->track_sets = 0;
        if (c->B > 0) {
            const size_t rb = c->f32 ? 4 : 8;
            if ((st = ensure(c, c->d_cost_part, (size_t)c->B * emi::node_chunks(M) * rb))) return st;
        }
        return EMI_OK;
    }
    c->points_only = false;
    if (!(tf > t0)) return fail(c, EMI_ERR_ARG, "emi_set_mesh: tf must exceed t0");
    HIP_TRY(c, hipSetDevice(c->device));
    const double h = (tf - t0) / 2.0;
    std::vector<double> nt(M), dd(M);
    for (int k = 0; k < M; ++k) {
        nt[k] = t0 + h * (tau[k] + 1.0);
        dd[k] = D[(size_t)k * M + k];
    }
    int st;
    if ((st = upload_real(c, c->d_w, w, M))) return st;
    if ((st = upload_real(c, c->d_t, nt.data(), M))) return st;
    if ((st = upload_real(c, c->d_Ddiag, dd.data(), M))) return st;
    if (c->f32) {
        // f32 copy of D whose rows sum to EXACTLY zero in f32 arithmetic terms: off-diagonals rounded,
        // diagonal = -(f64 sum of the rounded off-diagonals), rounded.  The shifted-difference form of
        // the f32 defect kernel (emi_defect_f32.hip) relies on it.
        std::vector<float> Df((size_t)M * M);
        for (int i = 0; i < M; ++i) {
            double rs = 0.0;
            for (int j = 0; j < M; ++j) {
                if (j == i) continue;
                Df[(size_t)i * M + j] = (float)D[(size_t)i * M + j];
                rs += (double)Df[(size_t)i * M + j];
            }
            Df[(size_t)i * M + i] = (float)(-rs);
        }
        if ((st = ensure(c, c->d_D, Df.size() * 4))) return st;
        HIP_TRY(c, hipMemcpyAsync(c->d_D.p, Df.data(), Df.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else if ((st = upload_real(c, c->d_D, D, (size_t)M * M))) {
        return st;
    }
    // even/odd split of D for the fused kernel: valid only for an exactly centro-antisymmetric D
    c->symmetric = false;
    if (M % 2 == 0 && !c->f32) {
        const int N = M - 1, Hh = M / 2;
        bool sym = true;
        for (int i = 0; i < M && sym; ++i)
            for (int j = 0; j < M; ++j)
                if (D[(size_t)i * M + j] != -D[(size_t)(N - i) * M + (N - j)]) { sym = false; break; }
        if (sym) {
            std::vector<double> De((size_t)Hh * Hh), Do((size_t)Hh * Hh);
            for (int i = 0; i < Hh; ++i)
                for (int j = 0; j < Hh; ++j) {
                    const double p = D[(size_t)i * M + j], q = D[(size_t)i * M + (N - j)];
                    De[(size_t)i * Hh + j] = 0.5 * (p + q);
                    Do[(size_t)i * Hh + j] = 0.5 * (p - q);
                }
            if ((st = upload_real(c, c->d_De, De.data(), De.size()))) return st;
            if ((st = upload_real(c, c->d_Do, Do.data(), Do.size()))) return st;
            c->symmetric = true;
        }
    }
    c->h_tau.assign(tau, tau + M);
    c->h_w.assign(w, w + M);
    c->M = M;
    c->t0 = t0;
    c->tf = tf;
    c->delay_dirty = true;      // W(delay) is built for one mesh: [nd][M][M] on these nodes and this horizon
    emi::kkt_mesh_changed(c->kkt);
    // tables sized by M are stale now
    c->ntracks = 0;
    c->track_sets = 0;
    // the per-block cost partials are sized B * node_chunks(M): keep them valid for the batch already set
    if (c->B > 0) {
        const size_t rb = c->f32 ? 4 : 8;
        if ((st = ensure(c, c->d_cost_part, (size_t)c->B * emi::node_chunks(M) * rb))) return st;
    }
    return EMI_OK;
}

int emi_set_model(emi_ctx_t c, int model, const double* params, int nparams, int maximize) {
    if (!c) return EMI_ERR_ARG;
    int ns, nc, np_expected;
    if (emi_model_dims(model, &ns, &nc, &np_expected)) return fail(c, EMI_ERR_ARG, "unknown model %d", model);
    if (nparams != np_expected || (nparams > 0 && !params))
        return fail(c, EMI_ERR_ARG, "model %d takes %d parameters, got %d", model, np_expected, nparams);
    if (c->rtc) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        emi::rtc_destroy(c->rtc);
        c->rtc = nullptr;
    }
    c->model = model;
    c->ns = ns;
    c->nc = nc;
    c->xh = c->uh = c->nch = 0;
    c->np_model = 0;
    c->pvars.clear();
    c->maximize = maximize ? 1 : 0;
    memset(c->params, 0, sizeof c->params);
    for (int i = 0; i < nparams; ++i) c->params[i] = params[i];
    // path rows name states (px, py) of the previous model, which this one may not have
    c->np = 0;
    c->path_sets = 0;
    return EMI_OK;
}

int emi_set_model_source(emi_ctx_t c, const char* struct_name, const char* source, int ns, int nc, int npath,
                         const int* path_vars, int n_path_vars, const double* params, int nparams, int maximize) {
    if (!c) return EMI_ERR_ARG;
    if (npath < 0 || npath > 64) return fail(c, EMI_ERR_ARG, "emi_set_model_source: npath must be in [0, 64]");
    if (npath > 0 && (!path_vars || n_path_vars < 1 || n_path_vars > ns + nc))
        return fail(c, EMI_ERR_ARG, "emi_set_model_source: %d traced rows need the list of variables they depend on", npath);
    for (int q = 0; q < (npath > 0 ? n_path_vars : 0); ++q)
        if (path_vars[q] < 0 || path_vars[q] >= ns + nc || (q > 0 && path_vars[q] <= path_vars[q - 1]))
            return fail(c, EMI_ERR_ARG, "emi_set_model_source: path_vars must be ascending node-variable indices below %d", ns + nc);
    if (nparams < 0 || nparams > EMI_MAX_PARAMS || (nparams > 0 && !params))
        return fail(c, EMI_ERR_ARG, "emi_set_model_source: at most %d parameters", EMI_MAX_PARAMS);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    emi::RtcModel* m = nullptr;
    std::string log;
    const int st = emi::rtc_build(c->f32, struct_name, source, ns, nc, npath, npath > 0 ? n_path_vars : 0, &m, &log);
    if (st) {
        c->err = log;
        return st;
    }
    emi::rtc_destroy(c->rtc);
    c->rtc = m;
    c->model = EMI_MODEL_SOURCE;
    c->ns = ns;
    c->nc = nc;
    c->xh = c->uh = c->nch = 0;     // delayed values belong to the model they were declared for: emi_set_delays again
    c->np_model = npath;
    c->pvars.assign(path_vars, path_vars + (npath > 0 ? n_path_vars : 0));
    c->maximize = maximize ? 1 : 0;
    memset(c->params, 0, sizeof c->params);
    for (int i = 0; i < nparams; ++i) c->params[i] = params[i];
    // path rows name states of the previous model
    c->np = 0;
    c->path_sets = 0;
    return EMI_OK;
}

int emi_check_model_source(const char* struct_name, const char* source, int ns, int nc, int npath, int n_path_vars, int f32,
                           char* log, size_t log_len) {
    std::string l;
    const int st = emi::rtc_check(f32 != 0, struct_name, source, ns, nc, npath, npath > 0 ? n_path_vars : 0, &l);
    if (log && log_len) {
        strncpy(log, l.c_str(), log_len - 1);
        log[log_len - 1] = '\0';
    }
    return st;
}

int emi_set_batch(emi_ctx_t c, int B) {
    if (!c || B < 1) return fail(c, EMI_ERR_ARG, "emi_set_batch: B must be >= 1");
    if (c->M <= 0) return fail(c, EMI_ERR_STATE, "emi_set_mesh must precede emi_set_batch");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t rb = c->f32 ? 4 : 8;
    int st = ensure(c, c->d_cost_part, (size_t)B * emi::node_chunks(c->M) * rb);
    if (st) return st;
    c->B = B;
    return EMI_OK;
}

int emi_set_delays(emi_ctx_t c, int x_horizon, int u_horizon, double dt) {
    if (!c || x_horizon < 0 || u_horizon < 0) return fail(c, EMI_ERR_ARG, "emi_set_delays: horizons must be >= 0");
    if (c->model < 0) return fail(c, EMI_ERR_STATE, "emi_set_model / emi_set_model_source must precede emi_set_delays");
    const int nxd = std::max(x_horizon - 1, 0) * c->ns;
    // nc_free + nxd + uh * nc_free = nc  (the model's control count includes the delayed values)
    const int rest = c->nc - nxd;
    if (nxd + u_horizon == 0) { c->xh = x_horizon; c->uh = 0; c->nch = 0; return EMI_OK; }
    if (!(dt > 0)) return fail(c, EMI_ERR_ARG, "emi_set_delays: dt must be positive");
    if (rest < 1 || rest % (1 + u_horizon) != 0)
        return fail(c, EMI_ERR_ARG, "emi_set_delays: the model has %d controls, which is not nc + %d delayed states + %d x nc delayed controls for any nc >= 1",
                    c->nc, nxd, u_horizon);
    if (c->f32) return fail(c, EMI_ERR_UNSUPPORTED, "emi_set_delays: f64 contexts only");
    c->xh = x_horizon;
    c->uh = u_horizon;
    c->nch = c->nc - rest / (1 + u_horizon);
    c->delay_dt = dt;
    c->delay_dirty = true;
    return EMI_OK;
}

int emi_get_delays(emi_ctx_t c, int* x_horizon, int* u_horizon, int* n_delayed) {
    if (!c) return EMI_ERR_ARG;
    if (x_horizon) *x_horizon = c->xh;
    if (u_horizon) *u_horizon = c->uh;
    if (n_delayed) *n_delayed = c->nch;
    return EMI_OK;
}

int emi_delay_matrix(int M, const double* tau, const double* w, double t0, double tf, double delay, double* W) {
    if (M < 2 || !tau || !w || !W || !(tf > t0) || delay < 0) return EMI_ERR_ARG;
    delay_matrix(std::vector<double>(tau, tau + M), std::vector<double>(w, w + M), t0, tf, delay, W);
    return EMI_OK;
}

int emi_set_path(emi_ctx_t c, int np, int nsets, const double* recs, int px_state, int py_state) {
    if (!c || np < 0) return fail(c, EMI_ERR_ARG, "emi_set_path: bad argument");
    if (c->model < 0) return fail(c, EMI_ERR_STATE, "emi_set_model must precede emi_set_path");
    if (np > 0 && (!recs || nsets < 1)) return fail(c, EMI_ERR_ARG, "emi_set_path: null table");
    if (px_state < 0 || px_state >= c->ns || py_state < 0 || py_state >= c->ns || px_state == py_state)
        return fail(c, EMI_ERR_ARG, "emi_set_path: state indices (%d,%d) out of range", px_state, py_state);
    HIP_TRY(c, hipSetDevice(c->device));
    for (size_t i = 0; i < (size_t)np * nsets; ++i) {
        const int kind = (int)recs[i * EMI_PATH_REC];
        if (kind != EMI_PATH_ELLIPSE && kind != EMI_PATH_DISC && kind != EMI_PATH_TRACK)
            return fail(c, EMI_ERR_ARG, "emi_set_path: record %zu has unknown kind %d", i, kind);
    }
    int st = upload_real(c, c->d_path, recs, (size_t)np * nsets * EMI_PATH_REC);
    if (st) return st;
    c->np = np;
    c->path_sets = np > 0 ? nsets : 0;
    c->px = px_state;
    c->py = py_state;
    return EMI_OK;
}

int emi_set_tracks(emi_ctx_t c, int ntracks, int nsets, const double* xc, const double* yc) {
    if (!c || ntracks < 0) return fail(c, EMI_ERR_ARG, "emi_set_tracks: bad argument");
    if (c->M <= 0) return fail(c, EMI_ERR_STATE, "emi_set_mesh must precede emi_set_tracks");
    if (ntracks > 0 && (!xc || !yc || nsets < 1)) return fail(c, EMI_ERR_ARG, "emi_set_tracks: null table");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)ntracks * nsets * c->M;
    int st;
    if ((st = upload_real(c, c->d_trkx, xc, n))) return st;
    if ((st = upload_real(c, c->d_trky, yc, n))) return st;
    c->ntracks = ntracks;
    c->track_sets = ntracks > 0 ? nsets : 0;
    return EMI_OK;
}

int emi_get_layout(emi_ctx_t c, emi_layout_t* o) {
    if (!c || !o) return EMI_ERR_ARG;
    o->model = c->model;
    o->ns = c->ns;
    o->nc = c->nc;
    o->np = np_total(c);
    o->M = c->M;
    o->B = c->B;
    o->nres = nres_of(c);
    o->nvals = nvals_of(c);
    o->nhess = nhess_of(c);
    o->real_bytes = c->f32 ? 4 : 8;
    o->px = c->px;
    o->py = c->py;
    o->t0 = c->t0;
    o->tf = c->tf;
    return EMI_OK;
}

// Per-instance NLP numbering (DESIGN.md "NLP layout"):
//   variables   z: state i node k -> i*M + k ; control c node k -> (ns+c)*M + k
//   constraints g: defect (i,k) -> i*M + k ; events ns*M + e (e < 2 ns) ;
//                  path (j,k) -> ns*M + 2 ns + j*M + k
int emi_jac_structure(emi_ctx_t c, int* rows, int* cols) {
    if (!c || !rows || !cols) return EMI_ERR_ARG;
    if (c->M <= 0 || c->model < 0) return fail(c, EMI_ERR_STATE, "mesh and model must be set");
    const int M = c->M, ns = c->ns, nv = c->ns + c->nc;
    size_t e = 0;
    for (int i = 0; i < ns; ++i)
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < M; ++k, ++e) { rows[e] = i * M + k; cols[e] = v * M + k; }
    for (int j = 0; j < c->np; ++j)                      // rows of the record table: two partials, (px, py)
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < M; ++k, ++e) {
                rows[e] = ns * M + 2 * ns + j * M + k;
                cols[e] = (s == 0 ? c->px : c->py) * M + k;
            }
    for (int j = 0; j < c->np_model; ++j)                // traced rows: one partial per variable of the model's list
        for (size_t q = 0; q < c->pvars.size(); ++q)
            for (int k = 0; k < M; ++k, ++e) {
                rows[e] = ns * M + 2 * ns + (c->np + j) * M + k;
                cols[e] = c->pvars[q] * M + k;
            }
    for (int v = 0; v < nv; ++v)
        for (int k = 0; k < M; ++k, ++e) { rows[e] = -1; cols[e] = v * M + k; }
    return EMI_OK;
}

int emi_dev_alloc(emi_ctx_t c, size_t bytes, void** dptr) {
    if (!c || !dptr) return EMI_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(dptr, bytes));
    return EMI_OK;
}
int emi_dev_free(emi_ctx_t c, void* dptr) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipFree(dptr));
    return EMI_OK;
}
int emi_h2d(emi_ctx_t c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return EMI_ERR_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMI_OK;
}
int emi_d2h(emi_ctx_t c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return EMI_ERR_ARG;
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMI_OK;
}

static int eval_dev_slice(emi_ctx_t c, const void* dX, const void* dU, void* dRES, void* dVALS, void* dCOST, unsigned flags);

int emi_eval_dev(emi_ctx_t c, const void* dX, const void* dU, void* dRES, void* dVALS, void* dCOST,
                 unsigned flags) {
    int st = ready(c);
    if (st) return st;
    if (c->nch > 0 && dX && dU && (st = extend_controls(c, dX, dU, &dU))) return st;     // delayed values appended to the controls
    // large batches go out in pieces only where something forces them (plan_piece); not while per-kernel profiling is on
    int piece = 0;
    if (!c->profile && !c->f32 && (flags & EMI_EVAL_ALL) == EMI_EVAL_ALL && overlapped_path(c) && dX && dU && dRES && dCOST &&
        (dVALS || (flags & EMI_EVAL_NOJAC)))
        piece = plan_piece(c);
    if (piece > 0) {
        const int SL = piece;
        const int Btot = c->B;
        const size_t rb = 8, M = c->M;
        const size_t nres = nres_of(c), nvals = nvals_of(c);
        for (int first = 0; first < Btot && st == EMI_OK; first += SL) {
            c->B = std::min(SL, Btot - first);
            c->slice_first = first;
            st = eval_dev_slice(c, (const char*)dX + (size_t)first * c->ns * M * rb, (const char*)dU + (size_t)first * c->nc * M * rb,
                                (char*)dRES + (size_t)first * nres * M * rb, dVALS ? (char*)dVALS + (size_t)first * nvals * M * rb : nullptr,
                                (char*)dCOST + (size_t)first * rb, flags);
        }
        c->B = Btot;
        c->slice_first = 0;
        return st;
    }
    return eval_dev_slice(c, dX, dU, dRES, dVALS, dCOST, flags);
}

static int eval_dev_slice(emi_ctx_t c, const void* dX, const void* dU, void* dRES, void* dVALS, void* dCOST, unsigned flags) {
    const bool nodes = flags & EMI_EVAL_NODES, defect = flags & EMI_EVAL_DEFECT;
    const bool jac = !(flags & EMI_EVAL_NOJAC);
    if (!nodes && !defect) return fail(c, EMI_ERR_ARG, "emi_eval: empty flags");
    if (defect && c->points_only) return fail(c, EMI_ERR_STATE, "emi_eval: the mesh has no differentiation matrix (points-only mesh): EMI_EVAL_NODES only");
    if (!dX || !dRES || (nodes && (!dU || !dCOST || (jac && !dVALS))))
        return fail(c, EMI_ERR_ARG, "emi_eval: null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->attr_set) {
        HIP_TRY(c, emi::defect_f64_set_attr());
        c->attr_set = true;
    }
    // a handful of instances: the stand-alone MFMA kernels would have a few workgroups to run and a skinny streaming product wins
    // (21 us at B = 1) -- unless the whole pass can go as ONE launch with its K range sliced, which is faster still (13 - 15 us for
    // any batch up to 16 instances: profiles/r03_notes.md section 7)
    const bool small = defect && !c->f32 && c->B * c->ns <= c->small_rows && emi::defect_small_supported(c->B * c->ns) &&
                       !(nodes && jac && pass_takes_small_batches(c));
    const bool fused = nodes && defect && !small && overlapped_path(c);
    ProfEvents* pe = nullptr;
    if (c->profile) {
        if (c->prof_used == c->prof.size()) {
            ProfEvents n;
            for (int i = 0; i < 3; ++i) HIP_TRY(c, hipEventCreate(&n.e[i]));
            for (int i = 0; i < 4; ++i) HIP_TRY(c, hipEventCreate(&n.k[i]));
            c->prof.push_back(n);
        }
        pe = &c->prof[c->prof_used++];
        pe->has_node = nodes;
        pe->has_defect = defect;
        pe->fused = fused;
        pe->level = c->profile;
        if (pe->level == 1 || (!fused && pe->level == 3)) HIP_TRY(c, hipEventRecord(pe->e[0], c->stream));
    }
    const int plv = pe ? pe->level : 0;
    if (fused) {
        // fork: the two kernels read X,U and write disjoint outputs, so they run concurrently.
        // The MFMA kernel goes first and takes one workgroup per CU (LDS-shaped); the streaming
        // kernel's waves fill the rest of every CU.
        emi::SymDefectArgs sa;
        sa.X = (const double*)dX;
        sa.U = (const double*)dU;
        sa.RES = (double*)dRES;
        sa.node_t = (const double*)c->d_t.p;
        sa.De = (const double*)c->d_De.p;
        sa.Do = (const double*)c->d_Do.p;
        sa.M = c->M;
        sa.B = c->B;
        sa.nres = nres_of(c);
        sa.h = (c->tf - c->t0) / 2.0;
        sa.order = c->sym_order;
        sa.ablate = c->sym_ablate;
        sa.ksplit = 1;
        sa.slab = nullptr;
        sa.tile_ticket = nullptr;
        sa.cpart = sa.cx = 0;
        sa.mfma_first = 0;
        for (int i = 0; i < EMI_MAX_PARAMS; ++i) sa.P.p[i] = c->params[i];
        emi::NodeArgs<double> na;
        fill_node_args(c, na, dX, dU, dRES, dVALS, dCOST);
        const PassPlan pp = plan_pass(c, c->B, jac);
        if (pp.one_launch) {
            const emi::SymPlan& plan = pp.sym;
            sa.mfma_first = pp.mfma_first;
            sa.cpart = plan.cpart;
            sa.cx = plan.cx;
            if (plan.ks > 1) {
                int est = ensure(c, c->d_slab, plan.slab_bytes);
                if (est) return est;
                if (c->d_tile_ticket.bytes < (size_t)plan.tiles * 4) {
                    est = ensure(c, c->d_tile_ticket, (size_t)plan.tiles * 4);
                    if (est) return est;
                    HIP_TRY(c, hipMemsetAsync(c->d_tile_ticket.p, 0, c->d_tile_ticket.bytes, c->stream));
                }
                sa.ksplit = plan.ks;
                sa.slab = (double*)c->d_slab.p;
                sa.tile_ticket = (unsigned*)c->d_tile_ticket.p;
            }
            if (c->d_ticket.bytes < (size_t)c->B * 4) {
                int est = ensure(c, c->d_ticket, (size_t)c->B * 4);
                if (est) return est;
                HIP_TRY(c, hipMemsetAsync(c->d_ticket.p, 0, (size_t)c->B * 4, c->stream));
            }
            na.cost_ticket = (unsigned*)c->d_ticket.p;
            if (plv) HIP_TRY(c, hipEventRecord(pe->k[0], c->stream));
            if (c->rtc) HIP_TRY(c, emi::rtc_launch_pass(c->rtc, sa, na, plan.sw, c->stream));
            else HIP_TRY(c, emi::launch_pass(c->model, sa, na, c->stream, plan));
            if (plv) HIP_TRY(c, hipEventRecord(pe->k[1], c->stream));
            if (pe) pe->level = -1;                 // one bracket: the pass kernel
            c->last_defect_kernel = "emi_pass_f64_kernel<SW=" + std::to_string(plan.sw) + "> (MFMA + node roles, one launch" +
                                    (plan.ks > 1 ? ", " + std::to_string(plan.ks) + " K slices per tile" : "") + ")" +
                                    (plan.bk == 16 ? " [K tiles of 16]" : "") + (plan.ct == 2 ? " [128-column tiles]" : "") + (plan.hs == 2 ? " [K range in two halves per workgroup]" : "");
            return EMI_OK;
        }
        const bool two = c->overlap_mode != 1;
        const bool split = two && c->cu_split > 0;
        if (two && !split) { if (int st = need_stream2(c)) return st; }
        hipStream_t s1 = split ? c->s_mfma : c->stream;
        hipStream_t s2 = split ? c->s_node : (two ? c->stream2 : c->stream);
        const unsigned bit = 1u << c->sym_ct;
        if (two) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_fork, 0));
            if (split) HIP_TRY(c, hipStreamWaitEvent(s1, c->ev_fork, 0));
        }
        if (plv == 1 || plv == 2) HIP_TRY(c, hipEventRecord(pe->k[0], s1));
        if (c->rtc) {
            HIP_TRY(c, emi::rtc_launch_symdefect(c->rtc, sa, s1));
            c->last_defect_kernel = "emi_symdefect_ring_f64_kernel";
        } else {
            const emi::SymPlan plan = emi::plan_symdefect(c->ns, c->B, c->M, c->sym_ct, c->sym_ksplit, c->sym_cpart, c->sym_gblk, c->sym_cx);
            if (plan.slab_bytes) {
                int est = ensure(c, c->d_slab, plan.slab_bytes);
                if (est) return est;
            }
            sa.ksplit = plan.ring1 ? 1 : plan.ks;
            sa.slab = (double*)c->d_slab.p;
            sa.cpart = plan.cpart; sa.cx = plan.cx;
            if (sa.ksplit > 1 && c->sym_combine) {
                if (c->d_tile_ticket.bytes < (size_t)plan.tiles * 4) {
                    int est = ensure(c, c->d_tile_ticket, (size_t)plan.tiles * 4);
                    if (est) return est;
                    HIP_TRY(c, hipMemsetAsync(c->d_tile_ticket.p, 0, c->d_tile_ticket.bytes, s1));
                }
                sa.tile_ticket = (unsigned*)c->d_tile_ticket.p;
            }
            HIP_TRY(c, emi::launch_symdefect(c->model, sa, s1, !(c->fused_attr_mask & bit), c->sym_ct, plan));
            c->fused_attr_mask |= bit;
            c->last_defect_kernel = !plan.ring1 ? "emi_symdefect_ring2_f64_kernel<SW=" + std::to_string(plan.sw) + ">" +
                                                      (plan.ks > 1 ? " x" + std::to_string(plan.ks) + (sa.tile_ticket ? " K slices (in-kernel combine)" : " K slices + emi_symdefect_combine_kernel") : "")
                                                : (c->sym_ct == 1 || c->sym_ct == 2 ? "emi_symdefect_f64_kernel" : "emi_symdefect_ring_f64_kernel");
        }
        if (plv == 1 || plv == 2) HIP_TRY(c, hipEventRecord(pe->k[1], s1));
        // COST is finished inside the node kernel (last workgroup of an instance, by ticket, in chunk order): one launch
        // and one kernel boundary less at the end of every pass (emi_cost_finish_kernel alone was 5 us)
        if (c->cost_in_kernel) {
            if (c->d_ticket.bytes < (size_t)c->B * 4) {
                int est = ensure(c, c->d_ticket, (size_t)c->B * 4);
                if (est) return est;
                HIP_TRY(c, hipMemsetAsync(c->d_ticket.p, 0, (size_t)c->B * 4, s2));
            }
            na.cost_ticket = (unsigned*)c->d_ticket.p;
        }
        if (plv == 1 || plv == 3) HIP_TRY(c, hipEventRecord(pe->k[2], s2));
        if (c->rtc && jac && na.store_mode == 2 && c->M % 2 == 0) HIP_TRY(c, emi::rtc_launch_nodes_nt(c->rtc, na, s2));
        else if (c->rtc) HIP_TRY(c, emi::rtc_launch_nodes<double>(c->rtc, na, jac, false, s2));
        else HIP_TRY(c, emi::launch_nodes<double>(c->model, na, jac, false, s2));
        if (plv == 1 || plv == 3) HIP_TRY(c, hipEventRecord(pe->k[3], s2));
        if (!c->cost_in_kernel)
            HIP_TRY(c, emi::launch_cost_finish<double>(na.cost_part, na.cost, c->B, emi::node_chunks(c->M),
                                                       na.sgn * na.h, s2));
        if (two) {
            HIP_TRY(c, hipEventRecord(c->ev_join, s2));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            if (split) {
                HIP_TRY(c, hipEventRecord(c->ev_join2, s1));
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join2, 0));
            }
        }
        if (plv == 1) {
            HIP_TRY(c, hipEventRecord(pe->e[1], c->stream));
            HIP_TRY(c, hipEventRecord(pe->e[2], c->stream));
        }
        return EMI_OK;
    }
    if (c->f32 && nodes && defect && jac && !c->rtc && c->allow_fused && (c->overlap_mode == 3 || (c->overlap_mode == 0 && c->f32_one_launch)) &&
        emi::pass_f32_supported(c->model, c->B * c->ns, c->M, c->B)) {
        // fp32 contexts (config 5): the pass as ONE launch -- MFMA-role and node-role workgroups in one grid, the defect rows zeroed
        // here and completed by float atomics from both roles (emi_defect_f32.hip), COST finished in-kernel by ticket
        emi::NodeArgs<float> na;
        fill_node_args(c, na, dX, dU, dRES, dVALS, dCOST);
        const size_t rowb = (size_t)c->M * 4;
        HIP_TRY(c, hipMemset2DAsync(dRES, (size_t)nres_of(c) * rowb, 0, (size_t)c->ns * rowb, c->B, c->stream));
        if (c->d_ticket.bytes < (size_t)c->B * 4) {
            int est = ensure(c, c->d_ticket, (size_t)c->B * 4);
            if (est) return est;
            HIP_TRY(c, hipMemsetAsync(c->d_ticket.p, 0, (size_t)c->B * 4, c->stream));
        }
        na.cost_ticket = (unsigned*)c->d_ticket.p;
        emi::DefectArgsF32 da{(const float*)dX, (const float*)c->d_D.p, (float*)dRES, c->B * c->ns, c->M, c->ns, nres_of(c)};
        if (plv) HIP_TRY(c, hipEventRecord(pe->k[0], c->stream));
        HIP_TRY(c, emi::launch_pass_f32(c->model, da, na, c->pass_order >= 0 ? c->pass_order : 0, c->stream));
        if (plv) HIP_TRY(c, hipEventRecord(pe->k[1], c->stream));
        if (pe) { pe->level = -1; pe->fused = true; }
        c->last_defect_kernel = "emi_pass_f32_kernel (MFMA + node roles, one launch)";
        return EMI_OK;
    }
    if (c->f32 && nodes && defect && jac && !c->rtc && c->allow_fused && c->overlap_mode == 2 && emi::defect_f32_mfma_supported(c->M)) {
        // fp32 contexts (config 5), only when asked for ("overlap_mode" 2): the f32 MFMA defect kernel ACCUMULATES onto
        // -h f, so a values-only node kernel writes -h f first and the MFMA kernel follows it on the context's stream, while
        // the full node kernel (Jacobian values, cost; no defect rows) runs beside them on the second stream.  Measured at
        // B = 256, M = 4096: 1.076 ms against 1.082 ms back to back -- both kernels stretch (MFMA 0.93 -> 1.03 ms, node
        // 0.16 -> 0.80 ms), nothing is gained, so the default stays sequential (profiles/r02_notes.md)
        emi::NodeArgs<float> pre, full;
        fill_node_args(c, pre, dX, dU, dRES, nullptr, dCOST);
        fill_node_args(c, full, dX, dU, dRES, dVALS, dCOST);
        int est = ensure(c, c->d_cost_part2, (size_t)c->B * emi::node_chunks(c->M) * 4);
        if (est) return est;
        if (int st2 = need_stream2(c)) return st2;
        pre.cost_part = (float*)c->d_cost_part2.p;      // its cost partials go nowhere
        pre.np = 0;                                      // ... and it leaves the path rows to the full kernel
        // (round 4, "f32_ring_wgs" 1: the ring kernel at one workgroup per CU, which costs it nothing, leaves the node kernel's waves
        // room on every SIMD; the node kernel is then released only once the values-only kernel is through, so that it does not fill the
        // chip before the ring kernel's workgroups arrive)
        if (c->f32_ring_wgs != 1) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        }
        HIP_TRY(c, emi::launch_nodes<float>(c->model, pre, false, true, c->stream));
        if (c->f32_ring_wgs == 1) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        }
        emi::DefectArgsF32 da{(const float*)dX, (const float*)c->d_D.p, (float*)dRES, c->B * c->ns, c->M, c->ns, nres_of(c)};
        if (plv == 1 || plv == 2) HIP_TRY(c, hipEventRecord(pe->k[0], c->stream));
        HIP_TRY(c, emi::launch_defect_f32_mfma(da, c->stream, c->f32_ring, c->f32_ring_wgs));
        if (plv == 1 || plv == 2) HIP_TRY(c, hipEventRecord(pe->k[1], c->stream));
        c->last_defect_kernel = c->f32_ring ? "emi_defect_f32_ring_kernel" : "emi_defect_f32_mfma_kernel";
        if (plv == 1 || plv == 3) HIP_TRY(c, hipEventRecord(pe->k[2], c->stream2));
        HIP_TRY(c, emi::launch_nodes<float>(c->model, full, true, false, c->stream2));
        if (plv == 1 || plv == 3) HIP_TRY(c, hipEventRecord(pe->k[3], c->stream2));
        HIP_TRY(c, emi::launch_cost_finish<float>(full.cost_part, full.cost, c->B, emi::node_chunks(c->M), full.sgn * full.h, c->stream2));
        HIP_TRY(c, hipEventRecord(c->ev_join, c->stream2));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if (pe) pe->fused = true;
        if (plv == 1) {
            HIP_TRY(c, hipEventRecord(pe->e[1], c->stream));
            HIP_TRY(c, hipEventRecord(pe->e[2], c->stream));
        }
        return EMI_OK;
    }
    if (nodes) {
        if (c->f32) {
            emi::NodeArgs<float> a;
            fill_node_args(c, a, dX, dU, dRES, dVALS, dCOST);
            if (c->rtc) HIP_TRY(c, emi::rtc_launch_nodes<float>(c->rtc, a, jac, true, c->stream));
            else HIP_TRY(c, emi::launch_nodes<float>(c->model, a, jac, true, c->stream));
            HIP_TRY(c, emi::launch_cost_finish<float>(a.cost_part, a.cost, c->B, emi::node_chunks(c->M), a.sgn * a.h,
                                                      c->stream));
        } else {
            emi::NodeArgs<double> a;
            fill_node_args(c, a, dX, dU, dRES, dVALS, dCOST);
            if (c->rtc) HIP_TRY(c, emi::rtc_launch_nodes<double>(c->rtc, a, jac, true, c->stream));
            else HIP_TRY(c, emi::launch_nodes<double>(c->model, a, jac, true, c->stream));
            HIP_TRY(c, emi::launch_cost_finish<double>(a.cost_part, a.cost, c->B, emi::node_chunks(c->M), a.sgn * a.h,
                                                       c->stream));
        }
    }
    if (plv) HIP_TRY(c, hipEventRecord(pe->e[1], c->stream));
    if (defect) {
        if (c->f32) {
            emi::DefectArgsF32 a{(const float*)dX, (const float*)c->d_D.p, (float*)dRES, c->B * c->ns,
                                 c->M, c->ns, nres_of(c)};
            if (emi::defect_f32_mfma_supported(c->M) && c->allow_fused) { HIP_TRY(c, emi::launch_defect_f32_mfma(a, c->stream, c->f32_ring, c->f32_ring_wgs)); c->last_defect_kernel = c->f32_ring ? "emi_defect_f32_ring_kernel" : "emi_defect_f32_mfma_kernel"; }
            else { HIP_TRY(c, emi::launch_defect_f32(a, c->stream)); c->last_defect_kernel = "emi_defect_f32_kernel"; }
        } else {
            emi::DefectArgs a{(const double*)dX, (const double*)c->d_D.p, (double*)dRES, c->B * c->ns,
                              c->M, c->ns, nres_of(c)};
            if (small) { HIP_TRY(c, emi::launch_defect_small_f64(a, c->stream)); c->last_defect_kernel = "emi_defect_small_f64_kernel"; }
            else { HIP_TRY(c, emi::launch_defect_f64(a, c->stream)); c->last_defect_kernel = "emi_defect_f64_kernel"; }
        }
    }
    if (plv == 1 || plv == 2) HIP_TRY(c, hipEventRecord(pe->e[2], c->stream));
    return EMI_OK;
}

int emi_eval_host(emi_ctx_t c, const double* X, const double* U, double* RES, double* VALS,
                  double* COST, unsigned flags) {
    int st = ready(c);
    if (st) return st;
    if (!X || !U) return fail(c, EMI_ERR_ARG, "emi_eval_host: null input");
    const size_t rb = c->f32 ? 4 : 8;
    const size_t nX = (size_t)c->B * c->ns * c->M, nU = (size_t)c->B * (c->nc - c->nch) * c->M;
    const size_t nR = (size_t)c->B * nres_of(c) * c->M, nV = (size_t)c->B * nvals_of(c) * c->M;
    if ((st = upload_real(c, c->s_X, X, nX))) return st;
    if ((st = upload_real(c, c->s_U, U, nU))) return st;
    if ((st = ensure(c, c->s_RES, nR * rb))) return st;
    if ((st = ensure(c, c->s_VALS, nV * rb))) return st;
    if ((st = ensure(c, c->s_COST, (size_t)c->B * rb))) return st;
    if (!(flags & EMI_EVAL_NODES)) {
        // accumulate-only form: the caller's RES is the starting value
        if (!RES) return fail(c, EMI_ERR_ARG, "emi_eval_host: defect-only needs RES in/out");
        if ((st = upload_real(c, c->s_RES, RES, nR))) return st;
    }
    if ((st = emi_eval_dev(c, c->s_X.p, c->s_U.p, c->s_RES.p, c->s_VALS.p, c->s_COST.p, flags))) return st;
    if ((st = download_real(c, RES, c->s_RES.p, nR))) return st;
    if (!(flags & EMI_EVAL_NOJAC) && (flags & EMI_EVAL_NODES))
        if ((st = download_real(c, VALS, c->s_VALS.p, nV))) return st;
    if (flags & EMI_EVAL_NODES)
        if ((st = download_real(c, COST, c->s_COST.p, c->B))) return st;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EMI_OK;
}

int emi_hess_dev(emi_ctx_t c, const void* dX, const void* dU, const void* dLamF, const void* dLamC,
                 double sigma, void* dH) {
    int st = ready(c);
    if (st) return st;
    if (!dX || !dU || !dLamF || !dH || (np_total(c) > 0 && !dLamC))
        return fail(c, EMI_ERR_ARG, "emi_hess: null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->nch > 0 && (st = extend_controls(c, dX, dU, &dU))) return st;
    auto fill = [&](auto& a) {
        using T = typename std::remove_reference<decltype(a.h)>::type;
        a.X = (const T*)dX; a.U = (const T*)dU; a.lamF = (const T*)dLamF; a.lamC = (const T*)dLamC;
        a.H = (T*)dH; a.w = (const T*)c->d_w.p; a.node_t = (const T*)c->d_t.p;
        a.path = (const T*)c->d_path.p;
        a.M = c->M; a.B = c->B; a.np = np_total(c); a.path_sets = c->path_sets; a.px = c->px; a.py = c->py;
        a.h = (T)((c->tf - c->t0) / 2.0); a.sgn = c->maximize ? T(-1) : T(1); a.sigma = (T)sigma;
        for (int i = 0; i < EMI_MAX_PARAMS; ++i) a.P.p[i] = (T)c->params[i];
    };
    if (c->f32) {
        emi::HessArgs<float> a;
        fill(a);
        if (c->rtc) HIP_TRY(c, emi::rtc_launch_hess<float>(c->rtc, a, c->stream));
        else HIP_TRY(c, emi::launch_hess<float>(c->model, a, c->stream));
    } else {
        emi::HessArgs<double> a;
        fill(a);
        if (c->rtc) HIP_TRY(c, emi::rtc_launch_hess<double>(c->rtc, a, c->stream));
        else HIP_TRY(c, emi::launch_hess<double>(c->model, a, c->stream));
    }
    return EMI_OK;
}

int emi_hess_host(emi_ctx_t c, const double* X, const double* U, const double* LamF,
                  const double* LamC, double sigma, double* H) {
    int st = ready(c);
    if (st) return st;
    if (!X || !U || !LamF || !H || (np_total(c) > 0 && !LamC)) return fail(c, EMI_ERR_ARG, "emi_hess_host: null pointer");
    const size_t rb = c->f32 ? 4 : 8;
    const size_t nX = (size_t)c->B * c->ns * c->M, nU = (size_t)c->B * (c->nc - c->nch) * c->M;
    const size_t nC = (size_t)c->B * np_total(c) * c->M, nH = (size_t)c->B * nhess_of(c) * c->M;
    if ((st = upload_real(c, c->s_X, X, nX))) return st;
    if ((st = upload_real(c, c->s_U, U, nU))) return st;
    if ((st = upload_real(c, c->s_LF, LamF, nX))) return st;
    if (nC && (st = upload_real(c, c->s_LC, LamC, nC))) return st;
    if ((st = ensure(c, c->s_H, nH * rb))) return st;
    if ((st = emi_hess_dev(c, c->s_X.p, c->s_U.p, c->s_LF.p, c->s_LC.p, sigma, c->s_H.p))) return st;
    return download_real(c, H, c->s_H.p, nH);
}

int emi_kkt_factor(emi_ctx_t c, const double* Qblk, const double* Jblk, const unsigned char* fixed, double dc,
                   int* info) {
    if (!c) return EMI_ERR_ARG;
    if (c->M <= 0 || c->model < 0) return fail(c, EMI_ERR_STATE, "emi_kkt_factor: mesh and model must be set");
    if (c->f32) return fail(c, EMI_ERR_UNSUPPORTED, "emi_kkt_factor: f64 contexts only");
    if (c->points_only) return fail(c, EMI_ERR_STATE, "emi_kkt_factor: the mesh has no differentiation matrix");
    if (!Qblk || !Jblk || !fixed || !info || !(dc >= 0.0)) return fail(c, EMI_ERR_ARG, "emi_kkt_factor: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    std::string err;
    const int st = emi::kkt_factor(&c->kkt, c->stream, (const double*)c->d_D.p, c->M, c->ns, c->ns + c->nc, Qblk, Jblk,
                                   fixed, dc, c->kkt_method, info, &err);
    if (st) c->err = err;
    return st;
}

int emi_kkt_lowrank(emi_ctx_t c, int r, const int* node, const double* vec, const double* delta, int* exact) {
    if (!c || r < 0 || !exact || (r > 0 && (!node || !vec || !delta))) return fail(c, EMI_ERR_ARG, "emi_kkt_lowrank: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int a = 0; a < r; ++a)
        if (node[a] < 0 || node[a] >= c->M || !(delta[a] > 0.0))
            return fail(c, EMI_ERR_ARG, "emi_kkt_lowrank: column %d has node %d / delta %g", a, node[a], delta[a]);
    std::string err;
    const int st = emi::kkt_lowrank(c->kkt, c->stream, (c->ns + c->nc) * c->M, r, node, vec, delta, exact, &err);
    if (st) c->err = err;
    return st;
}

// ---- batched Newton steps: n contexts (one scenario each) on the same mesh and model ---------------------------------------
static int batch_compatible(int n, const emi_ctx_t* ctxs, const char* what, bool factorising) {
    if (n < 1 || !ctxs || !ctxs[0]) return EMI_ERR_ARG;
    emi_ctx_t c0 = ctxs[0];
    for (int b = 0; b < n; ++b) {
        emi_ctx_t c = ctxs[b];
        if (!c) return EMI_ERR_ARG;
        if (c->M <= 0 || c->model < 0 || c->f32 || c->points_only)
            return fail(c0, EMI_ERR_STATE, "%s: context %d is not an f64 context with a collocation mesh and a model", what, b);
        if (factorising && c->kkt_method != 1)
            return fail(c0, EMI_ERR_UNSUPPORTED, "%s: context %d is set to the LU method (\"kkt_method\" 0): emi_kkt_factor", what, b);
        if (c->device != c0->device || c->M != c0->M || c->ns != c0->ns || c->nc != c0->nc)
            return fail(c0, EMI_ERR_ARG, "%s: context %d differs from context 0 in device, mesh size or model dimensions", what, b);
        for (int a = 0; a < b; ++a)
            if (ctxs[a] == c) return fail(c0, EMI_ERR_ARG, "%s: context %d appears twice", what, b);
    }
    return EMI_OK;
}

int emi_kkt_factor_batch(int n, const emi_ctx_t* ctxs, const double* const* Qblk, const double* const* Jblk,
                         const unsigned char* const* fixed, const double* dc, int* info) {
    int st = batch_compatible(n, ctxs, "emi_kkt_factor_batch", true);
    if (st) return st;
    emi_ctx_t c0 = ctxs[0];
    if (!Qblk || !Jblk || !fixed || !dc || !info) return fail(c0, EMI_ERR_ARG, "emi_kkt_factor_batch: null argument");
    for (int b = 0; b < n; ++b)
        if (!Qblk[b] || !Jblk[b] || !fixed[b] || !(dc[b] >= 0.0)) return fail(c0, EMI_ERR_ARG, "emi_kkt_factor_batch: bad argument for scenario %d", b);
    HIP_TRY(c0, hipSetDevice(c0->device));
    std::vector<emi::KktWorkspace**> pws(n);
    std::vector<const double*> dD(n);
    for (int b = 0; b < n; ++b) {
        HIP_TRY(c0, hipStreamSynchronize(ctxs[b]->stream));     // whatever the scenario's own stream still holds (its last evaluation)
        pws[b] = &ctxs[b]->kkt;
        dD[b] = (const double*)ctxs[b]->d_D.p;
    }
    std::string err;
    st = emi::kkt_factor_batch(n, pws.data(), c0->stream, dD.data(), c0->M, c0->ns, c0->ns + c0->nc, Qblk, Jblk, fixed, dc, info, &err);
    if (st) { c0->err = err; return st; }
    // scenarios the batch could not take (a node block not positive definite, the ladder exhausted): the single path with its LU
    for (int b = 0; b < n; ++b)
        if (info[b] < 0) {
            const int s1 = emi_kkt_factor(ctxs[b], Qblk[b], Jblk[b], fixed[b], dc[b], &info[b]);
            if (s1) { c0->err = ctxs[b]->err; return s1; }
        }
    return EMI_OK;
}

int emi_kkt_solve_batch(int n, const emi_ctx_t* ctxs, double* const* rhs) {
    int st = batch_compatible(n, ctxs, "emi_kkt_solve_batch", false);
    if (st) return st;
    emi_ctx_t c0 = ctxs[0];
    if (!rhs) return fail(c0, EMI_ERR_ARG, "emi_kkt_solve_batch: null argument");
    HIP_TRY(c0, hipSetDevice(c0->device));
    // scenarios whose factorisation is the LU fallback (or none) go through the single entry point; the rest as one batch
    std::vector<emi::KktWorkspace*> ws;
    std::vector<double*> rb;
    for (int b = 0; b < n; ++b) {
        if (!rhs[b]) return fail(c0, EMI_ERR_ARG, "emi_kkt_solve_batch: null right-hand side %d", b);
        HIP_TRY(c0, hipStreamSynchronize(ctxs[b]->stream));
        if (emi::kkt_is_schur(ctxs[b]->kkt)) { ws.push_back(ctxs[b]->kkt); rb.push_back(rhs[b]); }
        else if ((st = emi_kkt_solve(ctxs[b], rhs[b], 1))) { c0->err = ctxs[b]->err; return st; }
    }
    if (ws.empty()) return EMI_OK;
    std::string err;
    st = emi::kkt_solve_batch((int)ws.size(), ws.data(), c0->stream, (c0->ns + c0->nc) * c0->M, rb.data(), &err);
    if (st) c0->err = err;
    return st;
}

int emi_kkt_solve_refined_batch(int n, const emi_ctx_t* ctxs, double* const* rhs, const double* dc_nominal, int max_steps, double* rel,
                                int* nsolve, int* reverted, int* status) {
    int st = batch_compatible(n, ctxs, "emi_kkt_solve_refined_batch", false);
    if (st) return st;
    emi_ctx_t c0 = ctxs[0];
    if (!rhs || !dc_nominal || !rel || !nsolve || !reverted || !status || max_steps < 0)
        return fail(c0, EMI_ERR_ARG, "emi_kkt_solve_refined_batch: bad argument");
    HIP_TRY(c0, hipSetDevice(c0->device));
    std::vector<emi::KktWorkspace*> ws(n);
    for (int b = 0; b < n; ++b) {
        if (!rhs[b] || !(dc_nominal[b] >= 0.0)) return fail(c0, EMI_ERR_ARG, "emi_kkt_solve_refined_batch: bad argument for scenario %d", b);
        if (!emi::kkt_is_schur(ctxs[b]->kkt))
            return fail(c0, EMI_ERR_UNSUPPORTED, "emi_kkt_solve_refined_batch: scenario %d holds no factorisation of the Schur path (the LU fallback "
                                                 "is refined by the caller: emi_kkt_solve)", b);
        HIP_TRY(c0, hipStreamSynchronize(ctxs[b]->stream));
        ws[b] = ctxs[b]->kkt;
    }
    std::string err;
    st = emi::kkt_solve_refined_batch(n, ws.data(), c0->stream, rhs, dc_nominal, max_steps, rel, nsolve, reverted, status, &err);
    if (st) c0->err = err;
    return st;
}

int emi_kkt_solve_refined(emi_ctx_t c, double* rhs, double dc_nominal, int max_steps, double* rel, int* nsolve, int* reverted, int* status) {
    if (!c) return EMI_ERR_ARG;
    double* r1 = rhs;
    return emi_kkt_solve_refined_batch(1, &c, &r1, &dc_nominal, max_steps, rel, nsolve, reverted, status);
}

int emi_kkt_is_schur(emi_ctx_t c) { return c && emi::kkt_is_schur(c->kkt) ? 1 : 0; }

int emi_kkt_last_regularisation(emi_ctx_t c, double* dc, double* dw) {
    if (!c || (!dc && !dw)) return EMI_ERR_ARG;
    if (!c->kkt) return fail(c, EMI_ERR_STATE, "emi_kkt_last_regularisation: no factorisation");
    emi::kkt_last_regularisation(c->kkt, dc, dw);
    return EMI_OK;
}

int emi_kkt_solve(emi_ctx_t c, double* rhs, int nrhs) {
    if (!c || !rhs || nrhs < 1) return EMI_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    std::string err;
    const int st = emi::kkt_solve(c->kkt, c->stream, (c->ns + c->nc) * c->M, rhs, nrhs, &err);
    if (st) c->err = err;
    return st;
}

int emi_timer_start(emi_ctx_t c) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipEventRecord(c->t_start, c->stream));
    return EMI_OK;
}

int emi_timer_stop(emi_ctx_t c, float* ms) {
    if (!c || !ms) return EMI_ERR_ARG;
    HIP_TRY(c, hipEventRecord(c->t_stop, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->t_stop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->t_start, c->t_stop));
    return EMI_OK;
}

int emi_profile_enable(emi_ctx_t c, int on) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->profile = (on >= 0 && on <= 3) ? on : 1;
    c->prof_used = 0;
    return EMI_OK;
}

int emi_profile_read(emi_ctx_t c, float* node_ms, int* node_launches, float* defect_ms,
                     int* defect_launches, float* pass_ms, int* overlapped_passes) {
    if (!c) return EMI_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(c, hipStreamSynchronize(c->stream2));
    float nm = 0, dm = 0, fm = 0;
    int nl = 0, dl = 0, fl = 0;
    for (size_t i = 0; i < c->prof_used; ++i) {
        ProfEvents& pe = c->prof[i];
        float ms = 0;
        if (pe.fused && pe.level == -1) {
            HIP_TRY(c, hipEventElapsedTime(&ms, pe.k[0], pe.k[1]));
            dm += ms; ++dl; fm += ms; ++fl;
            continue;
        }
        if (pe.fused) {
            if (pe.level == 1) {
                HIP_TRY(c, hipEventElapsedTime(&ms, pe.e[0], pe.e[1]));
                fm += ms;
            }
            ++fl;
            if (pe.level == 1 || pe.level == 2) {
                HIP_TRY(c, hipEventElapsedTime(&ms, pe.k[0], pe.k[1]));
                dm += ms;
                ++dl;
            }
            if (pe.level == 1 || pe.level == 3) {
                HIP_TRY(c, hipEventElapsedTime(&ms, pe.k[2], pe.k[3]));
                nm += ms;
                ++nl;
            }
            continue;
        }
        if (pe.has_node && pe.level != 2) {
            HIP_TRY(c, hipEventElapsedTime(&ms, pe.e[0], pe.e[1]));
            nm += ms;
            ++nl;
        }
        if (pe.has_defect && pe.level != 3) {
            HIP_TRY(c, hipEventElapsedTime(&ms, pe.e[1], pe.e[2]));
            dm += ms;
            ++dl;
        }
    }
    c->prof_used = 0;
    if (node_ms) *node_ms = nm;
    if (node_launches) *node_launches = nl;
    if (defect_ms) *defect_ms = dm;
    if (defect_launches) *defect_launches = dl;
    if (pass_ms) *pass_ms = fm;
    if (overlapped_passes) *overlapped_passes = fl;
    return EMI_OK;
}

int emi_set_option(emi_ctx_t c, const char* name, int value) {
    if (!c || !name) return EMI_ERR_ARG;
    if (strcmp(name, "overlap") == 0 || strcmp(name, "fused") == 0) { c->allow_fused = value != 0; return EMI_OK; }
    if (strcmp(name, "sym_ct") == 0) {
        if (value < 0 || value > 8) return fail(c, EMI_ERR_ARG, "sym_ct must be 0..8 (0/4 = chosen from the batch, 3 = LDS-DMA ring, 5..8 = state-split ring with SW = NS/2/1/3)");
        c->sym_ct = value;
        return EMI_OK;
    }
    if (strcmp(name, "cu_split") == 0) {
        // experiment: spatial instead of temporal sharing of the chip between the MFMA and the streaming kernel
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        const int ncu = prop.multiProcessorCount;
        if (value < 0 || value >= ncu) return fail(c, EMI_ERR_ARG, "cu_split must be in [0, %d)", ncu);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->s_mfma) { HIP_TRY(c, hipStreamDestroy(c->s_mfma)); c->s_mfma = nullptr; }
        if (c->s_node) { HIP_TRY(c, hipStreamDestroy(c->s_node)); c->s_node = nullptr; }
        c->cu_split = value;
        if (value > 0) {
            const int words = (ncu + 31) / 32;
            std::vector<uint32_t> m1(words, 0u), m2(words, 0u);
            for (int i = 0; i < ncu; ++i) (i < value ? m1 : m2)[i / 32] |= 1u << (i % 32);
            HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->s_mfma, words, m1.data()));
            HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->s_node, words, m2.data()));
            if (!c->ev_join2) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
        }
        return EMI_OK;
    }
    if (strncmp(name, "kkt_", 4) == 0 && strcmp(name, "kkt_method") != 0) {
        if (emi::kkt_set_option(name, value)) return EMI_OK;
        return fail(c, EMI_ERR_ARG, "unknown option %s", name);
    }
    if (strcmp(name, "kkt_method") == 0) {
        if (value != 0 && value != 1) return fail(c, EMI_ERR_ARG, "kkt_method must be 0 (LU) or 1 (Schur complement + Cholesky)");
        c->kkt_method = value;
        return EMI_OK;
    }
    if (strcmp(name, "small_rows") == 0) {
        if (value < 0) return fail(c, EMI_ERR_ARG, "small_rows must be >= 0 (0 disables the skinny defect kernel)");
        c->small_rows = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_order") == 0) { c->sym_order = value != 0; return EMI_OK; }
    if (strcmp(name, "f32_ring") == 0) { c->f32_ring = value != 0; return EMI_OK; }
    if (strcmp(name, "f32_ring_wgs") == 0) { c->f32_ring_wgs = value == 1 ? 1 : 2; return EMI_OK; }
    if (strcmp(name, "f32_one_launch") == 0) { c->f32_one_launch = value != 0; return EMI_OK; }
    if (strcmp(name, "slice") == 0) {
        if (value < 0 || (value > 0 && value % 16 != 0)) return fail(c, EMI_ERR_ARG, "slice must be 0 (never) or a multiple of 16 instances");
        c->slice = value;
        return EMI_OK;
    }
    if (strcmp(name, "pass_order") == 0) { c->pass_order = value < 0 ? -1 : (value >= 100 ? value : (value != 0)); return EMI_OK; }
    if (strcmp(name, "sym_ablate") == 0) { c->sym_ablate = value; return EMI_OK; }   // diagnostics only
    if (strcmp(name, "cost_in_kernel") == 0) { c->cost_in_kernel = value != 0; return EMI_OK; }
    if (strcmp(name, "sym_nst") == 0) {
        if (value != 3 && value != 4) return fail(c, EMI_ERR_ARG, "sym_nst must be 3 or 4");
        c->sym_nst = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_hs") == 0) {
        if (value < 0 || value > 2) return fail(c, EMI_ERR_ARG, "sym_hs must be 0 (by batch size), 1 or 2");
        c->sym_hs = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_ctc") == 0) {
        if (value < 0 || value > 2) return fail(c, EMI_ERR_ARG, "sym_ctc must be 0 (by batch size), 1 or 2");
        c->sym_ctc = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_bk") == 0) {
        if (value != 0 && value != 8 && value != 16) return fail(c, EMI_ERR_ARG, "sym_bk must be 0 (by batch size), 8 or 16");
        c->sym_bk = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_cpart") == 0) {
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(c, EMI_ERR_ARG, "sym_cpart must be -1 (plain order), 0 (by mesh size), 1, 2, 4 or 8");
        c->sym_cpart = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_gblk") == 0) {
        if (value < 0 || value > 64) return fail(c, EMI_ERR_ARG, "sym_gblk must be 0 (off) .. 64 instance groups per super-block");
        c->sym_gblk = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_cx") == 0) {
        if (value < 0 || value > 64) return fail(c, EMI_ERR_ARG, "sym_cx must be 0 (default) .. 64 column tiles per block");
        c->sym_cx = value;
        return EMI_OK;
    }
    if (strcmp(name, "sym_combine") == 0) { c->sym_combine = value != 0; return EMI_OK; }
    if (strcmp(name, "sym_ksplit") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(c, EMI_ERR_ARG, "sym_ksplit must be 0 (by batch size), 1, 2, 4 or 8");
        c->sym_ksplit = value;
        return EMI_OK;
    }
    if (strcmp(name, "node_store") == 0) {
        if (value < -1 || value > 3) return fail(c, EMI_ERR_ARG, "node_store must be -1 (by size), 0 (plain), 1 (write-through sc1), 2 (non-temporal) or 3 (nt sc1; the one-launch pass only)");
        c->node_store = value;
        return EMI_OK;
    }
    if (strcmp(name, "overlap_mode") == 0) {
        if (value < 0 || value > 3) return fail(c, EMI_ERR_ARG, "overlap_mode must be 0 (by batch size), 1 (one stream), 2 (two streams) or 3 (one launch)");
        c->overlap_mode = value;
        return EMI_OK;
    }
    return fail(c, EMI_ERR_ARG, "unknown option '%s'", name);
}

int emi_plan_pass(emi_ctx_t c, int B, emi_pass_plan_t* out) {
    if (!c || !out || B < 1) return EMI_ERR_ARG;
    if (c->M <= 0 || c->model < 0) return fail(c, EMI_ERR_STATE, "emi_plan_pass: mesh and model must be set");
    memset(out, 0, sizeof *out);
    const int Bkeep = c->B;
    c->B = B;                                   // the policy reads the batch from the context (piece sizes, per-instance tables)
    const int piece = (!c->f32 && overlapped_path(c)) ? plan_piece(c) : 0;
    const int first = piece > 0 ? piece : B;    // instances of the first launch
    out->piece = piece;
    out->tail = piece > 0 ? B % piece : 0;
    if (!c->f32 && overlapped_path(c)) {
        const PassPlan p = plan_pass(c, first, true);
        out->one_launch = p.one_launch ? 1 : 0;
        out->sw = p.sym.sw;
        out->ksplit = p.sym.ks;
        out->ring_stages = p.sym.nst;
        out->k_tile = p.sym.bk;
        out->column_tiles = p.sym.ct;
        out->k_halves = p.sym.hs;
        out->cpart = p.sym.cpart;
        out->cx = p.sym.cx;
        out->mfma_workgroups = p.sym.tiles * p.sym.ks;
        out->store_mode = p.store_mode;
        out->block_order = p.mfma_first;
        out->tiles16 = p.tiles16;
    }
    c->B = Bkeep;
    return EMI_OK;
}

int emi_last_path(emi_ctx_t c, int* fused) {
    if (!c || !fused) return EMI_ERR_ARG;
    const bool small = !c->f32 && c->B > 0 && c->B * c->ns <= c->small_rows && emi::defect_small_supported(c->B * c->ns) &&
                       !pass_takes_small_batches(c);
    *fused = (!small && overlapped_path(c)) ? 1 : 0;
    return EMI_OK;
}

/* name of the kernel that produced the defect rows in the last emi_eval_dev of this context (for reports) */
int emi_debug_pass_roles(int nm, int nn, int order, int* out_role, int out_cap) {
    if (nm < 0 || nn < 0 || nm + nn < 1 || !out_role || out_cap < nm + nn) return EMI_ERR_ARG;
    for (int j = 0; j < nm + nn; ++j) {
        const emi::PassRole r = emi::pass_role_of(j, nm, nn, order);
        out_role[j] = r.mfma ? r.index : -1 - r.index;
    }
    return EMI_OK;
}

int emi_debug_tile_order(int ns, int B, int M, int sym_ct, int sym_cpart, int* out_tile, int out_cap, int* ntiles_total, int* cpart,
                         int* cx) {
    return emi_debug_tile_order2(ns, B, M, sym_ct, sym_cpart, 0, 0, out_tile, out_cap, ntiles_total, cpart, cx);
}

int emi_debug_tile_order2(int ns, int B, int M, int sym_ct, int sym_cpart, int sym_gblk, int sym_cx, int* out_tile, int out_cap,
                          int* ntiles_total, int* cpart, int* cx) {
    if (ns < 1 || B < 1 || M < 128 || M % 128 != 0 || !ntiles_total) return EMI_ERR_ARG;
    const emi::SymPlan p = emi::plan_symdefect(ns, B, M, sym_ct, 1, sym_cpart, sym_gblk, sym_cx);
    if (p.ring1 || p.sw < 1) return EMI_ERR_UNSUPPORTED;
    const int ntiles = (M / 2) / 64, ngrp = ((B + 15) / 16) * (ns / p.sw), total = ntiles * ngrp;
    *ntiles_total = total;
    if (cpart) *cpart = p.cpart;
    if (cx) *cx = p.cx;
    for (int t = 0; t < total && t < out_cap && out_tile; ++t) {
        const emi::RingTile rt = emi::ring_tile_of(t, ntiles, ngrp, p.cpart, p.cx, ns / p.sw);
        out_tile[t] = rt.ntile + ntiles * rt.grp;
    }
    return EMI_OK;
}

const char* emi_last_defect_kernel(emi_ctx_t c) {
    if (!c) return "";
    return c->last_defect_kernel.c_str();
}

}  // extern "C"
