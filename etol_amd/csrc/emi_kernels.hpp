// emi_kernels.hpp -- launcher prototypes shared by the kernel translation units and the C-ABI
// implementation (emi_api.hip).  Argument blocks and tile constants: emi_args.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <string>

#include "emi355x.h"
#include "emi_args.hpp"

// emi_args.hpp repeats these for run-time compiled model programs
static_assert(EMI_PATH_ELLIPSE == 0 && EMI_PATH_DISC == 1 && EMI_PATH_TRACK == 2 && EMI_PATH_REC == 8,
              "emi_args.hpp and emi355x.h disagree");

namespace emi {

int node_chunks(int M);
template <typename T>
hipError_t launch_nodes(int model, const NodeArgs<T>& a, bool jac, bool defect_rows, hipStream_t s);
template <typename T> hipError_t launch_hess(int model, const HessArgs<T>& a, hipStream_t s);
hipError_t launch_defect_f64(const DefectArgs& a, hipStream_t s);
bool defect_small_supported(int R);
hipError_t launch_defect_small_f64(const DefectArgs& a, hipStream_t s);
hipError_t launch_defect_f32(const DefectArgsF32& a, hipStream_t s);
bool defect_f32_mfma_supported(int M);
hipError_t launch_defect_f32_mfma(const DefectArgsF32& a, hipStream_t s, int ring = 0, int wgs_per_cu = 2);    // ring: the LDS-DMA operand ring form
// the fp32 pass as one launch (MFMA role + node role, defect rows by float atomics onto zeroed rows): emi_defect_f32.hip
bool pass_f32_supported(int model, int R, int M, int B);
hipError_t launch_pass_f32(int model, const DefectArgsF32& d, const NodeArgs<float>& n, int order, hipStream_t s);
hipError_t defect_f64_set_attr();
template <typename T>
hipError_t launch_cost_finish(const T* part, T* cost, int B, int nchunks, T scale, hipStream_t s);
bool fused_supported(int model, int M, int ct);
// which even/odd MFMA defect kernel a launch uses (emi_symdefect.hip, plan_symdefect)
struct SymPlan {
    bool ring1 = false;       // the one-workgroup-per-CU ring kernel / register-staged forms (sym_ct 1..3)
    int sw = 0;               // state-split ring: states per workgroup
    int ks = 1;               // ... and K slices per tile (> 1: partial sums through a slab, combined in-kernel by ticket or by a second launch)
    int nst = 3;              // ring stages of the one-launch pass (3 or 4: K tiles in flight = nst - 1)
    int bk = 8;               // depth of a K tile of the one-launch pass (8, or 16: SW 1 / 2 with three stages)
    int hs = 1;               // 2: the K range of a tile in two halves inside a 512-thread workgroup, combined through LDS (SW 1 / 2, unsplit, one sub-tile)
    int ct = 1;               // 64-column sub-tiles per MFMA workgroup of the one-launch pass (2: SW = 2, unsplit, M % 256 == 0)
    size_t slab_bytes = 0;
    int cpart = 0, cx = 0;    // tile order (SymDefectArgs::cpart, cx): 0 = plain, > 0 column partitions, < 0 grouped (-G)
    int tiles = 0;            // tiles of the launch (instance groups x column tiles x state groups): tickets of an in-kernel combine
};
SymPlan plan_symdefect(int ns, int B, int M, int ct, int ksplit_opt, int cpart_opt = 0, int gblk_opt = 0, int cx_opt = 0, int bk_opt = 8, int ct_cols = 1);
hipError_t launch_symdefect(int model, const SymDefectArgs& a, hipStream_t s, bool set_attr, int ct, const SymPlan& plan);
// the whole pass as one launch (MFMA-role and node-role workgroups in one grid; COST finished in-kernel)
bool pass_supported(int model, int ns, int B, int M, const SymPlan& plan);
hipError_t launch_pass(int model, const SymDefectArgs& sa, const NodeArgs<double>& na, hipStream_t s, const SymPlan& plan);

// model programs compiled at run time (emi_rtc.hip); the int results are EMI_* status codes
struct RtcModel;
int rtc_check(bool f32, const char* struct_name, const char* source, int ns, int nc, int npath, int pw, std::string* log);
int rtc_build(bool f32, const char* struct_name, const char* source, int ns, int nc, int npath, int pw, RtcModel** out, std::string* log);
void rtc_destroy(RtcModel* m);
bool rtc_has_symdefect(const RtcModel* m);
template <typename T>
hipError_t rtc_launch_nodes(RtcModel* m, const NodeArgs<T>& a, bool jac, bool defect_rows, hipStream_t s);
template <typename T> hipError_t rtc_launch_hess(RtcModel* m, const HessArgs<T>& a, hipStream_t s);
hipError_t rtc_launch_symdefect(RtcModel* m, const SymDefectArgs& a, hipStream_t s);
// the one-launch pass of a run-time compiled model: which state split it was compiled with for large batches (small ones take
// SW = 1), whether (sw, store mode) is available, and the launch
int rtc_pass_sw_large(const RtcModel* m);
bool rtc_pass_supported(const RtcModel* m, int B, int M, int sw, int ks, int store_mode);
hipError_t rtc_launch_pass(RtcModel* m, const SymDefectArgs& sa, const NodeArgs<double>& na, int sw, hipStream_t s);
hipError_t rtc_launch_nodes_nt(RtcModel* m, const NodeArgs<double>& a, hipStream_t s);   // vec2, Jacobian, no defect rows, non-temporal stores

// Newton step on the device (emi_kkt.hip)
struct KktWorkspace;
int kkt_factor(KktWorkspace** w, hipStream_t stream, const double* dD, int M, int ns, int nv, const double* Qblk,
               const double* Jblk, const unsigned char* fixed, double dc, int method, int* info, std::string* err);
int kkt_solve(KktWorkspace* w, hipStream_t stream, int nz, double* rhs, int nrhs, std::string* err);
bool kkt_set_option(const char* name, int value);   // process-wide diagnostics of the factorisation ("kkt_cholesky", ...)
int kkt_lowrank(KktWorkspace* w, hipStream_t stream, int nz, int r, const int* node, const double* vec, const double* delta,
                int* exact, std::string* err);
void kkt_destroy(KktWorkspace* w);
void kkt_last_regularisation(const KktWorkspace* w, double* dc, double* dw);
void kkt_mesh_changed(KktWorkspace* w);
bool kkt_is_schur(const KktWorkspace* w);            // holds a factorisation of the Schur path (what the batched solve takes)
// the Newton steps of n scenarios on one mesh at once (emi_kkt.hip, "Batched entry points")
int kkt_factor_batch(int n, KktWorkspace** const* pws, hipStream_t stream, const double* const* dD, int M, int ns, int nv,
                     const double* const* Qblk, const double* const* Jblk, const unsigned char* const* fixed, const double* dc, int* info,
                     std::string* err);
int kkt_solve_batch(int n, KktWorkspace* const* ws, hipStream_t stream, int nz, double* const* rhs, std::string* err);
int kkt_solve_refined_batch(int n, KktWorkspace* const* ws, hipStream_t stream, double* const* rhs, const double* dc_nominal, int max_steps,
                            double* rel, int* nsolve, int* reverted, int* status, std::string* err);

}  // namespace emi
