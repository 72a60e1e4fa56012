// emi_rtc.hip -- model programs compiled at run time.
//
// ePSOPT evaluates the user's callbacks and their derivatives by interpreting an ADOL-C tape on the
// CPU at every NLP evaluation (reference src/ePSOPT/ePSOPT.cpp:64-65, 186-276).  Here a model
// arrives as the text of a struct with the Model interface of emi_models.hpp (written by hand, or
// generated from traced callbacks by etol_amd/host/emi_trace.cpp); it is compiled ONCE for gfx950
// with hiprtc against the same kernel templates libemi355x.so was built from (their headers are
// embedded as text, tools/embed_src.py), and the resulting code object is launched exactly like
// the built-in instantiations: node kernel, Hessian kernel and the even/odd MFMA defect kernel.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "emi_kernels.hpp"
#include "emi_rtc_sources.inc"

namespace emi {

struct RtcModel {
    hipModule_t mod = nullptr;
    hipFunction_t nodes[2][2][2] = {};   // [vec2][jac][defect rows]
    hipFunction_t hess = nullptr;
    hipFunction_t ring = nullptr;        // even/odd MFMA defect kernel (f64, LDS permitting)
    hipFunction_t pass_small = nullptr;  // the pass as one launch: SW = 1, plain stores (small batches)
    hipFunction_t pass_large = nullptr;  //                          SW = sw_large, non-temporal stores (large batches)
    hipFunction_t node_nt = nullptr;     // node kernel (vec2, Jacobian, no defect rows) with non-temporal stores
    bool f32 = false;
    int ns = 0, nc = 0, sw_large = 0;
    size_t ring_lds = 0;
};

namespace {

bool valid_identifier(const char* s) {
    if (!s || !*s || (*s >= '0' && *s <= '9')) return false;
    for (; *s; ++s)
        if (!((*s >= 'a' && *s <= 'z') || (*s >= 'A' && *s <= 'Z') || (*s >= '0' && *s <= '9') || *s == '_')) return false;
    return true;
}

size_t ring_lds_bytes(int ns) { return (size_t)3 * (2 * ns * FUSED_TI + 2 * 64) * 16 * sizeof(double); }
// the ring kernel holds 16 accumulator registers per state and needs three stages in 160 KB of LDS
bool ring_fits(int ns) { return ns <= 8 && ring_lds_bytes(ns) <= 160 * 1024; }

struct Names {
    std::string nodes[2][2][2], hess, ring, pass_small, pass_large, node_nt;
};
int pass_sw_large(int ns) { return (ns % 2 == 0 && ns > 2) ? 2 : 1; }

Names kernel_names(const char* sn, bool f32, bool with_ring, int ns) {
    Names n;
    const std::string T = f32 ? "float" : "double";
    const std::string model = std::string("emi::") + sn + "<" + T + ">";
    for (int v = 0; v < 2; ++v)
        for (int j = 0; j < 2; ++j)
            for (int d = 0; d < 2; ++d)
                n.nodes[v][j][d] = "emi::emi_nodes_kernel<" + T + ", " + model + ", " + (v ? "2" : "1") + ", " +
                                   (j ? "true" : "false") + ", " + (d ? "true" : "false") + ">";
    n.hess = "emi::emi_hess_kernel<" + T + ", " + model + ">";
    if (with_ring) {
        n.ring = "emi::emi_symdefect_ring_f64_kernel<" + model + ">";
        n.pass_small = "emi::emi_pass_f64_kernel<" + model + ", 1, 2, 0, 3>";
        n.pass_large = "emi::emi_pass_f64_kernel<" + model + ", " + std::to_string(pass_sw_large(ns)) + ", 2, 2, 3>";
        n.node_nt = "emi::emi_nodes_kernel<double, " + model + ", 2, true, false, 2>";
    }
    return n;
}

// compile; on success *code holds the gfx950 code object and *lowered the mangled kernel names in
// the order nodes[0][0][0..1], nodes[0][1][..], nodes[1][..][..], hess, ring, pass (small, large), node kernel with nt stores
int compile(bool f32, const char* sn, const char* source, int ns, int nc, int npath, int pw, std::vector<char>* code,
            std::vector<std::string>* lowered, bool* has_ring, std::string* log) {
    if (!valid_identifier(sn) || !source || ns < 1 || nc < 0 || ns + nc > 64) {
        if (log) *log = "emi_set_model_source: bad struct name, null source or dimensions out of range";
        return EMI_ERR_ARG;
    }
    const bool with_ring = !f32 && ring_fits(ns);
    std::string prog = "#include \"emi_args.hpp\"\n#include \"emi_node_kernels.hpp\"\n";
    if (with_ring) prog += "#include \"emi_symdefect_kernels.hpp\"\n";
    prog += "namespace emi {\n";
    prog += source;
    prog += "\n}  // namespace emi\n";
    char chk[600];
    snprintf(chk, sizeof chk, "static_assert(emi::%s<double>::NS == %d && emi::%s<double>::NC == %d && emi::%s<double>::NV == %d && "
             "emi::%s<double>::NPATH == %d, \"model struct dimensions differ from emi_set_model_source(ns, nc, npath)\");\n", sn, ns,
             sn, nc, sn, ns + nc, sn, npath);
    prog += chk;
    if (npath > 0) {
        snprintf(chk, sizeof chk, "static_assert(emi::%s<double>::PW == %d, \"the model's traced rows depend on another number of "
                 "variables than emi_set_model_source(n_path_vars) says\");\n", sn, pw);
        prog += chk;
    }

    hiprtcProgram p;
    if (hiprtcCreateProgram(&p, prog.c_str(), "emi_model_program.hip", emi_rtc_nfiles, emi_rtc_texts, emi_rtc_names) !=
        HIPRTC_SUCCESS) {
        if (log) *log = "hiprtcCreateProgram failed";
        return EMI_ERR_HIP;
    }
    const Names nm = kernel_names(sn, f32, with_ring, ns);
    std::vector<const std::string*> order;
    for (int v = 0; v < 2; ++v)
        for (int j = 0; j < 2; ++j)
            for (int d = 0; d < 2; ++d) order.push_back(&nm.nodes[v][j][d]);
    order.push_back(&nm.hess);
    if (with_ring) {
        order.push_back(&nm.ring);
        order.push_back(&nm.pass_small);
        order.push_back(&nm.pass_large);
        order.push_back(&nm.node_nt);
    }
    for (const std::string* s : order) hiprtcAddNameExpression(p, s->c_str());

    // (-Wno-inline-asm: the DMA instructions of the MFMA kernels name m0 in their clobber list, which clang reports as a reserved register)
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-inline-asm"};
    const hiprtcResult r = hiprtcCompileProgram(p, 5, opts);
    size_t ls = 0;
    hiprtcGetProgramLogSize(p, &ls);
    std::string l(ls, '\0');
    if (ls) hiprtcGetProgramLog(p, &l[0]);
    while (!l.empty() && l.back() == '\0') l.pop_back();
    if (r != HIPRTC_SUCCESS) {
        if (log) *log = std::string("model program does not compile (") + hiprtcGetErrorString(r) + "):\n" + l;
        hiprtcDestroyProgram(&p);
        return EMI_ERR_ARG;
    }
    if (log) *log = l;
    lowered->clear();
    for (const std::string* s : order) {
        const char* low = nullptr;
        if (hiprtcGetLoweredName(p, s->c_str(), &low) != HIPRTC_SUCCESS || !low) {
            if (log) *log = "no lowered name for " + *s;
            hiprtcDestroyProgram(&p);
            return EMI_ERR_HIP;
        }
        lowered->push_back(low);
    }
    size_t cs = 0;
    hiprtcGetCodeSize(p, &cs);
    code->resize(cs);
    hiprtcGetCode(p, code->data());
    hiprtcDestroyProgram(&p);
    *has_ring = with_ring;
    return EMI_OK;
}

hipError_t launch(hipFunction_t f, dim3 grid, dim3 block, size_t lds, hipStream_t s, const void* args, size_t bytes) {
    if (!f) return hipErrorInvalidDeviceFunction;
    // one by-value struct parameter: hand it over as the packed argument buffer
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes,
                      HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, s, nullptr,
                                 config);
}

}  // namespace

int rtc_check(bool f32, const char* struct_name, const char* source, int ns, int nc, int npath, int pw, std::string* log) {
    std::vector<char> code;
    std::vector<std::string> low;
    bool ring = false;
    return compile(f32, struct_name, source, ns, nc, npath, pw, &code, &low, &ring, log);
}

namespace {
// code objects by (arithmetic type, struct name, dimensions, text): a Monte-Carlo run sets the same traced model up
// in many contexts and pays the ~4 s of hiprtc once per process
struct CachedProgram {
    std::vector<char> code;
    std::vector<std::string> lowered;
    bool ring = false;
};
std::mutex g_cache_mutex;
std::map<std::string, CachedProgram> g_cache;
}  // namespace

int rtc_build(bool f32, const char* struct_name, const char* source, int ns, int nc, int npath, int pw, RtcModel** out,
              std::string* log) {
    *out = nullptr;
    std::vector<char> code;
    std::vector<std::string> low;
    bool ring = false;
    const std::string key = std::string(f32 ? "f32|" : "f64|") + (struct_name ? struct_name : "") + "|" + std::to_string(ns) + "|" +
                            std::to_string(nc) + "|" + std::to_string(npath) + "|" + (source ? source : "");
    bool hit = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) {
            code = it->second.code;
            low = it->second.lowered;
            ring = it->second.ring;
            hit = true;
        }
    }
    if (!hit) {
        const int st = compile(f32, struct_name, source, ns, nc, npath, pw, &code, &low, &ring, log);
        if (st) return st;
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        CachedProgram& cp = g_cache[key];
        cp.code = code;
        cp.lowered = low;
        cp.ring = ring;
    }
    RtcModel* m = new RtcModel();
    m->f32 = f32;
    m->ns = ns;
    m->nc = nc;
    hipError_t e = hipModuleLoadData(&m->mod, code.data());
    size_t i = 0;
    for (int v = 0; v < 2 && e == hipSuccess; ++v)
        for (int j = 0; j < 2 && e == hipSuccess; ++j)
            for (int d = 0; d < 2 && e == hipSuccess; ++d) e = hipModuleGetFunction(&m->nodes[v][j][d], m->mod, low[i++].c_str());
    if (e == hipSuccess) e = hipModuleGetFunction(&m->hess, m->mod, low[i++].c_str());
    if (e == hipSuccess && ring) {
        e = hipModuleGetFunction(&m->ring, m->mod, low[i++].c_str());
        m->ring_lds = ring_lds_bytes(ns);
        if (e == hipSuccess) e = hipModuleGetFunction(&m->pass_small, m->mod, low[i++].c_str());
        if (e == hipSuccess) e = hipModuleGetFunction(&m->pass_large, m->mod, low[i++].c_str());
        if (e == hipSuccess) e = hipModuleGetFunction(&m->node_nt, m->mod, low[i++].c_str());
        m->sw_large = pass_sw_large(ns);
    }
    if (e != hipSuccess) {
        if (log) *log = std::string("loading the model code object failed: ") + hipGetErrorString(e);
        rtc_destroy(m);
        return EMI_ERR_HIP;
    }
    *out = m;
    return EMI_OK;
}

void rtc_destroy(RtcModel* m) {
    if (!m) return;
    if (m->mod) (void)hipModuleUnload(m->mod);
    delete m;
}

bool rtc_has_symdefect(const RtcModel* m) { return m && m->ring; }

template <typename T>
hipError_t rtc_launch_nodes(RtcModel* m, const NodeArgs<T>& a, bool jac, bool defect_rows, hipStream_t s) {
    const bool vec2 = a.M % 2 == 0;
    const int per_block = EMI_NODE_THREADS * (vec2 ? 2 : 1);
    dim3 grid((a.M + per_block - 1) / per_block, a.B), block(EMI_NODE_THREADS);
    return launch(m->nodes[vec2][jac][defect_rows], grid, block, 0, s, &a, sizeof a);
}
template hipError_t rtc_launch_nodes<double>(RtcModel*, const NodeArgs<double>&, bool, bool, hipStream_t);
template hipError_t rtc_launch_nodes<float>(RtcModel*, const NodeArgs<float>&, bool, bool, hipStream_t);

template <typename T> hipError_t rtc_launch_hess(RtcModel* m, const HessArgs<T>& a, hipStream_t s) {
    dim3 grid((a.M + EMI_NODE_THREADS - 1) / EMI_NODE_THREADS, a.B), block(EMI_NODE_THREADS);
    return launch(m->hess, grid, block, 0, s, &a, sizeof a);
}
template hipError_t rtc_launch_hess<double>(RtcModel*, const HessArgs<double>&, hipStream_t);
template hipError_t rtc_launch_hess<float>(RtcModel*, const HessArgs<float>&, hipStream_t);

hipError_t rtc_launch_symdefect(RtcModel* m, const SymDefectArgs& a, hipStream_t s) {
    const int mtiles = (a.B + FUSED_TI - 1) / FUSED_TI, ntiles = (a.M / 2) / 64;
    return launch(m->ring, dim3(mtiles * ntiles), dim3(256), m->ring_lds, s, &a, sizeof a);
}

int rtc_pass_sw_large(const RtcModel* m) { return m ? m->sw_large : 0; }

// as pass_supported (emi_symdefect.hip) for the two instantiations a model program holds
bool rtc_pass_supported(const RtcModel* m, int B, int M, int sw, int ks, int store_mode) {
    if (!m || !m->pass_small || ks < 1 || M % 128 != 0) return false;       // (K slices are a run-time argument of the kernel)
    if (!((sw == 1 && store_mode != 2) || (sw == m->sw_large && store_mode == 2))) return false;
    return B >= 1 && m->ns % sw == 0;
}

hipError_t rtc_launch_pass(RtcModel* m, const SymDefectArgs& sa, const NodeArgs<double>& na, int sw, hipStream_t s) {
    PassArgs a;
    a.s = sa;
    a.n = na;
    const int mtiles = (sa.B + FUSED_TI - 1) / FUSED_TI, ntiles = (sa.M / 2) / 64;
    const int nm = mtiles * ntiles * (m->ns / sw) * (sa.ksplit > 1 ? sa.ksplit : 1);
    a.nbx = (na.M + 2 * EMI_NODE_THREADS - 1) / (2 * EMI_NODE_THREADS);
    const int nn = a.nbx * na.B;
    a.nm = nm;
    a.nn = nn;
    a.nm8 = (nm + 7) / 8;
    a.nn8 = (nn + 7) / 8;
    const size_t lds = (size_t)3 * ((2 * sw * FUSED_TI + 2 * 64 + 63) / 64 * 64) * 8 * sizeof(double);     // 3 ring stages
    return launch(na.store_mode == 2 ? m->pass_large : m->pass_small, dim3(8 * (a.nm8 + a.nn8)), dim3(256), lds, s, &a, sizeof a);
}

hipError_t rtc_launch_nodes_nt(RtcModel* m, const NodeArgs<double>& a, hipStream_t s) {
    if (!m->node_nt || a.M % 2) return hipErrorInvalidDeviceFunction;
    dim3 grid((a.M + 2 * EMI_NODE_THREADS - 1) / (2 * EMI_NODE_THREADS), a.B), block(EMI_NODE_THREADS);
    return launch(m->node_nt, grid, block, 0, s, &a, sizeof a);
}

}  // namespace emi
