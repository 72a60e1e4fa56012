// ref_driver.cpp -- runs pieces of the REFERENCE itself (compiled where they lie under
// /root/reference, nothing copied) and prints their outputs, so that the oracle and the kernels
// are pinned by reference-executed vectors.  TEST INFRASTRUCTURE, build container only: the
// binary goes to oracle/_ref/ (git-ignored) and tests/golden/gen_ref_vectors.py turns its output
// into the committed fixtures tests/golden/ref_*.json.
//
// What is executed from the reference (oracle/Makefile, target `ref`):
//   * include/ETOL/TrajectoryOptimizer.hpp:239-258   linear_interpolation<double>   (header template)
//                                          :268-324   extractTraj / scaleTraj / offsetTraj
//   * src/Examples/Dymos/etol_dymos_example1.cpp     compiled as its own object with
//     -Dmain=etol_dymos_example1_main; this driver calls its node callbacks
//       objFunction :135-156, dxConstraint :158-176, dyConstraint :178-196,
//       obsConstraint :198-256, saaConstraint :258-306, linear_interpolation :362-379
//     with plain doubles in the std::any arguments, as eDymos does (src/eDymos/eDymos.cpp:115-151
//     for values, :241-266 with pnames = {"partials"}).  Its file-scope tables exz / mexz are
//     filled from the data this driver reads on stdin (the generator takes it from the shipped
//     resource/configs/ocp_2d_ex1.xml); setExz / setMexz need a TrajectoryOptimizer object, whose
//     translation unit needs CGAL and cannot be built here, so they are NOT run (the linker drops
//     them together with the example's main: -Wl,--gc-sections).
//
// stdin (numbers as C hex floats or decimals):
//   exz N          then N lines  xc yc radsq tt
//   mexz K         then per track: "radius nway" and nway lines "t x y"
//   points P       then P lines  x y t u0 u1
//   traj R C       then R lines  t v_1 .. v_C ; "idxs n i..", "scale n s..", "offset n o.."
// stdout: one line per result, "%a" floats.
#include <any>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <tuple>
#include <vector>

#include <ETOL/TrajectoryOptimizer.hpp>   // the reference's header (-I/root/reference/include)

// the example's file-scope objects and callbacks (declared here, defined in the reference's .cpp)
extern size_t exz_comp_idx, exz_partial_idx, mexz_comp_idx, mexz_partial_idx;
extern std::vector<std::tuple<double, double, double, double>> exz;
extern std::vector<std::tuple<double, ETOL::state_t, std::vector<ETOL::state_t>>> mexz;
ETOL::scalar_t objFunction(F_ARGS);
ETOL::scalar_t dxConstraint(F_ARGS);
ETOL::scalar_t dyConstraint(F_ARGS);
ETOL::scalar_t obsConstraint(F_ARGS);
ETOL::scalar_t saaConstraint(F_ARGS);
double linear_interpolation(const double&, const ETOL::state_t&, const ETOL::state_t&);

static double rd() {
    std::string s;
    if (!(std::cin >> s)) { std::fprintf(stderr, "ref_driver: short input\n"); std::exit(2); }
    return std::strtod(s.c_str(), nullptr);
}
static void put(const char* tag, const std::vector<double>& v) {
    std::printf("%s", tag);
    for (double d : v) std::printf(" %a", d);
    std::printf("\n");
}
static std::vector<double> call(ETOL::scalar_t (*f)(F_ARGS), double x, double y, double u0, double u1,
                                double t, bool partials) {
    ETOL::vector_t xs{x, y}, us{u0, u1};
    std::vector<std::string> pn{partials ? std::string("partials") : std::string()};
    return std::any_cast<std::vector<double>>(f(xs, us, ETOL::vector_t{}, pn, t, 0.5));
}

int main() {
    std::string key;
    while (std::cin >> key) {
        if (key == "exz") {
            const int n = (int)rd();
            exz.clear();
            for (int i = 0; i < n; ++i) {
                const double xc = rd(), yc = rd(), rs = rd(), tt = rd();
                exz.push_back({xc, yc, rs, tt});
            }
        } else if (key == "mexz") {
            const int n = (int)rd();
            mexz.clear();
            for (int i = 0; i < n; ++i) {
                const double radius = rd();
                const int nway = (int)rd();
                ETOL::state_t tv, xv, yv;
                for (int j = 0; j < nway; ++j) { tv.push_back(rd()); xv.push_back(rd()); yv.push_back(rd()); }
                mexz.push_back({radius, tv, std::vector<ETOL::state_t>{xv, yv}});
            }
        } else if (key == "points") {
            const int n = (int)rd();
            for (int p = 0; p < n; ++p) {
                const double x = rd(), y = rd(), t = rd(), u0 = rd(), u1 = rd();
                put("obj", call(objFunction, x, y, u0, u1, t, false));
                put("obj_p", call(objFunction, x, y, u0, u1, t, true));
                put("dx", call(dxConstraint, x, y, u0, u1, t, false));
                put("dx_p", call(dxConstraint, x, y, u0, u1, t, true));
                put("dy", call(dyConstraint, x, y, u0, u1, t, false));
                put("dy_p", call(dyConstraint, x, y, u0, u1, t, true));
                exz_comp_idx = exz_partial_idx = 0;      // the callbacks walk the tables round-robin
                for (size_t i = 0; i < exz.size(); ++i) put("obs", call(obsConstraint, x, y, u0, u1, t, false));
                for (size_t i = 0; i < exz.size(); ++i) put("obs_p", call(obsConstraint, x, y, u0, u1, t, true));
                mexz_comp_idx = mexz_partial_idx = 0;
                for (size_t i = 0; i < mexz.size(); ++i) put("saa", call(saaConstraint, x, y, u0, u1, t, false));
                // the partial branch reads the radius at mexz_comp_idx (:287): keep both counters in step
                for (size_t i = 0; i < mexz.size(); ++i) {
                    mexz_comp_idx = mexz_partial_idx;
                    put("saa_p", call(saaConstraint, x, y, u0, u1, t, true));
                }
                for (size_t i = 0; i < mexz.size(); ++i) {
                    const ETOL::state_t& tv = std::get<1>(mexz[i]);
                    const ETOL::state_t& xv = std::get<2>(mexz[i])[0];
                    const ETOL::state_t& yv = std::get<2>(mexz[i])[1];
                    put("interp_hdr", {ETOL::TrajectoryOptimizer::linear_interpolation<double>(t, tv, xv),
                                       ETOL::TrajectoryOptimizer::linear_interpolation<double>(t, tv, yv)});
                    put("interp_ex", {linear_interpolation(t, tv, xv), linear_interpolation(t, tv, yv)});
                }
            }
        } else if (key == "traj") {
            const int R = (int)rd(), C = (int)rd();
            ETOL::traj_t tr;
            for (int r = 0; r < R; ++r) {
                const double t = rd();
                ETOL::state_t s;
                for (int c = 0; c < C; ++c) s.push_back(rd());
                tr.push_back({t, s});
            }
            std::vector<size_t> idxs;
            std::vector<double> sc, of;
            std::cin >> key; for (int n = (int)rd(); n > 0; --n) idxs.push_back((size_t)rd());
            std::cin >> key; for (int n = (int)rd(); n > 0; --n) sc.push_back(rd());
            std::cin >> key; for (int n = (int)rd(); n > 0; --n) of.push_back(rd());
            ETOL::traj_t ex = ETOL::TrajectoryOptimizer::extractTraj(tr, idxs);
            for (auto& e : ex) { std::vector<double> row{e.first}; row.insert(row.end(), e.second.begin(), e.second.end()); put("extract", row); }
            ETOL::traj_t s2 = tr;
            ETOL::TrajectoryOptimizer::scaleTraj(&s2, sc);
            for (auto& e : s2) { std::vector<double> row{e.first}; row.insert(row.end(), e.second.begin(), e.second.end()); put("scale", row); }
            ETOL::traj_t o2 = tr;
            ETOL::TrajectoryOptimizer::offsetTraj(&o2, of);
            for (auto& e : o2) { std::vector<double> row{e.first}; row.insert(row.end(), e.second.begin(), e.second.end()); put("offset", row); }
        } else {
            std::fprintf(stderr, "ref_driver: unknown section %s\n", key.c_str());
            return 2;
        }
    }
    return 0;
}
