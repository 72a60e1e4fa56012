// ETOL_Types.hpp -- the type vocabulary shared by every ETOL eSolver.
//
// Interface re-declaration for the standalone eMI355X build: the NAMES and
// their meaning are the contract (reference include/ETOL/ETOL_Types.hpp:25-117);
// inside the ETOL tree the reference's own header is used instead
// (INTEGRATION.md).  Callback arguments travel as std::any, and what each any
// holds is decided by the eSolver (for eMI355X: include/ETOL/eMI355X_Types.hpp).
#ifndef ETOL_MI355X_ETOL_TYPES_HPP_
#define ETOL_MI355X_ETOL_TYPES_HPP_

#include <any>
#include <array>
#include <cmath>
#include <functional>
#include <list>
#include <map>
#include <string>
#include <utility>
#include <vector>

// argument pack of every ETOL callback; user code spells callbacks `f(F_ARGS)`
#define PARAM_PAIR ETOL::param_name_t, ETOL::param_configs_t
#define F_ARGS                                                                   \
    ETOL::vector_t x, ETOL::vector_t u, ETOL::vector_t params,                   \
        std::vector<std::string> pnames, std::any k, std::any dt

namespace ETOL {

// ---- variables ---------------------------------------------------------------
enum var_t { CONTINUOUS = 0, INTERGER = 1, BINARY = 2 };  // (sic) reference spelling

using param_name_t = std::string;
struct param_configs_t {
    var_t varType = var_t::CONTINUOUS;
    double lbnd = 0.;    // lower bound of the row / variable
    double ubnd = 0.;    // upper bound
    double tStart = 0.;  // active from
    double tStop = 0.;   // active until
};
using param_t = std::pair<PARAM_PAIR>;
using paramset_t = std::map<PARAM_PAIR>;   // name-sorted: eSolvers rely on this order

// ---- geometry -----------------------------------------------------------------
using coord_t = std::pair<double, double>;
using line_t = std::vector<coord_t>;
using lines_t = std::vector<line_t>;
using corner_t = std::array<double, 3>;
struct edge_prop_ {
    double slope = 0.;
    double length = 0.;
};
using edge_prop_t = edge_prop_;
using edge_t = std::pair<corner_t, edge_prop_t>;
using seg_t = std::vector<edge_t>;
using closure_t = std::pair<std::vector<seg_t>, std::vector<seg_t>>;
using border_t = std::list<corner_t>;      // polygon as an ordered corner list
struct boundary_t {
    border_t lower;
    border_t upper;
};
using region_t = std::list<boundary_t>;    // convex pieces of one exclusion zone

// ---- trajectories ---------------------------------------------------------------
using state_t = std::vector<double>;
using state_var_t = std::vector<var_t>;
using traj_elem_t = std::pair<double, state_t>;   // (time, values)
using traj_t = std::vector<traj_elem_t>;
struct track_t {
    double radius = 0.0;          // keep-out radius around the moving object
    traj_t trajectory = traj_t(); // its waypoints
};

// ---- callbacks --------------------------------------------------------------------
using scalar_t = std::any;
using vector_t = std::vector<scalar_t>;
using f_t = std::function<scalar_t(F_ARGS)>;

}  // namespace ETOL

#endif  // ETOL_MI355X_ETOL_TYPES_HPP_
