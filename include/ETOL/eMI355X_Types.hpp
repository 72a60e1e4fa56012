// eMI355X_Types.hpp -- what the std::any arguments of ETOL callbacks carry for eMI355X.
//
// Every eSolver fixes the content of the callback anys in its own *_Types.hpp
// (ePSOPT: adouble* in, adouble / fout_psopt_t out, reference
// include/ETOL/ePSOPT_Types.hpp:20 and src/ePSOPT/ePSOPT.cpp:196-205,256-269;
// eGurobi: GRBVar in, GRBLinExpr out; "functions must be created with the eSolver
// datatypes", reference src/docs/source/tutorials/vgp.rst:155).
//
// User closures are host code and cannot run on the GPU, and ePSOPT's pattern of
// calling them once per node per NLP evaluation is exactly the cost this backend
// removes.  eMI355X therefore calls each callback ONCE, inside setup() (as the
// MILP eSolvers do to build their expressions, reference src/eGurobi/eGurobi.cpp:
// 149-157), with
//     x[i], u[j] : mi355x::Var      (handle of state i / control j; alias mi355x::Symbol)
//     k          : mi355x::Var      (kind TIME)
//     dt         : double
// and expects back
//     objective      -> mi355x::ModelTerm with row == -1,  or the mi355x::Var of the integrand
//     gradient[i]    -> mi355x::ModelTerm with row == i,   or the mi355x::Var of d x_i / dt
//     constraints[c] -> fout_mi355x_t      (keep-out rows from the library's row kinds, in path-row order)
//                       or fout_mi355x_vars_t  (rows computed with the handles, traced like the model)
// A ModelTerm names one of the hand-written device models (include/emi355x.h,
// EMI_MODEL_*) and its parameter block; Vars are traced expressions that are
// differentiated and compiled into the same kernels at setup(); the rows of a
// fout_mi355x_t become the path table of the device evaluator.  A wrong type in an any ends, like in
// ePSOPT, in errorHandler(): message on stderr and exit(EXIT_FAILURE).
#ifndef ETOL_MI355X_EMI355X_TYPES_HPP_
#define ETOL_MI355X_EMI355X_TYPES_HPP_

#include <array>
#include <list>
#include <string>
#include <vector>

#include <ETOL/ETOL_Types.hpp>
#include <emi355x.h>

namespace ETOL {
namespace mi355x {

// What x[i], u[j] and k hold when eMI355X calls a callback: a handle into the expression trace of
// the current setup().  A callback may either ignore the arithmetic and return a descriptor of a
// built-in device model (mi355x::objective / mi355x::derivative), or compute with the handles --
// +, -, *, /, sin, cos, tan, exp, log, sqrt, pow(x, c), max, min, abs, interp1 -- and return the resulting Var.  In the
// second case eMI355X differentiates the recorded expressions symbolically (Jacobian, cost
// gradient, Lagrangian Hessian), generates a model struct for the hand-written kernel templates
// and compiles them for gfx950 at setup() (hiprtc): any smooth user model runs in the same
// kernels as the built-in ones.  `kind`/`index` identify input handles (states, controls, time).
class Var {
 public:
    enum Kind { STATE = 0, CONTROL = 1, TIME = 2, EXPR = 3 };
    Kind kind = EXPR;
    size_t index = 0;
    int node = -1;                       // node of the active trace (-1: not part of a trace)
    Var() {}
    Var(double constant);                // NOLINT: numbers mix freely with handles
    Var(Kind k, size_t i);               // input handle (registers it in the active trace)
};
typedef Var Symbol;                      // the name used by descriptor-only callbacks

Var operator+(const Var& a, const Var& b);
Var operator-(const Var& a, const Var& b);
Var operator*(const Var& a, const Var& b);
Var operator/(const Var& a, const Var& b);
Var operator-(const Var& a);
Var sin(const Var& a);
Var cos(const Var& a);
Var tan(const Var& a);
Var exp(const Var& a);
Var log(const Var& a);
Var sqrt(const Var& a);
Var pow(const Var& a, double c);
Var max(const Var& a, const Var& b);     // piecewise-linear operations: derivative of the active branch
Var min(const Var& a, const Var& b);
Var abs(const Var& a);
// linear interpolation of a waypoint table at t (constant outside the table), e.g. the centre of a moving
// exclusion zone at the node time (reference etol_psopt_example1.cpp:233-241)
Var interp1(const std::vector<double>& t_table, const std::vector<double>& v_table, const Var& t);

struct ModelTerm {
    int model = -1;                 // EMI_MODEL_*
    std::vector<double> params;     // model parameter block (emi_model_dims)
    int row = -1;                   // -1: integrand cost; i >= 0: d/dt of state i
};
inline ModelTerm objective(int model, const std::vector<double>& params = {}) {
    ModelTerm t;
    t.model = model;
    t.params = params;
    t.row = -1;
    return t;
}
inline ModelTerm derivative(int model, int state, const std::vector<double>& params = {}) {
    ModelTerm t;
    t.model = model;
    t.params = params;
    t.row = state;
    return t;
}

struct TrackTable {                 // waypoints of one moving keep-out
    double radius = 0;
    state_t t, x, y;
};

// Result of a constraint callback: path rows in evaluation order.
struct PathBlock {
    std::vector<std::array<double, EMI_PATH_REC>> rows;   // EMI_PATH_ELLIPSE / EMI_PATH_DISC records
    std::vector<TrackTable> tracks;                        // one EMI_PATH_TRACK row each, after `rows`
    size_t px = 0, py = 1;                                 // states the rows act on (from the Symbols)
};

// One ellipse row per polygon edge, edges in corner order with wrap-around
// (reference src/Examples/PSOPT/etol_psopt_example1.cpp:159-186).
PathBlock ellipse_rows(const std::vector<border_t>& zones, const Symbol& sx, const Symbol& sy);
// One fixed disc row per (xc, yc, r).
PathBlock disc_rows(const std::vector<std::array<double, 3>>& discs, const Symbol& sx, const Symbol& sy);
// One moving-disc row per track (reference etol_psopt_example1.cpp:208-224, 232-249);
// waypoint values 0 and 1 are the centre coordinates.
PathBlock track_rows(const std::list<track_t>& tracks, const Symbol& sx, const Symbol& sy);

}  // namespace mi355x

typedef mi355x::PathBlock fout_mi355x_t;
// Constraint rows computed with the handles, the counterpart of ePSOPT's fout_psopt_t (a vector of adoubles,
// include/ETOL/ePSOPT_Types.hpp:20): every entry is one path row c(x, t); the rows of one problem may depend
// on two of the states (and on time).
typedef std::vector<mi355x::Var> fout_mi355x_vars_t;

}  // namespace ETOL
#endif
