// etol_mi355x_example1.cpp -- loads an ETOL configuration XML and solves it with eMI355X.
//
// The eMI355X twin of the reference's src/Examples/PSOPT/etol_psopt_example1.cpp
// (same OCP, same call sequence, same outputs):
//    minimize    integral (u0^2 + u1^2) dt
//    subject to  dx/dt = u0,  dy/dt = u1
//                (x,y) outside every exclusion zone (one ellipse per polygon edge)
//                (x,y) outside every moving exclusion disc
// Callbacks use the eMI355X datatypes (include/ETOL/eMI355X_Types.hpp) and are called once, at
// setup().  Objective and dynamics are written as in the reference example -- arithmetic on the
// solver's scalar type (there adouble, here mi355x::Var): eMI355X records the expressions,
// differentiates them and compiles the model for the GPU.  (A callback may instead name one of the
// library's hand-written kernels: return mx::objective(EMI_MODEL_POINTMASS2D) /
// mx::derivative(EMI_MODEL_POINTMASS2D, i).)
#include <ETOL/eMI355X.hpp>

#include <iostream>

namespace mx = ETOL::mi355x;

void editAlgo(ETOL::TrajectoryOptimizer* t);
ETOL::scalar_t objFunction(F_ARGS);
ETOL::scalar_t dxdt(F_ARGS);
ETOL::scalar_t dydt(F_ARGS);
ETOL::f_t obsConstraint(ETOL::TrajectoryOptimizer* t);
ETOL::f_t saaConstraint(ETOL::TrajectoryOptimizer* t);
std::string paramName(const std::string& name, size_t i, size_t j, size_t k);

int main(int argc, char** argv) {
    if (argc != 2) {
        printf("Usage: %s <ETOL configuration xml filepath>\n", argv[0]);
        exit(EXIT_FAILURE);
    }
    ETOL::eMI355X solver;
    ETOL::TrajectoryOptimizer* t = &solver;

    t->loadConfigs(argv[1]);
    t->printConfigs();

    t->setMaximize(false);
    ETOL::f_t obj = &objFunction;
    t->setObjective(&obj);
    ETOL::f_t xdot = &dxdt, ydot = &dydt;
    t->setGradient({&xdot, &ydot});

    ETOL::f_t obs = obsConstraint(t);
    ETOL::f_t saa = saaConstraint(t);
    t->setConstraints({&obs, &saa});

    t->setup();
    editAlgo(t);
    t->debug();
    t->solve();

    printf("\n!!!!!!!!!!!!!!!!!Results!!!!!!!!!!!!!!!!!\n");
    printf("Minimization Score:\t%f\n", t->getScore());
    printf("State variables saved in %s\n",
           ETOL::TrajectoryOptimizer::save(t->getXtraj(), "state_mi355x1.csv").c_str());
    printf("Control variables saved in %s\n",
           ETOL::TrajectoryOptimizer::save(t->getUtraj(), "control_mi355x1.csv").c_str());
    t->close();
    printf("\n!!!!!!!!!!!!!!Graceful Exit!!!!!!!!!!!!!!\n");
    return EXIT_SUCCESS;
}

// per-solver knobs through the solver's own handle, like the ePSOPT example
void editAlgo(ETOL::TrajectoryOptimizer* t) {
    ETOL::eMI355X* ptr = dynamic_cast<ETOL::eMI355X*>(t);
    if (!ptr) {
        std::cout << "EditAlgo only works for eMI355X!" << std::endl;
        exit(EXIT_FAILURE);
    }
    mx::Alg* algo = ptr->getAlgorithm();
    algo->nlp_tolerance = 1.e-8;
    algo->defect_scaling = "jacobian-based";       // as the reference example asks of PSOPT (etol_psopt_example1.cpp:91)
    algo->max_cpu_time = 100;
}

// reference etol_psopt_example1.cpp:101-138, with mx::Var in place of adouble
ETOL::scalar_t objFunction(F_ARGS) {
    const mx::Var u0 = std::any_cast<mx::Var>(u.at(0)), u1 = std::any_cast<mx::Var>(u.at(1));
    return u0 * u0 + u1 * u1;
}
ETOL::scalar_t dxdt(F_ARGS) { return std::any_cast<mx::Var>(u.at(0)); }
ETOL::scalar_t dydt(F_ARGS) { return std::any_cast<mx::Var>(u.at(1)); }

ETOL::f_t obsConstraint(ETOL::TrajectoryOptimizer* t) {
    const double tspan = t->getDt() * t->getNSteps();
    const std::vector<ETOL::border_t>* zones = t->getObstacles_Raw();
    size_t i = 0;
    for (const ETOL::border_t& zone : *zones) {
        for (size_t j = 0; j < zone.size(); ++j)
            t->addParams({std::pair<PARAM_PAIR>(paramName("side", i, j, 0),
                                                {ETOL::var_t::CONTINUOUS, -1000., 0., 0., tspan})});
        ++i;
    }
    // One row per polygon edge, computed with the handles like the reference's callback computes it with adoubles
    // (etol_psopt_example1.cpp:153-190): eMI355X traces the arithmetic, differentiates it and compiles it into the
    // kernels.  (mx::ellipse_rows(*zones, x0, x1) would hand the same rows to the library's built-in row kind.)
    return [zones](F_ARGS) -> ETOL::scalar_t {
        try {
            ETOL::fout_mi355x_vars_t rows;
            const mx::Var px = std::any_cast<mx::Var>(x.at(0)), py = std::any_cast<mx::Var>(x.at(1));
            for (const ETOL::border_t& zone : *zones) {
                const std::vector<ETOL::corner_t> corner(zone.begin(), zone.end());
                for (size_t i = 0; i < corner.size(); ++i) {
                    const ETOL::corner_t& p0 = corner[i];
                    const ETOL::corner_t& p1 = corner[(i + 1) % corner.size()];
                    double e[EMI_PATH_REC];          // {kind, xc, yc, cos, sin, a^2, b^2, -} of the edge's ellipse
                    emi_edge_ellipse(p0.at(0), p0.at(1), p1.at(0), p1.at(1), e);
                    const mx::Var ox = px - e[1], oy = py - e[2];
                    const mx::Var delx = e[3] * ox - e[4] * oy, dely = e[4] * ox + e[3] * oy;
                    rows.push_back(e[5] * e[6] - (e[6] * mx::pow(delx, 2.) + e[5] * mx::pow(dely, 2.)));
                }
            }
            return rows;
        } catch (std::bad_any_cast& e) {
            std::cout << "Error in obs" << std::endl << e.what() << std::endl;
            exit(EXIT_FAILURE);
        }
    };
}

ETOL::f_t saaConstraint(ETOL::TrajectoryOptimizer* t) {
    const double tspan = t->getDt() * t->getNSteps();
    const std::list<ETOL::track_t>* tracks = t->getTracks();
    for (size_t i = 0; i < tracks->size(); ++i)
        t->addParams({std::pair<PARAM_PAIR>(paramName("ball", i, 0, 0),
                                            {ETOL::var_t::CONTINUOUS, -1000., 0., 0., tspan})});
    // One row per moving zone, again as arithmetic on the handles: the zone's centre is its waypoint table
    // interpolated at the node time k (reference etol_psopt_example1.cpp:233-247).
    // (mx::track_rows(*tracks, x0, x1) would hand the zones to the library's built-in moving-disc row kind.)
    return [tracks](F_ARGS) -> ETOL::scalar_t {
        try {
            ETOL::fout_mi355x_vars_t rows;
            const mx::Var px = std::any_cast<mx::Var>(x.at(0)), py = std::any_cast<mx::Var>(x.at(1));
            const mx::Var time = std::any_cast<mx::Var>(k);
            for (const ETOL::track_t& trk : *tracks) {
                std::vector<double> tw, xw, yw;
                for (const ETOL::traj_elem_t& wp : trk.trajectory) {
                    tw.push_back(wp.first);
                    xw.push_back(wp.second.at(0));
                    yw.push_back(wp.second.at(1));
                }
                const mx::Var ox = px - mx::interp1(tw, xw, time), oy = py - mx::interp1(tw, yw, time);
                rows.push_back(trk.radius * trk.radius - (ox * ox + oy * oy));
            }
            return rows;
        } catch (std::bad_any_cast& e) {
            std::cout << "Error in saa" << std::endl << e.what() << std::endl;
            exit(EXIT_FAILURE);
        }
    };
}

std::string paramName(const std::string& name, size_t i, size_t j, size_t k) {
    return name + "_" + std::to_string(i) + "_" + std::to_string(j) + "_" + std::to_string(k);
}
