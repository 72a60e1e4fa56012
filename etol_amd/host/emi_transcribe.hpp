// emi_transcribe.hpp -- mi355x::Prob (the transcribed ETOL problem) -> NlpProblem.
// Split out of eMI355X::solve() so that the unit tests can exercise the NLP
// iteration on the same problem construction the eSolver uses.
#ifndef ETOL_MI355X_EMI_TRANSCRIBE_HPP_
#define ETOL_MI355X_EMI_TRANSCRIBE_HPP_

#include <vector>

#include <ETOL/eMI355X.hpp>

#include "emi_nlp.hpp"

namespace ETOL {
namespace mi355x {

// Variable bounds per node (state/control boxes intersected with the event
// bounds at the first and last node), path-row bounds, mesh.
NlpProblem make_nlp(const Prob& P, NlpEvaluator* ev);
// zeros unless Prob::guess_* were pre-filled (reference ePSOPT.cpp:47-56)
std::vector<double> initial_guess(const Prob& P);
// positions of the nodes along a shortest path through the free space of the static keep-outs (the last cold-start guess of solve()):
// false if the problem has no such rows or no route exists
// clearance_weight x span: the clearance at which a step of the route costs twice its length (0: the shortest route)
bool planned_path_guess(const Prob& P, double* xs, double* ys, double clearance_weight = 0.01);
// variable scales of Alg::scaling = "automatic": max(|lower|, |upper|) per state / control (1 without a finite bound)
std::vector<double> bound_scales(const Prob& P);

}  // namespace mi355x
}  // namespace ETOL
#endif
