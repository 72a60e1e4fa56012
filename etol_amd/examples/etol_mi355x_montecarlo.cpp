// etol_mi355x_montecarlo.cpp -- Monte-Carlo of vehicle-guidance problems over random keep-out fields.
//
// Config 4 of the scope table in miniature: independent scenarios, sharded over
// processes (one per GPU: RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run or mpirun set them) and,
// inside a process, over host threads that each own an ETOL::eMI355X (one device context per thread: the
// per-iteration kernels of a single solve leave most of the GPU idle, concurrent solves fill it).
// No communication while solving; every rank writes one summary line per scenario.
//
//   etol_mi355x_montecarlo <scenarios> <nsteps> <keep-outs per scenario> <threads> [traced]
//
// Scenario s draws its discs from SplitMix64(0xE70100 + 0x100*4 + s): centres U([1,9]^2), radii U(0.2,0.6),
// redrawn while they cover the start or the goal.  Model: the 6-state planar quadrotor, as a built-in
// device model or (5th argument "traced") written with mi355x::Var arithmetic and compiled at setup().
#include <ETOL/eMI355X.hpp>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mx = ETOL::mi355x;

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform(double a, double b) { return a + (b - a) * ((next() >> 11) * (1.0 / 9007199254740992.0)); }
};

std::vector<std::array<double, 3>> scenario_discs(int s, int ndiscs) {
    SplitMix64 g(0xE70100ull + 0x100ull * 4 + (uint64_t)s);
    std::vector<std::array<double, 3>> d;
    while ((int)d.size() < ndiscs) {
        const double x = g.uniform(1, 9), y = g.uniform(1, 9), r = g.uniform(0.2, 0.6);
        if (std::hypot(x - 1, y - 1) < r + 0.4 || std::hypot(x - 8, y - 6) < r + 0.4) continue;
        d.push_back({x, y, r});
    }
    return d;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct Result {
    int scenario = -1, rc = 0, nodes = 0, iterations = 0;
    double cost = 0, seconds = 0;
    std::string message;
};

Result solve_scenario(int s, int nsteps, int ndiscs, int device, bool traced) {
    Result R;
    R.scenario = s;
    const auto t0 = std::chrono::steady_clock::now();
    ETOL::eMI355X solver;
    ETOL::TrajectoryOptimizer* t = &solver;
    const double tf = 4.0;
    t->setNSteps(nsteps); t->setDt(tf / nsteps); t->setNStates(6); t->setNControls(2);
    t->setX0({1, 1, 0, 0, 0, 0}); t->setXf({8, 6, 0, 0, 0, 0}); t->setXtol({0.01, 0.01, 0.01, 0.05, 0.05, 0.05});
    t->setXlower({0, 0, -1.2, -6, -6, -4}); t->setXupper({10, 10, 1.2, 6, 6, 4});
    t->setUlower({0, -1}); t->setUupper({25, 1});
    t->setMaximize(false);
    const std::vector<double> mp = {1.0, 0.01, 9.81, 1.0, 1.0};     // m, I, g, cost weights
    ETOL::f_t obj = [mp, traced](F_ARGS) -> ETOL::scalar_t {
        if (!traced) return mx::objective(EMI_MODEL_QUADROTOR2D, mp);
        const mx::Var T = std::any_cast<mx::Var>(u.at(0)), tau = std::any_cast<mx::Var>(u.at(1));
        return T * T + tau * tau;
    };
    std::vector<ETOL::f_t> grad(6);
    std::vector<ETOL::f_t*> gp;
    for (int i = 0; i < 6; ++i) {
        grad[i] = [mp, i, traced](F_ARGS) -> ETOL::scalar_t {
            if (!traced) return mx::derivative(EMI_MODEL_QUADROTOR2D, i, mp);
            const mx::Var th = std::any_cast<mx::Var>(x.at(2)), T = std::any_cast<mx::Var>(u.at(0));
            switch (i) {
                case 0: return std::any_cast<mx::Var>(x.at(3));
                case 1: return std::any_cast<mx::Var>(x.at(4));
                case 2: return std::any_cast<mx::Var>(x.at(5));
                case 3: return -(T / mp[0]) * mx::sin(th);
                case 4: return (T / mp[0]) * mx::cos(th) - mp[2];
                default: return std::any_cast<mx::Var>(u.at(1)) / mp[1];
            }
        };
        gp.push_back(&grad[i]);
    }
    t->setObjective(&obj);
    t->setGradient(gp);
    const auto discs = scenario_discs(s, ndiscs);
    for (int i = 0; i < ndiscs; ++i)
        t->addParams({std::pair<PARAM_PAIR>("disc_" + std::to_string(i), {ETOL::var_t::CONTINUOUS, -1000., 0., 0., tf})});
    ETOL::f_t obs = [discs](F_ARGS) -> ETOL::scalar_t {
        return mx::disc_rows(discs, std::any_cast<mx::Symbol>(x.at(0)), std::any_cast<mx::Symbol>(x.at(1)));
    };
    if (ndiscs > 0) t->setConstraints({&obs});
    solver.getAlgorithm()->device = device;
    t->setup();
    solver.getAlgorithm()->nlp_tolerance = 1e-7;
    solver.getAlgorithm()->nlp_iter_max = 400;
    solver.getAlgorithm()->mesh_refinement = "none";
    solver.getAlgorithm()->print_level = env_int("EMI_MC_PRINT_LEVEL", 0);
    t->solve();
    const mx::Sol* sol = solver.getSolution();
    R.rc = sol->error_flag;
    R.message = sol->error_msg;
    R.nodes = (int)sol->nodes;
    R.iterations = sol->nlp_iterations_total;
    R.cost = sol->error_flag ? 0.0 : t->getScore();
    t->close();
    R.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return R;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        printf("Usage: %s <scenarios> <nsteps> <keep-outs per scenario> <threads> [traced]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const int nscen = atoi(argv[1]), nsteps = atoi(argv[2]), ndiscs = atoi(argv[3]), nthreads = std::max(1, atoi(argv[4]));
    const bool traced = argc > 5 && std::string(argv[5]) == "traced";
    const int rank = env_int("RANK", 0), world = std::max(1, env_int("WORLD_SIZE", 1)), device = env_int("LOCAL_RANK", 0);
    // static block partition, as etol_amd/batch.py shard_range: scenario s -> rank floor(s * world / nscen)
    int lo = (int)((long long)rank * nscen / world), hi = (int)((long long)(rank + 1) * nscen / world);
    if (env_int("EMI_MC_ONLY", -1) >= 0) { lo = env_int("EMI_MC_ONLY", 0); hi = lo + 1; }   // diagnostics: one scenario

    std::vector<Result> results(hi - lo);
    std::atomic<int> next(lo);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int th = 0; th < nthreads; ++th)
        pool.emplace_back([&] {
            for (int s = next++; s < hi; s = next++) results[s - lo] = solve_scenario(s, nsteps, ndiscs, device, traced);
            ETOL::eMI355X::releaseDevices();     // this thread's idle device contexts
        });
    for (auto& th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    int ok = 0;
    double iters = 0;
    for (const Result& r : results) {
        printf("scenario %4d  rank %d  rc %d  nodes %d  iterations %4d  cost %.8f  %.2f s%s%s\n", r.scenario, rank, r.rc, r.nodes,
               r.iterations, r.cost, r.seconds, r.rc ? "  " : "", r.rc ? r.message.c_str() : "");
        ok += r.rc == 0;
        iters += r.iterations;
    }
    printf("{\"rank\": %d, \"world\": %d, \"scenarios\": %d, \"solved\": %d, \"nodes\": %d, \"keepouts\": %d, \"threads\": %d, "
           "\"model\": \"%s\", \"wall_s\": %.3f, \"solves_per_s\": %.3f, \"mean_iterations\": %.1f}\n",
           rank, world, hi - lo, ok, nsteps + 1, ndiscs, nthreads, traced ? "traced" : "built-in", wall,
           (hi - lo) / wall, results.empty() ? 0.0 : iters / results.size());
    return ok == hi - lo ? EXIT_SUCCESS : 2;
}
