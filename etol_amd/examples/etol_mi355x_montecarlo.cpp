// etol_mi355x_montecarlo.cpp -- Monte-Carlo of vehicle-guidance problems over random keep-out fields.
//
// Config 4 of the scope table in miniature: independent scenarios, sharded over
// processes (one per GPU: RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run or mpirun set them) and,
// inside a process, over host threads that each own an ETOL::eMI355X (one device context per thread: the
// per-iteration kernels of a single solve leave most of the GPU idle, concurrent solves fill it).
// No communication while solving; every rank writes one summary line per scenario, then the solved
// trajectories are gathered on rank 0 over RCCL (emi_comm_gather, include/emi355x.h): the only collective of
// the path.  The RCCL id travels through a file that rank 0 writes (single node; EMI_COMM_FILE overrides the
// name).  With EMI_MC_SAVE=<dir> rank 0 writes every gathered trajectory as ETOL CSV files there.
//
//   etol_mi355x_montecarlo <scenarios> <nsteps> <keep-outs per scenario> <threads> [traced]
//
// Scenario s draws its discs from SplitMix64(0xE70100 + 0x100*4 + s): centres U([1,9]^2), radii U(0.2,0.6),
// redrawn while they cover the start or the goal.  Model: the 6-state planar quadrotor, as a built-in
// device model or (5th argument "traced") written with mi355x::Var arithmetic and compiled at setup().
#include <ETOL/eMI355X.hpp>
#include <emi355x.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
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
    std::vector<double> states, controls, time;   // [6][nodes], [2][nodes], [nodes] of a solved scenario
    std::vector<mx::Sol::NlpRun> runs;            // the NLP solves of this scenario (ladder rungs, restarts), in order
};

// one gathered record per scenario: header {scenario, rc, nodes, iterations, cost} then X[6][M], U[2][M], t[M]
constexpr int REC_HEAD = 5;
size_t record_doubles(int M) { return REC_HEAD + (size_t)(6 + 2 + 1) * M; }

// The 128-byte RCCL id goes from rank 0 to the others through a file (one node, shared /tmp), AT START-UP, before any
// solving: the ranks of a launch start within seconds of each other, so a bounded wait means something there (after the
// solves rank 0 may be minutes behind the others).  The name carries what the launcher gives to tell runs apart (launcher
// pid, MASTER_PORT, torchrun's run id and restart count).  The file holds, behind the id, that same launch key and rank 0's
// start time: a reader takes the id only from a file whose key is its own AND whose writer started within
// EMI_COMM_SKEW_S (15 s) of the reader -- a leftover of a crashed earlier launch under the same name (a quick relaunch from
// the same shell) fails the second test, where a test on the file's age alone would let it through and the ranks would
// then wait in ncclCommInitRank on different ids with no diagnostic.  Rank 0 removes leftovers before it writes and its own
// file once the communicator exists on every rank (comm_file_done).
std::string comm_launch_key() {
    auto env = [](const char* n, const char* d) { const char* v = getenv(n); return std::string(v ? v : d); };
    return std::to_string((long)getppid()) + "_" + env("MASTER_PORT", "0") + "_" + env("TORCHELASTIC_RUN_ID", "run") + "_" +
           env("TORCHELASTIC_RESTART_COUNT", "0");
}
std::string comm_file_path() {
    if (getenv("EMI_COMM_FILE")) return getenv("EMI_COMM_FILE");
    return "/tmp/emi_comm_" + comm_launch_key() + ".id";
}
constexpr size_t COMM_KEY_BYTES = 160;
bool exchange_id(int rank, char* id, double process_start_s, std::string* why) {
    const std::string path = comm_file_path(), key = comm_launch_key();
    if (rank == 0) {
        unlink(path.c_str());                            // a leftover of an earlier run under the same name
        if (emi_comm_unique_id(id) != EMI_OK) { *why = std::string("emi_comm_unique_id: ") + emi_comm_last_error(nullptr); return false; }
        char keybuf[COMM_KEY_BYTES] = {0};
        strncpy(keybuf, key.c_str(), COMM_KEY_BYTES - 1);
        const std::string tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, EMI_COMM_ID_BYTES, f) != EMI_COMM_ID_BYTES || fwrite(keybuf, 1, COMM_KEY_BYTES, f) != COMM_KEY_BYTES ||
            fwrite(&process_start_s, sizeof process_start_s, 1, f) != 1) {
            *why = "cannot write " + tmp;
            if (f) fclose(f);
            return false;
        }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str()) != 0) { *why = "cannot rename " + tmp; return false; }
        return true;
    }
    const int timeout_s = std::max(1, env_int("EMI_COMM_TIMEOUT_S", 120));
    const double skew_s = std::max(1, env_int("EMI_COMM_SKEW_S", 15));
    bool stale = false;
    for (int tries = 0; tries < timeout_s * 10; ++tries) {
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            char keybuf[COMM_KEY_BYTES] = {0};
            double writer_start = 0;
            const bool whole = fread(id, 1, EMI_COMM_ID_BYTES, f) == EMI_COMM_ID_BYTES && fread(keybuf, 1, COMM_KEY_BYTES, f) == COMM_KEY_BYTES &&
                               fread(&writer_start, sizeof writer_start, 1, f) == 1;
            fclose(f);
            keybuf[COMM_KEY_BYTES - 1] = 0;
            if (whole && key == keybuf && std::fabs(writer_start - process_start_s) <= skew_s) return true;
            if (whole) stale = true;                     // not this launch's file: wait for rank 0 to replace it
        }
        usleep(100000);
    }
    *why = "no RCCL id from rank 0 in " + path + " after " + std::to_string(timeout_s) + " s (EMI_COMM_TIMEOUT_S)" +
           (stale ? "; only a file of another launch is there (launch key or start time differ: EMI_COMM_SKEW_S)" : "") +
           ": is rank 0 running, and is /tmp shared between the ranks?";
    return false;
}
void comm_file_done(int rank) {
    if (rank == 0 && !getenv("EMI_COMM_FILE_KEEP")) unlink(comm_file_path().c_str());
}

Result solve_scenario(int s, int nsteps, int ndiscs, int device, bool traced, const std::shared_ptr<mx::KktBatcher>& batcher = nullptr) {
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
    // the reference's own NLP tolerance (ePSOPT.cpp:67: 1e-6).  Rounds 1 - 3 and the first record runs of round 4 used 1e-7; at 1e-6 the
    // 1024-node rung of the 64-scenario set takes 1051 iterations instead of 1164 (profiles/r04_notes.md section 15)
    solver.getAlgorithm()->nlp_tolerance = getenv("EMI_MC_TOL") ? atof(getenv("EMI_MC_TOL")) : 1e-6;
    solver.getAlgorithm()->nlp_iter_max = 400;
    solver.getAlgorithm()->mesh_refinement = "none";
    if (getenv("EMI_MC_SCALING")) solver.getAlgorithm()->scaling = getenv("EMI_MC_SCALING");                 // "none" (default) / "automatic"
    if (getenv("EMI_MC_LADDER_RATIO")) solver.getAlgorithm()->ladder_ratio = env_int("EMI_MC_LADDER_RATIO", 2);
    if (getenv("EMI_MC_RUNG_TOL")) solver.getAlgorithm()->rung_tolerance = atof(getenv("EMI_MC_RUNG_TOL"));
    if (getenv("EMI_MC_RUNG_PATIENCE")) solver.getAlgorithm()->rung_patience = env_int("EMI_MC_RUNG_PATIENCE", 0);                                   // iterations a ladder rung may take (0: nlp_iter_max)
    // iterations per scenario over all its meshes, rungs and restarts (0: no limit).  1000 by default here: in the 1024-scenario run of
    // config 4 one scenario spent 3287 iterations (91 of the run's 394 s) to end "locally infeasible" (profiles/r04_notes.md section 7)
    solver.getAlgorithm()->nlp_iter_budget = env_int("EMI_MC_BUDGET", 1000);
    if (getenv("EMI_MC_TARGET_PATIENCE")) solver.getAlgorithm()->target_patience = env_int("EMI_MC_TARGET_PATIENCE", 200);
    if (getenv("EMI_MC_PLAN")) solver.getAlgorithm()->plan_first_start = env_int("EMI_MC_PLAN", 2) == 2;     // (A/B: 2 planned route first (default), 1 second, 0 last)
    if (getenv("EMI_MC_PLAN_CLEARANCE")) solver.getAlgorithm()->plan_clearance = atof(getenv("EMI_MC_PLAN_CLEARANCE"));
    if (getenv("EMI_MC_PLAN")) solver.getAlgorithm()->plan_second_start = env_int("EMI_MC_PLAN", 1) != 0;     // (A/B: 0 = the planned route as the last cold-start attempt only)
    if (getenv("EMI_MC_DEFECT_SCALING")) solver.getAlgorithm()->defect_scaling = getenv("EMI_MC_DEFECT_SCALING");   // "state-based" (default) / "jacobian-based"
    if (getenv("EMI_MC_WARM_MU")) solver.getAlgorithm()->warm_mu_init = atof(getenv("EMI_MC_WARM_MU"));      // experiments
    if (getenv("EMI_MC_WARM_PATIENCE")) solver.getAlgorithm()->warm_patience = atoi(getenv("EMI_MC_WARM_PATIENCE"));
    if (getenv("EMI_MC_MU_RESTART")) solver.getAlgorithm()->mu_restart = atof(getenv("EMI_MC_MU_RESTART"));
    if (getenv("EMI_MC_WARM_PUSH")) solver.getAlgorithm()->warm_bound_push = atof(getenv("EMI_MC_WARM_PUSH"));
    solver.getAlgorithm()->print_level = env_int("EMI_MC_PRINT_LEVEL", 0);
    solver.getAlgorithm()->kkt_batcher = batcher;          // the Newton steps of every scenario in flight share their launches
    t->solve();
    const mx::Sol* sol = solver.getSolution();
    R.rc = sol->error_flag;
    R.message = sol->error_msg;
    R.nodes = (int)sol->nodes;
    R.iterations = sol->nlp_iterations_total;
    R.runs = sol->nlp_runs;
    R.cost = sol->error_flag ? 0.0 : t->getScore();
    if (!sol->error_flag) { R.states = sol->states; R.controls = sol->controls; R.time = sol->time; }
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

    // diagnostics: process-wide factorisation switches (emi_set_option "kkt_*"), e.g. EMI_MC_KKT_DEBUG=1 prints retries and
    // fallbacks on stderr, EMI_MC_KKT_STICKY=0 starts every factorisation at the nominal regularisation again
    const int kkt_debug = env_int("EMI_MC_KKT_DEBUG", 0), kkt_sticky = env_int("EMI_MC_KKT_STICKY", -1), kkt_primal = env_int("EMI_MC_KKT_PRIMAL", -1);
    const int kkt_chol = env_int("EMI_MC_KKT_CHOLESKY", -1), kkt_refine = env_int("EMI_MC_KKT_REFINE_EXP", -1);
    if (kkt_debug > 0 || kkt_sticky >= 0 || kkt_primal >= 0 || kkt_chol >= 0 || kkt_refine >= 0) {
        emi_ctx_t sw = nullptr;
        if (emi_create(device, &sw) == EMI_OK) {
            if (kkt_debug > 0) emi_set_option(sw, "kkt_debug", kkt_debug);
            if (kkt_sticky >= 0) emi_set_option(sw, "kkt_sticky_reg", kkt_sticky);
            if (kkt_primal >= 0) emi_set_option(sw, "kkt_primal_levels", kkt_primal);
            if (kkt_chol >= 0) emi_set_option(sw, "kkt_cholesky", kkt_chol);
            if (kkt_refine >= 0) emi_set_option(sw, "kkt_refine_exp", kkt_refine);
            emi_destroy(sw);
        }
    }
    char comm_id[EMI_COMM_ID_BYTES] = {0};
    const bool gather = env_int("EMI_MC_GATHER", 1) != 0;
    if (gather) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        std::string why;
        if (!exchange_id(rank, comm_id, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, &why)) {
            fprintf(stderr, "rank %d: RCCL id exchange: %s\n", rank, why.c_str());
            return EXIT_FAILURE;
        }
    }
    std::vector<Result> results(hi - lo);
    std::atomic<int> next(lo);
    // EMI_MC_BATCH = g > 0: the worker threads form g groups, each sharing a KktBatcher -- the factorisations and solves of a
    // group's scenarios go out as batched launches (emi_kkt_factor_batch); while one group's batch runs on the device the other
    // groups do their host work.  0: every thread launches for itself (the round-3 form).
    const int groups = std::max(0, std::min(env_int("EMI_MC_BATCH", 0), nthreads));
    std::vector<std::shared_ptr<mx::KktBatcher>> batchers;
    for (int g = 0; g < groups; ++g) {
        batchers.push_back(std::make_shared<mx::KktBatcher>());
        batchers.back()->flush_us = env_int("EMI_MC_FLUSH_US", 20000);
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int th = 0; th < nthreads; ++th)
        pool.emplace_back([&, th] {
            const std::shared_ptr<mx::KktBatcher> mine = groups > 0 ? batchers[th % groups] : nullptr;
            {
                mx::KktBatcher::Member member(mine);
                for (int s = next++; s < hi; s = next++) results[s - lo] = solve_scenario(s, nsteps, ndiscs, device, traced, mine);
            }
            ETOL::eMI355X::releaseDevices();     // this thread's idle device contexts
        });
    for (auto& th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t g = 0; g < batchers.size(); ++g)
        printf("batcher %zu: %ld factor calls carrying %ld factorisations, %ld solve calls carrying %ld solves, largest batch %d\n", g,
               batchers[g]->factor_calls, batchers[g]->factor_items, batchers[g]->solve_calls, batchers[g]->solve_items, batchers[g]->largest_batch);

    int ok = 0;
    double iters = 0;
    const bool show_runs = env_int("EMI_MC_RUNS", 0) != 0;
    for (const Result& r : results) {
        printf("scenario %4d  rank %d  rc %d  nodes %d  iterations %4d  cost %.8f  %.2f s%s%s", r.scenario, rank, r.rc, r.nodes,
               r.iterations, r.cost, r.seconds, r.rc ? "  " : "", r.rc ? r.message.c_str() : "");
        if (show_runs) {                                  // EMI_MC_RUNS=1: every NLP solve of the scenario as nodes:iterations ('!' = did not converge)
            printf("  runs");
            for (const auto& u : r.runs) printf(" %zu:%d%s", u.nodes, u.iterations, u.converged ? "" : "!");
        }
        printf("\n");
        ok += r.rc == 0;
        iters += r.iterations;
    }
    // where the iterations (and the solver's time) went: per mesh size, over all scenarios of this rank
    std::map<size_t, std::array<double, 14>> by_mesh;     // nodes -> NLP solves, iterations, seconds, not converged, the phases of Sol::NlpRun
    for (const Result& r : results)
        for (const auto& u : r.runs) {
            auto& a = by_mesh[u.nodes];
            const double v[14] = {1.0, (double)u.iterations, u.seconds, u.converged ? 0.0 : 1.0, u.t_eval, u.t_hess, u.t_factor, u.t_solve,
                                  u.t_lowrank, u.t_blocks, u.t_jt, u.t_matvec, (double)u.factorisations, (double)u.solves};
            for (int i = 0; i < 14; ++i) a[i] += v[i];
        }
    std::string mesh_json = "{";
    for (const auto& kv : by_mesh) {
        char buf[512];
        const auto& a = kv.second;
        snprintf(buf, sizeof buf,
                 "%s\"%zu\": {\"nlp_solves\": %.0f, \"iterations\": %.0f, \"seconds\": %.2f, \"not_converged\": %.0f, \"eval_s\": %.2f, \"hess_s\": %.2f, "
                 "\"factor_s\": %.2f, \"solve_s\": %.2f, \"lowrank_s\": %.2f, \"blocks_s\": %.2f, \"jt_s\": %.2f, \"matvec_s\": %.2f, "
                 "\"factorisations\": %.0f, \"solves\": %.0f}",
                 mesh_json.size() > 1 ? ", " : "", kv.first, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]);
        mesh_json += buf;
    }
    mesh_json += "}";
    printf("{\"by_mesh\": %s, ", mesh_json.c_str());
    printf("\"rank\": %d, \"world\": %d, \"scenarios\": %d, \"solved\": %d, \"nodes\": %d, \"keepouts\": %d, \"threads\": %d, "
           "\"model\": \"%s\", \"kkt_batch_groups\": %d, \"wall_s\": %.3f, \"solves_per_s\": %.3f, \"mean_iterations\": %.1f}\n",
           rank, world, hi - lo, ok, nsteps + 1, ndiscs, nthreads, traced ? "traced" : "built-in", groups, wall,
           (hi - lo) / wall, results.empty() ? 0.0 : iters / results.size());

    // ---- the one collective: every rank's trajectories to rank 0 over RCCL ------------------------------
    if (gather) {
        const int M = nsteps + 1, per_rank = (nscen + world - 1) / world;    // equal-sized blocks, padded
        const size_t rec = record_doubles(M), bytes = (size_t)per_rank * rec * sizeof(double);
        std::vector<double> block((size_t)per_rank * rec, 0.0);
        for (size_t q = 0; q < block.size(); q += rec) block[q] = -1.0;      // scenario -1 = padding
        for (size_t i = 0; i < results.size(); ++i) {
            const Result& r = results[i];
            double* p = &block[i * rec];
            p[0] = r.scenario; p[1] = r.rc; p[2] = r.nodes; p[3] = r.iterations; p[4] = r.cost;
            if (r.rc == 0 && r.nodes == M) {
                std::copy(r.states.begin(), r.states.end(), p + REC_HEAD);
                std::copy(r.controls.begin(), r.controls.end(), p + REC_HEAD + 6 * M);
                std::copy(r.time.begin(), r.time.end(), p + REC_HEAD + 8 * M);
            }
        }
        auto die = [&](const char* what, const char* why) { fprintf(stderr, "rank %d: %s: %s\n", rank, what, why); return EXIT_FAILURE; };
        emi_ctx_t ctx = nullptr;
        emi_comm_t comm = nullptr;
        if (emi_create(device, &ctx) != EMI_OK) return die("emi_create", "no device");
        if (emi_comm_create(device, world, rank, comm_id, &comm) != EMI_OK) return die("emi_comm_create", emi_comm_last_error(nullptr));
        comm_file_done(rank);                           // every rank holds the id by now (ncclCommInitRank is collective)
        void *dsend = nullptr, *drecv = nullptr;
        if (emi_dev_alloc(ctx, bytes, &dsend) != EMI_OK || (rank == 0 && emi_dev_alloc(ctx, bytes * world, &drecv) != EMI_OK))
            return die("emi_dev_alloc", emi_last_error(ctx));
        if (emi_h2d(ctx, dsend, block.data(), bytes) != EMI_OK) return die("emi_h2d", emi_last_error(ctx));
        const auto g0 = std::chrono::steady_clock::now();
        if (emi_comm_gather(comm, dsend, drecv, bytes, 0, nullptr) != EMI_OK) return die("emi_comm_gather", emi_comm_last_error(comm));
        const double gather_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        if (rank == 0) {
            std::vector<double> all((size_t)world * per_rank * rec);
            if (emi_d2h(ctx, all.data(), drecv, bytes * world) != EMI_OK) return die("emi_d2h", emi_last_error(ctx));
            int got = 0, solved = 0;
            double cost_sum = 0;
            const char* save_dir = getenv("EMI_MC_SAVE");
            for (size_t q = 0; q < all.size(); q += rec) {
                if (all[q] < 0) continue;
                ++got;
                if (all[q + 1] != 0) continue;
                ++solved;
                cost_sum += all[q + 4];
                if (save_dir) {
                    ETOL::traj_t xt, ut;
                    const double* X = &all[q + REC_HEAD];
                    for (int k = 0; k < M; ++k) {
                        ETOL::state_t xs, us;
                        for (int i = 0; i < 6; ++i) xs.push_back(X[i * M + k]);
                        for (int j = 0; j < 2; ++j) us.push_back(X[(6 + j) * M + k]);
                        xt.push_back({X[8 * M + k], xs});
                        ut.push_back({X[8 * M + k], us});
                    }
                    const std::string stem = std::string(save_dir) + "/scenario" + std::to_string((int)all[q]);
                    ETOL::TrajectoryOptimizer::save(&xt, stem + "_state.csv");
                    ETOL::TrajectoryOptimizer::save(&ut, stem + "_control.csv");
                }
            }
            printf("{\"gathered\": %d, \"gathered_solved\": %d, \"cost_sum\": %.10f, \"world\": %d, \"bytes_per_rank\": %zu, "
                   "\"gather_ms\": %.3f, \"collective\": \"RCCL send/recv group (emi_comm_gather)\"}\n",
                   got, solved, cost_sum, world, bytes, gather_ms);
            if (got != nscen && env_int("EMI_MC_ONLY", -1) < 0) return die("gather", "scenario count mismatch");
        }
        emi_dev_free(ctx, dsend);
        if (drecv) emi_dev_free(ctx, drecv);
        emi_comm_destroy(comm);
        emi_destroy(ctx);
    }
    return ok == hi - lo ? EXIT_SUCCESS : 2;
}
