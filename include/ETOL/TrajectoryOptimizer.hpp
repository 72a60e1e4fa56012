// TrajectoryOptimizer.hpp -- abstract base every ETOL eSolver derives from.
//
// API-identical re-declaration for the standalone eMI355X build (the reference
// header, include/ETOL/TrajectoryOptimizer.hpp:27-693, needs nothing this
// declaration does not also provide; its implementation, however, needs CGAL
// and gnuplot-iostream, which do not exist here).  The hot-path parts are
// implemented in etol_amd/host/TrajectoryOptimizer.cpp: the config store, the
// XML loader, addParams/addExclZone/addAdjTrack, the callback setters, save().
// Out-of-scope members (plots, animation, CGAL partition) are declared so user
// code keeps compiling; they report that the build has no such backend.
#ifndef ETOL_MI355X_TRAJECTORYOPTIMIZER_HPP_
#define ETOL_MI355X_TRAJECTORYOPTIMIZER_HPP_

#include <algorithm>
#include <cfloat>
#include <iterator>
#include <list>
#include <string>
#include <vector>

#include <ETOL/ETOL_Types.hpp>

namespace ETOL {

class TrajectoryOptimizer {
 public:
    TrajectoryOptimizer();
    virtual ~TrajectoryOptimizer() {}

    // ---- what an eSolver must provide (reference :39-54) -----------------------
    virtual void setup() = 0;
    virtual void solve() = 0;
    virtual void debug() = 0;
    virtual void close() = 0;

    // ---- configuration I/O ---------------------------------------------------------
    const double getScore() const;
    void resetConfigs();
    void printConfigs();
    void loadConfigs(const char* filepath);
    void saveConfigs(const char* filepath);
    void addParams(std::list<param_t> params);
    void addExclZone(border_t* border);
    void addAdjTrack(track_t* track);
    void plotX(const size_t idx);
    void plotU(const size_t idx);

    // ---- static helpers ---------------------------------------------------------------
    static region_t genRegion(border_t* border);
    static void calcSlopes(const region_t& region, std::vector<seg_t>* lower, std::vector<seg_t>* upper);
    static void plot(traj_t* traj, const std::string title = "Time History", const std::string tlab = "t",
                     const std::string xlab = "", double tmin = DBL_MAX, double tmax = DBL_MIN,
                     double xmin = DBL_MAX, double xmax = DBL_MIN);
    static void plotXY(traj_t* traj, size_t xIdx = 0, size_t yIdx = 1, const std::string title = "XY Plot",
                       const std::string xlab = "x", const std::string ylab = "y", double xmin = DBL_MAX,
                       double xmax = DBL_MIN, double ymin = DBL_MAX, double ymax = DBL_MIN);
    static void plotXY_wExclZones(traj_t* traj, std::list<region_t>* zones = NULL, size_t xIdx = 0,
                                  size_t yIdx = 1, const std::string title = "XY Plot",
                                  const std::string xlab = "x", const std::string ylab = "y",
                                  double xmin = DBL_MAX, double xmax = DBL_MIN, double ymin = DBL_MAX,
                                  double ymax = DBL_MIN);
    static std::string animate2D(traj_t* traj, const int framerate = 2, bool toFile = false,
                                 std::string outFile = "animation.mp4", std::list<region_t>* obstacles = NULL,
                                 std::list<track_t>* tracks = NULL, size_t xIdx = 0, size_t yIdx = 1,
                                 const std::string title = "notitle", const std::string xlab = "x",
                                 const std::string ylab = "y", double xmin = DBL_MAX, double xmax = DBL_MIN,
                                 double ymin = DBL_MAX, double ymax = DBL_MIN);
    // CSV writer: "time,traj0,..." header, std::to_string formatting, never overwrites
    static std::string save(traj_t* traj, std::string fp = "traj.csv");

    // ---- trajectory templates (reference :239-324) -----------------------------------------
    // Piecewise-linear lookup.  Segment choice: before the table -> first
    // segment, after it -> last segment, inside -> the LAST segment whose
    // closed interval holds tval.
    template <class T>
    static T linear_interpolation(const T& tval, const state_t& tvec, const state_t& ref) {
        size_t seg = 0;
        const size_t n = tvec.size();
        if (tval > tvec.back()) {
            seg = n - 2;
        } else if (tval >= tvec.front()) {
            for (size_t s = 0; s + 1 < n; ++s)
                if (tval >= tvec[s] && tval <= tvec[s + 1]) seg = s;
        }
        return (tval - tvec.at(seg)) * (ref.at(seg + 1) - ref.at(seg)) / (tvec.at(seg + 1) - tvec.at(seg)) +
               ref.at(seg);
    }
    // Column selection; index 0 means "time", i>0 means value i-1.
    template <typename T> static traj_t extractTraj(const traj_t& traj, const std::vector<T>& idxs) {
        traj_t out;
        out.reserve(traj.size());
        for (const traj_elem_t& e : traj) {
            state_t s;
            for (const T& i : idxs) s.push_back(i == 0 ? e.first : e.second.at(i - 1));
            out.push_back(traj_elem_t(e.first, s));
        }
        return out;
    }
    // values[i] *= scalers[i] while scalers last, untouched afterwards
    template <typename T> static void scaleTraj(traj_t* traj, const std::vector<T>& scalers) {
        for (traj_elem_t& e : *traj)
            for (size_t i = 0; i < e.second.size() && i < scalers.size(); ++i) e.second[i] *= scalers[i];
    }
    template <typename T> static void offsetTraj(traj_t* traj, const std::vector<T>& offsets) {
        for (traj_elem_t& e : *traj)
            for (size_t i = 0; i < e.second.size() && i < offsets.size(); ++i) e.second[i] += offsets[i];
    }

    // ---- getters / setters (reference :326-649) ------------------------------------------------
    state_t& getX0();
    void setX0(const state_t& x0);
    state_t& getXf();
    void setXf(const state_t& xf);
    const size_t getNControls() const;
    const size_t getNStates() const;
    state_t& getXlower();
    void setXlower(const state_t& xlower);
    state_t& getXupper();
    void setXupper(const state_t& xupper);
    state_var_t& getXvartype();
    void setXvartype(const state_var_t& xvartype);
    const double getDt() const;
    void setDt(double dt);
    const size_t getNSteps() const;
    void setNSteps(const size_t nSteps);
    state_t& getXtol();
    void setXtol(const state_t& xtol);
    state_t& getUlower();
    void setUlower(const state_t& ulower);
    state_t& getUupper();
    void setUupper(const state_t& uupper);
    state_var_t& getUvartype();
    void setUvartype(const state_var_t& uvartype);
    const size_t getUrhorizon() const;
    void setUrhorizon(const size_t nu4dyn);
    const size_t getXrhorizon() const;
    void setXrhorizon(const size_t nx4dyn);
    size_t getRhorizon() const;
    void setNControls(const size_t nControls);
    void setNStates(const size_t nStates);
    void setEqConstraints(std::vector<f_t*> constraints);
    void setLessEqConstraints(std::vector<f_t*> constraints);
    void setConstraints(std::vector<f_t*> constraints);
    void setGradient(std::vector<f_t*> gradient);
    void setObjective(f_t* objective);
    traj_t* getUtraj();
    traj_t* getXtraj();
    const f_t* getObjective() const;
    std::vector<f_t*>* getGradient();
    std::vector<f_t*>* getEqConstraints();
    std::vector<f_t*>* getLessEqConstraints();
    std::vector<f_t*>* getConstraints();
    std::vector<border_t>* getObstacles_Raw();
    std::list<region_t>* getObstacles();
    std::list<track_t>* getTracks();
    bool isMaximized() const;
    void setMaximize(const bool maximize);
    size_t getNExclZones();
    size_t getNTracks();

 protected:
    void errorHandler();                  // bad any_cast -> message on stderr, exit(EXIT_FAILURE)
    void setScore(const double score);

    bool _maximize;
    double _score;
    double _dt;
    size_t _nSteps;
    size_t _nStates;
    size_t _nControls;
    state_t _x0;
    state_t _xlower;
    state_t _xupper;
    state_var_t _xvartype;
    state_t _xtol;
    state_t _xf;
    state_t _ulower;
    state_t _uupper;
    state_var_t _uvartype;
    size_t _xrhorizon;
    size_t _urhorizon;
    size_t _rhorizon;
    paramset_t _parameters;
    std::vector<border_t> _obstacles_raw;
    std::list<region_t> _obstacles;
    std::list<track_t> _tracks;
    std::vector<f_t*> _constraints;
    std::vector<f_t*> _eq;
    std::vector<f_t*> _lesseq;
    std::vector<f_t*> _gradient;
    f_t* _objective;
    traj_t _xtraj;
    traj_t _utraj;
    std::bad_any_cast* _eAny;
};

}  // namespace ETOL

#endif  // ETOL_MI355X_TRAJECTORYOPTIMIZER_HPP_
