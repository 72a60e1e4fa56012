// TrajectoryOptimizer.cpp -- base-class implementation for the standalone eMI355X build.
//
// Behaviour follows the reference (src/TrajectoryOptimizer/TrajectoryOptimizer.cpp):
//   config store / setters     :1637-1875
//   resetConfigs               :678-697   (note: leaves _xlower/_xupper/_parameters alone)
//   loadConfigs (XML schema)   :787-1117  (resource/configs/*.xml, docs tutorials/vgp.rst:82-153)
//   save (CSV)                 :626-674
//   saveConfigs / printConfigs :699-785, :1119-1635 (wire format only)
// Not built here (no CGAL / gnuplot in this environment, and not on the OCP
// path: SURVEY.md section 2 rows 6-7): genRegion's convex partition, every plot.
#include <ETOL/TrajectoryOptimizer.hpp>

#include <libxml/parser.h>
#include <libxml/tree.h>
#include <libxml/xpath.h>
#include <sys/stat.h>

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>

namespace ETOL {

TrajectoryOptimizer::TrajectoryOptimizer()
    : _maximize(false), _score(0.), _dt(0.0), _nSteps(0), _nStates(0), _nControls(0), _xrhorizon(0),
      _urhorizon(0), _rhorizon(0), _objective(NULL), _eAny(NULL) {}

// ---------------------------------------------------------------------------------------------
// XML loader
// ---------------------------------------------------------------------------------------------
namespace {

double xnum(const xmlChar* s) { return xmlXPathCastStringToNumber(s); }

// visit every attribute of an element as (name, value-text)
void each_attr(xmlNodePtr n, const std::function<void(const std::string&, const xmlChar*)>& fn) {
    for (xmlAttrPtr a = n->properties; a; a = a->next) {
        xmlChar* v = xmlNodeListGetString(a->doc, a->children, 1);
        fn(std::string(reinterpret_cast<const char*>(a->name)), v);
        xmlFree(v);
    }
}

std::string nname(xmlNodePtr n) { return std::string(reinterpret_cast<const char*>(n->name)); }

var_t vartype_of(const xmlChar* v, const char* what) {
    switch (v ? *reinterpret_cast<const char*>(v) : '\0') {
        case 'C': return var_t::CONTINUOUS;
        case 'B': return var_t::BINARY;
        case 'I': return var_t::INTERGER;
    }
    std::cout << "Invalid " << what << std::endl;
    exit(EXIT_FAILURE);
}

}  // namespace

void TrajectoryOptimizer::resetConfigs() {
    setNStates(0);
    setNControls(0);
    setDt(0);
    setXrhorizon(0);
    _xvartype.clear();
    _x0.clear();
    _xf.clear();
    _xtol.clear();
    setUrhorizon(0);
    _uvartype.clear();
    _ulower.clear();
    _uupper.clear();
    _obstacles_raw.clear();
    _obstacles.clear();
    _tracks.clear();
}

void TrajectoryOptimizer::loadConfigs(const char* filepath) {
    resetConfigs();
    LIBXML_TEST_VERSION
    xmlDocPtr doc = xmlParseFile(filepath);
    if (doc == NULL) {
        fprintf(stderr, "Document not parsed successfully. \n");
        exit(EXIT_FAILURE);
    }
    // first <etol> anywhere in the document ("//etol")
    xmlXPathContextPtr xctx = xmlXPathNewContext(doc);
    xmlXPathObjectPtr found = xctx ? xmlXPathEvalExpression(BAD_CAST "//etol", xctx) : NULL;
    if (xctx) xmlXPathFreeContext(xctx);
    if (!found || xmlXPathNodeSetIsEmpty(found->nodesetval)) {
        fprintf(stderr, "No <etol> element in %s\n", filepath);
        exit(EXIT_FAILURE);
    }
    xmlNodePtr root = found->nodesetval->nodeTab[0];

    each_attr(root, [&](const std::string& a, const xmlChar* v) {
        if (a == "nsteps") setNSteps((size_t)xnum(v));
        else if (a == "dt") setDt(xnum(v));
    });
    assert(getNSteps() != 0);
    assert(getDt() != 0);

    for (xmlNodePtr sec = root->children; sec; sec = sec->next) {
        const std::string section = nname(sec);
        if (section == "states") {
            // nstates caps how many <state> children are taken
            size_t cap = 0;
            each_attr(sec, [&](const std::string& a, const xmlChar* v) {
                if (a == "nstates") cap = (size_t)xnum(v);
                else if (a == "rhorizon") setXrhorizon(std::max(getXrhorizon(), (size_t)xnum(v)));
            });
            for (xmlNodePtr st = sec->children; st; st = st->next) {
                if (!(cap > getNStates())) continue;
                each_attr(st, [&](const std::string& a, const xmlChar* v) {
                    if (a == "vartype") {
                        setNStates(getNStates() + 1);
                        _xvartype.push_back(vartype_of(v, "xVartype"));
                    } else if (a == "lower") _xlower.push_back(xnum(v));
                    else if (a == "upper") _xupper.push_back(xnum(v));
                    else if (a == "initial") _x0.push_back(xnum(v));
                    else if (a == "terminal") _xf.push_back(xnum(v));
                    else if (a == "tolerance") _xtol.push_back(xnum(v));
                });
            }
        } else if (section == "controls") {
            size_t cap = 0;
            each_attr(sec, [&](const std::string& a, const xmlChar* v) {
                if (a == "ncontrols") cap = (size_t)xnum(v);
                else if (a == "rhorizon") setUrhorizon(std::max(getUrhorizon(), (size_t)xnum(v)));
            });
            for (xmlNodePtr ct = sec->children; ct; ct = ct->next) {
                if (!(cap > getNControls())) continue;
                each_attr(ct, [&](const std::string& a, const xmlChar* v) {
                    if (a == "vartype") {
                        setNControls(getNControls() + 1);
                        _uvartype.push_back(vartype_of(v, "uVartype"));
                    } else if (a == "lower") _ulower.push_back(xnum(v));
                    else if (a == "upper") _uupper.push_back(xnum(v));
                });
            }
        } else if (section == "exzones") {
            size_t zone_cap = SIZE_MAX, taken = 0;
            each_attr(sec, [&](const std::string& a, const xmlChar* v) {
                if (a == "nzones") zone_cap = (size_t)xnum(v);
            });
            for (xmlNodePtr bd = sec->children; bd; bd = bd->next) {
                size_t corner_cap = SIZE_MAX;
                each_attr(bd, [&](const std::string& a, const xmlChar* v) {
                    if (a == "ncorners") corner_cap = (size_t)xnum(v);
                });
                border_t border;
                for (xmlNodePtr cn = bd->children; cn; cn = cn->next) {
                    // a corner counts only if x, y and z were all given
                    double x(DBL_MIN), y(DBL_MIN), z(DBL_MIN);
                    each_attr(cn, [&](const std::string& a, const xmlChar* v) {
                        if (a == "x") x = xnum(v);
                        else if (a == "y") y = xnum(v);
                        else if (a == "z") z = xnum(v);
                    });
                    if (x != DBL_MIN && y != DBL_MIN && z != DBL_MIN && !(border.size() > corner_cap))
                        border.push_back(corner_t{x, y, z});
                }
                if (!border.empty() && zone_cap > taken) {
                    addExclZone(&border);
                    ++taken;
                }
            }
        } else if (section == "mexzones") {
            double zone_cap = 0;   // absent attribute -> no track is taken
            each_attr(sec, [&](const std::string& a, const xmlChar* v) {
                if (a == "nzones") zone_cap = (size_t)xnum(v);
            });
            for (xmlNodePtr tk = sec->children; tk; tk = tk->next) {
                size_t way_cap = 0;
                track_t track;
                each_attr(tk, [&](const std::string& a, const xmlChar* v) {
                    if (a == "radius") track.radius = xnum(v);
                    else if (a == "nwaypoints") way_cap = (size_t)xnum(v);
                });
                traj_t traj;
                for (xmlNodePtr wp = tk->children; wp; wp = wp->next) {
                    size_t datum_cap = 0;
                    traj_elem_t elem;
                    each_attr(wp, [&](const std::string& a, const xmlChar* v) {
                        if (a == "t") elem.first = xnum(v);
                        else if (a == "ndatums") datum_cap = (size_t)xnum(v);
                    });
                    state_t values;
                    for (xmlNodePtr dm = wp->children; dm; dm = dm->next) {
                        if (nname(dm) != "datum") continue;
                        if (datum_cap != 0 && values.size() >= datum_cap) continue;
                        xmlChar* txt = xmlNodeListGetString(dm->doc, dm->children, 1);
                        values.push_back(xnum(txt));
                        xmlFree(txt);
                    }
                    if (!values.empty() && way_cap > traj.size()) {
                        elem.second = values;
                        traj.push_back(elem);
                    }
                }
                if (!traj.empty() && zone_cap > getTracks()->size()) {
                    track.trajectory = traj;
                    addAdjTrack(&track);
                }
            }
        }
    }
    xmlXPathFreeObject(found);
    xmlFreeDoc(doc);
}

// ---------------------------------------------------------------------------------------------
// writers
// ---------------------------------------------------------------------------------------------
namespace {
char vchar(var_t v) { return v == var_t::BINARY ? 'B' : (v == var_t::INTERGER ? 'I' : 'C'); }
}  // namespace

void TrajectoryOptimizer::printConfigs() {
    using std::cout;
    using std::endl;
    cout << endl << "ETOL Information" << endl;
    cout << "# Steps:\t" << getNSteps() << endl << "dt:\t\t" << getDt() << endl;
    cout << "# States:\t" << getNStates() << endl << "# Controls:\t" << getNControls() << endl;
    cout << "Xrhorizon:\t" << getXrhorizon() << endl << "Urhorizon:\t" << getUrhorizon() << endl;
    cout << "#Obstacles:\t" << getNExclZones() << endl << "#Tracks:\t" << getNTracks() << endl << endl;
    cout << "\t\tvartype\tlower\tupper\tinitial\tfinal\ttol" << endl;
    for (size_t i = 0; i < _xvartype.size(); ++i)
        cout << "XInfo:\t\t" << vchar(_xvartype[i]) << "\t" << _xlower.at(i) << "\t" << _xupper.at(i) << "\t"
             << _x0.at(i) << "\t" << _xf.at(i) << "\t" << _xtol.at(i) << endl;
    for (size_t i = 0; i < _uvartype.size(); ++i)
        cout << "UInfo:\t\t" << vchar(_uvartype[i]) << "\t" << _ulower.at(i) << "\t" << _uupper.at(i) << endl;
    cout << endl << "\t\tExclusion Zone Corners (raw)..." << endl;
    for (const border_t& b : _obstacles_raw) {
        cout << "Border\t\t";
        for (const corner_t& c : b) cout << "(" << c[0] << "," << c[1] << "," << c[2] << ")  ";
        cout << endl;
    }
    for (const track_t& t : _tracks) {
        cout << endl << "\t\tradius\t#points" << endl;
        cout << "TrackInfo\t" << t.radius << "\t" << t.trajectory.size() << endl << endl;
        cout << "\t\ttime\tElements..." << endl;
        for (const traj_elem_t& e : t.trajectory) {
            cout << "Waypoint\t" << e.first << "\t";
            for (double d : e.second) cout << d << "\t";
            cout << endl;
        }
    }
    cout << endl;
}

// Same element/attribute names and "%.02f" number format as the reference writer,
// so files round-trip through either loader.
void TrajectoryOptimizer::saveConfigs(const char* filepath) {
    FILE* f = fopen(filepath, "w");
    if (!f) {
        printf("saveConfigs: cannot open %s\n", filepath);
        return;
    }
    fprintf(f, "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n");
    fprintf(f, "<etol nsteps=\"%zu\" dt=\"%.02f\">\n", getNSteps(), getDt());
    fprintf(f, "\t<states nstates=\"%zu\" rhorizon=\"%zu\">\n", getNStates(), getXrhorizon());
    for (size_t i = 0; i < getNStates(); ++i)
        fprintf(f, "\t\t<state name=\"x%zu\" vartype=\"%c\" lower=\"%.02f\" upper=\"%.02f\" initial=\"%.02f\" "
                   "terminal=\"%.02f\" tolerance=\"%.02f\"/>\n",
                i, vchar(_xvartype.at(i)), _xlower.at(i), _xupper.at(i), _x0.at(i), _xf.at(i), _xtol.at(i));
    fprintf(f, "\t</states>\n\t<controls ncontrols=\"%zu\" rhorizon=\"%zu\">\n", getNControls(), getUrhorizon());
    for (size_t i = 0; i < getNControls(); ++i)
        fprintf(f, "\t\t<control name=\"u%zu\" vartype=\"%c\" lower=\"%.02f\" upper=\"%.02f\"/>\n", i,
                vchar(_uvartype.at(i)), _ulower.at(i), _uupper.at(i));
    fprintf(f, "\t</controls>\n\t<exzones nzones=\"%zu\">\n", _obstacles_raw.size());
    size_t z = 0;
    for (const border_t& b : _obstacles_raw) {
        fprintf(f, "\t\t<border name=\"exz%zu\" ncorners=\"%zu\">\n", z++, b.size());
        for (const corner_t& c : b)
            fprintf(f, "\t\t\t<corner x=\"%.02f\" y=\"%.02f\" z=\"%.02f\"/>\n", c[0], c[1], c[2]);
        fprintf(f, "\t\t</border>\n");
    }
    fprintf(f, "\t</exzones>\n\t<mexzones nzones=\"%zu\">\n", _tracks.size());
    z = 0;
    for (const track_t& t : _tracks) {
        fprintf(f, "\t\t<track name=\"mexz%zu\" radius=\"%.02f\" nwaypoints=\"%zu\">\n", z++, t.radius,
                t.trajectory.size());
        size_t p = 0;
        for (const traj_elem_t& e : t.trajectory) {
            fprintf(f, "\t\t\t<waypoint name=\"pt%zu\" t=\"%.02f\" ndatums=\"%zu\">\n", p++, e.first, e.second.size());
            for (double d : e.second) fprintf(f, "\t\t\t\t<datum>%.02f</datum>\n", d);
            fprintf(f, "\t\t\t</waypoint>\n");
        }
        fprintf(f, "\t\t</track>\n");
    }
    fprintf(f, "\t</mexzones>\n</etol>\n");
    fclose(f);
}

// "time,traj0,..,trajN-1" then one row per waypoint, std::to_string (6 decimals), no
// trailing newline; an existing file is never overwritten: the stem's trailing number is
// incremented until the name is free (traj.csv -> traj1.csv -> traj2.csv).
std::string TrajectoryOptimizer::save(traj_t* traj, std::string fp) {
    if (traj->empty()) {
        std::cout << "No Data to Save!!!" << std::endl;
        return fp;
    }
    const size_t dot = fp.find('.');
    const std::string ext = fp.substr(dot);
    struct stat sb;
    while (stat(fp.c_str(), &sb) != -1) {
        const std::string stem = fp.substr(0, fp.find('.'));
        const size_t digits_at = stem.find_last_not_of("0123456789") + 1;
        const int idx = digits_at == stem.size() ? 0 : std::atoi(stem.substr(digits_at).c_str());
        fp = stem.substr(0, digits_at) + std::to_string(idx + 1) + ext;
    }
    std::ofstream out(fp, std::ios::out);
    const size_t width = traj->front().second.size();
    out << "time";
    for (size_t c = 0; c < width; ++c) out << ",traj" << std::to_string(c);
    out << "\n";
    for (size_t r = 0; r < traj->size(); ++r) {
        const traj_elem_t& e = (*traj)[r];
        out << std::to_string(e.first);
        for (double v : e.second) out << "," << std::to_string(v);
        if (r + 1 != traj->size()) out << "\n";
    }
    return fp;
}

// ---------------------------------------------------------------------------------------------
// members that need CGAL / gnuplot in the reference: not on the OCP path
// ---------------------------------------------------------------------------------------------
namespace {
void no_backend(const char* what) {
    std::cout << what << ": not available in the eMI355X standalone build (needs gnuplot-iostream / CGAL)"
              << std::endl;
}
}  // namespace

region_t TrajectoryOptimizer::genRegion(border_t*) { return region_t(); }
void TrajectoryOptimizer::calcSlopes(const region_t&, std::vector<seg_t>*, std::vector<seg_t>*) {}
void TrajectoryOptimizer::plot(traj_t*, const std::string, const std::string, const std::string, double, double,
                               double, double) { no_backend("plot"); }
void TrajectoryOptimizer::plotXY(traj_t*, size_t, size_t, const std::string, const std::string,
                                 const std::string, double, double, double, double) { no_backend("plotXY"); }
void TrajectoryOptimizer::plotXY_wExclZones(traj_t*, std::list<region_t>*, size_t, size_t, const std::string,
                                            const std::string, const std::string, double, double, double,
                                            double) { no_backend("plotXY_wExclZones"); }
std::string TrajectoryOptimizer::animate2D(traj_t*, const int, bool, std::string, std::list<region_t>*,
                                           std::list<track_t>*, size_t, size_t, const std::string,
                                           const std::string, const std::string, double, double, double, double) {
    no_backend("animate2D");
    return "";
}
void TrajectoryOptimizer::plotX(const size_t) { no_backend("plotX"); }
void TrajectoryOptimizer::plotU(const size_t) { no_backend("plotU"); }

// ---------------------------------------------------------------------------------------------
// problem additions
// ---------------------------------------------------------------------------------------------
void TrajectoryOptimizer::addParams(std::list<param_t> params) {
    for (const param_t& p : params) _parameters.insert(p);   // existing names keep their first value
}

void TrajectoryOptimizer::addExclZone(border_t* border) {
    _obstacles_raw.push_back(*border);
    region_t pieces = TrajectoryOptimizer::genRegion(border);
    if (!pieces.empty()) _obstacles.push_back(pieces);
}

void TrajectoryOptimizer::addAdjTrack(track_t* track) { _tracks.push_back(*track); }

// ---------------------------------------------------------------------------------------------
// config store
// ---------------------------------------------------------------------------------------------
const double TrajectoryOptimizer::getScore() const { return _score; }
void TrajectoryOptimizer::setScore(const double score) { _score = score; }
state_t& TrajectoryOptimizer::getX0() { return _x0; }
void TrajectoryOptimizer::setX0(const state_t& x0) { _x0 = x0; }
state_t& TrajectoryOptimizer::getXf() { return _xf; }
void TrajectoryOptimizer::setXf(const state_t& xf) { _xf = xf; }
const size_t TrajectoryOptimizer::getNControls() const { return _nControls; }
const size_t TrajectoryOptimizer::getNStates() const { return _nStates; }
state_t& TrajectoryOptimizer::getXlower() { return _xlower; }
void TrajectoryOptimizer::setXlower(const state_t& v) { _xlower = v; }
state_t& TrajectoryOptimizer::getXupper() { return _xupper; }
void TrajectoryOptimizer::setXupper(const state_t& v) { _xupper = v; }
state_var_t& TrajectoryOptimizer::getXvartype() { return _xvartype; }
void TrajectoryOptimizer::setXvartype(const state_var_t& v) { _xvartype = v; }
const double TrajectoryOptimizer::getDt() const { return _dt; }
void TrajectoryOptimizer::setDt(const double dt) { _dt = dt; }
const size_t TrajectoryOptimizer::getNSteps() const { return _nSteps; }
void TrajectoryOptimizer::setNSteps(const size_t n) { _nSteps = n; }
state_t& TrajectoryOptimizer::getXtol() { return _xtol; }
void TrajectoryOptimizer::setXtol(const state_t& xtol) { _xtol = xtol; }   // stored as given
state_t& TrajectoryOptimizer::getUlower() { return _ulower; }
void TrajectoryOptimizer::setUlower(const state_t& v) { _ulower = v; }
state_t& TrajectoryOptimizer::getUupper() { return _uupper; }
void TrajectoryOptimizer::setUupper(const state_t& v) { _uupper = v; }
state_var_t& TrajectoryOptimizer::getUvartype() { return _uvartype; }
void TrajectoryOptimizer::setUvartype(const state_var_t& v) { _uvartype = v; }
const size_t TrajectoryOptimizer::getUrhorizon() const { return _urhorizon; }
void TrajectoryOptimizer::setUrhorizon(const size_t n) {
    _urhorizon = n;
    _rhorizon = std::max(_xrhorizon, _urhorizon);
}
const size_t TrajectoryOptimizer::getXrhorizon() const { return _xrhorizon; }
void TrajectoryOptimizer::setXrhorizon(const size_t n) {
    _xrhorizon = n;
    _rhorizon = std::max(_xrhorizon, _urhorizon);
}
size_t TrajectoryOptimizer::getRhorizon() const { return _rhorizon; }
void TrajectoryOptimizer::setNControls(const size_t n) { _nControls = n; }
void TrajectoryOptimizer::setNStates(const size_t n) { _nStates = n; }
void TrajectoryOptimizer::setConstraints(std::vector<f_t*> c) { _constraints = c; }
void TrajectoryOptimizer::setEqConstraints(std::vector<f_t*> c) { _eq = c; }
void TrajectoryOptimizer::setLessEqConstraints(std::vector<f_t*> c) { _lesseq = c; }
void TrajectoryOptimizer::setGradient(std::vector<f_t*> g) { _gradient = g; }
void TrajectoryOptimizer::setObjective(f_t* o) { _objective = o; }

void TrajectoryOptimizer::errorHandler() {
    if (_eAny != NULL) {
        fprintf(stderr, "%s", _eAny->what());
        exit(EXIT_FAILURE);
    }
}

traj_t* TrajectoryOptimizer::getUtraj() { return &_utraj; }
traj_t* TrajectoryOptimizer::getXtraj() { return &_xtraj; }
const f_t* TrajectoryOptimizer::getObjective() const { return _objective; }
std::vector<f_t*>* TrajectoryOptimizer::getGradient() { return &_gradient; }
std::vector<f_t*>* TrajectoryOptimizer::getEqConstraints() { return &_eq; }
std::vector<f_t*>* TrajectoryOptimizer::getLessEqConstraints() { return &_lesseq; }
std::vector<f_t*>* TrajectoryOptimizer::getConstraints() { return &_constraints; }
std::vector<border_t>* TrajectoryOptimizer::getObstacles_Raw() { return &_obstacles_raw; }
std::list<region_t>* TrajectoryOptimizer::getObstacles() { return &_obstacles; }
std::list<track_t>* TrajectoryOptimizer::getTracks() { return &_tracks; }
bool TrajectoryOptimizer::isMaximized() const { return _maximize; }
void TrajectoryOptimizer::setMaximize(const bool m) { _maximize = m; }
size_t TrajectoryOptimizer::getNExclZones() { return _obstacles.size(); }
size_t TrajectoryOptimizer::getNTracks() { return _tracks.size(); }

}  // namespace ETOL
