// Source-level mirror of the reference's plugin surface for the hot path (hemocell.h:68-253,
// core/hemoCellFields.h, core/hemoCellField.h, helper/cellInfo.h, helper/hemoCellStretch.h, config/config.h),
// implemented as a thin host layer over the C ABI of libhemocell_amd.so.  Same names, argument meaning and
// error behaviour (log + exit(1); std::invalid_argument for a missing XML tag, config/config.cpp:78-85).
// Everything numerical happens on the GPU inside the library; this header only translates calls.
#pragma once
#include "palabos3D.h"

#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>
#include <sys/stat.h>

#define DESCRIPTOR plb::descriptors::ForcedD3Q19Descriptor   // config/constant_defaults.h:58-61

// config/constant_defaults.h:77-112
#define RBC_FROM_SPHERE 1
#define ELLIPSOID_FROM_SPHERE 6
#define STRING_FROM_VERTEXES 7
#define WBC_SPHERE 0
#define MESH_FROM_STL 2
#define OUTPUT_POSITION 1
#define OUTPUT_FORCE 2
#define OUTPUT_FORCE_VOLUME 21
#define OUTPUT_FORCE_BENDING 22
#define OUTPUT_FORCE_AREA 23
#define OUTPUT_FORCE_LINK 24
#define OUTPUT_FORCE_VISC 25
#define OUTPUT_FORCE_INNER_LINK 26
#define OUTPUT_FORCE_REPULSION 27
#define OUTPUT_TRIANGLES 3
#define OUTPUT_VELOCITY 4
#define OUTPUT_DENSITY 5
#define OUTPUT_VERTEX_ID 7
#define OUTPUT_CELL_ID 8
#define OUTPUT_CELL_DENSITY 9
#define OUTPUT_SHEAR_STRESS 10
#define OUTPUT_INNER_LINKS 11
#define OUTPUT_BOUNDARY 13
#define OUTPUT_RES_TIME 18
#define param Parameters
#ifndef PI
#define PI 3.14159265358979323846
#endif

namespace hemo {
using namespace plb;
using std::cout;
using std::endl;
using std::map;
using std::string;
using std::vector;

// ------------------------------------------------------------------ helper/array.h
template <typename U, std::size_t N>
struct Array : std::array<U, N> {
  Array() { this->fill(U()); }
  Array(std::initializer_list<U> l) { std::size_t i = 0; for (U v : l) if (i < N) (*this)[i++] = v; for (; i < N; i++) (*this)[i] = U(); }
  template <int M> Array(const plb::Array<U, M> &o) { for (std::size_t i = 0; i < N; i++) (*this)[i] = o[(int)i]; }
  Array operator/(U s) const { Array r; for (std::size_t i = 0; i < N; i++) r[i] = (*this)[i] / s; return r; }
  Array operator*(U s) const { Array r; for (std::size_t i = 0; i < N; i++) r[i] = (*this)[i] * s; return r; }
  Array operator-(const Array &o) const { Array r; for (std::size_t i = 0; i < N; i++) r[i] = (*this)[i] - o[i]; return r; }
  Array operator+(const Array &o) const { Array r; for (std::size_t i = 0; i < N; i++) r[i] = (*this)[i] + o[i]; return r; }
};

// ------------------------------------------------------------------ config/logfile.h
struct Logger {
  bool to_stdout; std::ofstream file;
  explicit Logger(bool s) : to_stdout(s) {}
  template <typename V> Logger &operator<<(const V &v) { if (to_stdout) std::cout << v; if (file.is_open()) file << v; return *this; }
  Logger &operator<<(std::ostream &(*f)(std::ostream &)) { if (to_stdout) std::cout << f; if (file.is_open()) file << f; return *this; }
};
inline Logger &hlog_instance() { static Logger l(true); return l; }
inline Logger &hlogfile_instance() { static Logger l(false); return l; }
#define hlog hemo::hlog_instance()
#define hlogfile hemo::hlogfile_instance()

// ------------------------------------------------------------------ config/config.h (XML subset)
struct XMLNode {
  string name, text;
  vector<std::unique_ptr<XMLNode>> children;
  XMLNode *child(const string &n) const { for (auto &c : children) if (c->name == n) return c.get(); return nullptr; }
};

inline std::unique_ptr<XMLNode> parse_xml(const string &src) {
  std::unique_ptr<XMLNode> root(new XMLNode()); root->name = "#document";
  vector<XMLNode *> stack = {root.get()};
  size_t i = 0;
  while (i < src.size()) {
    if (src[i] != '<') { size_t j = src.find('<', i); if (j == string::npos) j = src.size(); stack.back()->text += src.substr(i, j - i); i = j; continue; }
    if (src.compare(i, 4, "<!--") == 0) { size_t j = src.find("-->", i); i = (j == string::npos) ? src.size() : j + 3; continue; }
    if (src.compare(i, 2, "<?") == 0) { size_t j = src.find("?>", i); i = (j == string::npos) ? src.size() : j + 2; continue; }
    size_t j = src.find('>', i);
    if (j == string::npos) break;
    string tag = src.substr(i + 1, j - i - 1);
    i = j + 1;
    if (!tag.empty() && tag[0] == '/') { if (stack.size() > 1) stack.pop_back(); continue; }
    const bool selfclose = !tag.empty() && tag.back() == '/';
    if (selfclose) tag.pop_back();
    std::istringstream ts(tag); string nm; ts >> nm;
    std::unique_ptr<XMLNode> n(new XMLNode()); n->name = nm;
    XMLNode *raw = n.get();
    stack.back()->children.push_back(std::move(n));
    if (!selfclose) stack.push_back(raw);
  }
  return root;
}

class XMLElement {
 public:
  explicit XMLElement(XMLNode *n = nullptr) : node(n) {}
  XMLElement operator[](const string &name) const {
    XMLNode *c = node ? node->child(name) : nullptr;
    if (!c) throw std::invalid_argument("XML element " + name + " not found");   // config/config.cpp:78-85
    return XMLElement(c);
  }
  template <typename V> V read() const { std::istringstream s(node->text); V v; s >> v; if (s.fail()) throw std::invalid_argument("cannot read XML element " + node->name); return v; }
  XMLNode *getOrig() const { return node; }
  XMLNode *node;
};
template <> inline string XMLElement::read<string>() const { std::istringstream s(node->text); string v; s >> v; return v; }

class Config {
 public:
  explicit Config(const string &fileName) { load(fileName); }
  void load(const string &fileName) {
    std::ifstream f(fileName.c_str());
    if (!f.is_open()) { std::cerr << "(HemoCell) (Config) cannot open " << fileName << std::endl; std::exit(1); }
    std::stringstream ss; ss << f.rdbuf();
    doc = parse_xml(ss.str());
    root = doc->child("hemocell");
    if (!root) { root = doc->child("Checkpoint"); checkpointed = root != nullptr; if (root) root = root->child("hemocell"); }   // config/config.cpp:45-60
    if (!root) { std::cerr << "(HemoCell) (Config) " << fileName << " has no <hemocell> root" << std::endl; std::exit(1); }
  }
  XMLElement operator[](const string &name) const { return XMLElement(root)[name]; }
  bool checkpointed = false;
 private:
  std::unique_ptr<XMLNode> doc; XMLNode *root = nullptr;
};

// ------------------------------------------------------------------ mechanics/constantConversion.h
struct Parameters {
  static T dt, dx, dm, df, nu_p, rho_p, tau, re, nu_lbm, u_lbm_max, shearrate_lbm, kBT_lbm, kBT_p, ef_lbm, f_limit, pipe_radius;
  static hc_params &raw() { static hc_params p; return p; }
  static void lbm_base_parameters(Config &cfg) {
    hc_params &P = raw();
    hc_params_base(&P, cfg["domain"]["dx"].read<T>(), cfg["domain"]["dt"].read<T>(), cfg["domain"]["nuP"].read<T>(),
                   cfg["domain"]["rhoP"].read<T>(), cfg["domain"]["kBT"].read<T>());
    if (cfg["domain"]["dt"].read<T>() < 0.0) hlog << "(HemoCell) dt is set to *auto*. Tau will be set to 1!" << endl;
    dt = P.dt; dx = P.dx; dm = P.dm; df = P.df; nu_p = P.nu_p; rho_p = P.rho_p; tau = P.tau; nu_lbm = P.nu_lbm; kBT_lbm = P.kBT_lbm; kBT_p = P.kBT_p; f_limit = P.f_limit;
  }
  // mechanics/constantConversion.cpp:61-73
  static void lbm_pipe_parameters(Config &cfg, MultiScalarField3D<int> *sf) {
    lbm_base_parameters(cfg);
    re = cfg["domain"]["Re"].read<T>();
    T fluidArea = 0;
    for (plint y = 0; y < sf->getNy(); y++) for (plint z = 0; z < sf->getNz(); z++) fluidArea += sf->get(0, y, z);
    pcout << fluidArea << endl;
    pipe_radius = std::sqrt(fluidArea / PI);
    hlog << "(Parameters) The channel has a calculated radius of " << pipe_radius << " LU, assuming a perfect circle cross-section." << endl;
    u_lbm_max = re * nu_lbm / (pipe_radius * 2);
  }
  static void lbm_pipe_parameters(Config &cfg, int nY) {   // :75-83
    lbm_base_parameters(cfg);
    re = cfg["domain"]["Re"].read<T>();
    pipe_radius = nY;
    hlog << "(Parameters) The channel has a predefined radius of " << pipe_radius << " LU." << endl;
    u_lbm_max = re * nu_lbm / (pipe_radius * 2);
  }
  static void lbm_shear_parameters(Config &cfg, T nx) {     // :85-91
    lbm_base_parameters(cfg);
    T shearrate_p = cfg["domain"]["shearrate"].read<T>();
    re = (nx * (shearrate_p * (nx * 0.5))) / nu_p;
    shearrate_lbm = shearrate_p * dt;
    u_lbm_max = shearrate_lbm;
  }
  static void printParameters() {                            // :104-116
    hlog << "(HemoCell) System parameters:" << endl;
    hlog << "\t dx: \t" << dx << endl << "\t dt: \t" << dt << endl << "\t dm: \t" << dm << endl << "\t dN: \t" << df << endl;
    hlog << "\t tau: \t" << tau << endl << "\t nu_lbm: \t" << nu_lbm << endl << "\t u_lb_max: \t" << u_lbm_max << endl << "\t f_limit: \t" << f_limit << endl;
  }
};
#ifdef HEMOCELL_COMPAT_MAIN
T Parameters::dt = 0, Parameters::dx = 0, Parameters::dm = 0, Parameters::df = 0, Parameters::nu_p = 0, Parameters::rho_p = 0, Parameters::tau = 0,
  Parameters::re = 0, Parameters::nu_lbm = 0, Parameters::u_lbm_max = 0, Parameters::shearrate_lbm = 0, Parameters::kBT_lbm = 0, Parameters::kBT_p = 0,
  Parameters::ef_lbm = 0, Parameters::f_limit = 0, Parameters::pipe_radius = 0;
#endif

// helper/profiler.h:46-77 / config/config.h:80-96: hemo::global.statistics; backed by the library's hipEvent timers
struct ProfilerView {
  void printStatistics() {
    static const char *k[] = {"collide_stream", "ibm_spread", "ibm_interpolate", "advance", "mechanics"};
    static const char *ref[] = {"collideAndStream", "spreadParticleForce", "interpolateFluidVelocity", "advanceParticles", "applyConstitutiveModel"};
    hlog << "(Profiler) GPU kernel time by phase (reference timer names):" << endl;
    for (int i = 0; i < 5; i++) { double ms = 0; long n = 0; if (hc_profile_read(k[i], &ms, &n) == 0) hlog << "\t" << ref[i] << ": " << ms * 1e-3 << " s in " << n << " launches" << endl; }
  }
  void outputStatistics() { printStatistics(); }
  void start() { hc_profile_reset(); hc_profile_enable(1); }
};
struct Global { ProfilerView statistics; bool cellsDeletedInfo = false; };
static Global global;

class HemoCell;
class HemoCellFields;
class HemoCellField;

// ------------------------------------------------------------------ mechanics: the plugin classes of hemocell.h:122-128
struct MeshMetricsView { T volume = 0, surface = 0; T getVolume() const { return volume; } T getSurface() const { return surface; } };

class CellMechanics {
 public:
  virtual ~CellMechanics() {}
  virtual void statistics() = 0;
  T k_volume = 0, k_area = 0, k_link = 0, k_bend = 0, eta_m = 0;
};

class HemoCellField {
 public:
  string name; unsigned int ctype = 0; int constructType = 0;
  Config *materialCfg = nullptr;
  unsigned int timescale = 1;
  T minimumDistanceFromSolid = 0;   // micrometres.  core/hemoCellField.h:64 declares it unsigned int (0.5 -> 0); the fraction is kept here
                                    // because only then does examples/pipeflow keep the 42 cells the reference's tests assert
  int numVertex = 0, numTriangles = 0;
  MeshMetricsView *meshmetric = nullptr;
  CellMechanics *mechanics = nullptr;
  hc_celltype *dev = nullptr;
  vector<double> vertices; vector<long> triangles; vector<long> innerEdges;
  HemoCellFields *cellFields = nullptr;
  vector<int> desiredOutputVariables;
  ~HemoCellField() { delete materialCfg; delete meshmetric; delete mechanics; if (dev) hcp_celltype_destroy(dev); }
  hemo::Array<T, 6> getOriginalBoundingBox() const {
    hemo::Array<T, 6> b;
    for (int d = 0; d < 3; d++) { b[2 * d] = 1e300; b[2 * d + 1] = -1e300; }
    for (int i = 0; i < numVertex; i++) for (int d = 0; d < 3; d++) { b[2 * d] = std::min(b[2 * d], vertices[3 * i + d]); b[2 * d + 1] = std::max(b[2 * d + 1], vertices[3 * i + d]); }
    return b;
  }
  void statistics();
  void create_device_type(int model);
};

template <int MODEL>
class DeviceMechanics : public CellMechanics {
 public:
  DeviceMechanics(Config &, HemoCellField &field) : cellField(field) {
    field.create_device_type(MODEL);
    double sc[9];
    hc_check(hcp_celltype_tables(field.dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sc), "hcp_celltype_tables");
    k_volume = sc[4]; k_area = sc[5]; k_link = sc[6]; k_bend = sc[7]; eta_m = sc[8];
  }
  void statistics() override {   // mechanics/rbcHighOrderModel.cpp:209-216
    hlog << "(Cell-mechanics model) parameters for " << cellField.name << " cellfield" << endl;
    hlog << "\t k_link:   " << k_link << endl << "\t k_area:   " << k_area << endl << "\t k_bend: : " << k_bend << endl
         << "\t k_volume: " << k_volume << endl << "\t eta_m:    " << eta_m << endl;
  }
  HemoCellField &cellField;
};
typedef DeviceMechanics<HC_MODEL_RBC_HO> RbcHighOrderModel;   // mechanics/rbcHighOrderModel.h
typedef DeviceMechanics<HC_MODEL_PLT_SIMPLE> PltSimpleModel;  // mechanics/pltSimpleModel.h

class HemoCellFields {
 public:
  explicit HemoCellFields(HemoCell &h) : hemocell(h) {}
  ~HemoCellFields() { for (auto *f : cellFields) delete f; if (dev) hcp_destroy(dev); }
  HemoCellField *operator[](unsigned int i) { return cellFields[i]; }
  HemoCellField *operator[](const string &name) {
    for (auto *f : cellFields) if (f->name == name) return f;
    hlog << "(HemoCell) (CellFields) cell type " << name << " does not exist" << endl; std::exit(1);   // core/hemoCellFields.cpp:155-171
  }
  unsigned int size() const { return (unsigned int)cellFields.size(); }
  HemoCellField *addCellType(const string &name, int constructType) {
    HemoCellField *f = new HemoCellField();
    f->name = name; f->ctype = (unsigned int)cellFields.size(); f->constructType = constructType; f->cellFields = this;
    f->materialCfg = new Config(name + ".xml");                // core/hemoCellField.cpp:46-47
    cellFields.push_back(f);
    return f;
  }
  hc_cells *device();
  HemoCell &hemocell;
  vector<HemoCellField *> cellFields;
  int number_of_cells = 0;
  unsigned int particleVelocityUpdateTimescale = 1;
  hc_cells *dev = nullptr;
  bool types_bound = false;
};

// ------------------------------------------------------------------ hemocell.h:68-253
class HemoCell {
 public:
  enum class MPIHandle { Internal, External };
  HemoCell(char *configFileName, int argc, char *argv[]) : HemoCell(configFileName, argc, argv, MPIHandle::Internal) {}
  HemoCell(char *configFileName, int, char *[], MPIHandle) {
    hc_check(hc_init(0), "hc_init");       // replaces plb::plbInit (core/hemoCell.cpp:80-86)
    cfg = new Config(configFileName);
    configFile = configFileName;
    try { outDir = (*cfg)["parameters"]["outputDirectory"].read<string>(); } catch (std::invalid_argument &) { outDir = "tmp"; }
    mkdir(outDir.c_str(), 0755); mkdir((outDir + "/log").c_str(), 0755); mkdir((outDir + "/csv").c_str(), 0755);
    hlog_instance().file.open((outDir + "/log/logfile").c_str());
    hlogfile_instance().file.open((outDir + "/log/logfile.detail").c_str());
  }
  ~HemoCell() { delete cellfields; delete lattice; delete cfg; }   // core/hemoCell.cpp:97-127: the facade owns the driver's lattice

  void latticeEquilibrium(T rho, hemo::Array<T, 3> vel) { lattice->eq_rho = rho; for (int d = 0; d < 3; d++) lattice->eq_u[d] = vel[d]; lattice->dirty_layout = true; }
  void initializeCellfield() { cellfields = new HemoCellFields(*this); }

  template <class Mechanics>
  void addCellType(string name, int constructType) {
    HemoCellField *cellfield = cellfields->addCellType(name, constructType);
    Mechanics *mechanics = new Mechanics(*cellfield->materialCfg, *cellfield);
    cellfield->mechanics = mechanics;
    cellfield->statistics();
  }
  void setOutputs(string name, vector<int> outputs) { (*cellfields)[name]->desiredOutputVariables = outputs; }
  void setFluidOutputs(vector<int> outputs) { fluidOutputs = outputs; }
  void setMaterialTimeScaleSeparation(string name, unsigned int separation) {
    hlog << "(HemoCell) (Timescale Seperation) Setting seperation of " << name << " to " << separation << " timesteps" << endl;
    (*cellfields)[name]->timescale = separation;
  }
  void setParticleVelocityUpdateTimeScaleSeparation(unsigned int separation) {
    hlog << "(HemoCell) (Timescale separation) Setting update separation of all particles to " << separation << " timesteps" << endl;
    cellfields->particleVelocityUpdateTimescale = separation;
  }
  void setRepulsion(T repulsionConstant, T repulsionCutoff) {   // core/hemoCell.cpp:420-426 (cut-off in micrometres)
    hlog << "(HemoCell) (Repulsion) Setting repulsion constant to " << repulsionConstant << ". repulsionCutoff to" << repulsionCutoff << " µm" << endl;
    repulsionConstant_ = repulsionConstant; repulsionCutoff_ = repulsionCutoff * (1e-6 / Parameters::dx);
    repulsionEnabled = true; repulsionPushed = false;
  }
  void setRepulsionTimeScaleSeperation(unsigned int separation) {   // :394-397
    hlog << "(HemoCell) (Repulsion Timescale Seperation) Setting seperation to " << separation << " timesteps" << endl;
    repulsionTimescale = separation; repulsionPushed = false;
  }
  void enableBoundaryParticles(T boundaryRepulsionConstant, T boundaryRepulsionCutoff, unsigned int timestep = 1) {   // core/hemoCell.cpp:428-436
    hlog << "(HemoCell) (Repulsion) Setting boundary repulsion constant to " << boundaryRepulsionConstant << ". boundary repulsionCutoff to" << boundaryRepulsionCutoff << " µm" << endl;
    boundaryRepulsionConstant_ = boundaryRepulsionConstant; boundaryRepulsionCutoff_ = boundaryRepulsionCutoff * (1e-6 / Parameters::dx);
    boundaryRepulsionTimescale = timestep; boundaryRepulsionEnabled = true; boundaryRepulsionPushed = false;
  }
  void setInitialMinimumDistanceFromSolid(string name, T distance) {   // core/hemoCell.cpp:410-418 (micrometres, stored as unsigned int)
    (*cellfields)[name]->minimumDistanceFromSolid = distance;
  }
  void setSystemPeriodicity(unsigned int axis, bool bePeriodic) { lattice->periodicity().toggle((int)axis, bePeriodic); }
  void setSystemPeriodicityLimit(unsigned int, int) {}
  void loadParticles();
  void loadCheckPoint();
  void saveCheckPoint();
  void writeOutput();
  void iterate() {
    hc_cells *c = cellfields->device();
    if (boundaryRepulsionEnabled && !boundaryRepulsionPushed) { hc_check(hcp_set_boundary_repulsion(c, boundaryRepulsionConstant_, boundaryRepulsionCutoff_, (int)boundaryRepulsionTimescale), "hcp_set_boundary_repulsion"); boundaryRepulsionPushed = true; }
    if (repulsionEnabled && !repulsionPushed) { hc_check(hcp_set_repulsion(c, repulsionConstant_, repulsionCutoff_, (int)repulsionTimescale), "hcp_set_repulsion"); repulsionPushed = true; }
    long it = iter;
    hc_check(hc_iterate(lattice->device(), c, &it, 1, (int)cellfields->particleVelocityUpdateTimescale, /*force_limit=*/1, /*deletion check=*/1), "iterate");
    lattice->mark_stepped();
    iter = (unsigned int)it;
  }

  bool outputInSiUnits = true;
  bool repulsionEnabled = false, repulsionPushed = false; T repulsionConstant_ = 0, repulsionCutoff_ = 0; unsigned int repulsionTimescale = 1;
  bool boundaryRepulsionEnabled = false, boundaryRepulsionPushed = false; T boundaryRepulsionConstant_ = 0, boundaryRepulsionCutoff_ = 0; unsigned int boundaryRepulsionTimescale = 1;
  MultiBlockLattice3D<T, DESCRIPTOR> *lattice = nullptr;
  Config *cfg = nullptr;
  HemoCellFields *cellfields = nullptr;
  unsigned int iter = 0;
  void *preInlet = nullptr;
  string outDir, configFile;
  vector<int> fluidOutputs;
};

inline void HemoCellField::statistics() { if (mechanics) mechanics->statistics(); }

// core/hemoCellField.cpp:38-118: material XML -> mesh + tables on the device
inline void HemoCellField::create_device_type(int model) {
  Config &m = *materialCfg;
  hc_material M; std::memset(&M, 0, sizeof(M));
  M.kLink = m["MaterialModel"]["kLink"].read<T>(); M.kArea = m["MaterialModel"]["kArea"].read<T>();
  M.kVolume = m["MaterialModel"]["kVolume"].read<T>(); M.kBend = m["MaterialModel"]["kBend"].read<T>();
  M.eta_m = m["MaterialModel"]["eta_m"].read<T>(); M.radius = m["MaterialModel"]["radius"].read<T>();
  M.min_triangles = (int)m["MaterialModel"]["minNumTriangles"].read<T>();
  M.aspect_ratio = 0.3;
  if (constructType == ELLIPSOID_FROM_SPHERE) M.aspect_ratio = m["MaterialModel"]["aspectRatio"].read<T>();
  vector<long> inner;
  try {   // mechanics/commonCellConstants.cpp:139-153
    XMLNode *ie = m["MaterialModel"]["InnerEdges"].getOrig();
    for (auto &e : ie->children) { long a, b; if (std::sscanf(e->text.c_str(), "%ld %ld", &a, &b) == 2) { inner.push_back(a); inner.push_back(b); } else pcout << "Inner Edges not read, somethings wrong" << endl; }
  } catch (std::invalid_argument &) {}
  M.inner_edges = inner.empty() ? nullptr : inner.data(); M.n_inner = (int)inner.size() / 2;
  innerEdges = inner;
  if (constructType != RBC_FROM_SPHERE && constructType != ELLIPSOID_FROM_SPHERE) { hlog << "(HemoCell) (AddCellType) construct type " << constructType << " is not supported by the GPU back end" << endl; std::exit(1); }
  hc_check(hcp_celltype_create(&dev, model, constructType, &Parameters::raw(), &M), "hcp_celltype_create");
  int sz[4]; hcp_celltype_sizes(dev, sz);
  numVertex = sz[0]; numTriangles = sz[1];
  vertices.resize(3 * (size_t)sz[0]); triangles.resize(3 * (size_t)sz[1]);
  vector<double> area((size_t)sz[1]); double sc[9];
  hc_check(hcp_celltype_tables(dev, vertices.data(), triangles.data(), nullptr, nullptr, nullptr, area.data(), nullptr, nullptr, sc), "hcp_celltype_tables");
  meshmetric = new MeshMetricsView();
  meshmetric->volume = sc[0];
  for (double a : area) meshmetric->surface += a;
}

inline hc_cells *HemoCellFields::device() {
  if (!dev) hc_check(hcp_create(&dev, hemocell.lattice->device(), &Parameters::raw()), "hcp_create");
  if (!types_bound) {
    for (auto *f : cellFields) hc_check(hcp_add_type(dev, f->dev, (int)f->timescale, nullptr), "hcp_add_type");
    types_bound = true;
  }
  return dev;
}

// io/readPositionsBloodCells.cpp:205-361: "<name>.pos": N, then x y z (um) rx ry rz (deg) per cell
inline void HemoCell::loadParticles() {
  hc_cells *c = cellfields->device();
  int total = 0, cellid = 0;
  for (unsigned int j = 0; j < cellfields->size(); j++) { std::ifstream f(((*cellfields)[j]->name + ".pos").c_str()); int n = 0; if (f.is_open()) f >> n; total += n; }
  cellfields->number_of_cells = total;
  const T posRatio = 1e-6 / Parameters::dx;
  for (unsigned int j = 0; j < cellfields->size(); j++) {
    HemoCellField *field = (*cellfields)[j];
    std::ifstream f((field->name + ".pos").c_str());
    if (!f.is_open()) { std::cout << "*** WARNING! particle positions input file " << field->name << ".pos does not exist!" << std::endl; continue; }
    int n = 0; f >> n;
    hlog << "(readPositionsBloodCells) Particle count in file (" << field->name << "): " << n << "." << endl;
    int placed_n = 0;
    struct Entry { int id; double p[3], a[3]; long brick; };
    std::vector<Entry> entries;
    const plint dims[3] = {lattice->getNx(), lattice->getNy(), lattice->getNz()};
    for (int i = 0; i < n; i++) {
      Entry e; e.id = cellid++;
      f >> e.p[0] >> e.p[1] >> e.p[2] >> e.a[0] >> e.a[1] >> e.a[2];
      for (int d = 0; d < 3; d++) { e.a[d] *= PI / 180.0; e.a[d] *= -1.0; e.p[d] = e.p[d] * posRatio; }   // :228-229, :349
      // A .pos file may cover more than this domain (examples/pipeflow/RBC.pos does).  The reference places such cells
      // in the particle envelope, where no block owns them: they never count (centerLocal, helper/cellInfo.cpp:97)
      // and deleteNonLocalParticles (core/hemoCellFields.cpp:676-688) removes them at the first particle update.
      // Here they are not placed at all.
      bool local = true;
      for (int d = 0; d < 3; d++) if (!(e.p[d] > -0.5 && e.p[d] <= (double)dims[d] - 0.5)) local = false;
      if (!local) continue;
      // storage order = bricks of 32 lattice units, x-major: cells that are neighbours in space become neighbours in
      // memory, and the per-cell IBM kernels hand contiguous ranges of cells to one L2 (the cell ids stay those of the
      // file; the reference's 33 % hematocrit harness, whose .pos is in random order, runs 7 % faster this way)
      e.brick = ((long)(e.p[0] / 32.0) * 4096 + (long)(e.p[1] / 32.0)) * 4096 + (long)(e.p[2] / 32.0);
      entries.push_back(e);
    }
    std::stable_sort(entries.begin(), entries.end(), [](const Entry &x, const Entry &y) { return x.brick < y.brick || (x.brick == y.brick && x.p[2] < y.p[2]); });
    for (const Entry &e : entries) {
      int placed = 0;
      hc_check(hcp_add_cell(c, (int)j, e.id, e.p, e.a, (double)field->minimumDistanceFromSolid, &placed), "hcp_add_cell");
      placed_n += placed;
    }
    hlog << "(readPositionsBloodCells) " << placed_n << " complete " << field->name << " cells placed." << endl;
  }
  long it = iter;   // forces of the initial configuration, as the first applyConstitutiveModel would give them
  hc_check(hcp_mechanics(c, it, 1), "hcp_mechanics");
}

// ------------------------------------------------------------------ checkpoint / resume (core/hemoCellFields.cpp:240-319)
// The reference writes Palabos' own parallelIO dumps plus checkpoint.xml (a copy of the config under a <Checkpoint>
// root with the iteration); the binary format here is this back end's own: populations in the reference node
// order, then per type the cell ids and the vertex position / velocity / force arrays.
inline void HemoCell::saveCheckPoint() {
  const string dir = outDir + "/checkpoint";
  mkdir(dir.c_str(), 0755);
  rename((dir + "/checkpoint.bin").c_str(), (dir + "/checkpoint.bin.old").c_str());   // :283-290 keeps the previous one
  rename((dir + "/checkpoint.xml").c_str(), (dir + "/checkpoint.xml.old").c_str());
  hc_lattice *d = lattice->device(); hc_cells *c = cellfields->device();
  const size_t n = (size_t)lattice->nx * lattice->ny * lattice->nz;
  vector<double> f(n * HC_Q);
  hc_check(hcl_download_populations(d, f.data()), "hcl_download_populations");
  std::ofstream o((dir + "/checkpoint.bin").c_str(), std::ios::binary);
  const long hdr[6] = {0x48434b50, (long)iter, lattice->nx, lattice->ny, lattice->nz, (long)cellfields->size()};
  o.write((const char *)hdr, sizeof(hdr));
  o.write((const char *)f.data(), (std::streamsize)(f.size() * sizeof(double)));
  long nvt = 0, nct = 0; hcp_counts(c, &nvt, &nct, nullptr);
  vector<long> ids((size_t)nct); if (nct) hcp_download_cell_ids(c, ids.data());
  o.write((const char *)&nct, sizeof(long)); o.write((const char *)ids.data(), (std::streamsize)(ids.size() * sizeof(long)));
  for (unsigned int t = 0; t < cellfields->size(); t++) { long fv, nc; hcp_type_range(c, (int)t, &fv, &nc); o.write((const char *)&nc, sizeof(long)); }
  for (int what = 0; what < 3; what++) {
    vector<double> a(3 * (size_t)nvt); if (nvt) hc_check(hcp_download(c, what, a.data()), "hcp_download");
    o.write((const char *)a.data(), (std::streamsize)(a.size() * sizeof(double)));
  }
  std::ofstream x((dir + "/checkpoint.xml").c_str());
  x << "<?xml version=\"1.0\" ?>\n<Checkpoint>\n<General><Iteration>" << iter << "</Iteration><OutDirectory>" << outDir << "</OutDirectory></General>\n";
  std::ifstream cfgin(configFile.c_str()); string line; bool first = true;
  while (std::getline(cfgin, line)) { if (first && line.find("<?xml") != string::npos) { first = false; continue; } x << line << "\n"; }
  x << "</Checkpoint>\n";
  hlog << "(HemoCell) (saveCheckPoint) saved iteration " << iter << " to " << dir << endl;
}

inline void HemoCell::loadCheckPoint() {
  const string dir = outDir + "/checkpoint";
  std::ifstream in((dir + "/checkpoint.bin").c_str(), std::ios::binary);
  if (!in.is_open()) { hlog << "(HemoCell) (loadCheckPoint) " << dir << "/checkpoint.bin not found" << endl; std::exit(1); }
  long hdr[6]; in.read((char *)hdr, sizeof(hdr));
  if (hdr[0] != 0x48434b50 || hdr[2] != lattice->nx || hdr[3] != lattice->ny || hdr[4] != lattice->nz || hdr[5] != (long)cellfields->size()) {
    hlog << "(HemoCell) (loadCheckPoint) checkpoint does not match this case (lattice size / cell types)" << endl; std::exit(1);
  }
  hc_lattice *d = lattice->device(); hc_cells *c = cellfields->device();
  const size_t n = (size_t)lattice->nx * lattice->ny * lattice->nz;
  vector<double> f(n * HC_Q);
  in.read((char *)f.data(), (std::streamsize)(f.size() * sizeof(double)));
  hc_check(hcl_upload_populations(d, f.data()), "hcl_upload_populations");
  long nct = 0; in.read((char *)&nct, sizeof(long));
  vector<long> ids((size_t)nct); in.read((char *)ids.data(), (std::streamsize)(ids.size() * sizeof(long)));
  vector<long> per_type(cellfields->size()); for (auto &v : per_type) in.read((char *)&v, sizeof(long));
  // recreate the cells (placement is overwritten by the stored state right below)
  long k = 0, nvt = 0;
  for (unsigned int t = 0; t < cellfields->size(); t++)
    for (long i = 0; i < per_type[t]; i++, k++) {
      const double centre[3] = {lattice->nx * 0.5, lattice->ny * 0.5, lattice->nz * 0.5}, ang[3] = {0, 0, 0}; int placed = 0;
      hc_check(hcp_add_cell_unchecked(c, (int)t, ids[(size_t)k], centre, ang), "hcp_add_cell_unchecked");
      (void)placed; nvt += (*cellfields)[t]->numVertex;
    }
  cellfields->number_of_cells = (int)nct;
  for (int what = 0; what < 3; what++) {
    vector<double> a(3 * (size_t)nvt); in.read((char *)a.data(), (std::streamsize)(a.size() * sizeof(double)));
    if (nvt) hc_check(hcp_upload(c, what, a.data()), "hcp_upload");
  }
  iter = (unsigned int)hdr[1];
  lattice->mark_stepped();
  hlog << "(HemoCell) (loadCheckPoint) resumed at iteration " << iter << endl;
}

// ------------------------------------------------------------------ helper/cellInfo.h
struct CellInformation {
  hemo::Array<T, 3> position; T volume = 0, area = 0, stretch = 0; hemo::Array<T, 6> bbox; pluint cellType = 0; plint base_cell_id = 0; bool centerLocal = true; plint blockId = 0;
};
struct CellInformationFunctionals {
  static map<int, CellInformation> &info() { static map<int, CellInformation> m; return m; }
  static void fill(HemoCell *h, bool vol, bool area, bool pos, bool bbox, bool stretch) {
    hc_cells *c = h->cellfields->device();
    long nvt = 0, nct = 0; hcp_counts(c, &nvt, &nct, nullptr);
    vector<long> ids((size_t)nct); if (nct) hcp_download_cell_ids(c, ids.data());
    vector<double> allpos; if (stretch) { allpos.resize(3 * (size_t)nvt); if (nvt) hcp_download(c, 0, allpos.data()); }
    long first_cell = 0;
    for (unsigned int t = 0; t < h->cellfields->size(); t++) {
      long fv = 0, nc = 0; hcp_type_range(c, (int)t, &fv, &nc);
      if (nc == 0) continue;
      vector<double> V((size_t)nc), A((size_t)nc), B(6 * (size_t)nc), P(3 * (size_t)nc);
      hc_check(hcp_cell_info(c, (int)t, V.data(), A.data(), B.data(), P.data()), "hcp_cell_info");
      const int nv = (*h->cellfields)[t]->numVertex;
      for (long k = 0; k < nc; k++) {
        CellInformation &ci = info()[(int)ids[(size_t)(first_cell + k)]];
        ci.cellType = t; ci.base_cell_id = ids[(size_t)(first_cell + k)];
        if (vol) ci.volume = V[(size_t)k];
        if (area) ci.area = A[(size_t)k];
        if (pos) for (int d = 0; d < 3; d++) ci.position[d] = P[3 * (size_t)k + d];
        if (bbox) for (int d = 0; d < 6; d++) ci.bbox[d] = B[6 * (size_t)k + d];
        if (stretch) {   // helper/cellInfo.cpp:124-138: largest vertex-vertex distance
          T mx = 0; const double *pp = allpos.data() + 3 * (size_t)(fv + k * nv);
          for (int i = 0; i < nv - 1; i++) for (int jv = i + 1; jv < nv; jv++) {
            const T d2 = (pp[3 * i] - pp[3 * jv]) * (pp[3 * i] - pp[3 * jv]) + (pp[3 * i + 1] - pp[3 * jv + 1]) * (pp[3 * i + 1] - pp[3 * jv + 1]) + (pp[3 * i + 2] - pp[3 * jv + 2]) * (pp[3 * i + 2] - pp[3 * jv + 2]);
            mx = std::max(mx, d2);
          }
          ci.stretch = std::sqrt(mx);
        }
      }
      first_cell += nc;
    }
  }
  static void calculateCellVolume(HemoCell *h) { fill(h, true, false, false, false, false); }
  static void calculateCellArea(HemoCell *h) { fill(h, false, true, false, false, false); }
  static void calculateCellPosition(HemoCell *h) { fill(h, false, false, true, false, false); }
  static void calculateCellBoundingBox(HemoCell *h) { fill(h, false, false, false, true, false); }
  static void calculateCellStretch(HemoCell *h) { fill(h, false, false, false, false, true); }
  static void calculateCellInformation(HemoCell *h) { fill(h, true, true, true, true, false); }
  static void clear_list() { info().clear(); }
  static pluint getTotalNumberOfCells(HemoCell *h) { long nc = 0; hcp_counts(h->cellfields->device(), nullptr, &nc, nullptr); return (pluint)nc; }
  static pluint getNumberOfCellsFromType(HemoCell *h, string type) { long nc = 0; hcp_type_range(h->cellfields->device(), (int)(*h->cellfields)[type]->ctype, nullptr, &nc); return (pluint)nc; }
};
#define info_per_cell info()

// ------------------------------------------------------------------ helper/fluidInfo.h, helper/particleInfo.h
struct FluidStatistics { T min = 0, max = 0, avg = 0; pluint ncells = 0; };
struct FluidInfo {
  // helper/fluidInfo.cpp:33-118: device reductions (hcl_fluid_stats), folded deterministically
  static FluidStatistics stat(HemoCell *h, int what) {
    double o[3]; long n = 0;
    hc_check(hcl_fluid_stats(h->lattice->device(), what, o, &n), "hcl_fluid_stats");
    FluidStatistics s; s.min = o[0]; s.max = o[1]; s.ncells = (pluint)n; s.avg = n ? o[2] / (double)n : 0;
    return s;
  }
  static FluidStatistics calculateVelocityStatistics(HemoCell *h) { return stat(h, 0); }
  static FluidStatistics calculateForceStatistics(HemoCell *h) { return stat(h, 1); }
};
struct ParticleStatistics { T min = 0, max = 0, avg = 0; pluint ncells = 0; };
struct ParticleInfo {
  // helper/particleInfo.cpp:30-140: device reduction over the owned vertices (hcp_vertex_stats)
  static ParticleStatistics stat(HemoCell *h, int what) {
    double o[3]; long n = 0;
    hc_check(hcp_vertex_stats(h->cellfields->device(), what, o, &n), "hcp_vertex_stats");
    ParticleStatistics s; s.min = o[0]; s.max = o[1]; s.ncells = (pluint)n; s.avg = n ? o[2] / (double)n : 0;
    return s;
  }
  static ParticleStatistics calculateForceStatistics(HemoCell *h) { return stat(h, 2); }
  static ParticleStatistics calculateVelocityStatistics(HemoCell *h) { return stat(h, 1); }
};

// ------------------------------------------------------------------ io/writeCellInfoCSV.cpp:30-77
inline void writeCellInfo_CSV(HemoCell &h) {
  CellInformationFunctionals::clear_list();
  CellInformationFunctionals::calculateCellInformation(&h);
  for (unsigned int t = 0; t < h.cellfields->size(); t++) {
    char it[32]; std::snprintf(it, sizeof(it), "%012u", h.iter);
    std::ofstream f((h.outDir + "/csv/" + (*h.cellfields)[t]->name + "." + it + ".csv").c_str());
    f << "X,Y,Z,area,volume,atomic_block,cellId,baseCellId,velocity_x,velocity_y,velocity_z" << std::endl;   // :52
    for (auto &kv : CellInformationFunctionals::info()) {
      if (kv.second.cellType != t) continue;
      const CellInformation &c = kv.second;
      f << c.position[0] << "," << c.position[1] << "," << c.position[2] << "," << c.area << "," << c.volume << ",0," << kv.first << "," << c.base_cell_id << ",0,0,0" << std::endl;
    }
  }
  CellInformationFunctionals::clear_list();
}

}  // namespace hemo
#include "hdf5_output.h"
namespace hemo {

// core/hemoCell.cpp:221-287: <out>/hdf5/<iter>/ with one file per cell type plus the fluid file, and the CSV summary
inline void HemoCell::writeOutput() {
#ifdef HEMOCELL_WITH_HDF5
  mkdir((outDir + "/hdf5").c_str(), 0755);
  const string dir = outDir + "/hdf5/" + zeroPadNumber(iter);
  mkdir(dir.c_str(), 0755);
  hlog << "(HemoCell) (Output) writing output at timestep " << iter << " (" << iter * Parameters::dt << " s)" << endl;
  for (unsigned int t = 0; t < cellfields->size(); t++) writeCellField3D_HDF5(*this, *(*cellfields)[t], dir);
  writeFluidField_HDF5(*this, dir);
#else
  hlog << "(HemoCell) (Output) built without HEMOCELL_WITH_HDF5: only the CSV cell summary is written at " << iter << endl;
#endif
  writeCellInfo_CSV(*this);
}

// ------------------------------------------------------------------ helper/hemoCellStretch.h
class HemoCellStretch {
 public:
  HemoCellStretch(HemoCellField &cellfield_, unsigned int n_forced_lsps_, T external_force_) : cellfield(cellfield_) {
    HemoCellFields &cf = *cellfield.cellFields;
    if (cf.number_of_cells != 1) { pcout << "(HemoCellStretch) Refusing to run with more or less than 1 cell" << endl; std::exit(1); }
    n_forced_lsps = n_forced_lsps_; external_force = external_force_ / n_forced_lsps;
    // FindForcedLsps (helper/hemoCellStretch.cpp:30-60): the n vertices with the smallest / largest x
    hc_cells *c = cf.device();
    long nvt = 0; hcp_counts(c, &nvt, nullptr, nullptr);
    vector<double> pos(3 * (size_t)nvt); hc_check(hcp_download(c, 0, pos.data()), "hcp_download");
    vector<long> order((size_t)nvt); for (long i = 0; i < nvt; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return pos[3 * a] < pos[3 * b]; });
    lower_lsps.clear(); upper_lsps.clear();
    for (unsigned int i = 0; i < n_forced_lsps; i++) { lower_lsps.push_back(order[i]); upper_lsps.push_back(order[(size_t)nvt - 1 - i]); }
  }
  void applyForce() {   // :63-78, :99-107
    if (cellfield.timescale != 1) { pcout << "Refusing to stretch with particle update timestep larger than 1" << endl; std::exit(1); }
    vector<long> idx; vector<double> f;
    for (long v : lower_lsps) { idx.push_back(v); f.push_back(-external_force * scale); f.push_back(0); f.push_back(0); }
    for (long v : upper_lsps) { idx.push_back(v); f.push_back(external_force * scale); f.push_back(0); f.push_back(0); }
    hc_check(hcp_add_vertex_force(cellfield.cellFields->device(), idx.data(), (int)idx.size(), f.data()), "hcp_add_vertex_force");
  }
  HemoCellField &cellfield;
  vector<plint> lower_lsps, upper_lsps;
  unsigned int n_forced_lsps = 0; T external_force = 0; T scale = 1.0;
};

// ------------------------------------------------------------------ helper/voxelizeDomain.cpp:76-152 replacement for synthetic pipes
// analytic cylinder along x: flag 1 = fluid, 0 = outside, as getFlagMatrixFromSTL returns it
inline void getFlagMatrixCylinder(plint nx, plint ny, plint nz, std::unique_ptr<VoxelizedDomain3D<T>> &vox, std::unique_ptr<MultiScalarField3D<int>> &flags) {
  MultiBlockManagement3D m; m.nx = nx; m.ny = ny; m.nz = nz;
  vox.reset(new VoxelizedDomain3D<T>(m));
  flags.reset(new MultiScalarField3D<int>(nx, ny, nz, 0));
  const T R = (ny - 2) / 2.0, cy = (ny - 1) / 2.0, cz = (nz - 1) / 2.0;
  for (plint x = 0; x < nx; x++) for (plint y = 0; y < ny; y++) for (plint z = 0; z < nz; z++)
    flags->get(x, y, z) = ((y - cy) * (y - cy) + (z - cz) * (z - cz) > R * R) ? 0 : 1;
}

}  // namespace hemo
