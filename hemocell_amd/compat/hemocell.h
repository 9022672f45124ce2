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
#define OUTPUT_OMEGA 12
#define OUTPUT_BOUNDARY 13
#define OUTPUT_BINDING_SITES 14
#define OUTPUT_INTERIOR_POINTS 15
#define OUTPUT_SHEAR_RATE 16
#define OUTPUT_STRAIN_RATE 17
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
  bool to_stdout; std::ofstream file; Logger *sink = nullptr;   // hlogfile writes into hlog's file (config/logfile.h:61-76)
  string filename;
  explicit Logger(bool s) : to_stdout(s) {}
  std::ofstream &out() { return sink ? sink->file : file; }
  template <typename V> Logger &operator<<(const V &v) { if (to_stdout) std::cout << v; if (out().is_open()) out() << v; return *this; }
  Logger &operator<<(std::ostream &(*f)(std::ostream &)) { if (to_stdout) std::cout << f; if (out().is_open()) out() << f; return *this; }
};
inline Logger &hlog_instance() { static Logger l(true); return l; }
inline Logger &hlogfile_instance() { static Logger l(false); if (!l.sink) l.sink = &hlog_instance(); return l; }
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
  // <Checkpoint><General><name> of a checkpoint.xml (core/hemoCellFields.cpp:244-248); empty if absent
  string checkpointGeneral(const string &name) const {
    XMLNode *c = doc->child("Checkpoint"), *g = c ? c->child("General") : nullptr, *n = g ? g->child(name) : nullptr;
    if (!n) return string();
    const size_t a = n->text.find_first_not_of(" \t\r\n"), b = n->text.find_last_not_of(" \t\r\n");
    return a == string::npos ? string() : n->text.substr(a, b - a + 1);
  }
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
struct Global {   // config/config.h:80-96 (ConfigValues) plus this rank's place in the run
  ProfilerView statistics; bool cellsDeletedInfo = false; bool hemoCellInitialized = false;
  int rank = 0, world = 1;
  string checkpointDirectory;   // config/config.h:84, set by loadDirectories
};
static Global global;

class HemoCell;
class HemoCellFields;
class HemoCellField;
class HemoCellParticleField;
}  // namespace hemo
namespace plb { template <class PF> struct MultiParticleField3D; }
namespace hemo {
class HemoCellParticleField;
template <typename U> struct CEPAC_DESCRIPTOR_T { enum { d = 3, q = 19 }; };
#define CEPAC_DESCRIPTOR hemo::CEPAC_DESCRIPTOR_T   // config/constant_defaults.h:62-65 (only named in signatures here)

// ------------------------------------------------------------------ core/hemoCellParticle.h:38-187
// The plugin-facing view of one membrane vertex.  On this back end the vertices live on the GPU as structure-of-arrays;
// HemoCellParticleField::particles is filled from them on demand in the reference's own 120-byte record.
class HemoCellParticle {
 public:
  struct serializeValues_t {
    hemo::Array<T, 3> v, position, force, force_repulsion;
    plint cellId; uint16_t vertexId; unsigned int restime; unsigned char celltype;
  };
  serializeValues_t sv;
  hemo::Array<T, 3> force_total;
  plint tag = 0;
  hemo::Array<T, 3> *force_volume = &sv.force, *force_bending = &sv.force, *force_link = &sv.force, *force_area = &sv.force,
                    *force_visc = &sv.force, *force_inner_link = &sv.force;
  HemoCellParticle() { std::memset(&sv, 0, sizeof(sv)); }
  HemoCellParticle(const serializeValues_t &sv_) : sv(sv_) {}
  HemoCellParticle(hemo::Array<T, 3> position_, plint cellId_, plint vertexId_, pluint celltype_) {
    std::memset(&sv, 0, sizeof(sv)); sv.position = position_; sv.cellId = cellId_; sv.vertexId = (uint16_t)vertexId_; sv.celltype = (unsigned char)celltype_;
  }
  HemoCellParticle(const HemoCellParticle &o) : sv(o.sv), force_total(o.force_total), tag(o.tag) {}   // force pointers repoint to the copy's own sv.force
  HemoCellParticle &operator=(const HemoCellParticle &o) { sv = o.sv; force_total = o.force_total; tag = o.tag; return *this; }
  void repoint_force_vectors() { force_volume = force_bending = force_link = force_area = force_visc = force_inner_link = &sv.force; }
  void advance() { for (int d = 0; d < 3; d++) sv.position[d] += sv.v[d]; }   // core/hemoCellParticle.h:188-203 (Euler)
  plint getTag() const { return tag; }
  void setTag(plint t) { tag = t; }
};
static_assert(sizeof(HemoCellParticle::serializeValues_t) == 120, "serializeValues_t is the 120-byte record of the C ABI (hcp_download_records)");

// core/immersedBoundaryMethod.h:62: the default IBM kernel.  On this back end the phi2 stencil is evaluated inside the HIP
// kernels; the function only exists so that HemoCellField::kernelMethod has its reference default and a driver that
// installs another kernel can be recognised (and refused).
inline void interpolationCoefficientsPhi2(plb::BlockLattice3D<T, DESCRIPTOR> &, HemoCellParticle &) {
  std::cerr << "(HemoCell) (GPU backend) interpolationCoefficientsPhi2 runs on the device (csrc/ibm.hip); it cannot be called on the host" << std::endl;
  std::exit(1);
}

// ------------------------------------------------------------------ mechanics/commonCellConstants.h:38-85
class CommonCellConstants {
 public:
  static CommonCellConstants CommonCellConstantsConstructor(HemoCellField &, Config &modelCfg_);
  HemoCellField *cellFieldPtr = nullptr;
  std::vector<hemo::Array<plint, 3>> triangle_list;
  std::vector<hemo::Array<plint, 2>> edge_list;
  std::vector<T> edge_length_eq_list, edge_angle_eq_list, surface_patch_center_dist_eq_list;
  std::vector<hemo::Array<plint, 2>> edge_bending_triangles_list, edge_bending_triangles_outer_points;
  std::vector<T> triangle_area_eq_list;
  std::vector<hemo::Array<plint, 6>> vertex_vertexes, vertex_edges;
  std::vector<hemo::Array<signed int, 6>> vertex_edges_sign;
  std::vector<unsigned int> vertex_n_vertexes;
  T volume_eq = 0, area_mean_eq = 0, edge_mean_eq = 0, angle_mean_eq = 0;
  std::vector<hemo::Array<plint, 2>> inner_edge_list;
  std::vector<T> inner_edge_length_eq_list;
};

// ------------------------------------------------------------------ mechanics/cellMechanics.h:36-101: the plugin base class
// A model the GPU evaluates (RbcHighOrderModel, PltSimpleModel below) reports onDevice(); any other subclass compiles
// against the reference's interface unchanged and is refused by addCellType with the reference's log + exit(1), because
// its ParticleMechanics is host code working on HemoCellParticle pointers that this back end never materialises per step.
class CellMechanics {
 public:
  const CommonCellConstants cellConstants;
  Config &cfg;
  CellMechanics(HemoCellField &cellfield, Config &modelCfg_) : cellConstants(CommonCellConstants::CommonCellConstantsConstructor(cellfield, modelCfg_)), cfg(modelCfg_) {}
  virtual ~CellMechanics() {}
  virtual void ParticleMechanics(std::map<int, std::vector<HemoCellParticle *>> &, const std::map<int, bool> &, pluint ctype) = 0;
  virtual void statistics() = 0;
  virtual void solidifyMechanics(const std::map<int, std::vector<int>> &, std::vector<HemoCellParticle> &, plb::BlockLattice3D<T, DESCRIPTOR> *,
                                 plb::BlockLattice3D<T, CEPAC_DESCRIPTOR> *, pluint, HemoCellParticleField &) {}
  virtual bool onDevice() const { return false; }   // this back end's extension
  T calculate_kLink(Config &c, plb::MeshMetrics<T> &) { return c["MaterialModel"]["kLink"].read<T>() * Parameters::kBT_lbm / (7.5e-9 / Parameters::dx); }
  T calculate_kBend(Config &c, plb::MeshMetrics<T> &) { return c["MaterialModel"]["kBend"].read<T>() * Parameters::kBT_lbm / (5e-7 / Parameters::dx); }
  T calculate_kVolume(Config &c, plb::MeshMetrics<T> &) { return c["MaterialModel"]["kVolume"].read<T>() * (1280.0 / cellConstants.triangle_list.size()) * Parameters::kBT_lbm / (5e-7 / Parameters::dx); }
  T calculate_kArea(Config &c, plb::MeshMetrics<T> &) { return c["MaterialModel"]["kArea"].read<T>() * (1280.0 / cellConstants.triangle_list.size()) * Parameters::kBT_lbm / (5e-7 / Parameters::dx); }
  T calculate_etaM(Config &c) { return c["MaterialModel"]["eta_m"].read<T>() * Parameters::dx / Parameters::dt / Parameters::df; }
};
typedef plb::MeshMetrics<T> MeshMetricsView;

class HemoCellField {
 public:
  string name; unsigned int ctype = 0; int constructType = 0;
  Config *materialCfg = nullptr;
  unsigned int timescale = 1;
  T minimumDistanceFromSolid = 0;   // micrometres.  core/hemoCellField.h:64 declares it unsigned int (0.5 -> 0); the fraction is kept here
                                    // because only then does examples/pipeflow keep the 42 cells the reference's tests assert
  int numVertex = 0, numTriangles = 0;
  T volume = 0, volumeFractionOfLspPerNode = 0;   // core/hemoCellField.h:60-61, from <MaterialModel><Volume> (core/hemoCellField.cpp:92-98)
  plb::MeshMetrics<T> *meshmetric = nullptr;
  CellMechanics *mechanics = nullptr;
  void (*kernelMethod)(plb::BlockLattice3D<T, DESCRIPTOR> &, HemoCellParticle &) = interpolationCoefficientsPhi2;   // core/hemoCellField.h:67
  vector<hemo::Array<plint, 3>> triangle_list;
  int device_model = -1;   // HC_MODEL_* the device type was built for
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
  // the rest of core/hemoCellField.h:41-83
  T restingCellVolume = 0; bool outputTriangles = false, doSolidifyMechanics = false, doInteriorViscosity = false; T interiorViscosityTau = 1.0;
  string getIdentifier() const { return name; }
  int getNumberOfCells_Global() const { return 0; }   // core/hemoCellField.cpp:192 answers 0 as well
  void setOutputVariables(const vector<int> &outputs) {   // :139-147 (OUTPUT_TRIANGLES stays in the list here; the writer looks for it)
    desiredOutputVariables = outputs;
    outputTriangles = std::find(outputs.begin(), outputs.end(), OUTPUT_TRIANGLES) != outputs.end();
  }
  inline plb::MultiParticleField3D<HemoCellParticleField> *getParticleField3D();
  inline plb::MultiParticleField3D<HemoCellParticleField> *getParticleArg();
  inline plb::MultiBlockLattice3D<T, DESCRIPTOR> *getFluidField3D();
  void addSingleCell(hemo::Array<T, 3>, plint) { std::cerr << "(HemoCell) (ParticleType) addSingleCell not implemented, but might definitely be nice to have" << std::endl; std::exit(1); }   // :171-174, word for word
};

// selects the device model before the CellMechanics base builds the tables
template <int MODEL>
struct DeviceModelTag { explicit DeviceModelTag(HemoCellField &f) { f.device_model = MODEL; } };

template <int MODEL>
class DeviceMechanics : private DeviceModelTag<MODEL>, public CellMechanics {
 public:
  DeviceMechanics(Config &modelCfg_, HemoCellField &field) : DeviceModelTag<MODEL>(field), CellMechanics(field, modelCfg_), cellField(field) {
    double sc[9];
    hc_check(hcp_celltype_tables(field.dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sc), "hcp_celltype_tables");
    k_volume = sc[4]; k_area = sc[5]; k_link = sc[6]; k_bend = sc[7]; eta_m = sc[8];
  }
  bool onDevice() const override { return true; }
  // mechanics/rbcHighOrderModel.cpp:38-207 / mechanics/pltSimpleModel.cpp:44-208 are evaluated by mechanics_kernel<MODEL>
  // (csrc/mechanics.hip) through hcp_mechanics; nothing ever calls the host signature
  void ParticleMechanics(std::map<int, std::vector<HemoCellParticle *>> &, const std::map<int, bool> &, pluint) override {
    hlog << "(HemoCell) (GPU backend) " << cellField.name << ": ParticleMechanics is evaluated on the device by HemoCell::iterate(); the host entry point is not available" << endl;
    std::exit(1);
  }
  void statistics() override {   // mechanics/rbcHighOrderModel.cpp:209-216
    hlog << "(Cell-mechanics model) parameters for " << cellField.name << " cellfield" << endl;
    hlog << "\t k_link:   " << k_link << endl << "\t k_area:   " << k_area << endl << "\t k_bend: : " << k_bend << endl
         << "\t k_volume: " << k_volume << endl << "\t eta_m:    " << eta_m << endl;
  }
  HemoCellField &cellField;
  T k_volume = 0, k_area = 0, k_link = 0, k_bend = 0, eta_m = 0;
};
typedef DeviceMechanics<HC_MODEL_RBC_HO> RbcHighOrderModel;   // mechanics/rbcHighOrderModel.h
typedef DeviceMechanics<HC_MODEL_PLT_SIMPLE> PltSimpleModel;  // mechanics/pltSimpleModel.h

// ------------------------------------------------------------------ core/hemoCellParticleField.h:39-207
// The reference keeps std::vector<HemoCellParticle> per atomic block; here the vertices live on the GPU and this object is
// a host VIEW of this rank's block: refresh() fills `particles` from the device in the reference's record
// (hcp_download_records), the lookup maps are derived from it as the reference derives them (update_ppc / update_lpc,
// core/hemoCellParticleField.cpp:395-467), upload() hands edited records back.
class HemoCellParticleField {
 public:
  explicit HemoCellParticleField(HemoCellFields &f) : fields(f) {}
  vector<HemoCellParticle> particles;
  plb::Box3D localDomain, boundingBox;
  plb::BlockLattice3D<T, DESCRIPTOR> *atomicLattice = nullptr;
  pluint atomicBlockId = 0, envelopeSize = 0;
  static HemoCellFields *&cellFieldsRef() { static HemoCellFields *p = nullptr; return p; }
  pluint getsize() { refresh(); return particles.size(); }
  plb::Box3D &getBoundingBox() { return boundingBox; }
  inline void refresh();
  inline void upload();
  inline int deleteIncompleteCells(bool verbose = true);
  const map<int, vector<int>> &get_particles_per_cell() { refresh(); return _particles_per_cell; }
  const map<int, bool> &get_lpc() { refresh(); return _lpc; }
  const vector<vector<unsigned int>> &get_particles_per_type() { refresh(); return _particles_per_type; }
  bool isContainedABS(const hemo::Array<T, 3> &pos, const plb::Box3D &box) const {
    return (pos[0] > box.x0 - 0.5) && (pos[0] <= box.x1 + 0.5) && (pos[1] > box.y0 - 0.5) && (pos[1] <= box.y1 + 0.5) && (pos[2] > box.z0 - 0.5) && (pos[2] <= box.z1 + 0.5);
  }
  void invalidate() { fresh_iter = -1; }
 private:
  HemoCellFields &fields;
  long fresh_iter = -1;
  map<int, vector<int>> _particles_per_cell; map<int, bool> _lpc; vector<vector<unsigned int>> _particles_per_type;
};
}  // namespace hemo
namespace plb {
template <class PF>
struct MultiParticleField3D {   // particles/multiParticleField3D.h: one atomic block per rank here
  explicit MultiParticleField3D(PF *p) : pf(p) {}
  PF &getComponent(plint) { return *pf; }
  PF *pf;
};
}  // namespace plb
namespace hemo {

class HemoCellFields {
 public:
  explicit HemoCellFields(HemoCell &h) : hemocell(h), particleField(*this), immersedParticles(new plb::MultiParticleField3D<HemoCellParticleField>(&particleField)) { HemoCellParticleField::cellFieldsRef() = this; }
  ~HemoCellFields() { for (auto *f : cellFields) delete f; if (dev) hcp_destroy(dev); delete immersedParticles; }
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
  HemoCellParticleField particleField;
  plb::MultiParticleField3D<HemoCellParticleField> *immersedParticles;   // core/hemoCellFields.h:161
  // core/hemoCellFields.h:103-158: the phase methods drivers may call one by one
  inline void spreadParticleForce();
  inline void interpolateFluidVelocity();
  inline void advanceParticles();
  inline void applyConstitutiveModel(bool forced = false);
  inline void deleteIncompleteCells(bool verbose = true);
  inline void syncEnvelopes() {}   // part of the slab schedule inside hc_iterate (csrc/slab.hip)
  inline void deleteNonLocalParticles(int) {}   // likewise: a copy nobody refreshes any more is dropped at the next envelope synchronisation
  inline void applyRepulsionForce();            // core/hemoCellFields.cpp:527-540 (cadence is the caller's business, as in the reference)
  inline void applyBoundaryRepulsionForce();
  inline void populateBoundaryParticles() {}    // the flag map of the wall nodes is built by enableBoundaryParticles
  inline void separate_force_vectors() {}       // the separate vectors are evaluated when they are written (compat/hdf5_output.h)
  inline void unify_force_vectors() {}
  inline void updateResidenceTime(unsigned int) {}   // "Res Time" is written as zeros
  void refuse(const char *what);   // log + exit(1), like HemoCell::refuse
  inline void solidifyCells() { refuse("solidify mechanics (solidifyCells)"); }
  inline void prepareSolidification() { refuse("solidify mechanics (prepareSolidification)"); }
  inline void populateBindingSites(plb::Box3D * = nullptr) { refuse("binding sites (populateBindingSites)"); }
  inline void findInternalParticleGridPoints() { refuse("interior viscosity (findInternalParticleGridPoints)"); }
  inline void internalGridPointsMembrane() { refuse("interior viscosity (internalGridPointsMembrane)"); }
  inline void createCEPACfield() { refuse("the CEPAC field (createCEPACfield)"); }
};

// ------------------------------------------------------------------ hemocell.h:68-253
class HemoCell {
 public:
  enum class MPIHandle { Internal, External };
  HemoCell(char *configFileName, int argc, char *argv[]) : HemoCell(configFileName, argc, argv, MPIHandle::Internal) {}
  HemoCell(char *configFileName, int, char *[], MPIHandle) {
    // replaces plb::plbInit / MPI_Init (core/hemoCell.cpp:80-86): a run started as several processes (mpirun-style
    // RANK / WORLD_SIZE / LOCAL_RANK or OMPI_* / PMI_* variables) becomes one x-slab per process and GPU
    if (global.hemoCellInitialized) { std::cerr << "(HemoCell) only one HemoCell per process (core/hemoCell.cpp:75-79)" << std::endl; std::exit(1); }
    global.hemoCellInitialized = true;
    hc_check(hc_comm_init_env(), "hc_comm_init_env");
    int tr = 0; hc_comm_info(&global.rank, &global.world, &tr);
    if (global.world == 1) hc_check(hc_init(0), "hc_init");
    cfg = new Config(configFileName);
    configFile = configFileName;
    {   // the <hemocell> element as text, for checkpoint.xml (core/hemoCellFields.cpp:283-319); kept from now because a resumed
        // run's configuration IS the checkpoint.xml that saveCheckPoint rotates away before it writes the new one
      std::ifstream cfgin(configFileName); std::stringstream whole; whole << cfgin.rdbuf();
      const string text = whole.str();
      const size_t a = text.find("<hemocell"), b = text.rfind("</hemocell>");
      if (a != string::npos && b != string::npos && b > a) configElement = text.substr(a, b - a) + "</hemocell>\n";
    }
    try { global.cellsDeletedInfo = (*cfg)["verbose"]["cellsDeletedInfo"].read<int>() != 0; } catch (std::invalid_argument &) {}   // config/config.cpp:180
    if (global.rank != 0) hlog_instance().to_stdout = false;   // the log is rank 0's (config/logfile.h)
    loadDirectories(true);
    hlog << "(HemoCell) (Config) reading " << configFileName << endl;
    if (global.world > 1) hlog << "(HemoCell) (GPU backend) " << global.world << " atomic-blocks: one x-slab per rank and GPU, " << (tr == HC_TRANSPORT_RCCL ? "RCCL point-to-point" : "host-staged (ranks share a GPU)") << " neighbour exchange" << endl;
  }
  ~HemoCell() { flush(); hc_synchronize(); delete cellfields; delete lattice; delete cfg; hc_comm_finalize(); global.hemoCellInitialized = false; }   // core/hemoCell.cpp:97-127: the facade owns the driver's lattice

  // config/config.cpp:88-174 (loadDirectories): <parameters><outputDirectory> (default ./tmp) gets _0, _1, ... appended when it
  // exists already; the log goes to <logDirectory>/<logFile> below it (default log/logfile), with .0, .1, ... appended when
  // that exists; checkpoints to <checkpointDirectory> below it.  Rank 0 looks and creates, the others are told.
  static bool path_exists(const string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
  static void mkpath(const string &p) { for (size_t i = 1; i <= p.size(); i++) if (i == p.size() || p[i] == '/') ::mkdir(p.substr(0, i).c_str(), 0777); }
  void loadDirectories(bool edit_out_dir) {
    char names[2][1024]; std::memset(names, 0, sizeof(names));
    if (global.rank == 0) {
      if (edit_out_dir) {
        try {
          outDir = (*cfg)["parameters"]["outputDirectory"].read<string>();
          while (outDir.size() > 1 && outDir[outDir.size() - 1] == '/') outDir.pop_back();
          if (outDir[0] != '/') outDir = "./" + outDir;
        } catch (std::invalid_argument &) { outDir = "./tmp"; }
        if (path_exists(outDir))
          for (int i = 0;; i++) if (!path_exists(outDir + "_" + std::to_string(i))) { outDir += "_" + std::to_string(i); break; }
      }
      mkpath(outDir + "/hdf5");
      string logDir = "log", logName = "logfile";
      try { logDir = (*cfg)["parameters"]["logDirectory"].read<string>(); } catch (std::invalid_argument &) {}
      try { logName = (*cfg)["parameters"]["logFile"].read<string>(); } catch (std::invalid_argument &) {}
      mkpath(outDir + "/" + logDir);
      string file = outDir + "/" + logDir + "/" + logName;
      if (path_exists(file))
        for (int i = 0;; i++) if (!path_exists(file + "." + std::to_string(i))) { file += "." + std::to_string(i); break; }
      if (hlog_instance().file.is_open()) hlog_instance().file.close();
      hlog_instance().filename = file;
      hlog_instance().file.open(file.c_str());
      if (!hlog_instance().file.is_open()) { std::cerr << "(HemoCell) (LogFile) Error opening logfile, exiting" << std::endl; std::exit(1); }
      if (outDir.size() >= sizeof(names[0])) { std::cerr << "(HemoCell) output directory name too long" << std::endl; std::exit(1); }
      std::strcpy(names[0], outDir.c_str());
    }
    hc_check(hc_comm_bcast(names, sizeof(names), 0), "hc_comm_bcast");
    outDir = names[0];
    string chk = "checkpoint";
    try { chk = (*cfg)["parameters"]["checkpointDirectory"].read<string>(); } catch (std::invalid_argument &) {}
    global.checkpointDirectory = outDir + "/" + chk + "/";
  }

  void latticeEquilibrium(T rho, hemo::Array<T, 3> vel) { lattice->eq_rho = rho; for (int d = 0; d < 3; d++) lattice->eq_u[d] = vel[d]; lattice->dirty_layout = true; }
  void initializeCellfield() {
    cellfields = new HemoCellFields(*this);
    int env = 25;   // core/hemoCell.cpp:139 reads it here; core/hemoCellFields.cpp:43 prints it.  Used by loadParticles() on slab runs
    try { env = (*cfg)["domain"]["particleEnvelope"].read<int>(); } catch (const std::invalid_argument &) {}
    hlog << "(Hemocell) (HemoCellFields) (Init) particle envelope: " << env << " [lu]" << std::endl;
  }
  // core/hemoCell.cpp:438-474: the lattice from the block management alone, GuoExternalForceBGKdynamics(1/tau) as the bulk
  // dynamics.  <domain><mABx/y/z> (an explicit block layout) does not apply: every rank holds one x-slab.
  void initializeLattice(plb::MultiBlockManagement3D const &management) {
    if (lattice) { flush(); delete lattice; lattice = nullptr; }
    hlog << "(HemoCell) Using default domain management." << endl;
    lattice = new plb::MultiBlockLattice3D<T, DESCRIPTOR>(management, plb::defaultMultiBlockPolicy3D().getBlockCommunicator(),
                                                          plb::defaultMultiBlockPolicy3D().getCombinedStatistics(),
                                                          plb::defaultMultiBlockPolicy3D().getMultiCellAccess<T, DESCRIPTOR>(),
                                                          new plb::GuoExternalForceBGKdynamics<T, DESCRIPTOR>(1.0 / param::tau));
  }

  template <class Mechanics>
  void addCellType(string name, int constructType) {
    HemoCellField *cellfield = cellfields->addCellType(name, constructType);
    Mechanics *mechanics = new Mechanics(*cellfield->materialCfg, *cellfield);
    cellfield->mechanics = mechanics;
    if (!mechanics->onDevice()) {
      // the reference would call mechanics->ParticleMechanics(map<int, vector<HemoCellParticle*>>&, ...) on the host every
      // material step (core/hemoCellParticleField.cpp:669); this back end has no host particle objects in the loop
      hlog << "(HemoCell) (AddCellType) the mechanics model given for \"" << name << "\" is host code (it overrides CellMechanics::ParticleMechanics); "
           << "the GPU back end evaluates RbcHighOrderModel and PltSimpleModel on the device and cannot run it. Exiting." << endl;
      std::exit(1);
    }
    cellfield->statistics();
  }
  void setOutputs(string name, vector<int> outputs) { (*cellfields)[name]->desiredOutputVariables = outputs; }
  void setFluidOutputs(vector<int> outputs) { fluidOutputs = outputs; }
  void setMaterialTimeScaleSeparation(string name, unsigned int separation) {
    hlog << "(HemoCell) (Timescale Seperation) Setting seperation of " << name << " to " << separation << " timesteps" << endl;
    flush();
    (*cellfields)[name]->timescale = separation;
  }
  void setParticleVelocityUpdateTimeScaleSeparation(unsigned int separation) {
    hlog << "(HemoCell) (Timescale separation) Setting update separation of all particles to " << separation << " timesteps" << endl;
    flush();
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
  void setSystemPeriodicityLimit(unsigned int, int) {}   // how often a cell id may wrap (core/hemoCell.cpp): ids are not shifted here
  // ---- the rest of hemocell.h:86-253.  Features outside the path are refused the way the reference refuses
  // what it cannot do -- a line in the log and exit(1) -- never ignored
  void refuse(const char *what) { hlog << "(HemoCell) (GPU backend) " << what << " is not part of this back end (DESIGN.md section 0), exiting ..." << endl; std::exit(1); }
  void enableSolidifyMechanics(string) { refuse("solidify mechanics (enableSolidifyMechanics)"); }
  void setSolidifyTimeScaleSeperation(unsigned int) { refuse("solidify mechanics (setSolidifyTimeScaleSeperation)"); }
  void setInteriorViscosityTimeScaleSeperation(unsigned int, unsigned int) { refuse("interior viscosity (setInteriorViscosityTimeScaleSeperation)"); }
  void setCEPACOutputs(vector<int>) { refuse("the CEPAC field (setCEPACOutputs)"); }
  void checkExitSignals() {}                                   // core/hemoCell.cpp:300: SIGINT / SIGTERM handling of the reference's run scripts
  void sanityCheck() { sanityCheckDone = true; }               // the facade's own checks run where a setting is used
  // one x-slab per rank and GPU is the only layout: nothing to measure against, nothing to move
  T calculateFractionalLoadImbalance() { hlog << "(HemoCell) (LoadBalancer) one x-slab per rank: fractional load imbalance is reported as 0" << endl; return 0; }
  void doLoadBalance() { hlog << "(HemoCell) (LoadBalancer) one x-slab per rank: nothing to balance" << endl; }
  void doRestructure(bool = true) { hlog << "(HemoCell) (LoadBalancer) one x-slab per rank: nothing to restructure" << endl; }
  bool loadParticlesIsCalled = false, sanityCheckDone = false, partOfpreInlet = false, leesEdwardsBC = false;
  unsigned int lastOutputAt = 0;
  void loadParticles();
  void loadCheckPoint();
  void saveCheckPoint();
  void writeOutput();
  // HemoCell::iterate() (core/hemoCell.cpp:299-376).  A driver calls it once per iteration; the device runs fastest when it is
  // handed several iterations at once (advance, mechanics and the next spread then run beside the collide, DESIGN.md section
  // 4a).  So the call only queues the iteration; the queue is run -- one hc_iterate for all of it -- before anything can
  // observe or change the state: every path to the device goes through lattice->device() / cellfields->device(), which
  // flush first.  What a driver sees (iter, statistics, output, forces it adds between iterations) is unchanged.
  void iterate() {
    if (!lattice->before_access) lattice->before_access = [this] { flush(); };
    lattice->sync_force();   // what the driver wrote since the last iteration zeroed the field; usually the same force again
    if (!pending) {   // first queued iteration: make sure everything it needs exists and settings are pushed
      hc_cells *c = cellfields->device();
      if (boundaryRepulsionEnabled && !boundaryRepulsionPushed) { hc_check(hcp_set_boundary_repulsion(c, boundaryRepulsionConstant_, boundaryRepulsionCutoff_, (int)boundaryRepulsionTimescale), "hcp_set_boundary_repulsion"); boundaryRepulsionPushed = true; }
      if (repulsionEnabled && !repulsionPushed) { hc_check(hcp_set_repulsion(c, repulsionConstant_, repulsionCutoff_, (int)repulsionTimescale), "hcp_set_repulsion"); repulsionPushed = true; }
      lattice->device();
      queued_timescale = cellfields->particleVelocityUpdateTimescale;
    }
    const bool particle_step = iter % cellfields->particleVelocityUpdateTimescale == 0;
    pending++; iter++;
    lattice->force_cleared = true;   // core/hemoCell.cpp:369-371: the iteration ends by zeroing the external field
    // run at once where the reference does something between two iterations that the queue would skip, and bound the queue
    if ((global.cellsDeletedInfo && particle_step) || pending >= 1024) {
      flush();
      if (global.cellsDeletedInfo && particle_step) cellfields->deleteIncompleteCells(true);   // core/hemoCell.cpp:360-363
    }
  }
  void flush() {
    if (!pending || flushing) return;
    flushing = true;
    long it = (long)iter - (long)pending;
    const int n = (int)pending;
    pending = 0;
    hc_check(hc_iterate(lattice->device_now(), cellfields->dev, &it, n, (int)queued_timescale, /*force_limit=*/1, /*compaction look-up cadence=*/1), "iterate");
    lattice->mark_stepped();
    cellfields->particleField.invalidate();
    flushing = false;
  }
  unsigned int pending = 0, queued_timescale = 1; bool flushing = false;

  bool outputInSiUnits = true;
  bool repulsionEnabled = false, repulsionPushed = false; T repulsionConstant_ = 0, repulsionCutoff_ = 0; unsigned int repulsionTimescale = 1;
  bool boundaryRepulsionEnabled = false, boundaryRepulsionPushed = false; T boundaryRepulsionConstant_ = 0, boundaryRepulsionCutoff_ = 0; unsigned int boundaryRepulsionTimescale = 1;
  MultiBlockLattice3D<T, DESCRIPTOR> *lattice = nullptr;
  Config *cfg = nullptr;
  HemoCellFields *cellfields = nullptr;
  unsigned int iter = 0;
  void *preInlet = nullptr;
  string outDir, configFile, configElement;
  vector<int> fluidOutputs;
};

inline void HemoCellField::statistics() {   // core/hemoCellField.cpp:176-182
  hlog << "Cellfield  (+ material model) of " << name << std::endl;
  hlog << "  Volume :" << volume << " µm³ VolumeFraction of lsp per fluid node: " << volumeFractionOfLspPerNode << " %" << endl;
  hlog << "  Nvertex: " << numVertex << endl;
  if (mechanics) mechanics->statistics();
}

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
  triangle_list.resize((size_t)sz[1]);
  for (int t = 0; t < sz[1]; t++) for (int k = 0; k < 3; k++) triangle_list[(size_t)t][(size_t)k] = triangles[3 * (size_t)t + (size_t)k];
  meshmetric = new plb::MeshMetrics<T>();
  meshmetric->volume = sc[0]; meshmetric->meanLength = sc[2]; meshmetric->numVertices = sz[0]; meshmetric->numTriangles = sz[1];
  for (double a : area) meshmetric->surface += a;
  try {   // core/hemoCellField.cpp:92-98
    volume = m["MaterialModel"]["Volume"].read<T>();
    volumeFractionOfLspPerNode = (volume / numVertex) / std::pow(Parameters::dx * 1e6, 3);
  } catch (std::invalid_argument &) { hlog << "(HemoCell) (WARNING) (AddCellType) Volume of celltype " << name << " not present, volume set to zero" << endl; }
}

// mechanics/commonCellConstants.cpp:70-409: the tables are built by the library (csrc/mesh.cpp) together with the mesh;
// this copies them into the reference's container for models that read them
inline CommonCellConstants CommonCellConstants::CommonCellConstantsConstructor(HemoCellField &field, Config &) {
  if (!field.dev) field.create_device_type(field.device_model >= 0 ? field.device_model : HC_MODEL_RBC_HO);   // a host model still gets the mesh and its constants
  CommonCellConstants c; c.cellFieldPtr = &field;
  int sz[4]; hcp_celltype_sizes(field.dev, sz);
  const size_t nv = (size_t)sz[0], nt = (size_t)sz[1], ne = (size_t)sz[2], nie = (size_t)sz[3];
  vector<long> tri(3 * nt), edge(2 * ne), ring(6 * nv), ebt(2 * ne), ebo(2 * ne), ie(2 * nie);
  vector<int> nring(nv);
  c.edge_length_eq_list.resize(ne); c.edge_angle_eq_list.resize(ne); c.triangle_area_eq_list.resize(nt); c.surface_patch_center_dist_eq_list.resize(nv);
  c.inner_edge_length_eq_list.resize(nie);
  double sc[9];
  hc_check(hcp_celltype_tables(field.dev, nullptr, tri.data(), edge.data(), c.edge_length_eq_list.data(), c.edge_angle_eq_list.data(), c.triangle_area_eq_list.data(),
                               ring.data(), c.surface_patch_center_dist_eq_list.data(), sc), "hcp_celltype_tables");
  hc_check(hcp_celltype_tables2(field.dev, ebt.data(), ebo.data(), ie.data(), c.inner_edge_length_eq_list.data(), nring.data()), "hcp_celltype_tables2");
  c.triangle_list.resize(nt); for (size_t t = 0; t < nt; t++) for (size_t k = 0; k < 3; k++) c.triangle_list[t][k] = tri[3 * t + k];
  c.edge_list.resize(ne); c.edge_bending_triangles_list.resize(ne); c.edge_bending_triangles_outer_points.resize(ne);
  for (size_t e = 0; e < ne; e++) for (size_t k = 0; k < 2; k++) { c.edge_list[e][k] = edge[2 * e + k]; c.edge_bending_triangles_list[e][k] = ebt[2 * e + k]; c.edge_bending_triangles_outer_points[e][k] = ebo[2 * e + k]; }
  c.inner_edge_list.resize(nie); for (size_t e = 0; e < nie; e++) for (size_t k = 0; k < 2; k++) c.inner_edge_list[e][k] = ie[2 * e + k];
  c.vertex_vertexes.resize(nv); c.vertex_n_vertexes.resize(nv);
  for (size_t i = 0; i < nv; i++) { for (size_t k = 0; k < 6; k++) c.vertex_vertexes[i][k] = ring[6 * i + k]; c.vertex_n_vertexes[i] = (unsigned int)nring[i]; }
  // vertex_edges / vertex_edges_sign (mechanics/commonCellConstants.cpp:283-310): the edge towards each ring neighbour and whether this vertex is its first end
  c.vertex_edges.assign(nv, hemo::Array<plint, 6>()); c.vertex_edges_sign.assign(nv, hemo::Array<signed int, 6>());
  for (size_t i = 0; i < nv; i++) for (size_t k = 0; k < 6; k++) {
    c.vertex_edges[i][k] = -1; c.vertex_edges_sign[i][k] = 0;
    const plint nb = c.vertex_vertexes[i][k];
    if (nb < 0) continue;
    for (size_t e = 0; e < ne; e++) {
      if (c.edge_list[e][0] == (plint)i && c.edge_list[e][1] == nb) { c.vertex_edges[i][k] = (plint)e; c.vertex_edges_sign[i][k] = 1; break; }
      if (c.edge_list[e][1] == (plint)i && c.edge_list[e][0] == nb) { c.vertex_edges[i][k] = (plint)e; c.vertex_edges_sign[i][k] = -1; break; }
    }
  }
  c.volume_eq = sc[0]; c.area_mean_eq = sc[1]; c.edge_mean_eq = sc[2]; c.angle_mean_eq = sc[3];
  return c;
}

inline hc_cells *HemoCellFields::device() {
  hemocell.flush();   // queued iterations run before anybody looks at or edits the cells
  if (!dev) { hc_check(hcp_create(&dev, hemocell.lattice->device(), &Parameters::raw()), "hcp_create"); hemocell.lattice->cells_bound = true; }
  if (!types_bound) {
    for (auto *f : cellFields) {
      if (f->kernelMethod != interpolationCoefficientsPhi2) {   // core/hemoCellField.h:67: a user IBM kernel is host code
        hlog << "(HemoCell) (GPU backend) cell type " << f->name << " installs its own IBM kernelMethod; only interpolationCoefficientsPhi2 (core/immersedBoundaryMethod.h:62-138) runs on the device. Exiting." << endl;
        std::exit(1);
      }
      hc_check(hcp_add_type(dev, f->dev, (int)f->timescale, nullptr), "hcp_add_type");
    }
    types_bound = true;
  }
  return dev;
}

// ---- HemoCellParticleField: host view of this rank's particles (core/hemoCellParticleField.h:39-207)
inline void HemoCellParticleField::refresh() {
  if (fresh_iter == (long)fields.hemocell.iter && fresh_iter >= 0) return;
  hc_cells *c = fields.device();
  long nvt = 0, miss = 0; hcp_counts(c, &nvt, nullptr, nullptr); hcp_deletion_counts(c, nullptr, nullptr, nullptr, &miss);
  vector<HemoCellParticle::serializeValues_t> rec((size_t)(nvt - miss));
  if (!rec.empty()) hc_check(hcp_download_records(c, rec.data(), (long)rec.size()), "hcp_download_records");
  particles.clear(); particles.reserve(rec.size());
  for (auto &r : rec) particles.emplace_back(r);
  auto *L = fields.hemocell.lattice;
  localDomain = plb::Box3D(L->x0, L->x0 + L->nxl - 1, 0, L->ny - 1, 0, L->nz - 1); boundingBox = localDomain;
  // update_ppc / update_lpc / update_ppt (core/hemoCellParticleField.cpp:395-467)
  _particles_per_cell.clear(); _lpc.clear(); _particles_per_type.assign(fields.size(), vector<unsigned int>());
  for (unsigned int i = 0; i < particles.size(); i++) {
    const auto &sv = particles[i].sv;
    auto &v = _particles_per_cell[(int)sv.cellId];
    if (v.empty()) v.assign((size_t)fields[sv.celltype]->numVertex, -1);
    v[sv.vertexId] = (int)i;
    if (isContainedABS(sv.position, localDomain) || global.world == 1) _lpc[(int)sv.cellId] = true;
    _particles_per_type[sv.celltype].push_back(i);
  }
  fresh_iter = (long)fields.hemocell.iter;
}
// hand edited records back to the device (complete cells only, as the reference requires before mechanics)
inline void HemoCellParticleField::upload() {
  vector<HemoCellParticle::serializeValues_t> rec; rec.reserve(particles.size());
  for (auto &p : particles) rec.push_back(p.sv);
  hc_check(hcp_upload_records(fields.device(), rec.data(), (long)rec.size()), "hcp_upload_records");
  fresh_iter = -1;
}
inline int HemoCellParticleField::deleteIncompleteCells(bool verbose) {
  long n = 0; hc_check(hcp_delete_incomplete_cells(fields.device(), &n), "hcp_delete_incomplete_cells");
  if (verbose && n) hlog << "(HemoCell) (Delete Cells) " << n << " incomplete cell(s) removed: a particle of theirs had reached a wall" << endl;   // issueWarning, :497-503
  fresh_iter = -1;
  return (int)n;
}
inline void HemoCellFields::spreadParticleForce() { hc_check(hcp_spread(device(), 1), "hcp_spread"); }
inline void HemoCellFields::interpolateFluidVelocity() { hc_check(hcp_interpolate(device()), "hcp_interpolate"); particleField.invalidate(); }
inline void HemoCellFields::advanceParticles() { hc_check(hcp_advance(device(), 0), "hcp_advance"); particleField.invalidate(); }
inline void HemoCellFields::applyConstitutiveModel(bool forced) { hc_check(hcp_mechanics(device(), (long)hemocell.iter, forced ? 1 : 0), "hcp_mechanics"); particleField.invalidate(); }
inline void HemoCellFields::deleteIncompleteCells(bool verbose) { particleField.deleteIncompleteCells(verbose); }
inline void HemoCellFields::applyRepulsionForce() {
  HemoCell &h = hemocell; hc_cells *c = device();
  if (h.repulsionEnabled && !h.repulsionPushed) { hc_check(hcp_set_repulsion(c, h.repulsionConstant_, h.repulsionCutoff_, (int)h.repulsionTimescale), "hcp_set_repulsion"); h.repulsionPushed = true; }
  hc_check(hcp_repulsion(c), "hcp_repulsion"); particleField.invalidate();
}
inline void HemoCellFields::applyBoundaryRepulsionForce() {
  HemoCell &h = hemocell; hc_cells *c = device();
  if (h.boundaryRepulsionEnabled && !h.boundaryRepulsionPushed) { hc_check(hcp_set_boundary_repulsion(c, h.boundaryRepulsionConstant_, h.boundaryRepulsionCutoff_, (int)h.boundaryRepulsionTimescale), "hcp_set_boundary_repulsion"); h.boundaryRepulsionPushed = true; }
  hc_check(hcp_boundary_repulsion(c), "hcp_boundary_repulsion"); particleField.invalidate();
}
inline void HemoCellFields::refuse(const char *what) { hemocell.refuse(what); }
inline plb::MultiParticleField3D<HemoCellParticleField> *HemoCellField::getParticleField3D() { return cellFields->immersedParticles; }
inline plb::MultiParticleField3D<HemoCellParticleField> *HemoCellField::getParticleArg() { return cellFields->immersedParticles; }
inline plb::MultiBlockLattice3D<T, DESCRIPTOR> *HemoCellField::getFluidField3D() { return cellFields->hemocell.lattice; }

// io/readPositionsBloodCells.cpp:205-361: "<name>.pos": N, then x y z (um) rx ry rz (deg) per cell
inline void HemoCell::loadParticles() {
  loadParticlesIsCalled = true;
  hc_cells *c = cellfields->device();
  int total = 0, cellid = 0;
  for (unsigned int j = 0; j < cellfields->size(); j++) { std::ifstream f(((*cellfields)[j]->name + ".pos").c_str()); int n = 0; if (f.is_open()) f >> n; total += n; }
  cellfields->number_of_cells = total;
  vector<int> placed_per_type;
  const T posRatio = 1e-6 / Parameters::dx;
  if (global.world > 1 && cellfields->size() > 0) {
    // <domain><particleEnvelope> (core/hemoCell.cpp:139): how far from a slab face cells are replicated on the neighbour
    double env = 25.0, share = 0.0;
    try { env = (double)(*cfg)["domain"]["particleEnvelope"].read<int>(); } catch (const std::invalid_argument &) {}
    hc_check(hcp_set_envelope(c, env, &share), "hcp_set_envelope");
    // not in the logfile: the reference's CI compares the logfiles of runs with different rank counts (scripts/ci/pipeflow_sanity.sh:23-33)
    if (global.rank == 0) std::cout << "(HemoCell) (CellFields) whole cells are replicated within " << share << " [lu] of a slab face" << std::endl;
  }
  for (unsigned int j = 0; j < cellfields->size(); j++) {
    HemoCellField *field = (*cellfields)[j];
    std::ifstream f((field->name + ".pos").c_str());
    if (!f.is_open()) { if (global.rank == 0) std::cout << "*** WARNING! particle positions input file " << field->name << ".pos does not exist!" << std::endl; placed_per_type.push_back(0); continue; }
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
    placed_per_type.push_back(placed_n);
  }
  if (global.world > 1) {   // every rank offered every cell; agree on the rejected ones and count the distinct cells
    vector<long> g(cellfields->size() + 1, 0);
    hc_check(hcp_slab_sync_placement(c, g.data()), "hcp_slab_sync_placement");
    for (size_t t = 0; t < placed_per_type.size(); t++) placed_per_type[t] = (int)g[t];
  }
  for (unsigned int j = 0; j < cellfields->size() && j < placed_per_type.size(); j++)
    hlog << "(readPositionsBloodCells) " << placed_per_type[j] << " complete " << (*cellfields)[j]->name << " cells placed." << endl;
  long it = iter;   // forces of the initial configuration, as the first applyConstitutiveModel would give them
  hc_check(hcp_mechanics(c, it, 1), "hcp_mechanics");
}

// ------------------------------------------------------------------ checkpoint / resume (core/hemoCellFields.cpp:240-319)
// The reference writes Palabos' own parallelIO dumps plus checkpoint.xml (a copy of the config under a <Checkpoint>
// root with the iteration); the binary format here is this back end's own: populations in the reference node
// order, then per type the cell ids and the vertex position / velocity / force arrays.
inline string checkpoint_file(const string &dir) { return dir + (global.world > 1 ? "/checkpoint." + std::to_string(global.rank) + ".bin" : string("/checkpoint.bin")); }

inline void HemoCell::saveCheckPoint() {
  string dir = global.checkpointDirectory; while (dir.size() > 1 && dir[dir.size() - 1] == '/') dir.pop_back();
  if (global.rank == 0) {
    mkpath(dir);
    rename((dir + "/checkpoint.xml").c_str(), (dir + "/checkpoint.xml.old").c_str());
  }
  hc_comm_barrier();
  hc_lattice *d = lattice->device(); hc_cells *c = cellfields->device();
  const size_t n = (size_t)lattice->nxl * lattice->ny * lattice->nz;
  vector<double> f(n * HC_Q);
  hc_check(hcl_download_populations(d, f.data()), "hcl_download_populations");
  // written next to its final name and renamed into place when complete: a run that dies while saving leaves the previous
  // checkpoint (core/hemoCellFields.cpp:283-290 keeps it as .old) and never a truncated one under the name that is loaded
  const string final_name = checkpoint_file(dir), tmp_name = final_name + ".tmp";
  std::ofstream o(tmp_name.c_str(), std::ios::binary);
  const long hdr[8] = {0x48434b51, (long)iter, lattice->nx, lattice->ny, lattice->nz, (long)cellfields->size(), lattice->x0, lattice->nxl};
  o.write((const char *)hdr, sizeof(hdr));
  o.write((const char *)f.data(), (std::streamsize)(f.size() * sizeof(double)));
  // the particles in the reference's own record (core/hemoCellParticle.h:45-63): position, velocity, force AND
  // force_repulsion, so that a resume with repulsionTimescale > 1 continues bit for bit
  long nvt = 0, miss = 0; hcp_counts(c, &nvt, nullptr, nullptr); hcp_deletion_counts(c, nullptr, nullptr, nullptr, &miss);
  long nrec = nvt - miss;
  vector<HemoCellParticle::serializeValues_t> rec((size_t)nrec);
  if (nrec) hc_check(hcp_download_records(c, rec.data(), nrec), "hcp_download_records");
  o.write((const char *)&nrec, sizeof(long));
  o.write((const char *)rec.data(), (std::streamsize)(rec.size() * sizeof(rec[0])));
  o.close();
  if (!o) { hlog << "(HemoCell) (saveCheckPoint) could not write " << tmp_name << endl; std::exit(1); }
  rename(final_name.c_str(), (final_name + ".old").c_str());
  rename(tmp_name.c_str(), final_name.c_str());
  if (global.rank == 0) {
    std::ofstream x((dir + "/checkpoint.xml").c_str());
    x << "<?xml version=\"1.0\" ?>\n<Checkpoint>\n<General><Iteration>" << iter << "</Iteration><OutDirectory>" << outDir << "/</OutDirectory></General>\n";
    // the <hemocell> element of the configuration this run was started with (read at start-up): only the element, not the
    // <Checkpoint> wrapper a resumed run's configuration has around it
    x << configElement;
    x << "</Checkpoint>\n";
  }
  hlog << "(HemoCell) (saveCheckPoint) saved iteration " << iter << " to " << dir << endl;
}

inline void HemoCell::loadCheckPoint() {
  if (cfg->checkpointed) {   // core/hemoCellFields.cpp:244-250: continue in the directory the checkpoint names (the constructor has
                             // opened a fresh one, as the reference's does)
    string d = cfg->checkpointGeneral("OutDirectory");
    while (d.size() > 1 && d[d.size() - 1] == '/') d.pop_back();
    if (!d.empty()) { outDir = d; loadDirectories(false); }
  } else hlog << "(HemoCell) (CellFields) loading checkpoint from non-checkpoint Config" << endl;
  string dir = global.checkpointDirectory; while (dir.size() > 1 && dir[dir.size() - 1] == '/') dir.pop_back();
  std::ifstream in(checkpoint_file(dir).c_str(), std::ios::binary);
  if (!in.is_open()) { hlog << "(HemoCell) (loadCheckPoint) " << checkpoint_file(dir) << " not found" << endl; std::exit(1); }
  hc_lattice *d = lattice->device(); hc_cells *c = cellfields->device();
  in.seekg(0, std::ios::end); const long file_bytes = (long)in.tellg(); in.seekg(0, std::ios::beg);
  auto truncated = [&]() { hlog << "(HemoCell) (loadCheckPoint) " << checkpoint_file(dir) << " is truncated or damaged" << endl; std::exit(1); };
  long hdr[8] = {0}; in.read((char *)hdr, sizeof(hdr));
  if (!in.good()) truncated();
  if (hdr[0] != 0x48434b51 || hdr[2] != lattice->nx || hdr[3] != lattice->ny || hdr[4] != lattice->nz || hdr[5] != (long)cellfields->size() || hdr[6] != lattice->x0 || hdr[7] != lattice->nxl) {
    hlog << "(HemoCell) (loadCheckPoint) checkpoint does not match this case (lattice size / cell types / number of ranks)" << endl; std::exit(1);
  }
  const size_t n = (size_t)lattice->nxl * lattice->ny * lattice->nz;
  vector<double> f(n * HC_Q);
  in.read((char *)f.data(), (std::streamsize)(f.size() * sizeof(double)));
  if (!in.good()) truncated();
  long nrec = -1; in.read((char *)&nrec, sizeof(long));
  const long rec_bytes = (long)sizeof(HemoCellParticle::serializeValues_t);
  if (!in.good() || nrec < 0 || nrec > (file_bytes - (long)in.tellg()) / rec_bytes) truncated();
  vector<HemoCellParticle::serializeValues_t> rec((size_t)nrec);
  in.read((char *)rec.data(), (std::streamsize)(rec.size() * sizeof(rec[0])));
  if (nrec && !in.good()) truncated();
  if (global.world > 1) {   // every rank writes its own file: all of them must hold the same iteration
    double lo = (double)hdr[1], hi = (double)hdr[1];
    hc_check(hc_comm_allreduce(&lo, 1, 1), "hc_comm_allreduce"); hc_check(hc_comm_allreduce(&hi, 1, 2), "hc_comm_allreduce");
    if (lo != hi) { hlog << "(HemoCell) (loadCheckPoint) the ranks hold checkpoints of different iterations (" << (long)lo << " ... " << (long)hi << ")" << endl; std::exit(1); }
  }
  hc_check(hcl_upload_populations(d, f.data()), "hcl_upload_populations");
  // a repulsion the driver enabled has to exist before the records arrive, or their force_repulsion would be dropped
  if (boundaryRepulsionEnabled && !boundaryRepulsionPushed) { hc_check(hcp_set_boundary_repulsion(c, boundaryRepulsionConstant_, boundaryRepulsionCutoff_, (int)boundaryRepulsionTimescale), "hcp_set_boundary_repulsion"); boundaryRepulsionPushed = true; }
  if (repulsionEnabled && !repulsionPushed) { hc_check(hcp_set_repulsion(c, repulsionConstant_, repulsionCutoff_, (int)repulsionTimescale), "hcp_set_repulsion"); repulsionPushed = true; }
  hc_check(hcp_upload_records(c, rec.data(), nrec), "hcp_upload_records");
  // core/hemoCellFields.cpp:272-274: load, syncEnvelopes, deleteIncompleteCells -- a checkpoint written while a cell had lost
  // particles at a wall (removeParticles(1)) holds that cell incomplete
  hc_check(hcp_delete_incomplete_cells(c, nullptr), "hcp_delete_incomplete_cells");
  long nct = 0; hcp_counts(c, nullptr, &nct, nullptr);
  cellfields->number_of_cells = (int)nct;
  iter = (unsigned int)hdr[1];
  lattice->mark_stepped();
  hlog << "(HemoCell) (loadCheckPoint) resumed at iteration " << iter << endl;
}

// ------------------------------------------------------------------ helper/cellInfo.h
struct CellInformation {
  hemo::Array<T, 3> position; T volume = 0, area = 0, stretch = 0; hemo::Array<T, 6> bbox; pluint cellType = 0; plint base_cell_id = 0; bool centerLocal = true; plint blockId = 0;
  hemo::Array<T, 3> velocity;   // mean of the vertices' sv.v (helper/cellInfo.cpp:216-217, :252)
};
struct CellInformationFunctionals {
  static map<int, CellInformation> &info() { static map<int, CellInformation> m; return m; }
  // helper/cellInfo.cpp:97: a cell is counted / reported by the rank whose domain holds its centre
  static bool centre_local(HemoCell *h, const double *centroid) {
    if (global.world == 1) return true;
    const auto *L = h->lattice;
    double cx = centroid[0];
    if (L->per.p[0]) cx -= (double)L->nx * std::floor((cx + 0.5) / (double)L->nx);   // copies across the seam are unwrapped images
    const double g = std::floor(cx + 0.5);
    return g >= (double)L->x0 && g < (double)(L->x0 + L->nxl);
  }
  static void fill(HemoCell *h, bool vol, bool area, bool pos, bool bbox, bool stretch, bool vel = false) {
    hc_cells *c = h->cellfields->device();
    long nvt = 0, nct = 0; hcp_counts(c, &nvt, &nct, nullptr);
    vector<long> ids((size_t)nct); if (nct) hcp_download_cell_ids(c, ids.data());
    vector<double> allpos; if (stretch) { allpos.resize(3 * (size_t)nvt); if (nvt) hcp_download(c, 0, allpos.data()); }
    vector<double> allvel; if (vel) { allvel.resize(3 * (size_t)nvt); if (nvt) hcp_download(c, 1, allvel.data()); }
    long first_cell = 0;
    for (unsigned int t = 0; t < h->cellfields->size(); t++) {
      long fv = 0, nc = 0; hcp_type_range(c, (int)t, &fv, &nc);
      if (nc == 0) continue;
      vector<double> V((size_t)nc), A((size_t)nc), B(6 * (size_t)nc), P(3 * (size_t)nc);
      hc_check(hcp_cell_info(c, (int)t, V.data(), A.data(), B.data(), P.data()), "hcp_cell_info");
      const int nv = (*h->cellfields)[t]->numVertex;
      for (long k = 0; k < nc; k++) {
        if (!centre_local(h, &P[3 * (size_t)k])) continue;   // the other holder reports it
        CellInformation &ci = info()[(int)ids[(size_t)(first_cell + k)]];
        ci.cellType = t; ci.base_cell_id = ids[(size_t)(first_cell + k)]; ci.blockId = global.rank;
        if (vol) ci.volume = V[(size_t)k];
        if (area) ci.area = A[(size_t)k];
        if (pos) for (int d = 0; d < 3; d++) ci.position[d] = P[3 * (size_t)k + d];
        if (bbox) for (int d = 0; d < 6; d++) ci.bbox[d] = B[6 * (size_t)k + d];
        if (vel) {
          const double *vv = allvel.data() + 3 * (size_t)(fv + k * nv);
          for (int d = 0; d < 3; d++) { T sum = 0; for (int i = 0; i < nv; i++) sum += vv[3 * i + d]; ci.velocity[d] = sum / T(nv); }
        }
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
  static void calculateCellVelocity(HemoCell *h) { fill(h, false, false, false, false, false, true); }
  static void calculateCellAtomicBlock(HemoCell *h) { fill(h, false, false, false, false, false); }   // blockId and cellType come with every entry
  static void calculateCellType(HemoCell *h) { fill(h, false, false, false, false, false); }
  static void calculate_vol_pos_area(HemoCell *h) { fill(h, true, true, true, false, false); }        // helper/cellInfo.h:107: "excludes Stretch"
  static void calculateCellInformation(HemoCell *h) { fill(h, true, true, true, true, false, true); }
  static void calculateCellInformation(HemoCell *h, map<int, CellInformation> &out) { clear_list(); calculateCellInformation(h); out = info(); clear_list(); }   // :121
  static void clear_list() { info().clear(); }
  // helper/cellInfo.cpp:324-365: centre-local cells of every rank, summed (HemoCellGatheringFunctional)
  static vector<double> counts_per_type(HemoCell *h) {
    hc_cells *c = h->cellfields->device();
    vector<double> n(h->cellfields->size(), 0.0);
    for (unsigned int t = 0; t < h->cellfields->size(); t++) {
      long fv = 0, nc = 0; hcp_type_range(c, (int)t, &fv, &nc);
      if (global.world == 1 || nc == 0) { n[t] = (double)nc; continue; }
      vector<double> V((size_t)nc), A((size_t)nc), B(6 * (size_t)nc), P(3 * (size_t)nc);
      hc_check(hcp_cell_info(c, (int)t, V.data(), A.data(), B.data(), P.data()), "hcp_cell_info");
      for (long k = 0; k < nc; k++) n[t] += centre_local(h, &P[3 * (size_t)k]) ? 1.0 : 0.0;
    }
    if (global.world > 1 && !n.empty()) hc_check(hc_comm_allreduce(n.data(), (int)n.size(), 0), "hc_comm_allreduce");
    return n;
  }
  static pluint getTotalNumberOfCells(HemoCell *h) { double s = 0; for (double v : counts_per_type(h)) s += v; return (pluint)(s + 0.5); }
  static pluint getNumberOfCellsFromType(HemoCell *h, string type) { return (pluint)(counts_per_type(h)[(*h->cellfields)[type]->ctype] + 0.5); }
};
#define info_per_cell info()

// ------------------------------------------------------------------ helper/fluidInfo.h, helper/particleInfo.h
struct FluidStatistics { T min = 0, max = 0, avg = 0; pluint ncells = 0; };
// {min, max, sum}, n over all ranks (the reference gathers per-block results, helper/fluidInfo.cpp:98-118)
inline void reduce_stats(double o[3], long &n) {
  if (global.world == 1) return;
  double mn = n ? o[0] : 1e300, mx = n ? o[1] : -1e300, sm[2] = {o[2], (double)n};
  hc_check(hc_comm_allreduce(&mn, 1, 1), "hc_comm_allreduce"); hc_check(hc_comm_allreduce(&mx, 1, 2), "hc_comm_allreduce"); hc_check(hc_comm_allreduce(sm, 2, 0), "hc_comm_allreduce");
  n = (long)(sm[1] + 0.5); o[0] = n ? mn : 0.0; o[1] = n ? mx : 0.0; o[2] = sm[0];
}
struct FluidInfo {
  // helper/fluidInfo.cpp:33-118: device reductions (hcl_fluid_stats), folded deterministically
  static FluidStatistics stat(HemoCell *h, int what) {
    double o[3]; long n = 0;
    hc_check(hcl_fluid_stats(h->lattice->device(), what, o, &n), "hcl_fluid_stats");
    reduce_stats(o, n);
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
    reduce_stats(o, n);
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
  // every rank holds its centre-local cells; rank 0 gathers and writes (HemoCellGatheringFunctional, :45)
  struct Row { double v[11]; };   // x y z area volume block cellId type velocity
  vector<Row> mine;
  for (auto &kv : CellInformationFunctionals::info()) {
    const CellInformation &c = kv.second; Row r;
    const double sx = h.outputInSiUnits ? Parameters::dx : 1.0;
    r.v[0] = c.position[0] * sx; r.v[1] = c.position[1] * sx; r.v[2] = c.position[2] * sx; r.v[3] = c.area * sx * sx; r.v[4] = c.volume * sx * sx * sx;
    r.v[5] = (double)c.blockId; r.v[6] = (double)kv.first; r.v[7] = (double)c.cellType;
    const double sv = h.outputInSiUnits ? Parameters::dx / Parameters::dt : 1.0;
    for (int d = 0; d < 3; d++) r.v[8 + d] = c.velocity[d] * sv;
    mine.push_back(r);
  }
  vector<Row> all = mine;
  if (global.world > 1) {
    double cnt = (double)mine.size(), mx = cnt;
    hc_check(hc_comm_allreduce(&mx, 1, 2), "hc_comm_allreduce");
    const size_t cap = (size_t)mx;
    vector<Row> padded(cap + 1); padded[0].v[0] = cnt; std::copy(mine.begin(), mine.end(), padded.begin() + 1);
    vector<Row> gathered((cap + 1) * (size_t)global.world);
    hc_check(hc_comm_allgather(padded.data(), (cap + 1) * sizeof(Row), gathered.data()), "hc_comm_allgather");
    all.clear();
    for (int r = 0; r < global.world; r++) { const Row *b = gathered.data() + (size_t)r * (cap + 1); for (size_t k = 0; k < (size_t)b[0].v[0]; k++) all.push_back(b[1 + k]); }
    std::stable_sort(all.begin(), all.end(), [](const Row &a, const Row &b) { return a.v[6] < b.v[6]; });
  }
  if (global.rank == 0) ::mkdir((h.outDir + "/csv").c_str(), 0777);
  if (global.rank == 0)
    for (unsigned int t = 0; t < h.cellfields->size(); t++) {
      char it[32]; std::snprintf(it, sizeof(it), "%012u", h.iter);
      std::ofstream f((h.outDir + "/csv/" + (*h.cellfields)[t]->name + "." + it + ".csv").c_str());
      f << "X,Y,Z,area,volume,atomic_block,cellId,baseCellId,velocity_x,velocity_y,velocity_z" << std::endl;   // :52
      for (const Row &r : all) {
        if ((unsigned int)r.v[7] != t) continue;
        f << r.v[0] << "," << r.v[1] << "," << r.v[2] << "," << r.v[3] << "," << r.v[4] << "," << (long)r.v[5] << "," << (long)r.v[6] << "," << (long)r.v[6] << ","
          << r.v[8] << "," << r.v[9] << "," << r.v[10] << std::endl;
      }
    }
  CellInformationFunctionals::clear_list();
}

}  // namespace hemo
#include "hdf5_output.h"
namespace hemo {

// core/hemoCell.cpp:221-287: <out>/hdf5/<iter>/ with one file per cell type plus the fluid file, and the CSV summary
inline void HemoCell::writeOutput() {
  lastOutputAt = iter;
  cellfields->deleteIncompleteCells(global.cellsDeletedInfo);   // core/hemoCell.cpp:248-252: the writers expect whole cells
#ifdef HEMOCELL_WITH_HDF5
  const string dir = outDir + "/hdf5/" + zeroPadNumber(iter);
  if (global.rank == 0) { mkdir((outDir + "/hdf5").c_str(), 0755); mkdir(dir.c_str(), 0755); }
  hc_comm_barrier();
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
