// Ranks of a multi-GPU run: one process per GPU of one node.
//
// Replaces what the reference gets from MPI through Palabos (plb::plbInit, core/hemoCell.cpp:80-86; the MPI call sites
// listed in SURVEY.md section 2.3): a rank / world size, neighbour exchange of lattice faces (duplicateOverlaps,
// core/hemoCell.cpp:317) and of particle records (HemoCellParticleDataTransfer::send / receive,
// core/hemoCellParticleDataTransfer.cpp:33-180), and the few reductions of the information functionals
// (core/hemoCellFunctional.h:101-112).
//
// Control plane: a TCP mesh between the ranks (every pair connected once) for bootstrap, barrier, reductions of a few
// scalars and agreement on the placed cells -- blocking host calls, never on the stepping path.
// Data plane: RCCL point-to-point (ncclSend / ncclRecv in one group per exchange) on the stream the caller names, i.e.
// x-neighbours talk over xGMI without a host in between; librccl.so.1 is loaded on demand so that single-GPU users and
// the CPU-side tests of the mesh do not need it.  HC_TRANSPORT_TCP moves the same messages through pinned host memory and
// the mesh: RCCL refuses two ranks on one device, so this is what ranks SHARING a GPU use (rehearsals on a one-GPU box).
#include "comm.h"

#include <arpa/inet.h>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <thread>
#include <unistd.h>

namespace {

struct Rccl {   // entry points of librccl.so.1, bound on first use
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct State {
  bool inited = false;
  int rank = 0, world = 1, transport = HC_TRANSPORT_NONE;
  std::vector<int> fd;          // fd[r] = socket to rank r (-1 for myself)
  int listen_fd = -1;
  double timeout_s = 120.0;
  Rccl rccl; ncclComm_t comm = nullptr;
  // host staging of the TCP data plane: [send lo, send hi, recv lo, recv hi]
  char *stage[4] = {nullptr, nullptr, nullptr, nullptr}; size_t stage_cap[4] = {0, 0, 0, 0};
};
State g;

int fail(const std::string &msg) { hc::set_error("hc_comm: " + msg); return HC_ERR_STATE; }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void set_nonblocking(int fd) { const int fl = fcntl(fd, F_GETFL, 0); fcntl(fd, F_SETFL, fl | O_NONBLOCK); }
void set_nodelay(int fd) { int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }

// One transfer of a progress() call.  Transfers on the same socket and in the same direction complete in list order.
struct Xfer { int fd; char *ptr; size_t left; bool send; };

// Drives all transfers to completion with non-blocking sockets: sends and receives make progress together, so two ranks
// that both have megabytes for each other cannot block each other on full socket buffers.
int progress(std::vector<Xfer> &x) {
  const double t0 = now_s();
  std::vector<pollfd> pf; std::vector<size_t> who;
  while (true) {
    pf.clear(); who.clear();
    for (size_t i = 0; i < x.size(); i++) {
      if (x[i].left == 0) continue;
      bool blocked = false;
      for (size_t k = 0; k < i && !blocked; k++) blocked = x[k].left > 0 && x[k].fd == x[i].fd && x[k].send == x[i].send;
      if (blocked) continue;
      pollfd p; p.fd = x[i].fd; p.events = x[i].send ? POLLOUT : POLLIN; p.revents = 0;
      pf.push_back(p); who.push_back(i);
    }
    if (pf.empty()) return HC_OK;
    const int rc = poll(pf.data(), (nfds_t)pf.size(), 1000);
    if (rc < 0) { if (errno == EINTR) continue; return fail(std::string("poll: ") + std::strerror(errno)); }
    if (rc == 0) { if (now_s() - t0 > g.timeout_s) return fail("a neighbour did not answer within the time limit (HEMOCELL_COMM_TIMEOUT)"); continue; }
    for (size_t k = 0; k < pf.size(); k++) {
      if (!pf[k].revents) continue;
      Xfer &t = x[who[k]];
      const ssize_t n = t.send ? ::send(t.fd, t.ptr, t.left, MSG_NOSIGNAL) : ::recv(t.fd, t.ptr, t.left, 0);
      if (n > 0) { t.ptr += n; t.left -= (size_t)n; }
      else if (n == 0 && !t.send) return fail("a peer rank closed its connection (it probably stopped with an error)");
      else if (n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return fail(std::string(t.send ? "send: " : "recv: ") + std::strerror(errno));
    }
  }
}
int send_all(int fd, const void *p, size_t n) { std::vector<Xfer> x{{fd, (char *)p, n, true}}; return progress(x); }
int recv_all(int fd, void *p, size_t n) { std::vector<Xfer> x{{fd, (char *)p, n, false}}; return progress(x); }

int resolve(const char *host, int port, sockaddr_in &a) {
  std::memset(&a, 0, sizeof(a));
  a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, host, &a.sin_addr) == 1) return HC_OK;
  addrinfo hints; std::memset(&hints, 0, sizeof(hints)); hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
  addrinfo *res = nullptr;
  if (getaddrinfo(host, nullptr, &hints, &res) != 0 || !res) return fail(std::string("cannot resolve ") + host);
  a.sin_addr = ((sockaddr_in *)res->ai_addr)->sin_addr;
  freeaddrinfo(res);
  return HC_OK;
}

// every pair of ranks gets one connection: rank r listens on port + r, connects to every lower rank (retrying while the
// listener is not up yet) and accepts one connection from every higher rank
int connect_mesh(const char *addr, int port) {
  g.fd.assign((size_t)g.world, -1);
  if (g.world == 1) return HC_OK;
  g.listen_fd = socket(AF_INET, SOCK_STREAM, 0);
  if (g.listen_fd < 0) return fail(std::string("socket: ") + std::strerror(errno));
  int one = 1; setsockopt(g.listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in me; std::memset(&me, 0, sizeof(me)); me.sin_family = AF_INET; me.sin_addr.s_addr = htonl(INADDR_ANY); me.sin_port = htons((uint16_t)(port + g.rank));
  if (bind(g.listen_fd, (sockaddr *)&me, sizeof(me)) != 0) return fail("cannot bind port " + std::to_string(port + g.rank) + ": " + std::strerror(errno) + " (set HEMOCELL_PORT)");
  if (listen(g.listen_fd, g.world) != 0) return fail(std::string("listen: ") + std::strerror(errno));
  const double t0 = now_s();
  for (int peer = 0; peer < g.rank; peer++) {
    sockaddr_in a; int rc = resolve(addr, port + peer, a); if (rc != HC_OK) return rc;
    while (true) {
      const int fd = socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return fail(std::string("socket: ") + std::strerror(errno));
      if (connect(fd, (sockaddr *)&a, sizeof(a)) == 0) {
        set_nodelay(fd); set_nonblocking(fd);
        const int32_t hello = g.rank;
        g.fd[(size_t)peer] = fd;
        rc = send_all(fd, &hello, sizeof(hello)); if (rc != HC_OK) return rc;
        break;
      }
      close(fd);
      if (now_s() - t0 > g.timeout_s) return fail("rank " + std::to_string(peer) + " is not listening on " + addr + ":" + std::to_string(port + peer));
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  for (int k = g.rank + 1; k < g.world; k++) {
    pollfd p; p.fd = g.listen_fd; p.events = POLLIN; p.revents = 0;
    while (true) {
      const int rc = poll(&p, 1, 1000);
      if (rc > 0) break;
      if (rc < 0 && errno != EINTR) return fail(std::string("poll: ") + std::strerror(errno));
      if (now_s() - t0 > g.timeout_s) return fail("not every higher rank connected within the time limit");
    }
    const int fd = accept(g.listen_fd, nullptr, nullptr);
    if (fd < 0) return fail(std::string("accept: ") + std::strerror(errno));
    set_nodelay(fd); set_nonblocking(fd);
    int32_t hello = -1;
    const int rc = recv_all(fd, &hello, sizeof(hello)); if (rc != HC_OK) return rc;
    if (hello <= g.rank || hello >= g.world || g.fd[(size_t)hello] != -1) { close(fd); return fail("unexpected rank announced itself on the mesh"); }
    g.fd[(size_t)hello] = fd;
  }
  close(g.listen_fd); g.listen_fd = -1;
  return HC_OK;
}

int load_rccl() {
  if (g.rccl.handle) return HC_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);   // the copy already in the process (a host framework's) or the system one
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(std::string("cannot load librccl.so.1: ") + dlerror());
  g.rccl.handle = h;
#define BIND(field, name) \
  g.rccl.field = (decltype(g.rccl.field))dlsym(h, name); \
  if (!g.rccl.field) return fail(std::string("librccl.so.1 has no ") + name)
  BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
  BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  return HC_OK;
}

#define HC_NCCL(call)                                                                                     \
  do {                                                                                                    \
    const ncclResult_t r__ = (call);                                                                      \
    if (r__ != ncclSuccess) return fail(std::string("RCCL: ") + g.rccl.GetErrorString(r__) + " in " #call); \
  } while (0)

int ensure_stage(int k, size_t bytes) {
  if (g.stage_cap[k] >= bytes) return HC_OK;
  if (g.stage[k]) HC_HIP(hipHostFree(g.stage[k]));
  g.stage[k] = nullptr; g.stage_cap[k] = 0;
  const size_t cap = bytes + bytes / 4 + 4096;
  HC_HIP(hipHostMalloc((void **)&g.stage[k], cap, hipHostMallocDefault));
  g.stage_cap[k] = cap;
  return HC_OK;
}

// host buffers over the mesh, same routing as the data plane
int exchange_host(bool periodic, const void *s_lo, size_t n_lo, const void *s_hi, size_t n_hi, void *r_lo, size_t m_lo, void *r_hi, size_t m_hi) {
  int lo, hi; hcm::neighbours(periodic, lo, hi);
  if (lo == g.rank && hi == g.rank) {   // a periodic one-rank world is its own neighbour on both sides
    if (n_lo != m_hi || n_hi != m_lo) return fail("self exchange with unequal message sizes");
    if (n_lo) std::memcpy(r_hi, s_lo, n_lo);
    if (n_hi) std::memcpy(r_lo, s_hi, n_hi);
    return HC_OK;
  }
  // when lo and hi are the same peer (two ranks, periodic) its first message is MY low-face data, i.e. its high halo:
  // sends go out lo first, receives come in hi first
  std::vector<Xfer> x;
  if (lo >= 0 && n_lo) x.push_back({g.fd[(size_t)lo], (char *)s_lo, n_lo, true});
  if (hi >= 0 && n_hi) x.push_back({g.fd[(size_t)hi], (char *)s_hi, n_hi, true});
  if (hi >= 0 && m_hi) x.push_back({g.fd[(size_t)hi], (char *)r_hi, m_hi, false});
  if (lo >= 0 && m_lo) x.push_back({g.fd[(size_t)lo], (char *)r_lo, m_lo, false});
  return progress(x);
}

}  // namespace

namespace hcm {

bool active() { return g.inited; }
int rank() { return g.rank; }
int world() { return g.world; }
int transport() { return g.transport; }

void neighbours(bool periodic, int &lo, int &hi) {
  lo = (periodic || g.rank > 0) ? (g.rank - 1 + g.world) % g.world : -1;
  hi = (periodic || g.rank < g.world - 1) ? (g.rank + 1) % g.world : -1;
}

int exchange(hipStream_t s, bool periodic, const void *send_lo, size_t n_lo, const void *send_hi, size_t n_hi, void *recv_lo, size_t m_lo,
             void *recv_hi, size_t m_hi) {
  if (!g.inited) return fail("hc_comm_init / hc_comm_init_env has not been called");
  int lo, hi; neighbours(periodic, lo, hi);
  if (lo < 0) { n_lo = 0; m_lo = 0; }
  if (hi < 0) { n_hi = 0; m_hi = 0; }
  if (!n_lo && !n_hi && !m_lo && !m_hi) return HC_OK;
  if (g.transport == HC_TRANSPORT_RCCL) {
    // one group: the sends and receives of both neighbours progress together (order: send lo, send hi, receive hi,
    // receive lo -- with two ranks both neighbours are the same peer and RCCL matches messages of a pair in issue order)
    HC_NCCL(g.rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (r == ncclSuccess && n_lo) r = g.rccl.Send(send_lo, n_lo, ncclChar, lo, g.comm, s);
    if (r == ncclSuccess && n_hi) r = g.rccl.Send(send_hi, n_hi, ncclChar, hi, g.comm, s);
    if (r == ncclSuccess && m_hi) r = g.rccl.Recv(recv_hi, m_hi, ncclChar, hi, g.comm, s);
    if (r == ncclSuccess && m_lo) r = g.rccl.Recv(recv_lo, m_lo, ncclChar, lo, g.comm, s);
    const ncclResult_t e = g.rccl.GroupEnd();
    if (r != ncclSuccess) return fail(std::string("RCCL: ") + g.rccl.GetErrorString(r) + " in ncclSend / ncclRecv");
    if (e != ncclSuccess) return fail(std::string("RCCL: ") + g.rccl.GetErrorString(e) + " in ncclGroupEnd");
    return HC_OK;
  }
  if (g.transport != HC_TRANSPORT_TCP) return fail("no data plane was selected (transport none)");
  const size_t need[4] = {n_lo, n_hi, m_lo, m_hi};
  for (int k = 0; k < 4; k++) { const int rc = ensure_stage(k, need[k]); if (rc != HC_OK) return rc; }
  if (n_lo) HC_HIP(hipMemcpyAsync(g.stage[0], send_lo, n_lo, hipMemcpyDeviceToHost, s));
  if (n_hi) HC_HIP(hipMemcpyAsync(g.stage[1], send_hi, n_hi, hipMemcpyDeviceToHost, s));
  HC_HIP(hipStreamSynchronize(s));   // also: the uploads of the previous exchange have left the receive blocks
  const int rc = exchange_host(periodic, g.stage[0], n_lo, g.stage[1], n_hi, g.stage[2], m_lo, g.stage[3], m_hi);
  if (rc != HC_OK) return rc;
  if (m_lo) HC_HIP(hipMemcpyAsync(recv_lo, g.stage[2], m_lo, hipMemcpyHostToDevice, s));
  if (m_hi) HC_HIP(hipMemcpyAsync(recv_hi, g.stage[3], m_hi, hipMemcpyHostToDevice, s));
  return HC_OK;
}

int barrier() {
  if (!g.inited || g.world == 1) return HC_OK;
  char c = 0;
  if (g.rank == 0) {
    for (int r = 1; r < g.world; r++) { const int rc = recv_all(g.fd[(size_t)r], &c, 1); if (rc != HC_OK) return rc; }
    for (int r = 1; r < g.world; r++) { const int rc = send_all(g.fd[(size_t)r], &c, 1); if (rc != HC_OK) return rc; }
    return HC_OK;
  }
  int rc = send_all(g.fd[0], &c, 1); if (rc != HC_OK) return rc;
  return recv_all(g.fd[0], &c, 1);
}

int allreduce(double *v, int n, int op) {
  if (!g.inited || g.world == 1 || n == 0) return HC_OK;
  const size_t bytes = (size_t)n * sizeof(double);
  if (g.rank == 0) {
    std::vector<double> in((size_t)n);
    for (int r = 1; r < g.world; r++) {   // rank order: the same sum whatever arrives first
      const int rc = recv_all(g.fd[(size_t)r], in.data(), bytes); if (rc != HC_OK) return rc;
      for (int i = 0; i < n; i++) v[i] = op == 0 ? v[i] + in[(size_t)i] : op == 1 ? (in[(size_t)i] < v[i] ? in[(size_t)i] : v[i]) : (in[(size_t)i] > v[i] ? in[(size_t)i] : v[i]);
    }
    for (int r = 1; r < g.world; r++) { const int rc = send_all(g.fd[(size_t)r], v, bytes); if (rc != HC_OK) return rc; }
    return HC_OK;
  }
  int rc = send_all(g.fd[0], v, bytes); if (rc != HC_OK) return rc;
  return recv_all(g.fd[0], v, bytes);
}

int bcast(void *buf, size_t bytes, int root) {
  if (!g.inited || g.world == 1 || bytes == 0) return HC_OK;
  if (g.rank == root) {
    for (int r = 0; r < g.world; r++) if (r != root) { const int rc = send_all(g.fd[(size_t)r], buf, bytes); if (rc != HC_OK) return rc; }
    return HC_OK;
  }
  return recv_all(g.fd[(size_t)root], buf, bytes);
}

int allgatherv(const void *mine, size_t bytes, std::vector<std::vector<char>> &all) {
  all.assign((size_t)g.world, std::vector<char>());
  all[(size_t)g.rank].assign((const char *)mine, (const char *)mine + bytes);
  if (!g.inited || g.world == 1) return HC_OK;
  std::vector<unsigned long long> sizes((size_t)g.world, 0ULL);
  if (g.rank == 0) {
    sizes[0] = bytes;
    for (int r = 1; r < g.world; r++) { const int rc = recv_all(g.fd[(size_t)r], &sizes[(size_t)r], sizeof(unsigned long long)); if (rc != HC_OK) return rc; }
  } else { unsigned long long b = bytes; const int rc = send_all(g.fd[0], &b, sizeof(b)); if (rc != HC_OK) return rc; }
  int rc = bcast(sizes.data(), sizes.size() * sizeof(unsigned long long), 0); if (rc != HC_OK) return rc;
  for (int r = 0; r < g.world; r++) {
    all[(size_t)r].resize((size_t)sizes[(size_t)r]);
    if (r != 0 && g.rank == r) { rc = send_all(g.fd[0], all[(size_t)r].data(), all[(size_t)r].size()); if (rc != HC_OK) return rc; }
    if (r != 0 && g.rank == 0) { rc = recv_all(g.fd[(size_t)r], all[(size_t)r].data(), all[(size_t)r].size()); if (rc != HC_OK) return rc; }
  }
  for (int r = 0; r < g.world; r++) { rc = bcast(all[(size_t)r].data(), all[(size_t)r].size(), 0); if (rc != HC_OK) return rc; }
  return HC_OK;
}

}  // namespace hcm

extern "C" {

int hc_comm_init(int rank, int world, int local_rank, const char *master_addr, int port, int transport, int init_device) {
  HC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hc_comm_init: rank must be in [0, world)");
  HC_REQUIRE(transport == HC_TRANSPORT_NONE || transport == HC_TRANSPORT_RCCL || transport == HC_TRANSPORT_TCP || transport == HC_TRANSPORT_AUTO,
             "hc_comm_init: unknown transport");
  HC_REQUIRE(port > 0 && port + world < 65536, "hc_comm_init: port out of range");
  if (g.inited) return fail("already initialised (one world per process, core/hemoCell.cpp:75-79)");
  if (const char *t = std::getenv("HEMOCELL_COMM_TIMEOUT")) { const double v = std::atof(t); if (v > 0) g.timeout_s = v; }
  if (init_device) {
    int ndev = 0;
    int rc = hc_device_count(&ndev); if (rc != HC_OK) return rc;
    if (ndev < 1) return fail("no HIP device");
    rc = hc_init(local_rank % ndev); if (rc != HC_OK) return rc;
  }
  const bool automatic = transport == HC_TRANSPORT_AUTO;
  if (automatic) transport = HC_TRANSPORT_RCCL;
  g.rank = rank; g.world = world; g.transport = transport;
  int rc = connect_mesh(master_addr ? master_addr : "127.0.0.1", port);
  if (rc != HC_OK) return rc;
  g.inited = true;   // the control plane is up: broadcasts and reductions below go over the mesh
  if (transport == HC_TRANSPORT_RCCL) {
    // every rank learns whether EVERY rank got its communicator and passed the self-test; they all take the same branch
    std::string why;
    double ok = 1.0;
    if (load_rccl() != HC_OK) { ok = 0.0; why = hc_last_error(); }
    ncclUniqueId id; std::memset(&id, 0, sizeof(id));
    if (ok > 0 && rank == 0) { const ncclResult_t r = g.rccl.GetUniqueId(&id); if (r != ncclSuccess) { ok = 0.0; why = std::string("ncclGetUniqueId: ") + g.rccl.GetErrorString(r); } }
    rc = hcm::allreduce(&ok, 1, 1); if (rc != HC_OK) { g.inited = false; return rc; }
    if (ok > 0) {
      rc = hcm::bcast(&id, sizeof(id), 0); if (rc != HC_OK) { g.inited = false; return rc; }
      const ncclResult_t r = g.rccl.CommInitRank(&g.comm, world, id, rank);
      if (r != ncclSuccess) { ok = 0.0; g.comm = nullptr; why = std::string("ncclCommInitRank: ") + g.rccl.GetErrorString(r) + " (two ranks on one GPU?)"; }
      rc = hcm::allreduce(&ok, 1, 1); if (rc != HC_OK) { g.inited = false; return rc; }
    }
    if (ok > 0) {   // self-test: one small ring exchange must complete, or the run would hang in its first step instead of failing here
      char *d = nullptr; hipEvent_t ev = nullptr;
      hipError_t e = hipMalloc((void **)&d, 4 * 4096);
      if (e == hipSuccess) e = hipMemset(d, 1, 4 * 4096);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) { ok = 0.0; why = std::string("self-test set-up: ") + hipGetErrorString(e); }
      if (ok > 0 && hcm::exchange(hc::comm_stream(), true, d, 4096, d + 4096, 4096, d + 2 * 4096, 4096, d + 3 * 4096, 4096) != HC_OK) { ok = 0.0; why = hc_last_error(); }
      if (ok > 0) {
        hipEventRecord(ev, hc::comm_stream());
        const double t0 = now_s();
        while (hipEventQuery(ev) == hipErrorNotReady) {
          if (now_s() - t0 > 60.0) { ok = 0.0; why = "the point-to-point self-test did not complete within 60 s"; break; }
          std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
      }
      if (ok > 0) { if (ev) hipEventDestroy(ev); if (d) hipFree(d); }   // after a hung transfer the stream still owns them
      rc = hcm::allreduce(&ok, 1, 1); if (rc != HC_OK) { g.inited = false; return rc; }
    }
    if (ok <= 0) {
      if (g.comm) { g.rccl.CommDestroy(g.comm); g.comm = nullptr; }
      if (why.empty()) why = "another rank failed to set RCCL up";
      if (!automatic) { g.inited = false; return fail("RCCL data plane: " + why + " (HEMOCELL_TRANSPORT=tcp stages the messages through the host instead)"); }
      // automatic choice: say so loudly and stage the messages through the host -- a slower run, not a different result
      if (rank == 0) std::fprintf(stderr, "(hemocell_amd) RCCL point-to-point is not usable here (%s): the neighbour exchange is staged through pinned host memory and TCP instead\n", why.c_str());
      g.transport = HC_TRANSPORT_TCP;
    }
  }
  return hcm::barrier();
}

int hc_comm_init_env(void) {
  auto env_int = [](std::initializer_list<const char *> names, int dflt) {
    for (const char *n : names) if (const char *v = std::getenv(n)) if (*v) return std::atoi(v);
    return dflt;
  };
  const int world = env_int({"HEMOCELL_WORLD_SIZE", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE"}, 1);
  if (world <= 1) return HC_OK;
  const int rank = env_int({"HEMOCELL_RANK", "RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK"}, 0);
  const int local = env_int({"HEMOCELL_LOCAL_RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID"}, rank);
  const char *addr = std::getenv("MASTER_ADDR");
  const int port = env_int({"HEMOCELL_PORT"}, env_int({"MASTER_PORT"}, 29400) + 1017);
  int transport = HC_TRANSPORT_AUTO;   // RCCL, or -- said loudly -- host staging when RCCL cannot be set up (ranks sharing a GPU)
  if (const char *t = std::getenv("HEMOCELL_TRANSPORT")) {
    if (std::strcmp(t, "tcp") == 0) transport = HC_TRANSPORT_TCP;
    else if (std::strcmp(t, "rccl") == 0) transport = HC_TRANSPORT_RCCL;
    else if (*t) { hc::set_error(std::string("hc_comm_init_env: HEMOCELL_TRANSPORT must be rccl or tcp, got ") + t); return HC_ERR_ARG; }
  }
  return hc_comm_init(rank, world, local, (addr && *addr) ? addr : "127.0.0.1", port, transport, 1);
}

int hc_comm_finalize(void) {
  if (!g.inited) return HC_OK;
  hcm::barrier();
  if (g.comm) { g.rccl.CommDestroy(g.comm); g.comm = nullptr; }
  for (int &fd : g.fd) if (fd >= 0) { close(fd); fd = -1; }
  for (int k = 0; k < 4; k++) { if (g.stage[k]) hipHostFree(g.stage[k]); g.stage[k] = nullptr; g.stage_cap[k] = 0; }
  g.inited = false; g.rank = 0; g.world = 1; g.transport = HC_TRANSPORT_NONE;
  return HC_OK;
}

int hc_comm_info(int *rank, int *world, int *transport) {
  if (rank) *rank = g.rank;
  if (world) *world = g.world;
  if (transport) *transport = g.transport;
  return HC_OK;
}

int hc_comm_barrier(void) { return hcm::barrier(); }
int hc_comm_allreduce(double *v, int n, int op) {
  HC_REQUIRE((v || n == 0) && n >= 0 && op >= 0 && op <= 2, "hc_comm_allreduce: bad arguments");
  return hcm::allreduce(v, n, op);
}
int hc_comm_bcast(void *buf, size_t bytes, int root) {
  HC_REQUIRE((buf || bytes == 0) && root >= 0 && root < g.world, "hc_comm_bcast: bad arguments");
  return hcm::bcast(buf, bytes, root);
}
int hc_comm_allgather(const void *mine, size_t bytes, void *all) {
  HC_REQUIRE((mine && all) || bytes == 0, "hc_comm_allgather: null pointer");
  std::vector<std::vector<char>> blocks;
  const int rc = hcm::allgatherv(mine, bytes, blocks); if (rc != HC_OK) return rc;
  for (int r = 0; r < g.world; r++) {
    if (blocks[(size_t)r].size() != bytes) return fail("hc_comm_allgather: the ranks passed blocks of different size");
    if (bytes) std::memcpy((char *)all + (size_t)r * bytes, blocks[(size_t)r].data(), bytes);
  }
  return HC_OK;
}
int hc_comm_exchange_host(int periodic, const void *send_lo, size_t n_lo, const void *send_hi, size_t n_hi, void *recv_lo, size_t m_lo,
                          void *recv_hi, size_t m_hi) {
  if (!g.inited) return fail("hc_comm_init / hc_comm_init_env has not been called");
  int lo, hi; hcm::neighbours(periodic != 0, lo, hi);
  if (lo < 0) { n_lo = 0; m_lo = 0; }
  if (hi < 0) { n_hi = 0; m_hi = 0; }
  return exchange_host(periodic != 0, send_lo, n_lo, send_hi, n_hi, recv_lo, m_lo, recv_hi, m_hi);
}

}  // extern "C"
