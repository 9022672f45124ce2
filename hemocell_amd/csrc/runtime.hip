// Runtime plumbing of libhemocell_amd.so: error string, device selection,
// stream, per-kernel hipEvent profiling.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace hc {

static thread_local std::string g_error;
static hipStream_t g_own_stream = nullptr;
static hipStream_t g_stream = nullptr;       // main stream (own, or the caller's through hc_set_stream)
static hipStream_t g_side = nullptr;         // side stream of fork / join
static hipStream_t g_comm = nullptr;         // transfers of slab runs (data plane): a message must not hold up the side stream's kernels
static hipEvent_t g_fork_ev = nullptr, g_join_ev = nullptr;
static bool g_on_side = false, g_forked = false;
static bool g_initialised = false;
static bool g_profile = false;

struct ProfRecord { hipEvent_t a, b; };
static std::vector<ProfRecord> g_prof[PK_COUNT];
static double g_prof_ms[PK_COUNT];
static long g_prof_n[PK_COUNT];

void set_error(const std::string &msg) { g_error = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  g_error = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + what + " at " + file + ":" + std::to_string(line);
  return HC_ERR_HIP;
}
hipStream_t stream() { return g_on_side ? g_side : g_stream; }
hipStream_t comm_stream() { return g_comm; }
int fork() {
  HC_HIP(hipEventRecord(g_fork_ev, g_stream));
  HC_HIP(hipStreamWaitEvent(g_side, g_fork_ev, 0));
  g_forked = true;
  return HC_OK;
}
bool forked() { return g_forked; }
void route(int side) { g_on_side = side != 0; }
int join() {
  g_on_side = false; g_forked = false;
  HC_HIP(hipEventRecord(g_join_ev, g_side));
  HC_HIP(hipStreamWaitEvent(g_stream, g_join_ev, 0));
  return HC_OK;
}

ProfScope::ProfScope(int kernel) : k(kernel), on(g_profile), a(nullptr), b(nullptr) {
  if (!on) return;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
  hipEventRecord(a, stream());
}
ProfScope::~ProfScope() {
  if (!on) return;
  hipEventRecord(b, stream());
  g_prof[k].push_back({a, b});
}

static void prof_collect() {
  for (int k = 0; k < PK_COUNT; k++) {
    for (auto &r : g_prof[k]) {
      hipEventSynchronize(r.b);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_prof_ms[k] += ms; g_prof_n[k]++; }
      hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    g_prof[k].clear();
  }
}

int stat_finish(const double *d_partial, double out[3], long *n, double *h_pinned) {
  double h_stack[STAT_BLOCKS * 4];
  double *h = h_pinned ? h_pinned : h_stack;
  HC_HIP(hipMemcpyAsync(h, d_partial, sizeof(h_stack), hipMemcpyDeviceToHost, g_stream));
  HC_HIP(hipStreamSynchronize(g_stream));
  double mn = 0, mx = 0, sum = 0; long cnt = 0;
  for (int b = 0; b < STAT_BLOCKS; b++) {
    const long nb = (long)h[4 * b + 3];
    if (nb == 0) continue;
    if (cnt == 0) { mn = h[4 * b]; mx = h[4 * b + 1]; }
    else { mn = h[4 * b] < mn ? h[4 * b] : mn; mx = h[4 * b + 1] > mx ? h[4 * b + 1] : mx; }
    sum += h[4 * b + 2]; cnt += nb;
  }
  out[0] = mn; out[1] = mx; out[2] = sum; *n = cnt;
  return HC_OK;
}

}  // namespace hc

extern "C" {

const char *hc_last_error(void) { return hc::g_error.c_str(); }

int hc_device_count(int *count) {
  HC_REQUIRE(count, "hc_device_count: null pointer");
  HC_HIP(hipGetDeviceCount(count));
  return HC_OK;
}

int hc_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    hc::set_error("hc_init: no HIP device available -- libhemocell_amd has no CPU fallback");
    return HC_ERR_HIP;
  }
  HC_REQUIRE(device >= 0 && device < n, "hc_init: device index out of range");
  HC_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  HC_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    hc::set_error(std::string("hc_init: device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
    return HC_ERR_HIP;
  }
  if (!hc::g_own_stream) HC_HIP(hipStreamCreateWithFlags(&hc::g_own_stream, hipStreamNonBlocking));
  if (!hc::g_stream) hc::g_stream = hc::g_own_stream;
  if (!hc::g_side) {
    // The side stream must not share a hardware queue with the main stream, or its kernels queue up behind the collide
    // they are meant to run beside (seen with rocprofv3: two default-priority streams of this library landed on the same
    // queue).  Streams of different priority never share one, so the side stream is created with the highest priority.
    int lo = 0, hi = 0;
    HC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char *pe = std::getenv("HEMOCELL_SIDE_PRIORITY");   // A/B: "low" = the side stream below the main stream instead of above it
    HC_HIP(hipStreamCreateWithPriority(&hc::g_side, hipStreamNonBlocking, (pe && pe[0] == 'l') ? lo : hi));
    HC_HIP(hipStreamCreateWithPriority(&hc::g_comm, hipStreamNonBlocking, hi));
    HC_HIP(hipEventCreateWithFlags(&hc::g_fork_ev, hipEventDisableTiming));
    HC_HIP(hipEventCreateWithFlags(&hc::g_join_ev, hipEventDisableTiming));
  }
  hc::g_initialised = true;
  return HC_OK;
}

int hc_set_stream(void *hip_stream) {
  hc::g_stream = hip_stream ? (hipStream_t)hip_stream : hc::g_own_stream;
  return HC_OK;
}

int hc_synchronize(void) {
  HC_HIP(hipStreamSynchronize(hc::g_stream));
  if (hc::g_side) HC_HIP(hipStreamSynchronize(hc::g_side));
  if (hc::g_comm) HC_HIP(hipStreamSynchronize(hc::g_comm));
  return HC_OK;
}

int hc_fork(void) { return hc::fork(); }
int hc_route(int side) { hc::route(side); return HC_OK; }
int hc_join(void) { return hc::join(); }
int hc_side_stream(void **hip_stream) {
  HC_REQUIRE(hip_stream, "hc_side_stream: null pointer");
  HC_REQUIRE(hc::g_side, "hc_side_stream: hc_init() has not been called");
  *hip_stream = (void *)hc::g_side;
  return HC_OK;
}

#ifndef HC_KERNEL_TAG
#define HC_KERNEL_TAG "untagged"
#endif
const char *hc_build_tag(void) { return HC_KERNEL_TAG; }

int hc_measure_copy_bandwidth(size_t bytes, int repeats, double *gbytes_per_s) {
  HC_REQUIRE(bytes > 0 && repeats > 0 && gbytes_per_s, "hc_measure_copy_bandwidth: bad arguments");
  if (!hc::g_stream) { hc::set_error("hc_measure_copy_bandwidth: hc_init() has not been called"); return HC_ERR_STATE; }
  void *a = nullptr, *b = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&a, bytes);
  if (e == hipSuccess) e = hipMalloc(&b, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, hc::g_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, hc::g_stream);   // warm
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipEventRecord(e0, hc::g_stream);
  for (int k = 0; k < repeats && e == hipSuccess; k++) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, hc::g_stream);
  if (e == hipSuccess) e = hipEventRecord(e1, hc::g_stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (a) hipFree(a);
  if (b) hipFree(b);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  if (e != hipSuccess) return hc::hip_fail(e, "hc_measure_copy_bandwidth", __FILE__, __LINE__);
  *gbytes_per_s = 2.0 * (double)bytes * repeats / ((double)ms * 1e-3) / 1e9;
  return HC_OK;
}

int hc_profile_enable(int on) { hc::g_profile = on != 0; return HC_OK; }
int hc_profile_reset(void) {
  hc::prof_collect();
  for (int k = 0; k < hc::PK_COUNT; k++) { hc::g_prof_ms[k] = 0; hc::g_prof_n[k] = 0; }
  return HC_OK;
}
int hc_profile_read(const char *kernel, double *total_ms, long *launches) {
  HC_REQUIRE(kernel && total_ms && launches, "hc_profile_read: null pointer");
  static const char *names[hc::PK_COUNT] = {"collide_stream_alone", "ibm_spread", "ibm_interpolate", "advance", "mechanics", "collide_stream_beside"};
  hc::prof_collect();
  if (std::strcmp(kernel, "collide_stream") == 0) {   // every launch of the collide kernel
    *total_ms = hc::g_prof_ms[hc::PK_COLLIDE] + hc::g_prof_ms[hc::PK_COLLIDE_BESIDE];
    *launches = hc::g_prof_n[hc::PK_COLLIDE] + hc::g_prof_n[hc::PK_COLLIDE_BESIDE];
    return HC_OK;
  }
  for (int k = 0; k < hc::PK_COUNT; k++)
    if (std::strcmp(kernel, names[k]) == 0) { *total_ms = hc::g_prof_ms[k]; *launches = hc::g_prof_n[k]; return HC_OK; }
  hc::set_error(std::string("hc_profile_read: unknown kernel ") + kernel);
  return HC_ERR_ARG;
}

}  // extern "C"
