// Error reporting + live per-family kernel timing for libunirestore_hip.so.
#include <cstring>
#include <mutex>
#include <sstream>
#include <vector>
#include <map>

#include "common.h"

namespace {
__global__ void zero_kernel(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

namespace ur {

void zero_async(void* ptr, size_t bytes, hipStream_t s) {
  const size_t n = bytes / 4;
  if (n == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)ptr, n);
}

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UR_E_LAUNCH, std::string(what) + ": " + hipGetErrorString(e));
  return UR_OK;
}

struct ProfRec {
  const char* family;
  double flops, bytes;
  hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(const char* family, double flops, double bytes, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{family, flops, bytes, get_event(), get_event()};
  hipEventRecord(r.a, s);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEventRecord(g_recs[slot].b, stream);
}

}  // namespace ur

extern "C" {

int ur_version(void) { return 100; }
const char* ur_last_error(void) { return ur::g_err.c_str(); }

int ur_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(ur::g_mu);
  ur::g_prof_on = on != 0;
  if (on) {
    for (auto& r : ur::g_recs) {
      ur::g_pool.push_back(r.a);
      ur::g_pool.push_back(r.b);
    }
    ur::g_recs.clear();
  }
  return UR_OK;
}

int ur_profile_report(char* buf, size_t buf_bytes) {
  std::lock_guard<std::mutex> lk(ur::g_mu);
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : ur::g_recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    Agg& a = agg[r.family];
    a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  std::ostringstream os;
  os << "{";
  bool first = true;
  for (auto& kv : agg) {
    if (!first) os << ", ";
    first = false;
    os << "\"" << kv.first << "\": {\"launches\": " << kv.second.n << ", \"ms\": " << kv.second.ms
       << ", \"flops\": " << kv.second.flops << ", \"bytes\": " << kv.second.bytes << "}";
  }
  os << "}";
  std::string s = os.str();
  if (s.size() + 1 > buf_bytes) return ur::fail(UR_E_INVALID, "ur_profile_report: buffer too small");
  memcpy(buf, s.c_str(), s.size() + 1);
  return UR_OK;
}

}  // extern "C"
