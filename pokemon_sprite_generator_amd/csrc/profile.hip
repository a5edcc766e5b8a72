// Live per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: zero cost on the product path.  When enabled, every launch of a profiled
// family is bracketed by a hipEventRecord pair on ITS stream; nothing synchronises until
// psg_profile_end(), which drains the stream once and sums the elapsed times.
#include <mutex>
#include <vector>

#include "psg_common.h"

namespace psg {

struct ProfRec { hipEvent_t a, b; int kind; double work, bytes; };
static double g_last_bytes[PROF_KINDS] = {0, 0, 0, 0, 0};
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_prof_mu;

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int kind, double work, hipStream_t stream, double bytes) : idx_(-1), stream_(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{take_event(), take_event(), kind, work, bytes};
    if (!r.a || !r.b) return;
    hipEventRecord(r.a, stream);
    g_recs.push_back(r);
    idx_ = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_recs[idx_].b, stream_);
}

static int g_avail_cus = 256;
static int g_reserve_rounds = 0;
int avail_cus() { return g_avail_cus; }
int avail_cus_for(double rounds_full) { return (g_reserve_rounds > 0 && rounds_full > (double)g_reserve_rounds) ? 256 : g_avail_cus; }

}  // namespace psg
using namespace psg;

extern "C" {

int psg_set_available_cus(int n) {
    PSG_REQUIRE(n == 0 || (n >= 8 && n <= 256), PSG_ERR_ARG, "set_available_cus: %d", n);
    g_avail_cus = n == 0 ? 256 : n;
    return PSG_OK;
}

int psg_set_reserve_rounds(int r) {
    PSG_REQUIRE(r >= 0 && r <= 64, PSG_ERR_ARG, "set_reserve_rounds: %d", r);
    g_reserve_rounds = r;
    return PSG_OK;
}

int psg_stream_create_cu_mask(int n_cus, psg_stream_t* stream) {
    PSG_REQUIRE(stream && n_cus >= 8 && n_cus <= 256, PSG_ERR_ARG, "stream_create_cu_mask: n_cus %d", n_cus);
    // take 256 - n_cus CUs out, spread evenly over the 8 XCDs under either numbering of the mask bits (XCD-major: bit i ->
    // XCD i / 32; round-robin: bit i -> XCD i % 8): cleared bits 33 * j (mod 256)
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFFu;
    for (int j = 0; j < 256 - n_cus; ++j) { const int bit = (33 * j) & 255; mask[bit >> 5] &= ~(1u << (bit & 31)); }
    hipStream_t s = nullptr;
    PSG_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    *stream = (psg_stream_t)s;
    return PSG_OK;
}

int psg_stream_destroy(psg_stream_t stream) {
    PSG_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return PSG_OK;
}

int psg_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_prof_on = true;
    return PSG_OK;
}

int psg_profile_end(double* ms, double* work, int64_t* launches, int nkinds) {
    PSG_REQUIRE(ms && work && launches && nkinds > 0, PSG_ERR_ARG, "profile_end: null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    for (int k = 0; k < nkinds; ++k) { ms[k] = 0; work[k] = 0; launches[k] = 0; }
    for (int k = 0; k < PROF_KINDS; ++k) g_last_bytes[k] = 0;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return hip_fail(hipGetLastError(), "profile_end sync");
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return hip_fail(hipGetLastError(), "profile_end elapsed");
        if (r.kind >= 0 && r.kind < nkinds) { ms[r.kind] += t; work[r.kind] += r.work; launches[r.kind] += 1; }
        if (r.kind >= 0 && r.kind < PROF_KINDS) g_last_bytes[r.kind] += r.bytes;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    return PSG_OK;
}

int psg_profile_bytes(double* bytes, int nkinds) {
    PSG_REQUIRE(bytes && nkinds > 0, PSG_ERR_ARG, "profile_bytes: null pointer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int k = 0; k < nkinds; ++k) bytes[k] = k < PROF_KINDS ? g_last_bytes[k] : 0.0;
    return PSG_OK;
}

}  // extern "C"
