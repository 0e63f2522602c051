// plx_gateway.hip -- the library-owned state of the gateway tier: scratch slots, plan caches, counters, release.
// See plx_gateway.h.  Reference seams this serves: the MEX gateways cmaadaptivefilter.c:93-174,
// easiadaptivefilter.c:95-169, fastexp.c:46-67 and the new seams of SURVEY 8(b) (fiber.m:372-389, CDE_OFDE.m, the
// receiver front end), which MATLAB calls once per pass / per span with host arrays.
#include "plx_gateway.h"

#include <cstring>
#include <list>
#include <vector>

namespace plxgw {

namespace {
struct Buf { void *p = nullptr; size_t cap = 0; };
Buf g_dev[S_COUNT], g_pin[S_COUNT];
Stats g_stats = {0, 0, 0, 0, 0, 0, 0, 0, 0};
std::mutex g_mu;

// A cache entry is found by its 64-bit key and CONFIRMED by its fingerprint: the descriptor's scalars verbatim, and for
// every table its length, a second hash (another function of the bytes) and sampled words -- a hit means an equal
// descriptor, not just an equal hash (MATLAB sweeps produce many near-identical tables).
struct Print {
    std::vector<double> scalars;
    std::vector<uint64_t> tables;          // per table: bytes, second hash, first / middle / last word
    bool operator==(const Print &o) const { return scalars == o.scalars && tables == o.tables; }
};
template <class PLAN> struct Entry { uint64_t key; Print print; PLAN *plan; };
std::list<Entry<plx_ssfm>> g_ssfm;      // most recently used first
std::list<Entry<plx_cde>> g_cde;
std::list<Entry<plx_front>> g_front;
const size_t kMaxPlans = 8;

uint64_t hash2(const void *p, size_t n)    // FNV-1a over 8-byte words folded with a rotation: independent of hash_bytes
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 0xCBF29CE484222325ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001B3ull;
        h = (h << 23) | (h >> 41);
    }
    for (; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}
void print_table(Print &pr, const void *p, size_t n)
{
    pr.tables.push_back((uint64_t)n);
    if (!p || n == 0) { pr.tables.push_back(0); return; }
    pr.tables.push_back(hash2(p, n));
    const size_t words = n / 8;
    for (size_t w : {(size_t)0, words / 2, words ? words - 1 : 0}) {
        uint64_t v = 0;
        if (words) std::memcpy(&v, (const unsigned char *)p + 8 * w, 8);
        pr.tables.push_back(v);
    }
}
template <class PLAN, class DESTROY> PLAN *lookup(std::list<Entry<PLAN>> &lst, uint64_t key, const Print &print)
{
    for (auto it = lst.begin(); it != lst.end(); ++it)
        if (it->key == key && it->print == print) {
            lst.splice(lst.begin(), lst, it);
            return lst.front().plan;
        }
    return nullptr;
}
template <class PLAN, class DESTROY> void insert(std::list<Entry<PLAN>> &lst, uint64_t key, const Print &print, PLAN *p, DESTROY destroy)
{
    lst.push_front({key, print, p});
    while (lst.size() > kMaxPlans) {
        destroy(lst.back().plan);
        lst.pop_back();
    }
}
} // namespace

// The scratch slots and the cached plans live on ONE device: the one that was current when the first of them was made.
// A gateway call on another device (plx_set_device in between) first releases that state on its own device.
static int g_device = -1;
static void release_locked()
{
    int cur = -1;
    (void)hipGetDevice(&cur);
    const bool other = g_device >= 0 && cur >= 0 && cur != g_device;
    if (other) (void)hipSetDevice(g_device);
    (void)hipDeviceSynchronize();
    for (auto &e : g_ssfm) plx_ssfm_destroy(e.plan);
    for (auto &e : g_cde) plx_cde_destroy(e.plan);
    for (auto &e : g_front) plx_front_destroy(e.plan);
    g_ssfm.clear(); g_cde.clear(); g_front.clear();
    for (int s = 0; s < S_COUNT; s++) {
        if (g_dev[s].p) (void)hipFree(g_dev[s].p);
        if (g_pin[s].p) (void)hipHostFree(g_pin[s].p);
        g_dev[s] = Buf(); g_pin[s] = Buf();
    }
    g_stats.dev_bytes = 0; g_stats.host_bytes = 0;
    g_stats.releases++;
    g_device = -1;
    if (other) (void)hipSetDevice(cur);
}
// called (under the mutex) by everything that is about to touch or create device state
static void claim_device()
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (g_device >= 0 && cur != g_device) release_locked();
    g_device = cur;
}

std::mutex &mutex() { return g_mu; }
void count_call() { g_stats.calls++; }
void count_fallback() { g_stats.fallbacks++; }
void count_plan(bool built) { if (built) g_stats.plan_builds++; else g_stats.plan_hits++; }

void *dev(Slot s, size_t bytes)
{
    claim_device();
    Buf &b = g_dev[s];
    if (bytes <= b.cap && b.p) return b.p;
    if (b.p) { (void)hipFree(b.p); g_stats.dev_bytes -= (int64_t)b.cap; b.p = nullptr; b.cap = 0; }
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;                 // geometric growth: a sweep over sizes allocates O(log) times
    if (hipMalloc(&b.p, cap) != hipSuccess) {
        b.p = nullptr;
        plx_set_error("gateway: device allocation failed");
        return nullptr;
    }
    b.cap = cap;
    g_stats.dev_allocs++;
    g_stats.dev_bytes += (int64_t)cap;
    return b.p;
}

void *pinned(Slot s, size_t bytes)
{
    claim_device();
    Buf &b = g_pin[s];
    if (bytes <= b.cap && b.p) return b.p;
    if (b.p) { (void)hipHostFree(b.p); g_stats.host_bytes -= (int64_t)b.cap; b.p = nullptr; b.cap = 0; }
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    if (hipHostMalloc(&b.p, cap, hipHostMallocDefault) != hipSuccess) {
        b.p = nullptr;
        plx_set_error("gateway: pinned host allocation failed");
        return nullptr;
    }
    b.cap = cap;
    g_stats.host_allocs++;
    g_stats.host_bytes += (int64_t)cap;
    return b.p;
}

// 64-bit multiply-xorshift hash over 8-byte words (tables are arrays of doubles), four lanes so that a 512 KiB table
// hashes at memory speed; the tail bytes are folded in one by one.  Not cryptographic: a cache key.
uint64_t hash_bytes(const void *p, size_t n, uint64_t seed)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h[4] = {seed ^ 0x9E3779B97F4A7C15ull, seed + 0xBF58476D1CE4E5B9ull, seed ^ 0x94D049BB133111EBull, seed + 0xD6E8FEB86659FD93ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, b + i, 32);
        for (int k = 0; k < 4; k++) {
            h[k] = (h[k] ^ w[k]) * 0xFF51AFD7ED558CCDull;
            h[k] ^= h[k] >> 29;
        }
    }
    uint64_t t = h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7) ^ (uint64_t)n;
    for (; i < n; i++) t = (t ^ b[i]) * 0x100000001B3ull;
    t ^= t >> 33; t *= 0xC4CEB9FE1A85EC53ull; t ^= t >> 33;
    return t;
}

plx_ssfm *ssfm_plan(const plx_ssfm_desc &d, int *rc)
{
    *rc = PLX_OK;
    if (!d.gam || !d.betat || d.nfft < 1 || d.nfc < 1) { *rc = PLX_ERR_ARG; plx_set_error("ssfm gateway: gam and betat are required"); return nullptr; }
    // the scalars (field by field: no padding bytes) + the contents of the tables
    const double sc[] = {(double)d.nfft, (double)d.nfc, (double)d.dual_pol, (double)d.max_frames, (double)d.fls[0], (double)d.fls[1],
                         (double)d.fls[2], (double)d.fls[3], d.dzmaxt, d.dphimaxt, d.alphalin, d.length, (double)d.nplates,
                         (double)d.manakov, d.db1 ? 1.0 : 0.0};
    uint64_t key = hash_bytes(sc, sizeof(sc), 1);
    const size_t tab = (size_t)d.nfft * d.nfc * sizeof(double);
    key = hash_bytes(d.gam, (size_t)d.nfc * sizeof(double), key);
    key = hash_bytes(d.betat, tab, key);
    if (d.db1) key = hash_bytes(d.db1, tab, key ^ 0x5851F42D4C957F2Dull);
    claim_device();
    Print pr;
    pr.scalars.assign(sc, sc + sizeof(sc) / sizeof(sc[0]));
    print_table(pr, d.gam, (size_t)d.nfc * sizeof(double));
    print_table(pr, d.betat, tab);
    print_table(pr, d.db1, d.db1 ? tab : 0);
    auto destroy = [](plx_ssfm *p) { plx_ssfm_destroy(p); };
    if (plx_ssfm *p = lookup<plx_ssfm, decltype(destroy)>(g_ssfm, key, pr)) { count_plan(false); return p; }
    plx_ssfm *p = nullptr;
    *rc = plx_ssfm_create(&p, &d);
    if (*rc) return nullptr;
    count_plan(true);
    insert(g_ssfm, key, pr, p, destroy);
    return p;
}

plx_cde *cde_plan(int64_t fft_len, int64_t L, const double *H, int *rc)
{
    *rc = PLX_OK;
    if (!H || fft_len < 1) { *rc = PLX_ERR_ARG; plx_set_error("cde gateway: null transfer function"); return nullptr; }
    const int64_t sc[2] = {fft_len, L};
    uint64_t key = hash_bytes(sc, sizeof(sc), 2);
    key = hash_bytes(H, (size_t)fft_len * 2 * sizeof(double), key);
    claim_device();
    Print pr;
    pr.scalars = {(double)fft_len, (double)L};
    print_table(pr, H, (size_t)fft_len * 2 * sizeof(double));
    auto destroy = [](plx_cde *p) { plx_cde_destroy(p); };
    if (plx_cde *p = lookup<plx_cde, decltype(destroy)>(g_cde, key, pr)) { count_plan(false); return p; }
    plx_cde *p = nullptr;
    *rc = plx_cde_create(&p, fft_len, L, H);
    if (*rc) return nullptr;
    count_plan(true);
    insert(g_cde, key, pr, p, destroy);
    return p;
}

plx_front *front_plan(const plx_front_desc &d, int *rc)
{
    *rc = PLX_OK;
    if (!d.hopt_re || !d.hel_re || d.nfft < 1) { *rc = PLX_ERR_ARG; plx_set_error("plx_front_create: filter tables are required"); return nullptr; }
    const double sc[] = {(double)d.nfft, (double)d.dual_pol, (double)d.max_frames, (double)d.balanced, (double)d.adcbits,
                         (double)d.decim, (double)d.ntaps, d.elo_scalar};
    uint64_t key = hash_bytes(sc, sizeof(sc), 3);
    const size_t tab = (size_t)d.nfft * sizeof(double);
    const double *tabs[6] = {d.hopt_re, d.hopt_im, d.hel_re, d.hel_im, d.elo_re, d.elo_im};
    for (int i = 0; i < 6; i++) key = tabs[i] ? hash_bytes(tabs[i], tab, key + (uint64_t)i) : (key * 31 + (uint64_t)i);
    if (d.fir && d.ntaps > 0) key = hash_bytes(d.fir, (size_t)d.ntaps * sizeof(double), key);
    claim_device();
    Print pr;
    pr.scalars.assign(sc, sc + sizeof(sc) / sizeof(sc[0]));
    for (int i = 0; i < 6; i++) print_table(pr, tabs[i], tabs[i] ? tab : 0);
    print_table(pr, d.fir, (d.fir && d.ntaps > 0) ? (size_t)d.ntaps * sizeof(double) : 0);
    auto destroy = [](plx_front *p) { plx_front_destroy(p); };
    if (plx_front *p = lookup<plx_front, decltype(destroy)>(g_front, key, pr)) { count_plan(false); return p; }
    plx_front *p = nullptr;
    *rc = plx_front_create(&p, &d);
    if (*rc) return nullptr;
    count_plan(true);
    insert(g_front, key, pr, p, destroy);
    return p;
}

} // namespace plxgw

// ------------------------------------------------------------------------------------ C ABI ---
extern "C" int plx_release_all(void)
{
    using namespace plxgw;
    std::lock_guard<std::mutex> lk(g_mu);
    release_locked();           // (on the device that owns the state, whatever device is current)
    return PLX_OK;
}

extern "C" int plx_gateway_stats(int64_t *out)
{
    using namespace plxgw;
    if (!out) PLX_FAIL(PLX_ERR_ARG, "plx_gateway_stats: null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    out[0] = g_stats.calls; out[1] = g_stats.dev_allocs; out[2] = g_stats.host_allocs; out[3] = g_stats.plan_builds;
    out[4] = g_stats.plan_hits; out[5] = g_stats.dev_bytes; out[6] = g_stats.host_bytes; out[7] = g_stats.releases;
    return PLX_OK;
}

extern "C" int plx_gateway_stats_ex(int64_t *out, int n)
{
    using namespace plxgw;
    if (!out || n < 0) PLX_FAIL(PLX_ERR_ARG, "plx_gateway_stats_ex: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t v[9] = {g_stats.calls, g_stats.dev_allocs, g_stats.host_allocs, g_stats.plan_builds, g_stats.plan_hits,
                          g_stats.dev_bytes, g_stats.host_bytes, g_stats.releases, g_stats.fallbacks};
    for (int i = 0; i < n && i < 9; i++) out[i] = v[i];
    return PLX_OK;
}
