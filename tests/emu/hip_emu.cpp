// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"
#include <chrono>
#include <map>

namespace emu {
thread_local BlockCtx *t_ctx = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx;
thread_local unsigned t_linear;
int g_concurrency = 1;
bool starve_barriers() { return std::getenv("PLX_EMU_STARVE") != nullptr; }

namespace {
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    uint64_t gen = 0;
    unsigned active = 0, remaining = 0;
    const std::function<void()> *body = nullptr;
    dim3 bidx;
    BlockCtx ctx;
    bool stop = false;

    void worker(unsigned id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv_start.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (id >= active) continue;
            lk.unlock();
            const dim3 b = ctx.block;
            t_ctx = &ctx;
            t_linear = id;
            t_threadIdx = dim3(id % b.x, (id / b.x) % b.y, id / (b.x * b.y));
            t_blockIdx = bidx;
            (*body)();
            ctx.wave_bar[id / 64]->arrive_and_drop();
            ctx.bar->arrive_and_drop();
            lk.lock();
            if (--remaining == 0) cv_done.notify_all();
        }
    }
    void ensure(unsigned n)
    {
        while (th.size() < n) {
            unsigned id = (unsigned)th.size();
            th.emplace_back([this, id] { worker(id); });
        }
    }
    void start_block(dim3 grid, dim3 block, unsigned n, size_t shmem, dim3 b, const std::function<void()> &f)
    {
        ensure(n);
        std::unique_lock<std::mutex> lk(m);
        ctx.grid = grid;
        ctx.block = block;
        ctx.nthreads = n;
        ctx.xchg.assign(4 * (size_t)n, 0);   // (slots [0, n): the shuffles; [n, 4n): emu_shfl4's further values)
        ctx.lds.assign(shmem ? shmem : 16, (char)0x7f); // poisoned LDS: uninitialised reads show up
        ctx.bar = std::make_unique<std::barrier<>>((std::ptrdiff_t)n);
        ctx.wave_bar.clear();
        for (unsigned w = 0; w < (n + 63) / 64; w++) {
            unsigned lanes = std::min(64u, n - w * 64);
            ctx.wave_bar.push_back(std::make_unique<std::barrier<>>((std::ptrdiff_t)lanes));
        }
        active = n;
        remaining = n;
        body = &f;
        bidx = b;
        ++gen;
        cv_start.notify_all();
    }
    void wait_block()
    {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return remaining == 0; });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_start.notify_all();
        for (auto &t : th) t.join();
    }
};
// one set of pools per block size: a block start wakes every thread of its pool, and a pool grown to 1024 threads by one
// kernel would otherwise be woken 64 lanes at a time by the one-wave workgroups of another
std::vector<std::unique_ptr<Pool>> &pools(unsigned n)
{
    static std::map<unsigned, std::vector<std::unique_ptr<Pool>>> p;
    return p[n];
}
} // namespace

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
    unsigned n = block.x * block.y * block.z;
    if (n == 0 || n > 1024) { std::fprintf(stderr, "emu: bad block size %u\n", n); std::abort(); }
    if (shmem > 160 * 1024) { std::fprintf(stderr, "emu: LDS request %zu > 160 KiB\n", shmem); std::abort(); }
    static const bool trace = std::getenv("PLX_EMU_TRACE") != nullptr;
    if (trace) std::fprintf(stderr, "emu launch grid=(%u,%u,%u) block=%u lds=%zu conc=%d\n", grid.x, grid.y, grid.z, n, shmem, g_concurrency);
    const unsigned conc = (unsigned)(g_concurrency < 1 ? 1 : g_concurrency);
    auto &ps = pools(n);
    while (ps.size() < conc) ps.push_back(std::make_unique<Pool>());
    std::vector<dim3> ids;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) ids.push_back(dim3(bx, by, bz));
    for (size_t i = 0; i < ids.size(); i += conc) {
        const size_t k = std::min((size_t)conc, ids.size() - i);
        for (size_t j = 0; j < k; j++) ps[j]->start_block(grid, block, n, shmem, ids[i + j], body);
        for (size_t j = 0; j < k; j++) ps[j]->wait_block();
    }
}
} // namespace emu

double emu_now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
