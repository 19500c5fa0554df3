// C ABI entry points (include/hgs_rasterizer.h): argument validation, scratch layout, stage sequencing.
#include <atomic>
#include <sched.h>
#include <time.h>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hgs_common.h"

using namespace hgs;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

namespace {
// The A/B switches (hgs_common.h) are published as an IMMUTABLE snapshot: a reload builds a new one and swaps the pointer, and every
// entry point copies the snapshot once, at entry, into a thread-local block that the launch paths below it read (FrameSwitches) --
// a frame never sees a mix of two loads, whatever other threads render or reload meanwhile.
std::shared_ptr<const hgs::Switches> g_sw_snapshot;   // (std::atomic_load / std::atomic_store)
std::once_flag g_sw_once;
std::shared_ptr<const hgs::Switches> read_switches_from_env()
{
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };   // default on
    auto on = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };    // default off
    auto s = std::make_shared<hgs::Switches>();
    const char* e = getenv("HGS_BIN_MODE");
    s->bin_mode = e && (e[0] == 'c' || e[0] == 'o') ? e[0] : 0;
    s->bwd_two_launches = on("HGS_BWD_TWO_LAUNCHES");
    s->deep_forward = !off("HGS_DEEP_FORWARD");
    s->long_min_sparse = num("HGS_LONG_MIN_SPARSE"), s->long_min_dense = num("HGS_LONG_MIN_DENSE");
    s->emit_scan = !off("HGS_EMIT_SCAN");
    s->k1_stage_sh = on("HGS_K1_STAGE_SH");
    e = getenv("HGS_BIG_PER_GROUP");
    s->big_per_group = e ? std::max(0, std::min(atoi(e), 64)) : hgs::BIG_PER_GROUP;
    s->bwd_segmented = !off("HGS_BWD_SEGMENTED");
    s->fused_sort_blend = !off("HGS_FUSED_SORT_BLEND");
    e = getenv("HGS_BWD_WAVES_PER_TILE");
    s->bwd_waves_per_tile = e && (e[0] == '1' || e[0] == '4') ? e[0] - '0' : 0;
    e = getenv("HGS_K8_COOP");
    s->k8_coop = e ? atoi(e) : -1;
    s->deep_min = std::max(0, num("HGS_DEEP_MIN"));
    e = getenv("HGS_FRAME_KIND");
    s->frame_kind = e && (e[0] == 's' || e[0] == 'd') ? e[0] : 0;
    return s;
}
std::shared_ptr<const hgs::Switches> snapshot()
{
    std::call_once(g_sw_once, [] { std::atomic_store(&g_sw_snapshot, read_switches_from_env()); });
    return std::atomic_load(&g_sw_snapshot);
}
thread_local hgs::Switches t_sw;          // this thread's copy: the frame's, while an entry point is running
thread_local int t_sw_depth = 0;
struct FrameSwitches {   // RAII at the top of every entry point that launches
    FrameSwitches() { if (t_sw_depth++ == 0) t_sw = *snapshot(); }
    ~FrameSwitches() { --t_sw_depth; }
};
}  // namespace
const hgs::Switches& hgs::switches()
{
    if (t_sw_depth == 0) t_sw = *snapshot();   // (outside an entry point: a fresh copy per question)
    return t_sw;
}

// for the other translation units (densify.hip, knn.hip): message behind hgs_last_error() on this thread
void hgs::set_last_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

namespace {

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(HGS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// debug=True in the reference's settings means: synchronise and check after every stage
#define STAGE_CHECK(debug, st, name)                                                                     \
    do {                                                                                                 \
        hipError_t e_ = hipGetLastError();                                                               \
        if (e_ == hipSuccess && (debug)) e_ = hipStreamSynchronize(st);                                  \
        if (e_ != hipSuccess) return fail(HGS_ERR_HIP, "stage %s: %s", name, hipGetErrorString(e_));     \
    } while (0)

// ---- optional per-stage timing (hipEvent pairs on the launch stream) ----
struct ProfState {
    std::mutex mu;
    uint32_t mask = 0;
    uint32_t every = 1;                       // time every `every`-th launch of a stage (event pairs cost ~5 us of GPU time each)
    uint32_t seen[HGS_NUM_STAGES] = {0};
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[HGS_NUM_STAGES] = {0};
    int64_t launches[HGS_NUM_STAGES] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void drain()
    {
        for (auto& p : pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                total_ms[p.stage] += ms;
                launches[p.stage] += 1;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
} g_prof;

struct ProfScope {
    int stage; hipStream_t st; hipEvent_t a = nullptr;
    ProfScope(int stage_, hipStream_t st_) : stage(stage_), st(st_)
    {
        if (g_prof.mask & (1u << stage)) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            if (g_prof.seen[stage]++ % g_prof.every) return;
            a = g_prof.get();
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope()
    {
        if (a) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            hipEvent_t b = g_prof.get();
            (void)hipEventRecord(b, st);
            g_prof.pending.push_back({stage, a, b});
        }
    }
};

// Host-time accounting (hgs_debug_stat): nanoseconds spent inside the two entry points and, of the forward's, spinning for N --
// "host busy per frame" of a frame loop = (its wall time - forward_wait_ns) / frames.  Two clock reads per call.
std::atomic<uint64_t> g_stat_fwd_calls{0}, g_stat_fwd_ns{0}, g_stat_wait_ns{0}, g_stat_bwd_calls{0}, g_stat_bwd_ns{0};
std::atomic<uint64_t> g_stat_binning_reruns{0}, g_stat_ckpt_reruns{0};   // optimistically enqueued frames that were run again
inline uint64_t now_ns()
{
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct StatScope {
    std::atomic<uint64_t>&calls, &ns; uint64_t t0;
    StatScope(std::atomic<uint64_t>& c, std::atomic<uint64_t>& n) : calls(c), ns(n), t0(now_ns()) {}
    ~StatScope() { calls.fetch_add(1, std::memory_order_relaxed), ns.fetch_add(now_ns() - t0, std::memory_order_relaxed); }
};

// The host learns N from a pinned, host-coherent 64-bit slot that tile_scan_kernel writes -- (sparse-frame bit << 63 |
// long-tiles bit << 62 | 30-bit ticket << 32 | N) -- and the host polls: a ring of slots so that calls from several
// threads / streams do not collide.
struct HostSlot { volatile unsigned long long* word; uint32_t ticket; uint32_t index; };
constexpr unsigned SLOT_RING = 1024;  // deferred frames are polled later: far more slots than frames anyone keeps in flight
unsigned long long* g_slot_base = nullptr;
// (device, stream, shape) of the frame that holds each slot, for hgs_forward_poll: a deferred frame's counts feed the shape's
// launch-size record there (the waiting path records them in hgs_rasterize_forward itself)
struct SlotOwner { uint32_t ticket = 0; int dev = -1; void* st = nullptr; int P = 0, H = 0, W = 0; };
SlotOwner g_slot_owner[SLOT_RING];
std::mutex g_slot_owner_mu;
HostSlot host_slot()
{
    static std::mutex mu;
    static uint32_t next = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!g_slot_base) {
        if (hipHostMalloc((void**)&g_slot_base, SLOT_RING * 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            g_slot_base = nullptr;
            return {nullptr, 0, 0};
        }
        memset(g_slot_base, 0, SLOT_RING * 64);
    }
    next = (next + 1) & 0x3FFFFFFFu;
    if (next == 0) ++next;  // ticket 0 is the initial content of a slot
    return {g_slot_base + 8 * (next % SLOT_RING), next, next % SLOT_RING};
}

// What the slot says about ticket `mine`: 1 = it carries it (N and flags are there), 0 = not yet (it still holds an older
// frame's word, or nothing), -1 = EXPIRED: it carries a LATER ticket -- the ring went all the way round (SLOT_RING forwards)
// before this frame was looked at, and its N is gone.  Tickets are 30-bit serial numbers: "later" = ahead by less than 2^29.
int slot_state(unsigned long long v, uint32_t mine)
{
    const uint32_t t = (uint32_t)(v >> 32) & 0x3FFFFFFFu;
    if (t == mine) return 1;
    if (t == 0) return 0;
    return (((t - mine) & 0x3FFFFFFFu) < 0x20000000u) ? -1 : 0;
}

int expired()
{
    return fail(HGS_ERR_EXPIRED, "the deferred frame's result slot was reused by a later frame (more than %u forwards since): run it again", SLOT_RING);
}

// Wait until the slot carries this call's ticket.  Round 5 spun (a sched_yield now and then): N arrives 30-50 us after the frame's first
// kernel STARTS, but on a GPU-bound frame loop that kernel waits behind the frame before -- the host sits here for 300 us of a 476 us
// frame (C2) with nothing urgent to do, the whole frame being enqueued already -- and every rank of a box held a core doing so: two cores
// per rank with the runtime's event thread (below), 15.6 of a 16-CPU quota at eight ranks.  Round 6: the shape's record remembers how long
// its last waits were (FrameHistory::wait_ns); a wait expected to last SLEEP_FROM_NS and more is slept through in one nanosleep that ends
// WAKE_EARLY_NS before the expected arrival (timer slack and wake-up latency are ~60 us), then spun; shorter waits -- the human-only
// renders, 110-150 us a frame -- spin as before (20 us naps there cost C3 11 %).  A wait that outlasts its expectation by 1 ms goes on in
// 20 us naps.  HGS_WAIT_SLEEP=0: spin only.
// Every ~20 ms the runtime is asked about the stream: an error there (a faulted kernel) or an idle stream without the ticket means N is
// never going to arrive.  (Round 5 asked every 16 384 probes -- once or twice per frame -- and every hipStreamQuery leaves a marker whose
// completion the runtime's event thread handles: that thread ran at 40-75 % of a core.)
constexpr uint64_t SLEEP_FROM_NS = 300000ull, WAKE_EARLY_NS = 200000ull;
int wait_for_slot(const HostSlot& hs, hipStream_t st, uint32_t* n_out, bool* sparse_out, bool* long_out, int64_t expected_ns = 0, uint64_t* waited_ns = nullptr)
{
    struct WaitTime {
        uint64_t t0 = now_ns(); uint64_t* out; uint64_t report_instead = 0;   // (what the shape's record is told when the measured wait is not the wait)
        ~WaitTime() { const uint64_t d = now_ns() - t0; g_stat_wait_ns.fetch_add(d, std::memory_order_relaxed); if (out) *out = report_instead ? report_instead : d; }
    } wt;
    wt.out = waited_ns;
    static const bool may_sleep = [] { const char* e = getenv("HGS_WAIT_SLEEP"); return !(e && e[0] == '0'); }();
    uint64_t next_query = wt.t0 + 20000000ull;
    bool slept = false, napping = false, just_woke = false, decayed = false;
    // How early the sleep ends: WAKE_EARLY_NS, or three times what this process's sleeps have been overshooting by (a running mean; ~60 us of
    // timer slack + wake-up latency on a quiet box, several hundred on a loaded one -- where the margin then outgrows the waits and the
    // frames are spun for again).  Round 6's last 8-rank run met such a box: 300 000 Gaussians 1 790 -> 1 499 FPS with the fixed margin,
    // and the overslept waits fed the shape's expectation, which lengthened the next sleep.
    static std::atomic<uint64_t> oversleep_ns{0};
#ifdef HGS_WAIT_FIXED_MARGIN   // (A/B builds: the wait as it was before the margin adapted)
    const uint64_t margin = WAKE_EARLY_NS;
#else
    const uint64_t margin = std::max<uint64_t>(WAKE_EARLY_NS, 3ull * oversleep_ns.load(std::memory_order_relaxed));
#endif
    // (measurements only: HGS_WAIT_TEST_OVERSLEEP_US lengthens every sleep by that much -- a box with sluggish timers on demand)
    static const uint64_t test_oversleep_ns = [] { const char* e = getenv("HGS_WAIT_TEST_OVERSLEEP_US"); return e ? 1000ull * strtoull(e, nullptr, 10) : 0ull; }();
    uint64_t meant_to_wake_at = 0;
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long v = *hs.word;
        const int state = slot_state(v, hs.ticket);
        if (state > 0) {
            *n_out = (uint32_t)v, *sparse_out = (v >> 63) != 0, *long_out = ((v >> 62) & 1u) != 0;
            // (N was there at the first look after the sleep: it arrived at some point DURING the sleep, and the time until this look says
            //  nothing about when -- the shape's record gets the time the sleep was meant to end, which shortens the next one)
#ifndef HGS_WAIT_FIXED_MARGIN
            if (just_woke) wt.report_instead = std::max<uint64_t>(meant_to_wake_at - wt.t0, 1);
#endif
            return HGS_OK;
        }
        just_woke = false;
        if (state < 0) return expired();
        if (napping || (spins & 0xFFu) == 0u) {   // (a clock read per nap, or per 256 probes while spinning)
            const uint64_t now = now_ns(), waited = now - wt.t0;
            if (now >= next_query) {
                next_query = now + 20000000ull;
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    const unsigned long long v2 = *hs.word;
                    const int state2 = slot_state(v2, hs.ticket);
                    if (state2 > 0) {
                        *n_out = (uint32_t)v2, *sparse_out = (v2 >> 63) != 0, *long_out = ((v2 >> 62) & 1u) != 0;
                        return HGS_OK;
                    }
                    if (state2 < 0) return expired();
                    return fail(HGS_ERR_HIP, "stream went idle without publishing the number of rendered pairs");
                }
                if (q != hipErrorNotReady) return fail(HGS_ERR_HIP, "HIP error while waiting for tile_scan: %s", hipGetErrorString(q));
            }
            if (may_sleep && !slept && waited > 20000ull && expected_ns >= (int64_t)SLEEP_FROM_NS && waited + margin < (uint64_t)expected_ns) {
                // (the first 20 us are spun: a GPU that was idle delivers N at once)
                slept = true;
                const uint64_t d = (uint64_t)expected_ns - margin - waited;
                const uint64_t dt = d + test_oversleep_ns;
                struct timespec ts = {(time_t)(dt / 1000000000ull), (long)(dt % 1000000000ull)};
                nanosleep(&ts, nullptr);
                meant_to_wake_at = now + d;
                const uint64_t woke = now_ns(), over = woke > meant_to_wake_at ? woke - meant_to_wake_at : 0;
                oversleep_ns.store((3ull * oversleep_ns.load(std::memory_order_relaxed) + over) / 4ull, std::memory_order_relaxed);
                just_woke = true;
                continue;
            }
#ifndef HGS_WAIT_FIXED_MARGIN
            // (a wait the minimum margin would have slept through and the adapted one does not: the overshoot estimate only learns from sleeps,
            //  so it is let go of slowly -- after ~80 such frames one sleep probes the host again; one late frame in 80 if it still overshoots)
            else if (may_sleep && !slept && !decayed && waited > 20000ull && expected_ns >= (int64_t)SLEEP_FROM_NS && waited + WAKE_EARLY_NS < (uint64_t)expected_ns) {
                decayed = true;
                oversleep_ns.store(oversleep_ns.load(std::memory_order_relaxed) * 63ull / 64ull, std::memory_order_relaxed);
            }
#endif
            // a wait far beyond what the shape's record (or, without one, a millisecond) allows: the GPU is busy with someone else's work
            if (may_sleep && !napping && waited > (uint64_t)(expected_ns > 0 ? expected_ns : 0) + 1000000ull) napping = true;
        }
        if (napping) {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        } else {
            __builtin_ia32_pause();
            // (spin-only mode: a rank that shares its core hands it on now and then, as round 5 did)
            if (!may_sleep && spins > 8192u && (spins & 0xFFu) == 0u) sched_yield();
        }
    }
}

// Per-tile pair counters: one small device array per (device, stream), ZERO between frames -- the preprocess kernel adds
// into it and tile_scan_kernel, its only reader, zeroes what it read.  (A frame cannot zero them itself: the adds of the
// preprocess kernel's workgroups must not race with a fill by another workgroup of the same kernel; a memset node in
// front of every frame is a launch this design does without.)
//  * An entry is LEASED from the moment a forward call picks it until that call has enqueued the scan (or gives up): a second
//    host thread issuing frames on the same stream waits at the lease, so its preprocess kernel can never add into counters
//    the first one's scan has not consumed yet.  A call that ends between the two kernels leaves the entry marked dirty and
//    the next lease zeroes it first.
//  * The table is a hash map, and bounded: beyond TC_MAX_ENTRIES, entries of streams that are idle (or gone) are dropped and
//    their arrays freed.  An array that has become too small (a larger frame on the same stream) is freed after the
//    stream has drained -- growth is rare, one synchronisation then is cheap.
struct TileCounters {
    uint32_t* buf = nullptr; size_t tiles = 0; bool dirty = true;
    uint64_t last_use = 0;   // (table clock: eviction takes the least recently used)
    std::mutex lease;
    // (no hipFree here: the table is a namespace-scope static, and HIP calls from static destructors -- process exit, dlclose --
    //  run after the runtime may have shut down.  Arrays are freed explicitly where entries are evicted or grow; what the table
    //  still holds at exit goes with the process.)
};
struct TcKey {
    int dev; hipStream_t st;
    bool operator==(const TcKey& o) const { return dev == o.dev && st == o.st; }
};
struct TcHash {
    size_t operator()(const TcKey& k) const { return std::hash<const void*>()((const void*)k.st) * 31u + (size_t)k.dev; }
};
constexpr size_t TC_MAX_ENTRIES = 64;
std::mutex g_tc_mu;
uint64_t g_tc_clock = 0;
// (shared_ptr: a caller PINS its entry while it holds it -- an eviction by another thread between the table lookup and the lease
//  can then drop the table's reference but never the entry itself)
std::unordered_map<TcKey, std::shared_ptr<TileCounters>, TcHash> g_tc;

// RAII lease of a stream's counters (see above)
struct TileCounterLease {
    std::shared_ptr<TileCounters> e;
    bool scan_enqueued = false;
    void release()
    {
        if (!e) return;
        e->dirty = !scan_enqueued;
        e->lease.unlock();
        e.reset();
    }
    ~TileCounterLease() { release(); }
};

int acquire_tile_counters(hipStream_t st, size_t tiles, uint32_t** out, TileCounterLease* lease)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::shared_ptr<TileCounters> e;
    std::vector<std::shared_ptr<TileCounters>> evicted;
    {
        std::lock_guard<std::mutex> lk(g_tc_mu);
        auto it = g_tc.find(TcKey{dev, st});
        if (it == g_tc.end()) {
            if (g_tc.size() >= TC_MAX_ENTRIES) {
                // Drop the least recently used entries nobody holds (use_count 1 = the table's own reference), down to half
                // the table.  They leave the table here, under the lock (nobody can look them up any more); their arrays are
                // freed below, outside it.  (Rare: more than TC_MAX_ENTRIES streams have rendered on this device.)
                std::vector<std::pair<uint64_t, TcKey>> idle;
                for (auto& kv : g_tc)
                    if (kv.first.dev == dev && kv.second.use_count() == 1) idle.push_back({kv.second->last_use, kv.first});
                std::sort(idle.begin(), idle.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
                const size_t drop = idle.empty() ? 0 : std::min(idle.size(), g_tc.size() - TC_MAX_ENTRIES / 2);
                for (size_t k = 0; k < drop; ++k) {
                    auto victim = g_tc.find(idle[k].second);
                    evicted.push_back(std::move(victim->second));
                    g_tc.erase(victim);
                }
            }
            it = g_tc.emplace(TcKey{dev, st}, std::make_shared<TileCounters>()).first;
        }
        e = it->second;
        e->last_use = ++g_tc_clock;
    }
    if (!evicted.empty()) {
        // The evicted streams may still have frames in flight that use the arrays -- and may no longer exist, so their handles
        // are not probed: the device is drained once (no lock held: other threads' forwards go on), then the arrays are freed.
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        for (auto& d : evicted) {
            if (d->buf) (void)hipFree(d->buf);
            d->buf = nullptr, d->tiles = 0;
        }
    }
    e->lease.lock();  // (outside the table lock: another thread may be between its preprocess kernel and its scan)
    lease->e = e;   // (the lease holds the pin)
    const size_t want = (tiles + 7) / 8 * 8;  // the scan reads whole groups of eight
    if (e->tiles < want) {
        if (e->buf) {
            // earlier frames of this stream may still be using the smaller array
            const hipError_t er = hipStreamSynchronize(st);
            if (er != hipSuccess) return fail(HGS_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(er));
            (void)hipFree(e->buf);
            e->buf = nullptr, e->tiles = 0;
        }
        if (hipMalloc((void**)&e->buf, want * sizeof(uint32_t)) != hipSuccess) {
            e->buf = nullptr;
            return fail(HGS_ERR_ALLOC, "tile counter allocation failed");
        }
        e->tiles = want, e->dirty = true;
    }
    if (e->dirty) {
        const hipError_t er = hipMemsetAsync(e->buf, 0, e->tiles * sizeof(uint32_t), st);
        if (er != hipSuccess) return fail(HGS_ERR_HIP, "hipMemsetAsync(tile counters): %s", hipGetErrorString(er));
        e->dirty = false;
    }
    *out = e->buf;
    return HGS_OK;
}

// Launch-size memory per (device, stream, frame shape): a small direct-mapped table, overwritten on collision.  Hints: a frame's
// image, radii and lists never depend on them.  What CAN depend on them is which of two equivalent backward forms runs (a dense
// frame leaves checkpoints for the depth-segmented backward only when the shape's record shows lists beyond SORT_CAP_SMALL entries):
// gradients then agree up to fp32 summation order, as they do between any two runs of the atomics-based backward.
// (The shape belongs to the key: a HUGS step renders the joint frame and the human-only frame in turn on one stream.)
struct HistKey {
    int dev; hipStream_t st; int P, H, W;
    bool operator==(const HistKey& o) const { return dev == o.dev && st == o.st && P == o.P && H == o.H && W == o.W; }
};
struct HistEntry { HistKey k{-1, nullptr, 0, 0, 0}; FrameHistory h; };
std::mutex g_hist_mu;
HistEntry g_hist[128];
size_t hist_slot(const HistKey& k)
{
    size_t h = std::hash<const void*>()((const void*)k.st);
    for (int v : {k.dev, k.P, k.H, k.W}) h = h * 1000003u + (size_t)v;
    return h % 128u;
}
FrameHistory history_get(const HistKey& k)
{
    std::lock_guard<std::mutex> lk(g_hist_mu);
    const HistEntry& e = g_hist[hist_slot(k)];
    return e.k == k ? e.h : FrameHistory{};
}
void history_put(const HistKey& k, const HostSlot& hs, uint64_t waited_ns = 0)
{
    FrameHistory h;
    {   // (how long the shape's frames wait for N: a running mean, 3 : 1; a deferred frame -- polled long after -- leaves it alone)
        std::lock_guard<std::mutex> lk(g_hist_mu);
        const HistEntry& e = g_hist[hist_slot(k)];
        const int64_t before = e.k == k ? e.h.wait_ns : -1;
        h.wait_ns = waited_ns == 0 ? before : before < 0 ? (int64_t)waited_ns : (3 * before + (int64_t)waited_ns) / 4;
    }
    // (written by tile_scan_kernel before the word that carries the ticket)
    h.n_long = (int32_t)hs.word[1], h.n_huge = (int32_t)(hs.word[2] & 0xFFFFFFFFull), h.n_deep = (int32_t)(hs.word[2] >> 32);
    h.sparse = (int32_t)(hs.word[0] >> 63);
    h.n_nonempty = (int32_t)(hs.word[4] & 0xFFFFFFFFull), h.deep_blend = (int32_t)((hs.word[4] >> 32) & 1ull);
    h.no_ckpt = (int32_t)((hs.word[4] >> 33) & 1ull);
    std::lock_guard<std::mutex> lk(g_hist_mu);
    g_hist[hist_slot(k)] = HistEntry{k, h};
}

// 1: N and flags read; 0: not there yet; -1: expired
int slot_ready(const HostSlot& hs, uint32_t* n_out, bool* sparse_out, bool* long_out)
{
    const unsigned long long v = *hs.word;
    const int state = slot_state(v, hs.ticket);
    if (state > 0) *n_out = (uint32_t)v, *sparse_out = (v >> 63) != 0, *long_out = ((v >> 62) & 1u) != 0;
    return state;
}

int too_many_pairs()
{
    return fail(HGS_ERR_OVERFLOW, "the frame has 2^32 - 16 or more (tile, Gaussian) pairs: more than 32-bit list positions can address");
}

int bits_for(uint32_t n)  // number of bits needed to represent values in [0, n)
{
    int b = 0;
    while (b < 32 && (1ull << b) < (uint64_t)n) ++b;
    return b < 1 ? 1 : b;
}

int make_camera(const hgs_forward_args& a, Camera& cam)
{
    const hgs_settings& s = a.s;
    if (a.P < 0 || a.seg2.P < 0) return fail(HGS_ERR_INVALID_ARGUMENT, "P must be >= 0");
    if ((uint64_t)a.P + (uint64_t)a.seg2.P > GID_MASK / 4u)
        return fail(HGS_ERR_INVALID_ARGUMENT, "P must be < 2^26 (index packing / 32-bit record offsets)");
    if (a.seg2.P > 0) {
        // the second segment holds the same KINDS of inputs as the first (the kernels choose the array set per Gaussian)
        const hgs_segment& b = a.seg2;
        if (a.P == 0) return fail(HGS_ERR_INVALID_ARGUMENT, "a second segment needs a non-empty first one (pass the Gaussians as the first)");
        if (!b.means3D || !b.opacities) return fail(HGS_ERR_INVALID_ARGUMENT, "seg2: means3D and opacities are required");
        if ((b.shs != nullptr) != (a.shs != nullptr) || (b.colors_precomp != nullptr) != (a.colors_precomp != nullptr) ||
            (b.scales != nullptr) != (a.scales != nullptr) || (b.rotations != nullptr) != (a.rotations != nullptr) ||
            (b.cov3D_precomp != nullptr) != (a.cov3D_precomp != nullptr))
            return fail(HGS_ERR_INVALID_ARGUMENT, "seg2 must provide the same kinds of inputs as the first segment");
        if (b.shs && b.M < (s.sh_degree + 1) * (s.sh_degree + 1))
            return fail(HGS_ERR_INVALID_ARGUMENT, "seg2.shs holds %d coefficients, degree %d needs %d", b.M, s.sh_degree,
                        (s.sh_degree + 1) * (s.sh_degree + 1));
    }
    if (s.image_height <= 0 || s.image_width <= 0) return fail(HGS_ERR_INVALID_ARGUMENT, "image size must be positive");
    if (a.P > 0 && !a.means3D) return fail(HGS_ERR_INVALID_ARGUMENT, "means3D must have dimensions (num_points, 3)");
    if (!s.bg || !s.viewmatrix || !s.projmatrix || !s.campos)
        return fail(HGS_ERR_INVALID_ARGUMENT, "bg, viewmatrix, projmatrix and campos are required device pointers");
    if (a.P > 0) {
        if ((a.shs != nullptr) == (a.colors_precomp != nullptr))
            return fail(HGS_ERR_INVALID_ARGUMENT, "Please provide excatly one of either SHs or precomputed colors!");
        const bool sr = a.scales != nullptr && a.rotations != nullptr;
        if (sr == (a.cov3D_precomp != nullptr) || ((a.scales != nullptr) != (a.rotations != nullptr)))
            return fail(HGS_ERR_INVALID_ARGUMENT,
                        "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        if (!a.opacities) return fail(HGS_ERR_INVALID_ARGUMENT, "opacities is required");
        if (a.shs) {
            if (s.sh_degree < 0 || s.sh_degree > 3) return fail(HGS_ERR_INVALID_ARGUMENT, "sh_degree must be in 0..3");
            if (a.M < (s.sh_degree + 1) * (s.sh_degree + 1))
                return fail(HGS_ERR_INVALID_ARGUMENT, "shs holds %d coefficients, degree %d needs %d", a.M, s.sh_degree,
                            (s.sh_degree + 1) * (s.sh_degree + 1));
        }
    }
    cam.W = s.image_width, cam.H = s.image_height;
    cam.gx = (cam.W + TILE - 1) / TILE, cam.gy = (cam.H + TILE - 1) / TILE;
    cam.tanfovx = s.tanfovx, cam.tanfovy = s.tanfovy;
    cam.fx = (float)cam.W / (2.0f * s.tanfovx);
    cam.fy = (float)cam.H / (2.0f * s.tanfovy);
    cam.mod = s.scale_modifier;
    cam.D = s.sh_degree, cam.M = a.M;
    cam.scale_grad_factor = s.scale_modifier;
    return HGS_OK;
}

}  // namespace

extern "C" {

const char* hgs_last_error(void) { return g_err; }
int32_t hgs_abi_version(void) { return HGS_ABI_VERSION; }

void hgs_profile_enable(uint32_t stage_mask)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.mask = stage_mask;
}
void hgs_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    for (int i = 0; i < HGS_NUM_STAGES; ++i) g_prof.total_ms[i] = 0, g_prof.launches[i] = 0, g_prof.seen[i] = 0;
}
void hgs_profile_set_sampling(uint32_t every_nth)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.every = every_nth ? every_nth : 1;
}
int32_t hgs_profile_read(int32_t stage, double* total_ms, int64_t* launches)
{
    if (stage < 0 || stage >= HGS_NUM_STAGES) return fail(HGS_ERR_INVALID_ARGUMENT, "bad stage %d", stage);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    if (total_ms) *total_ms = g_prof.total_ms[stage];
    if (launches) *launches = g_prof.launches[stage];
    return HGS_OK;
}
const char* hgs_stage_name(int32_t stage)
{
    static const char* names[HGS_NUM_STAGES] = {"preprocess", "scan", "emit_keys", "sort",
                                                "blend_forward", "blend_backward", "preprocess_backward"};
    return stage >= 0 && stage < HGS_NUM_STAGES ? names[stage] : "?";
}

size_t hgs_geom_bytes(int32_t P, int32_t H, int32_t W) { return GeomLayout(P < 1 ? 1 : P, num_tiles_of(H, W)).total; }
size_t hgs_image_bytes(int32_t H, int32_t W) { return ImageLayout(H, W).total; }
size_t hgs_binning_bytes(int64_t N, int32_t, int32_t) { return BinningLayout(N).total; }
size_t hgs_ckpt_bytes(int64_t N, int32_t H, int32_t W) { return CkptLayout(N, num_tiles_of(H, W)).total; }
size_t hgs_ckpt_bytes_for_slots(int64_t slots) { return CkptLayout((size_t)(slots < 1 ? 1 : slots)).total; }

void hgs_reload_switches(void)
{
    (void)snapshot();   // (the first load must not overwrite this one afterwards)
    std::atomic_store(&g_sw_snapshot, read_switches_from_env());
}

int64_t hgs_debug_stat(const char* name)
{
    if (!name) return -1;
    if (!strcmp(name, "tile_counter_entries")) {
        std::lock_guard<std::mutex> lk(g_tc_mu);
        return (int64_t)g_tc.size();
    }
    if (!strcmp(name, "tile_counter_max_entries")) return (int64_t)TC_MAX_ENTRIES;
    if (!strcmp(name, "slot_ring")) return (int64_t)SLOT_RING;
    if (!strcmp(name, "forward_calls")) return (int64_t)g_stat_fwd_calls.load();
    if (!strcmp(name, "forward_ns")) return (int64_t)g_stat_fwd_ns.load();
    if (!strcmp(name, "forward_wait_ns")) return (int64_t)g_stat_wait_ns.load();
    if (!strcmp(name, "backward_calls")) return (int64_t)g_stat_bwd_calls.load();
    if (!strcmp(name, "backward_ns")) return (int64_t)g_stat_bwd_ns.load();
    if (!strcmp(name, "binning_reruns")) return (int64_t)g_stat_binning_reruns.load();
    if (!strcmp(name, "ckpt_reruns")) return (int64_t)g_stat_ckpt_reruns.load();
    return -1;
}

size_t hgs_scratch_offset(const char* name, int32_t P, int64_t N, int32_t H, int32_t W)
{
    GeomLayout g(P < 1 ? 1 : P, num_tiles_of(H, W));
    ImageLayout im(H, W);
    BinningLayout b(N);
    if (!strcmp(name, "splats")) return g.splats;
    if (!strcmp(name, "tiles_touched")) return g.tiles_touched;
    if (!strcmp(name, "list")) return b.list;
    if (!strcmp(name, "final_T")) return im.final_T;
    if (!strcmp(name, "n_contrib")) return im.n_contrib;
    if (!strcmp(name, "ranges")) return im.ranges;
    if (!strcmp(name, "seg_first")) return im.seg_first;
    if (!strcmp(name, "n_total")) return im.n_total;
    return (size_t)-1;
}

int64_t hgs_rasterize_forward(const hgs_forward_args* args, hgs_alloc_fn alloc, void* alloc_ctx,
                              hgs_forward_state* state, void* stream)
{
    if (!args || !alloc || !state) return fail(HGS_ERR_INVALID_ARGUMENT, "null argument");
    StatScope stat(g_stat_fwd_calls, g_stat_fwd_ns);
    FrameSwitches frame_switches;   // one snapshot of the A/B switches for everything this frame launches
    const hgs_forward_args& a = *args;
    hipStream_t st = (hipStream_t)stream;
    Camera cam;
    if (int rc = make_camera(a, cam)) return rc;
    memset(state, 0, sizeof *state);
    if (a.P == 0) return 0;  // nothing is launched: out_color keeps the caller's zeros (no background)
    if (!a.out_color || !a.radii) return fail(HGS_ERR_INVALID_ARGUMENT, "out_color and radii are required");
    const bool dbg = a.s.debug != 0;
    // HGS_BWD_SEGMENTED=0: never leave checkpoints (backward then runs one wave per quad on sparse frames; A/B measurements)
    bool want_ckpt = a.backward_checkpoints != 0 && switches().bwd_segmented;

    const int num_tiles = cam.gx * cam.gy;
    const int Ptot = a.P + a.seg2.P;  // Gaussian indices run over both segments
    int dev = 0;
    (void)hipGetDevice(&dev);
    const HistKey hkey{dev, st, Ptot, cam.H, cam.W};
    FrameHistory hist = history_get(hkey);   // (refreshed below when this frame's own counts arrive before it is enqueued)
    // A DENSE frame uses the checkpoint buffer its caller offers only when the shape's last frame had lists beyond
    // SORT_CAP_SMALL entries: the depth-segmented backward pays on a dense frame where such lists make the one-wave-per-tile
    // kernel chain-bound, and costs (C4: +78 us) where the deepest lists are merely long.
    if (hist.sparse == 0 && hist.n_deep == 0) want_ckpt = false;
    // ... nor a sparse frame whose shape's last frame left none (the scan's CKPT_KIND_NONE: many tiles, deep lists, no tail); should this
    // frame want them after all, its backward runs one wave per quad without them -- another summation order, the same gradients
    if (hist.no_ckpt == 1) want_ckpt = false;
    GeomLayout gl(Ptot, num_tiles);
    ImageLayout il(cam.H, cam.W);
    // caller-provided scratch when it suffices, else the allocation callback
    auto obtain = [&](int which, size_t bytes) -> char* {
        if (a.scratch[which] && a.scratch_bytes[which] >= bytes) return (char*)a.scratch[which];
        return (char*)alloc(alloc_ctx, which, bytes);
    };
    if (a.defer_n && a.binning_capacity_hint <= 0) return fail(HGS_ERR_INVALID_ARGUMENT, "defer_n needs a binning_capacity_hint");
    char* geom = obtain(HGS_BUF_GEOM, gl.total);
    char* image = obtain(HGS_BUF_IMAGE, il.total);
    if (!geom || !image) return fail(HGS_ERR_ALLOC, "scratch allocation failed (geom %zu B, image %zu B)", gl.total, il.total);
    state->geom = geom, state->geom_bytes = gl.total;
    state->image = image, state->image_bytes = il.total;

    Splat* splats = (Splat*)(geom + gl.splats);
    uint32_t* tiles_touched = (uint32_t*)(geom + gl.tiles_touched);
    // the preprocess kernel counts pairs per tile itself when the tile array fits LDS (any frame up to ~7 Mpixel)
    const int group = bin_group_for(Ptot, num_tiles);
    const int num_cells = num_cells_of(cam.gx, cam.gy);
    uint32_t* run_start = group ? (uint32_t*)(geom + gl.run_start) : nullptr;
    uint2* cell_slot = (uint2*)(geom + gl.cell_slot);
    uint32_t* order = (uint32_t*)(geom + gl.order);
    uint4* windows = (uint4*)(geom + gl.windows);
    uint2* ranges = (uint2*)(image + il.ranges);
    uint32_t* cursor = (uint32_t*)(image + il.cursor);
    uint32_t* n_total = (uint32_t*)(image + il.n_total);
    uint32_t* large_tiles = (uint32_t*)(image + il.large_tiles);
    uint32_t* tile_count = nullptr;
    TileCounterLease tc_lease;  // held until the scan is enqueued (or this call gives up)
    // (the per-tile counters, then -- from the next multiple of eight -- the per-cell counters of the counting sort)
    const size_t cell_counters_at = ((size_t)num_tiles + 7) / 8 * 8;
    // (+ the arrival counter of emit_scan_kernel behind them, zero between frames like the rest)
    const size_t arrival_at = cell_counters_at + (size_t)num_cells + 1;   // (num_cells + 1 cell counters: the last is the big splats')
    const int big_per_group = switches().big_per_group;
    if (int rc = acquire_tile_counters(st, arrival_at + 1, &tile_count, &tc_lease)) return rc;
    int bin_mode = bin_mode_for(Ptot, num_tiles, num_cells, group, hist.n_nonempty);
    // HGS_BIN_MODE=cell / order: force one of the two LDS binning paths (tests run the small parity scenes through both)
    if (const int forced = group ? switches().bin_mode : 0)
        bin_mode = (forced == 'c' && num_cells <= BIN_MAX_CELLS) ? BIN_BY_CELL : forced == 'o' ? BIN_IN_ORDER : bin_mode;
    uint32_t* cell_count = bin_mode == BIN_BY_CELL ? tile_count + cell_counters_at : nullptr;

    { ProfScope ps(HGS_STAGE_PREPROCESS, st);
      launch_preprocess(a, cam, splats, tiles_touched, bin_mode, bin_mode == BIN_BY_CELL ? cell_count : tile_count, cell_slot, run_start,
                        group, big_per_group, st); }
    STAGE_CHECK(dbg, st, "preprocess");
    // Binning capacity: exact (after waiting for N) or the caller's guess (frame enqueued before N is known).
    const int64_t hint = a.binning_capacity_hint > 0 ? a.binning_capacity_hint : 0;
    const uint32_t cap32 = hint > 0 ? (uint32_t)(hint > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : hint) : 0xFFFFFFFFu;
    const HostSlot slot = host_slot();  // tile_scan publishes N to the host through it
    if (!slot.word) return fail(HGS_ERR_HIP, "pinned host buffer allocation failed");
    // A frame of few tiles that is enqueued before N is known (a capacity hint) has no scan kernel: emit's workgroups scan the tile
    // counts themselves and one extra workgroup of that launch does the scan's bookkeeping (binning.hip, emit_scan_kernel).
    // HGS_EMIT_SCAN=0: always the stand-alone scan kernel (A/B measurements, the equivalence test).
    bool scan_pending = switches().emit_scan && hint > 0 && emit_scan_applies(bin_mode, num_tiles, group, Ptot);
    uint32_t* const seg_first_arg = want_ckpt ? (uint32_t*)(image + il.seg_first) : nullptr;
    // checkpoint slots an optimistically enqueued frame's buffer is laid out for: the caller's guess, else the most a frame that
    // fits the binning guess can need; the scan closes the gate on a frame that needs more.  (No binning guess: the buffer is
    // sized after N and the frame's kind are known.)
    const size_t ckpt_slots_guess = !want_ckpt || hint <= 0 ? 0 : a.ckpt_slots_hint > 0 ? (size_t)a.ckpt_slots_hint : CkptLayout::slots_for(hint, num_tiles);
    const uint32_t ckpt_cap32 = (uint32_t)std::min<size_t>(ckpt_slots_guess, 0xFFFFFFFFu);
    if (scan_pending && bin_mode == BIN_BY_CELL) {   // (the cell scatter and the per-group counts stay kernels of their own)
        { ProfScope ps(HGS_STAGE_SCAN, st);
          launch_spatial_groups(Ptot, cam, splats, cell_count, cell_slot, order, windows, tile_count, run_start, group, big_per_group, st); }
        STAGE_CHECK(dbg, st, "cell_scatter + group_count");
    }
    if (!scan_pending) {
        { ProfScope ps(HGS_STAGE_SCAN, st);
          if (bin_mode == BIN_BY_CELL) launch_spatial_groups(Ptot, cam, splats, cell_count, cell_slot, order, windows, tile_count, run_start, group, big_per_group, st);
          else if (bin_mode == BIN_NONE) launch_count(Ptot, cam, splats, tile_count, st);
          launch_tile_scan(tile_count, num_tiles, cell_count, num_cells + 1, ranges, cursor, n_total, large_tiles,
                           seg_first_arg, cap32, (unsigned long long*)slot.word, slot.ticket, ckpt_cap32, st); }
        STAGE_CHECK(dbg, st, "tile_scan");
        // the scan, which re-zeroes the counters, is enqueued: the next frame on this stream may have them
        tc_lease.scan_enqueued = true;
        tc_lease.release();
    }

    uint32_t* act_count = (uint32_t*)(image + il.act_count);
    const uint32_t* gate = n_total + 1;
    // the long-tile sort is launched unless the caller expects no long tile (its previous frame of this shape had none)
    const bool guess_no_long = a.expect_no_long_tiles != 0;
    // HGS_FUSED_SORT_BLEND=0: separate tile-sort and forward-blend kernels (A/B measurements); default: fused
    const bool fused = switches().fused_sort_blend;
    float* final_T = (float*)(image + il.final_T);
    uint32_t* n_contrib = (uint32_t*)(image + il.n_contrib);
    FusedBlend fb{cam, (uint32_t)(Ptot - 1), splats, a.s.bg, a.out_color, final_T, n_contrib, a.clamp_output != 0 ? 1 : 0, Ckpt{}};
    // checkpoints for the depth-segmented backward: laid out for the same capacity as the binning buffer
    bool known_dense = false;  // (set once N and the frame's flags are known: a dense frame needs no checkpoint buffer)
    size_t ckpt_slots_exact = 0;   // (set for the re-run of a frame whose needs are known)
    auto obtain_ckpt = [&](int64_t capacity) -> int {
        if (!want_ckpt || known_dense) return HGS_OK;
        CkptLayout cl(ckpt_slots_exact ? ckpt_slots_exact : (capacity == hint && ckpt_slots_guess) ? ckpt_slots_guess : CkptLayout::slots_for(capacity, num_tiles));
        char* ck = obtain(HGS_BUF_CKPT, cl.total);
        if (!ck) return fail(HGS_ERR_ALLOC, "scratch allocation failed (checkpoints %zu B)", cl.total);
        state->ckpt = ck, state->ckpt_bytes = cl.total, state->ckpt_slots = (int64_t)cl.slots;
        fb.ck = Ckpt{(float4*)(ck + cl.state), (uint32_t*)(ck + cl.slot_tile), (const uint32_t*)(image + il.seg_first),
                     (uint32_t*)(image + il.quad_nproc), n_total + 3};
        return HGS_OK;
    };
    // repair of a wrong "no long tiles" guess: the long tiles' own sort kernel, then the forward blend over the device-built
    // list of them (normally the long-tile sort runs in front of the small-tile kernel, which then blends every tile)
    auto enqueue_long_tiles = [&](const BinningLayout& bl, char* bin) -> int {
        uint64_t* act = (uint64_t*)(bin + bl.act) + ACT_PAD;
        { ProfScope ps(HGS_STAGE_SORT, st);
          launch_tile_sort(ranges, num_tiles, (uint64_t*)(bin + bl.keys), (uint64_t*)(bin + bl.list), (uint64_t*)(bin + bl.scratch), act,
                           bl.act_stride, act_count, large_tiles, n_total, bin + bl.parts, (uint32_t)bl.max_parts, false, true, nullptr, FrameHistory{}, st); }
        STAGE_CHECK(dbg, st, "tile_sort (long tiles)");
        { ProfScope ps(HGS_STAGE_BLEND_FORWARD, st);
          launch_blend_forward(cam, Ptot, ranges, act, bl.act_stride, act_count, splats, a.s.bg, a.out_color, final_T, n_contrib, n_total,
                               a.clamp_output != 0, large_tiles, false, true, fb.ck, st); }
        STAGE_CHECK(dbg, st, "blend_forward (long tiles)");
        return HGS_OK;
    };
    // enqueue emit -> sort -> blend for a binning buffer laid out for `capacity` entries
    auto enqueue_frame = [&](int64_t capacity, bool with_long_tiles) -> int {
        BinningLayout bl(capacity);
        char* bin = obtain(HGS_BUF_BINNING, bl.total);
        if (!bin) return fail(HGS_ERR_ALLOC, "scratch allocation failed (binning %zu B)", bl.total);
        state->binning = bin, state->binning_bytes = bl.total, state->binning_capacity = capacity;
        if (int rc = obtain_ckpt(capacity)) return rc;
        uint64_t* keys = (uint64_t*)(bin + bl.keys);
        uint64_t* list = (uint64_t*)(bin + bl.list);
        uint64_t* act = (uint64_t*)(bin + bl.act) + ACT_PAD;
        if (scan_pending) {   // (the first, optimistic enqueue of such a frame; a re-run after an overflow finds the scan's results in place)
            { ProfScope ps(HGS_STAGE_EMIT_KEYS, st);
              launch_emit_scan(Ptot, cam, splats, run_start, bin_mode == BIN_BY_CELL ? order : nullptr, windows, group, big_per_group, keys, tile_count,
                               cell_count, cell_count ? num_cells + 1 : 0, ranges, cursor, n_total, large_tiles, seg_first_arg, cap32,
                               (unsigned long long*)slot.word, slot.ticket, ckpt_cap32, tile_count + arrival_at, st); }
            STAGE_CHECK(dbg, st, "emit + tile_scan");
            scan_pending = false;
            tc_lease.scan_enqueued = true;   // (its last workgroup re-zeroes the counters)
            tc_lease.release();
        } else {
            { ProfScope ps(HGS_STAGE_EMIT_KEYS, st); launch_emit(Ptot, cam, splats, cursor, run_start, bin_mode == BIN_BY_CELL ? order : nullptr, windows, group, big_per_group, keys, gate, st); }
            STAGE_CHECK(dbg, st, "emit");
        }
        { ProfScope ps(HGS_STAGE_SORT, st);
          launch_tile_sort(ranges, num_tiles, keys, list, (uint64_t*)(bin + bl.scratch), act, bl.act_stride, act_count, large_tiles, n_total,
                           bin + bl.parts, (uint32_t)bl.max_parts, true, with_long_tiles, fused ? &fb : nullptr, hist, st); }
        STAGE_CHECK(dbg, st, fused ? "tile_sort + blend_forward" : "tile_sort");
        if (!fused) {
            // (when the long-tile sort was skipped, long tiles read as empty here: they are blended by the repair)
            { ProfScope ps(HGS_STAGE_BLEND_FORWARD, st);
              launch_blend_forward(cam, Ptot, ranges, act, bl.act_stride, act_count, splats, a.s.bg, a.out_color, final_T, n_contrib, n_total,
                                   a.clamp_output != 0, large_tiles, true, with_long_tiles, fb.ck, st); }
            STAGE_CHECK(dbg, st, "blend_forward");
        }
        return HGS_OK;
    };

    bool long_sort_done = false, enqueued = false;
    state->n_token = ((uint64_t)slot.index << 32) | slot.ticket;
    if (a.defer_n) {
        // deferred frame: never waits; nothing can be repaired later, so the long-tile sort is always part of it
        if (int rc = enqueue_frame(hint, true)) return rc;
        {
            std::lock_guard<std::mutex> lk(g_slot_owner_mu);
            g_slot_owner[slot.index] = SlotOwner{slot.ticket, dev, (void*)st, Ptot, cam.H, cam.W};
        }
        state->num_rendered = -1;
        return 0;
    }
    if (hint > 0) {
        if (int rc = enqueue_frame(hint, !guess_no_long)) return rc;  // optimistic: the GPU runs on while the host waits for N below
        long_sort_done = !guess_no_long, enqueued = true;
    }
    uint32_t n32 = 0;
    bool sparse = false, has_long = false;
    if (a.before_wait) a.before_wait(a.before_wait_ctx);   // (the caller's N-independent host work, under the GPU's way to N)
    uint64_t waited_ns = 0;
    if (int rc = wait_for_slot(slot, st, &n32, &sparse, &has_long, enqueued ? hist.wait_ns : 0, &waited_ns)) return rc;
    history_put(hkey, slot, enqueued ? std::max<uint64_t>(waited_ns, 1) : 0);   // (only the waits of frames that were enqueued ahead are of the kind the record predicts)
    hist = history_get(hkey);
    if (n32 == 0xFFFFFFFFu) return too_many_pairs();  // (every kernel behind the scan was gated off)
    const int64_t N = (int64_t)n32;
    state->num_rendered = N;
    state->sparse_frame = sparse ? 1 : 0;
    state->has_long_tiles = has_long ? 1 : 0;
    // checkpoint slots the frame needs (the high half of word 3 of its result slot, written by the scan)
    const size_t ckpt_needed = !want_ckpt ? 0 : (size_t)(slot.word[3] >> 32);
    state->ckpt_slots_used = (int64_t)ckpt_needed;
    if (((slot.word[4] >> 33) & 1ull) != 0ull) {
        // The scan decided that this SPARSE frame leaves no checkpoints (CKPT_KIND_NONE): whatever buffer it was enqueued with stays
        // unused -- its slot table is not even written --, a frame that is enqueued only now gets none, and the backward must not see one.
        state->ckpt = nullptr, state->ckpt_bytes = 0, state->ckpt_slots = 0, fb.ck = Ckpt{};
        want_ckpt = false;
        state->ckpt_slots_used = -1;   // (tells the binding not to offer this shape a buffer)
    }
    const bool ckpt_overflow = enqueued && state->ckpt && ckpt_needed > (size_t)state->ckpt_slots;   // (the scan closed the gate)
    if (!enqueued || N > hint || ckpt_overflow) {
        ckpt_slots_exact = ckpt_needed;
        // exact size known now (and whether there are long tiles); after a too-small guess the gated kernels above did
        // nothing, so the frame is simply enqueued again
        if (enqueued) HIP_TRY(hipMemsetAsync(n_total + 1, 0, sizeof(uint32_t), st));
        if (enqueued) (N > hint ? g_stat_binning_reruns : g_stat_ckpt_reruns).fetch_add(1, std::memory_order_relaxed);
        known_dense = !sparse && hist.n_deep == 0;  // (a dense frame WITH lists beyond SORT_CAP_SMALL entries keeps its checkpoints: its deep tiles use them)
        if (known_dense) state->ckpt = nullptr, state->ckpt_bytes = 0, fb.ck = Ckpt{};
        if (int rc = enqueue_frame(N, has_long)) return rc;
    } else if (has_long && !long_sort_done) {
        // guessed "no long tiles" wrongly: their lists read as empty so far -- sort and blend them now
        if (int rc = enqueue_long_tiles(BinningLayout(state->binning_capacity), (char*)state->binning)) return rc;
    }
    return N;
}

int64_t hgs_forward_poll(hgs_forward_state* state, int32_t block, void* stream)
{
    if (!state) return fail(HGS_ERR_INVALID_ARGUMENT, "null argument");
    if (state->num_rendered >= 0) return state->num_rendered;
    const uint32_t index = (uint32_t)(state->n_token >> 32), ticket = (uint32_t)state->n_token;
    if (!g_slot_base || index >= SLOT_RING || ticket == 0) return fail(HGS_ERR_INVALID_ARGUMENT, "state does not belong to a deferred frame");
    const HostSlot hs{g_slot_base + 8 * index, ticket, index};
    uint32_t n32 = 0;
    bool sparse = false, has_long = false;
    const int ready = slot_ready(hs, &n32, &sparse, &has_long);
    if (ready < 0) return expired();
    if (ready == 0) {
        if (!block) return HGS_PENDING;
        if (int rc = wait_for_slot(hs, (hipStream_t)stream, &n32, &sparse, &has_long)) return rc;
    }
    {   // the frame's counts feed its shape's launch-size record, as a waiting frame's do
        SlotOwner o;
        {
            std::lock_guard<std::mutex> lk(g_slot_owner_mu);
            o = g_slot_owner[index];
        }
        if (o.ticket == ticket && slot_state(*hs.word, ticket) == 1) history_put(HistKey{o.dev, (hipStream_t)o.st, o.P, o.H, o.W}, hs);
    }
    if (n32 == 0xFFFFFFFFu) return too_many_pairs();
    state->sparse_frame = sparse ? 1 : 0, state->has_long_tiles = has_long ? 1 : 0;
    if ((int64_t)n32 > state->binning_capacity)
        return fail(HGS_ERR_OVERFLOW, "deferred frame needed %u binning entries but was given %lld: its output is invalid, run it again",
                    n32, (long long)state->binning_capacity);
    if (state->ckpt && ((hs.word[4] >> 33) & 1ull) != 0ull && slot_state(*hs.word, ticket) == 1)
        state->ckpt = nullptr, state->ckpt_bytes = 0, state->ckpt_slots = 0, state->ckpt_slots_used = -1;   // (a sparse frame that left no checkpoints)
    if (state->ckpt) {
        const unsigned long long w3 = hs.word[3];
        if (slot_state(*hs.word, ticket) != 1) return expired();   // (the slot was handed on between the two reads)
        const int64_t needed = (int64_t)(w3 >> 32);   // (the exact figure the scan gated the frame with)
        state->ckpt_slots_used = needed;
        if (needed > state->ckpt_slots)
            return fail(HGS_ERR_OVERFLOW, "deferred frame needed %lld checkpoint slots but was given %lld: its output is invalid, run it again",
                        (long long)needed, (long long)state->ckpt_slots);
    }
    state->num_rendered = (int64_t)n32;
    return state->num_rendered;
}

int32_t hgs_rasterize_backward(const hgs_backward_args* args, void* stream)
{
    if (!args) return fail(HGS_ERR_INVALID_ARGUMENT, "null argument");
    StatScope stat(g_stat_bwd_calls, g_stat_bwd_ns);
    FrameSwitches frame_switches;
    const hgs_backward_args& a = *args;
    const hgs_forward_args& f = a.fwd;
    hipStream_t st = (hipStream_t)stream;
    Camera cam;
    if (int rc = make_camera(f, cam)) return rc;
    if (a.flags & HGS_BWD_UPSTREAM_SCALE_GRAD) cam.scale_grad_factor = 1.0f;
    if (f.P == 0) return HGS_OK;
    if (!a.state.geom || !a.state.image || !a.state.binning)
        return fail(HGS_ERR_INVALID_ARGUMENT, "forward state is missing");
    if (a.state.num_rendered < 0)
        return fail(HGS_ERR_INVALID_ARGUMENT, "forward state belongs to a deferred frame that hgs_forward_poll has not resolved");
    if (!a.dL_dout_color || !a.dL_dmeans2D || !a.grad_accum || !a.dL_dopacity || !a.dL_dcolors || !a.dL_dmeans3D ||
        !a.dL_dcov3D || !a.dL_dscales || !a.dL_drotations || (f.shs && !a.dL_dsh))
        return fail(HGS_ERR_INVALID_ARGUMENT, "gradient buffers are required");
    // the other render's gradients (add_*): all or none, checked BEFORE anything is launched -- a rejected call leaves the accumulator as it was
    if (a.add_dL_dopacity || a.add_dL_dcolors || a.add_dL_dmeans3D || a.add_dL_dcov3D || a.add_dL_dsh || a.add_dL_dscales || a.add_dL_drotations) {
        if (!a.add_dL_dopacity || !a.add_dL_dcolors || !a.add_dL_dmeans3D || !a.add_dL_dcov3D || !a.add_dL_dscales || !a.add_dL_drotations ||
            (f.shs && !a.add_dL_dsh))
            return fail(HGS_ERR_INVALID_ARGUMENT, "add_*: all of the other render's gradient buffers are required");
        if ((((uintptr_t)a.add_dL_drotations) & 15u) != 0) return fail(HGS_ERR_INVALID_ARGUMENT, "add_dL_drotations must be 16-byte aligned");
        // ... and none of them may BE this call's output of the same kind (the kernel reads the one while it writes the other;
        // equal base pointers are what a caller gets wrong -- partial overlaps of distinct allocations cannot be told from here)
        if (a.add_dL_dopacity == a.dL_dopacity || a.add_dL_dcolors == a.dL_dcolors || a.add_dL_dmeans3D == a.dL_dmeans3D ||
            a.add_dL_dcov3D == a.dL_dcov3D || (a.add_dL_dsh && a.add_dL_dsh == a.dL_dsh) || a.add_dL_dscales == a.dL_dscales ||
            a.add_dL_drotations == a.dL_drotations)
            return fail(HGS_ERR_INVALID_ARGUMENT, "add_*: the other render's gradient buffers may not alias this call's outputs");
    }
    const bool dbg = f.s.debug != 0;
    const int Ptot = f.P + f.seg2.P;
    if (f.seg2.P > 0 && (!a.seg2_dL_dopacity || !a.seg2_dL_dcolors || !a.seg2_dL_dmeans3D || !a.seg2_dL_dcov3D || !a.seg2_dL_dscales ||
                         !a.seg2_dL_drotations || (f.seg2.shs && !a.seg2_dL_dsh)))
        return fail(HGS_ERR_INVALID_ARGUMENT, "gradient buffers of the second segment are required");
    GeomLayout gl(Ptot, cam.gx * cam.gy);
    ImageLayout il(cam.H, cam.W);
    BinningLayout bl(a.state.binning_capacity > 0 ? a.state.binning_capacity : a.state.num_rendered);
    if (a.state.geom_bytes < gl.total || a.state.image_bytes < il.total || a.state.binning_bytes < bl.total)
        return fail(HGS_ERR_INVALID_ARGUMENT, "forward state has the wrong size");
    Ckpt ck{};
    if (a.state.ckpt) {
        CkptLayout cl(a.state.ckpt_slots > 0 ? (size_t)a.state.ckpt_slots
                                             : CkptLayout::slots_for(a.state.binning_capacity > 0 ? a.state.binning_capacity : a.state.num_rendered, cam.gx * cam.gy));
        if (a.state.ckpt_bytes < cl.total) return fail(HGS_ERR_INVALID_ARGUMENT, "forward state has the wrong size (checkpoints)");
        char* c = (char*)a.state.ckpt;
        ck = Ckpt{(float4*)(c + cl.state), (uint32_t*)(c + cl.slot_tile), (const uint32_t*)((const char*)a.state.image + il.seg_first),
                  (uint32_t*)((char*)a.state.image + il.quad_nproc), nullptr};
    }
    const char* geom = (const char*)a.state.geom;
    const char* image = (const char*)a.state.image;
    const char* bin = (const char*)a.state.binning;
    const Splat* splats = (const Splat*)(geom + gl.splats);

    // A dense frame with checkpoints: how many slots its deep tiles use is word 3 of the frame's pinned result slot (tile_scan_kernel),
    // still there unless the ring went round since (then the backward covers the layout's upper bound and its surplus workgroups leave)
    int64_t dense_slots = -1;
    if (ck.state && a.state.sparse_frame == 0 && g_slot_base) {
        const uint32_t index = (uint32_t)(a.state.n_token >> 32), ticket = (uint32_t)a.state.n_token;
        if (index < SLOT_RING && ticket != 0) {
            const volatile unsigned long long* w = g_slot_base + 8 * index;
            const unsigned long long before = w[0], v3 = w[3] & 0xFFFFFFFFull, after = w[0];   // (the slot may be handed to a later frame any time)
            if (slot_state(before, ticket) == 1 && slot_state(after, ticket) == 1 && v3 != 0xFFFFFFFFull) dense_slots = (int64_t)v3;
        }
    }
    { ProfScope ps(HGS_STAGE_BLEND_BACKWARD, st);
      launch_blend_backward(cam, Ptot, (const uint2*)(image + il.ranges), (const uint64_t*)(bin + bl.act) + ACT_PAD,
                            bl.act_stride, (const uint32_t*)(image + il.act_count), a.state.sparse_frame != 0, splats,
                            f.s.bg, (const float*)(image + il.final_T), (const uint32_t*)(image + il.n_contrib), a.dL_dout_color,
                            a.grad_accum, ck, a.state.num_rendered, dense_slots, st); }
    STAGE_CHECK(dbg, st, "blend_backward");
    if (a.wait_before_per_gaussian) HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)a.wait_before_per_gaussian, 0));
    { ProfScope ps(HGS_STAGE_PREPROCESS_BACKWARD, st); launch_preprocess_backward(a, cam, splats, st); }
    STAGE_CHECK(dbg, st, "preprocess_backward");
    return HGS_OK;
}

// Measurement aid (bench.py's `roofline.peak_measured`): a float4 copy, one element per thread -- the kernel shape the
// microarchitecture guide measures the practical HBM ceiling with (6.29 TB/s there; 6.23-6.26 on this pool's boxes,
// tools/microbench/copy_bw.hip: a grid-stride loop over the same data reaches only 4.5-5.7 TB/s, whatever its grid and
// unrolling -- every thread then walks its own far-apart stream).  bytes must be a multiple of 16; 16-byte aligned pointers.
__global__ void __launch_bounds__(256) copy_bandwidth_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int32_t hgs_copy_bandwidth(void* dst, const void* src, size_t bytes, void* stream)
{
    if (!dst || !src || (bytes & 15u) || (((uintptr_t)dst | (uintptr_t)src) & 15u)) return fail(HGS_ERR_INVALID_ARGUMENT, "bad arguments");
    const size_t n = bytes / 16;
    if (n == 0) return HGS_OK;
    if ((n + 255) / 256 > 0x7FFFFFFFull) return fail(HGS_ERR_INVALID_ARGUMENT, "too large for one launch");
    hipLaunchKernelGGL(copy_bandwidth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)dst,
                       (const float4*)src, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HGS_ERR_HIP, "copy_bandwidth: %s", hipGetErrorString(e));
    return HGS_OK;
}

int32_t hgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* present, void* stream)
{
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(HGS_ERR_INVALID_ARGUMENT, "bad arguments");
    if (P == 0) return HGS_OK;
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HGS_ERR_HIP, "mark_visible: %s", hipGetErrorString(e));
    return HGS_OK;
}

}  // extern "C"
