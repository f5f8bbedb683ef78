// Chain streams of the factorization schedules: see chain_streams.hpp.
#include "chain_streams.hpp"

#include <algorithm>
#include <mutex>

namespace ssa {
namespace {

// Which high-priority stream a panel chain runs on matters: the runtime maps the streams of a process onto hardware
// queues and those onto the four pipes of the command processor.  A queue that shares its pipe with the queue of the
// trailing updates waits for the pipe while an update's workgroups are being dispatched - every dependent launch of
// the chain then costs 35-40 us instead of 2-4 us - and so do two chains that share a pipe with each other
// (tools/probes/pipe_probe.hip, profiles/r03_pipe_probe.txt: one high-priority stream in four is slow beside a given
// stream, in no fixed order; config H factorization 97-100 ms with both chains on good streams, 105-110 ms with one
// on a bad one).  Nothing in the HIP API tells which is which, so the device's chain streams are measured once
// against the stream of the schedule's trailing updates: a short chain of dependent one-workgroup launches on each candidate beside
// back-to-back chip-filling launches on that stream, then the good candidates against each other in pairs
// (which of them share a pipe); about 10 ms, once per device and caller stream.  The lanes take one stream of each
// pipe first.
constexpr int kCalibLinks = 8;
constexpr long long kCalibLinkTicks = 500;       // alone: 5-us links (at the 100 MHz of wall_clock64())
constexpr long long kCalibPairLinkTicks = 2000;  // pairs: 20-us links, so that two streams on ONE queue show as well
constexpr long long kCalibLoadTicks = 2000;      // 20 us per workgroup of the load
constexpr int kCalibLoadGrid = 2080, kCalibLoadLaunches = 7;   // per test: 7 x ceil(2080 / 512) x 20 us = 0.56 ms

// a timed spin; workgroups of the load leave at once after the tests are over (*stop != 0)
__global__ __launch_bounds__(256) void pipe_calibration_kernel(long long ticks, const int32_t *stop) {
    if (stop != nullptr && __atomic_load_n(stop, __ATOMIC_RELAXED) != 0) return;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__global__ void pipe_calibration_stop_kernel(int32_t *stop) { __atomic_store_n(stop, 1, __ATOMIC_RELAXED); }


// The chain streams of one device and what was measured for them, per load stream (a few are remembered: the
// Cholesky schedule's updates run on its caller's stream, the LU schedule's on a stream of its own).
constexpr int kMaxMeasurements = 8;
struct Measurement {
    bool valid = false;
    unsigned long long used = 0;  // tick of the last chain_streams_get that returned it (least recently used goes first)
    hipStream_t load = nullptr;
    float us[kChainPool] = {};    // per dependent launch, alone beside the load stream
    int pipe[kChainPool] = {};    // 0: shares the load stream's pipe; 1, 2, ...: groups of streams that share a pipe
    int order[kChainPool] = {};   // pool indices in order of use
};
struct ChainSet {
    hipStream_t chain_pool[kChainPool] = {};
    bool pool_made = false;
    Measurement measured[kMaxMeasurements];
    unsigned long long tick = 0;
    int last = -1;
    int32_t *calib_stop = nullptr;   // device flag of the calibration load
};
ChainSet g_chain_sets[kMaxDevices];
std::mutex g_chain_mutex;
// One measurement: the chain streams a (and b, if >= 0) each run kCalibLinks dependent 5-us launches, at the same
// time, beside the load on the load stream; us = cost per dependent launch, the worse of the two.
struct ChainTest {
    int a, b;
    float us;
};

// Runs the tests one after the other on the device (enqueued behind a gate and timed with events: the host's launch
// rate does not enter), then waits for them.
inline int run_chain_tests(ChainSet &set, hipStream_t st, ChainTest *tests, int n, long long link_ticks) {
    if (n <= 0) return SSA_OK;
    constexpr int kMaxTests = 2 * kChainPool;
    if (n > kMaxTests) return SSA_ERR_INVALID_ARGUMENT;
    hipEvent_t gate = nullptr, ev[kMaxTests][4] = {};
    hipStream_t gate_s = nullptr;
    int rc = SSA_OK;
    auto ok = [&rc](hipError_t e) {
        if (e != hipSuccess) rc = SSA_ERR_HIP;
        return e == hipSuccess;
    };
    ok(hipStreamCreateWithFlags(&gate_s, hipStreamNonBlocking)) && ok(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    for (int t = 0; t < n && rc == SSA_OK; ++t)
        for (int e = 0; e < 4 && rc == SSA_OK; ++e) ok(hipEventCreate(&ev[t][e]));
    if (rc == SSA_OK && set.calib_stop == nullptr) ok(hipMalloc(reinterpret_cast<void **>(&set.calib_stop), sizeof(int32_t)));
    if (rc == SSA_OK) {
        ok(hipMemsetAsync(set.calib_stop, 0, sizeof(int32_t), gate_s));
        hipLaunchKernelGGL(pipe_calibration_kernel, dim3(1), dim3(256), 0, gate_s, 20000LL + 2500LL * n, nullptr);   // the enqueue below
        ok(hipEventRecord(gate, gate_s));
        ok(hipStreamWaitEvent(st, gate, 0));
        for (int i = 0; i < n * kCalibLoadLaunches; ++i)
            hipLaunchKernelGGL(pipe_calibration_kernel, dim3(kCalibLoadGrid), dim3(256), 64 << 10, st, kCalibLoadTicks,
                               set.calib_stop);
        for (int t = 0; t < n && rc == SSA_OK; ++t) {
            const int who[2] = {tests[t].a, tests[t].b};
            for (int h = 0; h < 2; ++h) {
                if (who[h] < 0) continue;
                hipStream_t c = set.chain_pool[who[h]];
                if (t == 0) {
                    ok(hipStreamWaitEvent(c, gate, 0));
                } else {
                    ok(hipStreamWaitEvent(c, ev[t - 1][1], 0));
                    if (tests[t - 1].b >= 0) ok(hipStreamWaitEvent(c, ev[t - 1][3], 0));
                }
                ok(hipEventRecord(ev[t][2 * h], c));
            }
            for (int l = 0; l < kCalibLinks; ++l)
                for (int h = 0; h < 2; ++h)
                    if (who[h] >= 0)
                        hipLaunchKernelGGL(pipe_calibration_kernel, dim3(1), dim3(256), 0, set.chain_pool[who[h]], link_ticks, nullptr);
            for (int h = 0; h < 2; ++h)
                if (who[h] >= 0) ok(hipEventRecord(ev[t][2 * h + 1], set.chain_pool[who[h]]));
        }
        if (rc == SSA_OK) {   // the load is not needed any longer once the last test is over
            hipStream_t last = set.chain_pool[tests[n - 1].a];
            if (tests[n - 1].b >= 0) ok(hipStreamWaitEvent(last, ev[n - 1][3], 0));
            hipLaunchKernelGGL(pipe_calibration_stop_kernel, dim3(1), dim3(1), 0, last, set.calib_stop);
        }
        if (hipGetLastError() != hipSuccess) rc = SSA_ERR_HIP;
    }
    if (rc == SSA_OK) {
        ok(hipEventSynchronize(ev[n - 1][1]));
        if (tests[n - 1].b >= 0) ok(hipEventSynchronize(ev[n - 1][3]));
    }
    for (int t = 0; t < n && rc == SSA_OK; ++t) {
        float ms = 0.f, ms2 = 0.f;
        ok(hipEventElapsedTime(&ms, ev[t][0], ev[t][1]));
        if (tests[t].b >= 0 && ok(hipEventElapsedTime(&ms2, ev[t][2], ev[t][3]))) ms = std::max(ms, ms2);
        tests[t].us = ms * 1000.f / kCalibLinks - static_cast<float>(link_ticks) / 100.f;
    }
    for (int t = 0; t < n; ++t)
        for (hipEvent_t e : ev[t])
            if (e) (void)hipEventDestroy(e);
    if (gate) (void)hipEventDestroy(gate);
    if (gate_s) {
        (void)hipStreamSynchronize(gate_s);
        (void)hipStreamDestroy(gate_s);
    }
    return rc;
}

// Groups the chain streams of `set` by pipe (see above) and hands them to the lanes: one stream of every pipe that
// is not the caller's first, then the second of each, ...; the streams on the caller's pipe last.
inline int calibrate_chain_streams(ChainSet &set, hipStream_t st, Measurement &mm) {
    ChainTest tests[kChainPool];
    for (int j = 0; j < kChainPool; ++j) tests[j] = ChainTest{j, -1, 0.f};
    int rc = run_chain_tests(set, st, tests, kChainPool, kCalibLinkTicks);
    if (rc != SSA_OK) return rc;
    float best = 1e30f;
    for (int j = 0; j < kChainPool; ++j) {
        mm.us[j] = tests[j].us;
        best = std::min(best, tests[j].us);
    }
    // a stream is "slow" above a limit well clear of both populations (2-4 us and 35-40 us; pairs on distinct
    // pipes: 6-9 us)
    const float limit = 2.f * std::max(best, 0.f) + 12.f;
    int unassigned = 0;
    for (int j = 0; j < kChainPool; ++j) {
        mm.pipe[j] = (tests[j].us > limit) ? 0 : -1;
        unassigned += mm.pipe[j] < 0;
    }
    int pipes = 0;
    while (unassigned > 0 && pipes < 8) {
        // the first unassigned stream founds a group; the others join it if the pair is slow together
        int lead = 0;
        while (mm.pipe[lead] >= 0) ++lead;
        mm.pipe[lead] = ++pipes;
        --unassigned;
        int n = 0;
        for (int j = 0; j < kChainPool; ++j)
            if (mm.pipe[j] < 0) tests[n++] = ChainTest{lead, j, 0.f};
        rc = run_chain_tests(set, st, tests, n, kCalibPairLinkTicks);
        if (rc != SSA_OK) return rc;
        for (int t = 0; t < n; ++t)
            if (tests[t].us > limit) {
                mm.pipe[tests[t].b] = pipes;
                --unassigned;
            }
    }
    for (int j = 0; j < kChainPool; ++j)
        if (mm.pipe[j] < 0) mm.pipe[j] = pipes;   // (more than 8 groups: not a real device; keep going)
    // order of use: round r takes the cheapest unused stream of every pipe (cheapest pipe first); the streams on the
    // load stream's pipe come last
    int by_cost[kChainPool];
    for (int j = 0; j < kChainPool; ++j) by_cost[j] = j;
    std::stable_sort(by_cost, by_cost + kChainPool, [&mm](int a, int b) { return mm.us[a] < mm.us[b]; });
    int m = 0;
    bool used[kChainPool] = {};
    for (int round = 0; round < kChainPool && m < kChainPool; ++round) {
        bool pipe_taken[kChainPool + 1] = {};
        for (int q = 0; q < kChainPool; ++q) {
            const int j = by_cost[q], g = mm.pipe[j];
            if (used[j] || g == 0 || pipe_taken[g]) continue;
            used[j] = pipe_taken[g] = true;
            mm.order[m++] = j;
        }
    }
    for (int q = 0; q < kChainPool; ++q)
        if (!used[by_cost[q]]) mm.order[m++] = by_cost[q];
    mm.valid = true;
    mm.load = st;
    return SSA_OK;
}


}  // namespace

int chain_streams_get(hipStream_t load, int count, hipStream_t *out) {
    if (count < 0 || (count > 0 && out == nullptr)) return SSA_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    ChainSet &set = g_chain_sets[dev];   // streams belong to the device they were made on
    if (!set.pool_made) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return SSA_ERR_HIP;
        for (int j = 0; j < kChainPool; ++j)
            if (hipStreamCreateWithPriority(&set.chain_pool[j], hipStreamNonBlocking, hi) != hipSuccess) return SSA_ERR_HIP;
        set.pool_made = true;
    }
    int slot = -1;
    for (int m = 0; m < kMaxMeasurements; ++m)
        if (set.measured[m].valid && set.measured[m].load == load) slot = m;
    if (slot < 0) {
        // a free slot, else the least recently used one: a caller that alternates between a few streams keeps its
        // measurements
        slot = 0;
        for (int m = 0; m < kMaxMeasurements; ++m) {
            if (!set.measured[m].valid) {
                slot = m;
                break;
            }
            if (set.measured[m].used < set.measured[slot].used) slot = m;
        }
        set.measured[slot].valid = false;
        // The measurement wants the chain streams and the load stream idle: schedules only join them into their
        // caller's stream asynchronously, so an earlier factorization (of the other route, or on another stream) may
        // still be running.
        for (hipStream_t c : set.chain_pool)
            if (hipStreamSynchronize(c) != hipSuccess) return SSA_ERR_HIP;
        if (hipStreamSynchronize(load) != hipSuccess) return SSA_ERR_HIP;
        const int rc = calibrate_chain_streams(set, load, set.measured[slot]);
        if (rc != SSA_OK) return rc;
    }
    set.measured[slot].used = ++set.tick;
    set.last = slot;
    for (int i = 0; i < count; ++i) out[i] = set.chain_pool[set.measured[slot].order[i % kChainPool]];
    return SSA_OK;
}

// Forgets every measurement of the current device: the next schedule measures again (after the process has created
// or destroyed streams -- which can move hardware queues between the command processor's pipes -- or when a
// stream handle that was measured has been destroyed and its value may be handed out again).
int chain_streams_invalidate() {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    ChainSet &set = g_chain_sets[dev];
    for (Measurement &m : set.measured) m.valid = false;
    set.last = -1;
    return SSA_OK;
}

int chain_streams_costs(double *microseconds, int32_t *pipe_group, int capacity) {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    int dev = 0;
    if (current_device(&dev) != SSA_OK || capacity < 0) return 0;
    const ChainSet &set = g_chain_sets[dev];
    if (set.last < 0 || !set.measured[set.last].valid) return 0;
    const Measurement &mm = set.measured[set.last];
    for (int j = 0; j < kChainPool && j < capacity; ++j) {
        if (microseconds != nullptr) microseconds[j] = mm.us[mm.order[j]];
        if (pipe_group != nullptr) pipe_group[j] = mm.pipe[mm.order[j]];
    }
    return kChainPool;
}

// Destroys the chain streams of every device (ssa_shutdown), after waiting for what is on them.
int chain_streams_shutdown() {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    int rc = SSA_OK;
    for (int d = 0; d < kMaxDevices; ++d) {
        ChainSet &set = g_chain_sets[d];
        if (set.pool_made) {
            for (hipStream_t &c : set.chain_pool) {
                if (hipStreamSynchronize(c) != hipSuccess || hipStreamDestroy(c) != hipSuccess) rc = SSA_ERR_HIP;
                c = nullptr;
            }
            set.pool_made = false;
        }
        for (Measurement &m : set.measured) m.valid = false;
        set.last = -1;
        if (set.calib_stop != nullptr) {
            if (hipFree(set.calib_stop) != hipSuccess) rc = SSA_ERR_HIP;
            set.calib_stop = nullptr;
        }
    }
    return rc;
}

}  // namespace ssa
