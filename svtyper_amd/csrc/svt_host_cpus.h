// svt_host_cpus.h -- how many host threads are worth starting.
//
// hardware_concurrency() counts the machine; a containerised process may only schedule on its
// affinity mask and only for its cgroup CPU quota (cpu.max "quota period"): a pool of 256 threads
// on a 16-CPU quota spends its time throttled.  Used by the host tiling and the native BAM reader.
#pragma once

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace svt {

// cgroup CPU quota as {seconds of CPU time, seconds of wall time} per accounting period; {0, 0} = no quota
struct CpuQuota { double cpu_s, period_s; };
inline CpuQuota cpu_quota()
{
    static const CpuQuota cached = [] {
        long long quota = -1, period = 0;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2
            char q[64] = {0};
            if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
            std::fclose(f);
        } else {                                                              // cgroup v1
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
                std::fclose(g);
            }
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%lld", &period) != 1) period = 0;
                std::fclose(g);
            }
        }
        if (quota > 0 && period > 0) return CpuQuota{(double)quota * 1e-6, (double)period * 1e-6};
        return CpuQuota{0.0, 0.0};
    }();
    return cached;
}

inline unsigned usable_cpus()
{
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) n = n ? std::min<unsigned>(n, (unsigned)c) : (unsigned)c;
        }
        if (n == 0) n = 1;
        const CpuQuota q = cpu_quota();
        if (q.period_s > 0.0) {
            const unsigned by_quota = (unsigned)std::max(1.0, std::ceil(q.cpu_s / q.period_s - 1e-9));
            n = std::min(n, by_quota);
        }
        return n;
    }();
    return cached;
}

// distinct physical cores among the CPUs this process may run on (SMT siblings counted once); 0 = unknown
inline unsigned physical_cores()
{
    static const unsigned cached = [] {
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) != 0) return 0u;
        std::vector<std::pair<long, long>> seen;
        for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
            if (!CPU_ISSET(cpu, &set)) continue;
            long ids[2] = {-1, -1};
            const char* what[2] = {"physical_package_id", "core_id"};
            for (int k = 0; k < 2; ++k) {
                char path[128];
                std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/%s", cpu, what[k]);
                if (FILE* f = std::fopen(path, "r")) {
                    if (std::fscanf(f, "%ld", &ids[k]) != 1) ids[k] = -1;
                    std::fclose(f);
                }
            }
            if (ids[1] < 0) return 0u;
            const std::pair<long, long> id(ids[0], ids[1]);
            if (std::find(seen.begin(), seen.end(), id) == seen.end()) seen.push_back(id);
        }
        return (unsigned)seen.size();
    }();
    return cached;
}

// CPU seconds this process's host stages (the reader's calls) spent recently: a cgroup quota is CPU time per accounting period,
// so a burst is only a burst while the calls before it have left something of the current period's allowance.  note_cpu_s()
// after a call; recent_cpu_s(w) = what was noted within the last w seconds.
struct CpuLedger {
    std::mutex lock;
    std::vector<std::pair<double, double>> spent;      // (when, CPU seconds)
    static CpuLedger& get() { static CpuLedger l; return l; }
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
inline void note_cpu_s(double cpu_s)
{
    CpuLedger& l = CpuLedger::get();
    std::lock_guard<std::mutex> g(l.lock);
    const double t = CpuLedger::now_s();
    size_t keep = 0;
    for (size_t k = 0; k < l.spent.size(); ++k)
        if (t - l.spent[k].first < 2.0) l.spent[keep++] = l.spent[k];
    l.spent.resize(keep);
    l.spent.emplace_back(t, cpu_s);
}
inline double recent_cpu_s(double window_s)
{
    CpuLedger& l = CpuLedger::get();
    std::lock_guard<std::mutex> g(l.lock);
    const double t = CpuLedger::now_s();
    double sum = 0.0;
    for (const auto& e : l.spent)
        if (t - e.first < window_s) sum += e.second;
    return sum;
}

// Threads for a burst of `est_cpu_s` seconds of CPU work.  A cgroup CPU quota is CPU time per accounting period (16 CPUs =
// 1.6 s per 100 ms), not a number of threads: a call whose whole work -- together with what the calls just before it spent
// (CpuLedger: chunk after chunk of one run, sample after sample of a joint run) -- fits well inside one period's allowance may
// run one worker per physical core (at most `cap`) and be done sooner; a longer one is bound by the quota whatever it starts
// and keeps to usable_cpus().  Without a quota usable_cpus() is the machine already.
inline unsigned burst_threads(double est_cpu_s, unsigned cap)
{
    const unsigned base = usable_cpus();
    const CpuQuota q = cpu_quota();
    if (q.period_s <= 0.0) return base;
    // (the calls of the last period and this one together within 80 % of a period's allowance: the rest is for the caller's
    // other threads -- the next block's parse, the text of the one before)
    const double allowance = 0.8 * q.cpu_s - recent_cpu_s(q.period_s);
    if (est_cpu_s > allowance) return base;
    // a run that has been spending more than three quarters of the quota over the last ten periods (sample after sample of a
    // joint run, block after block of a long VCF with a slow reader) is bound by the quota: bursts would only be paid for with
    // throttled periods.  (A run below that -- block after block of a VCF whose units are cheap -- gains from every burst.)
    if (recent_cpu_s(10.0 * q.period_s) + est_cpu_s > 7.5 * q.cpu_s) return base;
    return std::max(base, std::min(physical_cores(), cap));
}

}  // namespace svt
