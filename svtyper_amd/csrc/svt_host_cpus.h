// svt_host_cpus.h -- how many host threads are worth starting.
//
// hardware_concurrency() counts the machine; a containerised process may only schedule on its
// affinity mask and only for its cgroup CPU quota (cpu.max "quota period"): a pool of 256 threads
// on a 16-CPU quota spends its time throttled.  Used by the host tiling and the native BAM reader.
#pragma once

#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

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

}  // namespace svt
