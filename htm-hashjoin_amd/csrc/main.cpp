// main.cpp -- the `main` command line of the reference (main.cpp:21-128), kept
// flag for flag, in front of libhtmjoin_hip.so. Host C++ only: it sees the C ABI
// of include/htm_hashjoin.h and no HIP types.
//
//   ./main --algo atomic --rSize 134217728 --dataDistr uniform
//   ./main --algo prj    --rSize 1073741824 --dataDistr local_shuffle --shuffleRange 1024
//
// --algo values:
//   atomic | hip-atomic         open-addressing build+probe on the GPU (the CAS insert loop
//                               replaced by the index-priority kernels)
//   htm                         the reference's bucketised table (three tuples per 32-byte bucket,
//                               overflow chains) on the GPU, TSX replaced by the same protocol
//   prj | hip-prj | PRO         radix-partitioned join on the GPU
//   auto                        samples R for locality and runs one of the two above
//                               (the reference's adaptive idea, HTMHashBuild.hpp:98-154);
//                               "algoUsed" in the JSON line says which
//   nocc | cpu-atomic           the reference's own loops on host threads
//                               (NoCCHashBuild.hpp:37-81 / AtomicHashBuild.hpp:37-86):
//                               the plumbing / CPU-baseline path; never a fallback --
//                               GPU algos fail if there is no gfx950 device.
// Defaults are main.cpp:78-85. Unknown flag -> "Found Unknown Arg: X", exit 1
// (:64-65); unknown algo -> "Unknown Algo: X", exit 0 (:108).
// Output: one JSON object per run, the reference's fields first and in its order
// (NoCCHashBuild.hpp:127-146), extra fields appended.

#include "../../include/htm_hashjoin.h"
#include "../../include/htm_hashjoin_sharded.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

namespace {

struct param_t {  // main.cpp:21-41, plus extensions at the end
    std::string algo = "htm";
    uint64_t rSize = 1u << 28;
    uint32_t transactionSize = 16;
    uint32_t probeLength = 4;
    std::string dataDistr = "shuffle";
    uint32_t shuffleRange = 16;
    uint32_t scaleOutput = 2;
    uint32_t numPartitions = 64;
    // extensions
    int probe = 1;           // ENABLE_PROBE (config.h:4) as a run-time switch
    uint64_t sSize = 0;      // 0 = rSize (main.cpp:93)
    std::string sDistr;      // "" = reference behaviour (sorted, or copy of R for random)
    double zipfTheta = 0.9;
    uint32_t radixBits = 0;
    int device = 0;
    int repeat = 1;
    int gpus = 1;            // > 1: the radix-sharded join over devices 0 .. gpus-1 (libhtmjoin_sharded.so); 1 with --split: the same path on one GPU
    std::string split;       // "" (single-GPU operator), "low" (key & (G-1)), "high" (range split)
};

void parseArgs(int argc, char** argv, param_t* p)
{
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        const char* v = (i + 1 < argc) ? argv[i + 1] : "";
        if (strcmp(a, "--algo") == 0) p->algo = v;
        else if (strcmp(a, "--rSize") == 0) p->rSize = strtoull(v, nullptr, 10);
        else if (strcmp(a, "--transactionSize") == 0) p->transactionSize = atoi(v);
        else if (strcmp(a, "--probeLength") == 0) {
            // main.cpp:53-54 stores this into dataDistr, so the reference always runs
            // with probeLength 4; accepted and ignored here for script compatibility
        }
        else if (strcmp(a, "--dataDistr") == 0) p->dataDistr = v;
        else if (strcmp(a, "--shuffleRange") == 0) p->shuffleRange = atoi(v);
        else if (strcmp(a, "--scaleOutput") == 0) p->scaleOutput = atoi(v);
        else if (strcmp(a, "--numPartitions") == 0) p->numPartitions = atoi(v);
        else if (strcmp(a, "--probe") == 0) p->probe = atoi(v);
        else if (strcmp(a, "--sSize") == 0) p->sSize = strtoull(v, nullptr, 10);
        else if (strcmp(a, "--sDistr") == 0) p->sDistr = v;
        else if (strcmp(a, "--zipfTheta") == 0) p->zipfTheta = atof(v);
        else if (strcmp(a, "--radixBits") == 0) p->radixBits = atoi(v);
        else if (strcmp(a, "--device") == 0) p->device = atoi(v);
        else if (strcmp(a, "--repeat") == 0) p->repeat = atoi(v);
        else if (strcmp(a, "--gpus") == 0) p->gpus = atoi(v);
        else if (strcmp(a, "--split") == 0) p->split = v;
        else {
            std::cout << "Found Unknown Arg: " << a << std::endl;
            exit(1);
        }
        i++;
    }
}

double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// CPUs this process may really use: its affinity mask, capped by the cgroup's CPU quota. hardware_concurrency() reports
// the host's CPUs -- a GPU box hands a container 16 of 64+, and the sweep's CPU legs said "cpu_threads: 64" while running
// on 16 (round-2 VERDICT, weak #8). Same rule as bench.py's effective_cpus().
static unsigned effective_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return std::max(1u, n);
}

// ---- the reference's CPU loops on host threads ------------------------------
struct cpu_result { uint64_t conflicts = 0, matches = 0, inputSum = 0, outputSum = 0; double us = 0; int threads = 0; };

template <class F>
void for_chunks(uint32_t chunks, int nthreads, F f)
{
    // parallel_for(blocked_range(0, n, n/numPartitions)): <= numPartitions tasks
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&] { for (uint32_t c; (c = next.fetch_add(1)) < chunks;) f(c); });
    for (auto& x : th) x.join();
}

cpu_result cpu_build_probe(bool useCas, const uint64_t* R, uint64_t rSize, const uint64_t* S, uint64_t sSize,
                           uint32_t numPartitions, uint32_t probeLength)
{
    cpu_result res;
    const uint64_t tableSize = rSize * 2, mask = tableSize - 1;
    std::vector<std::atomic<uint64_t>> output(tableSize + 4);
    for (auto& s : output) s.store(0, std::memory_order_relaxed);
    std::vector<uint64_t> conf(numPartitions, 0), confSum(numPartitions, 0), match(numPartitions, 0);
    int nthreads = (int)std::min<unsigned>(numPartitions, effective_cpus());
    res.threads = nthreads;
    const double t0 = now_us();
    for_chunks(numPartitions, nthreads, [&](uint32_t c) {
        const uint64_t ps = rSize / numPartitions, b = c * ps, e = (c + 1 == numPartitions) ? rSize : b + ps;
        for (uint64_t i = b; i < e; ++i) {
            uint64_t cur = R[i] & mask;
            uint32_t budget = probeLength;
            while (budget != 0) {
                const uint64_t prev = output[cur].load(std::memory_order_relaxed);
                if (prev == 0) {
                    if (!useCas) { output[cur].store(R[i], std::memory_order_relaxed); break; }
                    uint64_t zero = 0;
                    if (output[cur].compare_exchange_strong(zero, R[i])) break;
                    budget--;  // AtomicHashBuild.hpp:54
                } else { cur = (cur + 1) & mask; budget--; }
            }
            if (budget == 0) { conf[c]++; confSum[c] += R[i]; }
        }
    });
    if (S)
        for_chunks(numPartitions, nthreads, [&](uint32_t c) {
            const uint64_t ps = sSize / numPartitions, b = c * ps, e = (c + 1 == numPartitions) ? sSize : b + ps;
            uint64_t m = 0;
            for (uint64_t i = b; i < e; ++i) {
                uint64_t cur = S[i] & mask;
                uint32_t budget = probeLength;
                while (budget-- && output[cur].load(std::memory_order_relaxed) != 0) {
                    if (output[cur].load(std::memory_order_relaxed) == S[i]) m++;
                    cur++;
                }
            }
            match[c] = m;
        });
    res.us = now_us() - t0;
    for (uint64_t i = 0; i < rSize; ++i) res.inputSum += R[i];
    const uint64_t upto = useCas ? tableSize : rSize;  // NoCCHashBuild.hpp:94 vs AtomicHashBuild.hpp:100
    for (uint64_t i = 0; i < upto; ++i) res.outputSum += output[i].load(std::memory_order_relaxed);
    for (uint32_t c = 0; c < numPartitions; ++c) { res.conflicts += conf[c]; res.outputSum += confSum[c]; res.matches += match[c]; }
    return res;
}

}  // namespace

int main(int argc, char* argv[])
{
    param_t p;
    parseArgs(argc, argv, &p);

    const bool gpuOA = p.algo == "atomic" || p.algo == "htm" || p.algo == "hip-atomic";
    bool gpuPRJ = p.algo == "prj" || p.algo == "hip-prj" || p.algo == "PRO";
    const bool gpuAuto = p.algo == "auto";
    const bool cpu = p.algo == "nocc" || p.algo == "cpu-atomic";
    if (!gpuOA && !gpuPRJ && !gpuAuto && !cpu) {
        std::cout << "Unknown Algo: " << p.algo << std::endl;  // main.cpp:108
        return 0;
    }

    const uint64_t sSize = p.sSize ? p.sSize : p.rSize;
    std::vector<uint64_t> relR(p.rSize), relS;
    if (hj_generate_data(p.dataDistr.c_str(), p.rSize, p.rSize, (int)p.shuffleRange, p.zipfTheta, relR.data()) != HJ_OK) {
        std::cout << "Unknown distribution" << std::endl;  // DataGen.hpp:117-118
        return 1;
    }
    if (p.probe) {
        relS.resize(sSize);
        if (!p.sDistr.empty()) {
            if (hj_generate_data(p.sDistr.c_str(), sSize, p.rSize, (int)p.shuffleRange, p.zipfTheta, relS.data()) != HJ_OK) {
                std::cout << "Unknown distribution" << std::endl;
                return 1;
            }
        } else if (p.dataDistr != "random") {  // main.cpp:92-97
            hj_generate_data("sorted", sSize, p.rSize, (int)p.shuffleRange, 0, relS.data());
        } else {
            for (uint64_t i = 0; i < sSize; ++i) relS[i] = relR[i % p.rSize];
        }
    }
    const uint64_t* S = p.probe ? relS.data() : nullptr;

    if (!cpu && (p.gpus > 1 || !p.split.empty())) {
        // ---- the radix-sharded join: rank g = device g holds the g-th contiguous piece of R and of S ----------------
        if (p.algo != "atomic" && p.algo != "hip-atomic") { std::cerr << "--gpus / --split: the open-addressing operator (--algo atomic) shards" << std::endl; return 1; }
        const int G = p.gpus;
        std::vector<int> devs(G);
        for (int g = 0; g < G; ++g) devs[g] = g;
        hj_sharded* sh = nullptr;
        int rc = hj_sharded_create(devs.data(), G, &sh);
        if (rc != HJ_OK) { std::cerr << "hj_sharded_create(" << G << " devices): " << hj_strerror(rc) << std::endl; return 2; }
        std::vector<const uint64_t*> dR(G), dS(G);
        std::vector<uint64_t> nR(G), nS(G);
        std::vector<void*> owned;
        const double t0 = now_us();
        for (int g = 0; g < G && rc == HJ_OK; ++g) {
            const uint64_t rb = p.rSize * g / G, re = p.rSize * (g + 1) / G, sb = sSize * g / G, se = sSize * (g + 1) / G;
            nR[g] = re - rb; nS[g] = S ? se - sb : 0;
            void* d = nullptr;
            if ((rc = hj_sharded_alloc(sh, g, (nR[g] + 2) * 8, &d)) != HJ_OK) break;
            owned.push_back(d); dR[g] = static_cast<const uint64_t*>(d);
            if ((rc = hj_sharded_copy_h2d(sh, g, d, relR.data() + rb, nR[g] * 8)) != HJ_OK) break;
            if (S) {
                if ((rc = hj_sharded_alloc(sh, g, (nS[g] + 2) * 8, &d)) != HJ_OK) break;
                owned.push_back(d); dS[g] = static_cast<const uint64_t*>(d);
                if ((rc = hj_sharded_copy_h2d(sh, g, d, S + sb, nS[g] * 8)) != HJ_OK) break;
            }
        }
        const double h2d = now_us() - t0;
        hj_params hp{};
        hp.algo = HJ_ALGO_ATOMIC; hp.scaleOutput = p.scaleOutput; hp.numPartitions = p.numPartitions; hp.probeLength = p.probeLength;
        const uint32_t split = p.split == "high" ? (uint32_t)HJ_SPLIT_HIGH : (uint32_t)HJ_SPLIT_LOW;
        for (int rep = 0; rep < p.repeat && rc == HJ_OK; ++rep) {
            hj_result r{};
            hj_sharded_stats st{};
            const double w0 = now_us();
            rc = hj_sharded_join(sh, &hp, split, p.rSize, 0, dR.data(), nR.data(), S ? dS.data() : nullptr, S ? nS.data() : nullptr, &r, &st);
            const double wall = now_us() - w0;
            if (rc != HJ_OK) break;
            std::cout << "{\"algo\": \"" << p.algo << "\",\"rSize\": " << p.rSize << ", \"probeLength\": " << p.probeLength
                      << ", \"hashBuildTimeInMicroseconds\": " << (uint64_t)wall << ", \"conflicts\": " << r.conflicts;
            if (p.probe) std::cout << ", \"totalMatches\": " << r.totalMatches;
            std::cout << ", \"inputSum\": " << r.inputSum << ", \"outputSum\": " << r.outputSum
                      << ", \"device\": \"hip\", \"n_gpus\": " << G << ", \"split\": \"" << (st.mode ? "high" : "low") << "\", \"mode\": " << st.mode
                      << ", \"homeShift\": " << st.homeShift << ", \"keysMovedR\": " << st.keysMovedR << ", \"keysMovedS\": " << st.keysMovedS
                      << ", \"maxMessageKeys\": " << st.maxMessageKeys << ", \"tableSizePerRank\": " << st.tableSizePerRank
                      << ", \"sSize\": " << (S ? sSize : 0) << ", \"mtuples_per_s\": " << (double)(p.rSize + (S ? sSize : 0)) / wall
                      << ", \"slowest_rank_build_us\": " << r.build_us << ", \"slowest_rank_probe_us\": " << r.probe_us
                      << ", \"h2d_us\": " << h2d << "}" << std::endl;
        }
        if (rc != HJ_OK) std::cerr << "sharded join: " << hj_strerror(rc) << " (" << hj_sharded_last_error(sh) << ")" << std::endl;
        for (size_t i = 0; i < owned.size(); ++i) hj_sharded_free(sh, (int)(i / (S ? 2 : 1)) , owned[i]);
        hj_sharded_destroy(sh);
        return rc == HJ_OK ? 0 : 2;
    }

    for (int rep = 0; rep < p.repeat; ++rep) {
        if (cpu) {
            const cpu_result r = cpu_build_probe(p.algo == "cpu-atomic", relR.data(), p.rSize, S, sSize, p.numPartitions, p.probeLength);
            std::cout << "{\"algo\": \"" << (p.algo == "nocc" ? "nocc" : "atomic") << "\",\"rSize\": " << p.rSize
                      << ", \"probeLength\": " << p.probeLength << ", \"hashBuildTimeInMicroseconds\": " << (uint64_t)r.us
                      << ", \"conflicts\": " << r.conflicts;
            if (p.probe) std::cout << ", \"totalMatches\": " << r.matches;
            std::cout << ", \"inputSum\": " << r.inputSum << ", \"outputSum\": " << r.outputSum
                      << ", \"device\": \"cpu\", \"cpu_threads\": " << r.threads << "}" << std::endl;
            continue;
        }
        hj_ctx* ctx = nullptr;
        int rc = hj_create(p.device, &ctx);
        if (rc != HJ_OK) {
            std::cerr << "hj_create: " << hj_strerror(rc) << std::endl;
            return 2;
        }
        hj_params hp{};
        hp.algo = gpuAuto ? HJ_ALGO_AUTO : gpuPRJ ? HJ_ALGO_PRJ : (p.algo == "htm" ? HJ_ALGO_HTM : HJ_ALGO_ATOMIC);
        hp.scaleOutput = p.scaleOutput; hp.numPartitions = p.numPartitions; hp.probeLength = p.probeLength;
        hp.transactionSize = p.transactionSize; hp.radixBits = p.radixBits;
        hj_result r{};
        rc = hj_run(ctx, &hp, relR.data(), p.rSize, S, S ? sSize : 0, &r);
        if (rc != HJ_OK) {
            std::cerr << "hj_run: " << hj_strerror(rc) << " (" << hj_last_error(ctx) << ")" << std::endl;
            hj_destroy(ctx);
            return 2;
        }
        if (gpuAuto) gpuPRJ = r.algoUsed == HJ_ALGO_PRJ;     // print the fields of the path that ran
        const double mt = (double)(p.rSize + (S ? sSize : 0)) / r.total_us;
        if (p.algo == "htm") {
            // the reference's htm line, field for field (HTMHashBuild.hpp:417-452). There are no transactions here, so
            // none fail; conflictCount = tuples that found their bucket full; outputSum = buckets + overflow chains
            std::cout << "{\"algo\": \"htm\",\"rSize\": " << p.rSize << ", \"transactionSize\": " << p.transactionSize
                      << ", \"probeLength\": " << p.probeLength << ", \"hashBuildTimeInMicroseconds\": " << (uint64_t)r.total_us
                      << ", \"firstRoundTime\": 0, \"firstRoundFailureFraction\": 0, \"conflictCount\": " << r.conflicts
                      << ", \"failedTransactions\": 0, \"failedTransactionPercentage\": 0, \"totalFailedPercentage\": "
                      << (double)r.conflicts / (double)p.rSize;
            if (p.probe) std::cout << ", \"totalMatches\": " << r.totalMatches;
            std::cout << ", \"inputSum\": " << r.inputSum << ", \"outputSum\": " << r.outputSum
                      << ", \"device\": \"hip\", \"numBuckets\": " << r.htmBuckets << ", \"overflowBuckets\": " << r.htmOverflowBuckets
                      << ", \"sSize\": " << (S ? sSize : 0) << ", \"mtuples_per_s\": " << mt << ", \"build_us\": " << r.build_us
                      << ", \"probe_us\": " << r.probe_us << ", \"h2d_us\": " << r.h2d_us << "}" << std::endl;
            hj_destroy(ctx);
            continue;
        }
        std::cout << "{\"algo\": \"" << p.algo << "\",\"rSize\": " << p.rSize;
        std::cout << ", \"probeLength\": " << p.probeLength << ", \"hashBuildTimeInMicroseconds\": " << (uint64_t)r.total_us;
        if (!gpuPRJ) std::cout << ", \"conflicts\": " << r.conflicts;
        if (p.probe) std::cout << ", \"totalMatches\": " << r.totalMatches;
        if (gpuPRJ) std::cout << ", \"results\": " << r.prjChecksum << ", \"radixBits\": " << r.radixBits;
        else std::cout << ", \"inputSum\": " << r.inputSum << ", \"outputSum\": " << r.outputSum;
        if (gpuAuto) std::cout << ", \"algoUsed\": \"" << (gpuPRJ ? "prj" : "atomic") << "\"";
        std::cout << ", \"device\": \"hip\", \"sSize\": " << (S ? sSize : 0) << ", \"mtuples_per_s\": " << mt
                  << ", \"clear_us\": " << r.clear_us << ", \"build_us\": " << r.build_us << ", \"probe_us\": " << r.probe_us
                  << ", \"partition_us\": " << r.partition_us << ", \"join_us\": " << r.join_us
                  << ", \"h2d_us\": " << r.h2d_us << "}" << std::endl;
        hj_destroy(ctx);
    }
    return 0;
}
