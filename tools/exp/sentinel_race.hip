// tools/exp/sentinel_race.hip — stand-alone reproducer for the near-tie queue's slot protocols ("The queue's slot protocol, and
// the race it had").  An experiment, not product code: nothing links it.
//
//     hipcc --offload-arch=gfx950 -O3 -o tools/exp/sentinel_race tools/exp/sentinel_race.hip
//     tools/exp/sentinel_race [--mode sentinel|epoch|sentinel-drained|sentinel-line] [--launches N] [--rays N] [--push-shift S] [--noise MB]
//
// What it models (round-3 k_closest_fast + DrainRetrace, wf_backend.hip before commit 4110f66):
//   * a persistent grid; the last SERVICE workgroups walk nothing and poll the queue, the others ("workers") fetch chunks of rays from a
//     cursor, "walk" each (a dependent chain of loads of random length through a large buffer: the latency profile of a BVH walk), and with
//     probability 2^-S push an entry: slot = atomicAdd(cnt, 1), then ONE 64-bit relaxed agent-scope store of the entry into slots[slot];
//   * takers (service waves, and workers once their rays are gone) claim [head, head + t) with a CAS and spin on each claimed slot with
//     relaxed agent-scope loads until it is published; the "re-walk" is a long single-lane chain; the wave that signs off last rewinds the
//     counters; slot numbers restart at 0 in every launch.
// Protocols:
//   sentinel          the old one: a slot is free when it holds ~0; the taker stores ~0 back after reading the entry (the only state that
//                     crosses a launch boundary);
//   sentinel-drained  the same, but the taker waits for its restoring store (s_waitcnt vmcnt(0)) before it goes on;
//   sentinel-line     the same as sentinel with one slot per 128-byte line (no two takers share a line);
//   epoch             the current one: the entry carries the launch's epoch in its high bits, a slot is published when that tag is the
//                     current epoch, nothing is restored.
// An entry is (epoch << 40) | ray id, so EVERY protocol can tell what it took: a taken word whose epoch is not the launch's is a STALE TAKE
// (logged with the slot, the word, the taker's XCC and the launch), a pushed ray that nobody re-walked is a LOST ENTRY, a slot that is not
// ~0 after a sentinel launch is a RESIDUE.  Exit status 1 when any of them occurred.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                                                   \
    do {                                                                                                           \
        hipError_t e_ = (x);                                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

constexpr int TBLOCK = 256, SERVICE = 16, STRIDE = 32;   // counters 128 bytes apart
constexpr int SPIN_LIMIT = 1 << 24;                      // a slot that stays unpublished this long is logged as a stale take of ~0
enum { C_CNT = 0, C_HEAD, C_DONE, C_CURSOR, C_STALE, C_TAKEN, C_PUSHED, C_LOG, C_SPINS, C_N };
enum Mode { SENTINEL = 0, SENTINEL_DRAINED, SENTINEL_LINE, EPOCH };

struct LogRec { uint32_t launch, slot, xcc, wave; unsigned long long word; };
struct Args {
    int *c;                       // counters
    unsigned long long *slots;    // the queue
    const uint32_t *maze;         // the "scene": a large table of next-indices
    uint32_t mazeMask;
    uint32_t *rewalked;           // per ray: epoch of the launch that re-walked it
    uint32_t *pushedBy;           // per ray: epoch of the launch that pushed it
    LogRec *log;
    int nRays, pushShift, mode;
    uint32_t epoch;
};

__device__ inline uint32_t Mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline int SlotIndex(const Args &a, int s) { return a.mode == SENTINEL_LINE ? s * 16 : s; }

__device__ inline uint32_t Chain(const Args &a, uint32_t p, int steps) {   // dependent loads: the latency of a walk
    for (int k = 0; k < steps; ++k) p = a.maze[p & a.mazeMask] + k;
    return p;
}

__device__ inline bool TakeSome(const Args &a, int lane, int wave) {
    int *cnt = a.c + C_CNT * STRIDE, *head = a.c + C_HEAD * STRIDE;
    int base = 0, take = 0;
    if (lane == 0) {
        while (true) {
            const int h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h >= c) break;
            const int t = c - h < 64 ? c - h : 64;
            if (atomicCAS(head, h, h + t) == h) { base = h; take = t; break; }
        }
    }
    base = __builtin_amdgcn_readfirstlane(base);
    take = __builtin_amdgcn_readfirstlane(take);
    if (take == 0) return false;
    if (lane < take) {
        unsigned long long *slot = a.slots + SlotIndex(a, base + lane);
        unsigned long long e;
        int spins = 0;
        if (a.mode == EPOCH) {
            while (((e = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 40) != a.epoch && spins < SPIN_LIMIT) ++spins;
        } else {
            while ((e = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull && spins < SPIN_LIMIT) ++spins;   // (bounded: an experiment must not hang the box)
            __hip_atomic_store(slot, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.mode == SENTINEL_DRAINED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (spins) atomicAdd(a.c + C_SPINS * STRIDE, 1);
        const uint32_t ray = (uint32_t)(e & 0xffffffffull), ep = (uint32_t)(e >> 40);
        if (ep != a.epoch || ray >= (uint32_t)a.nRays) {
            atomicAdd(a.c + C_STALE * STRIDE, 1);
            const int l = atomicAdd(a.c + C_LOG * STRIDE, 1);
            if (l < 256) a.log[l] = LogRec{a.epoch, (uint32_t)(base + lane), (uint32_t)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15), (uint32_t)wave, e};
        } else {
            const uint32_t r = Chain(a, Mix(ray), 600 + (int)(Mix(ray ^ a.epoch) & 2047));   // the long single-lane re-walk
            a.rewalked[ray] = a.epoch | (r == 0xffffffffu ? 1u << 31 : 0);
            atomicAdd(a.c + C_TAKEN * STRIDE, 1);
        }
    }
    return true;
}

__global__ void __launch_bounds__(TBLOCK) k_walk(Args a) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * TBLOCK + threadIdx.x) >> 6;
    const int totalWaves = gridDim.x * (TBLOCK / 64), workerWaves = ((int)gridDim.x - SERVICE) * (TBLOCK / 64);
    int *done = a.c + C_DONE * STRIDE;
    auto signOff = [&]() {
        int prev = 0;
        if (lane == 0) { __threadfence(); prev = atomicAdd(done, 1); }
        return __builtin_amdgcn_readfirstlane(prev);
    };
    bool last;
    if ((int)blockIdx.x >= (int)gridDim.x - SERVICE) {   // a service workgroup
        while (true) {
            if (TakeSome(a, lane, wave)) continue;
            int dn = 0;
            if (lane == 0) dn = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dn = __builtin_amdgcn_readfirstlane(dn);
            if (dn >= workerWaves) {
                if (TakeSome(a, lane, wave)) continue;
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        last = signOff() == totalWaves - 1;
    } else {
        while (true) {   // every wave fetches its own 64 rays (waves never meet)
            int base = 0;
            if (lane == 0) base = atomicAdd(a.c + C_CURSOR * STRIDE, 64);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= a.nRays) break;
            const int ray = base + lane;
            if (ray < a.nRays) {
                const uint32_t h = Mix((uint32_t)ray * 2654435761u ^ a.epoch * 40503u);
                const uint32_t r = Chain(a, h, 20 + (int)(h >> 25));   // 20..147 steps
                if (((h ^ r) & ((1u << a.pushShift) - 1)) == 0 || r == 0xffffffffu) {
                    const int s = atomicAdd(a.c + C_CNT * STRIDE, 1);
                    a.pushedBy[ray] = a.epoch;
                    atomicAdd(a.c + C_PUSHED * STRIDE, 1);
                    __hip_atomic_store(a.slots + SlotIndex(a, s), ((unsigned long long)a.epoch << 40) | (unsigned long long)(uint32_t)ray, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        while (TakeSome(a, lane, wave)) {}
        last = signOff() == totalWaves - 1;
    }
    if (last && lane == 0) {
        __hip_atomic_store(a.c + C_HEAD * STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.c + C_CNT * STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_reset(int *c) {   // the stage reset between two launches (k_reset of the backend): plain stores
    if (threadIdx.x < 4) c[(threadIdx.x == 3 ? C_CURSOR : (int)threadIdx.x) * STRIDE] = 0;
}
__global__ void k_noise(float4 *buf, size_t n, float s) {   // what runs between two closest-hit launches: streaming kernels
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = buf[i];
        v.x = v.x * s + 1; v.y += v.x; v.z -= v.y; v.w *= s;
        buf[i] = v;
    }
}
__global__ void k_verify(Args a, int *out, int maxSlots) {   // lost entries (pushed this launch, not re-walked) and residues (slot != ~0)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nRays && a.pushedBy[i] == a.epoch && (a.rewalked[i] & 0x7fffffffu) != a.epoch) atomicAdd(out + 0, 1);
    if (a.mode != EPOCH && i < maxSlots && a.slots[SlotIndex(a, i)] != ~0ull) atomicAdd(out + 1, 1);
}

int main(int argc, char **argv) {
    int launches = 2000, nRays = 1 << 20, pushShift = 8, noiseMB = 64, mode = SENTINEL;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : (char *)"0"; };
        if (s == "--launches") launches = atoi(next());
        else if (s == "--rays") nRays = atoi(next());
        else if (s == "--push-shift") pushShift = atoi(next());
        else if (s == "--noise") noiseMB = atoi(next());
        else if (s == "--mode") {
            const std::string m = next();
            mode = m == "epoch" ? EPOCH : m == "sentinel-drained" ? SENTINEL_DRAINED : m == "sentinel-line" ? SENTINEL_LINE : SENTINEL;
        }
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, grid = cus * 4;
    const int capacity = nRays;   // slots
    Args a{};
    CHECK(hipMalloc(&a.c, C_N * STRIDE * sizeof(int)));
    CHECK(hipMemset(a.c, 0, C_N * STRIDE * sizeof(int)));
    CHECK(hipMalloc(&a.slots, (size_t)capacity * 16 * sizeof(unsigned long long) / (mode == SENTINEL_LINE ? 1 : 16)));
    CHECK(hipMemset(a.slots, 0xff, (size_t)capacity * 16 * sizeof(unsigned long long) / (mode == SENTINEL_LINE ? 1 : 16)));
    const uint32_t mazeN = 1u << 26;   // 256 MB: misses L2 and most of the MALL
    std::vector<uint32_t> maze(mazeN);
    uint64_t s = 88172645463325252ull;
    for (uint32_t &m : maze) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; m = (uint32_t)(s >> 20); }
    uint32_t *dMaze;
    CHECK(hipMalloc(&dMaze, (size_t)mazeN * 4));
    CHECK(hipMemcpy(dMaze, maze.data(), (size_t)mazeN * 4, hipMemcpyHostToDevice));
    a.maze = dMaze; a.mazeMask = mazeN - 1;
    CHECK(hipMalloc(&a.rewalked, (size_t)nRays * 4));
    CHECK(hipMemset(a.rewalked, 0, (size_t)nRays * 4));
    CHECK(hipMalloc(&a.pushedBy, (size_t)nRays * 4));
    CHECK(hipMemset(a.pushedBy, 0, (size_t)nRays * 4));
    CHECK(hipMalloc(&a.log, 256 * sizeof(LogRec)));
    a.nRays = nRays; a.pushShift = pushShift; a.mode = mode;
    int *dOut;
    CHECK(hipMalloc(&dOut, 8));
    CHECK(hipMemset(dOut, 0, 8));
    float4 *noise = nullptr;
    const size_t noiseN = (size_t)noiseMB << 16;
    if (noiseMB) { CHECK(hipMalloc(&noise, noiseN * 16)); CHECK(hipMemset(noise, 0, noiseN * 16)); }
    const char *names[] = {"sentinel", "sentinel-drained", "sentinel-line", "epoch"};
    printf("mode %s: %d launches of %d rays (push 2^-%d), grid %d x %d (%d service workgroups), %d MB streamed between launches\n", names[mode], launches, nRays,
           pushShift, grid, TBLOCK, SERVICE, noiseMB);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int l = 1; l <= launches; ++l) {
        a.epoch = (uint32_t)l;
        k_reset<<<1, 64>>>(a.c);
        k_walk<<<grid, TBLOCK>>>(a);
        k_verify<<<(nRays + 255) / 256, 256>>>(a, dOut, capacity);
        if (noise) k_noise<<<cus * 8, 256>>>(noise, noiseN, 1.0001f);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> c(C_N * STRIDE);
    int out[2];
    CHECK(hipMemcpy(c.data(), a.c, c.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(out, dOut, 8, hipMemcpyDeviceToHost));
    const int stale = c[C_STALE * STRIDE];
    printf("  pushed %d, re-walked %d, spun on an unpublished slot %d times; STALE TAKES %d, LOST ENTRIES %d, RESIDUES %d; %.1f ms per launch\n", c[C_PUSHED * STRIDE],
           c[C_TAKEN * STRIDE], c[C_SPINS * STRIDE], stale, out[0], out[1], ms / launches);
    if (stale) {
        std::vector<LogRec> log(256);
        CHECK(hipMemcpy(log.data(), a.log, 256 * sizeof(LogRec), hipMemcpyDeviceToHost));
        for (int i = 0; i < (stale < 24 ? stale : 24); ++i)
            printf("    launch %u slot %u: word %016llx (epoch %llu, %lld launches old), taker wave %u on XCC %u\n", log[i].launch, log[i].slot, log[i].word,
                   log[i].word >> 40, (long long)log[i].launch - (long long)(log[i].word >> 40), log[i].wave, log[i].xcc);
    }
    return stale || out[0] || out[1] ? 1 : 0;
}
