// ShmBarrier.cpp -- a barrier for the ranks of ONE node, in POSIX shared memory (no reference counterpart: upstream is one process,
// EngineMain.cpp:11-17). bench.py brackets its timed region with it when every rank runs on the same host: a rank's share of a
// 3840x2160 frame on eight GPUs is 20 x 0.13 ms = 2.6 ms of timed region, and a TCP (gloo) or collective (RCCL) barrier's latency and
// exit skew -- tens to hundreds of microseconds, milliseconds on a loaded host -- would be counted as rendering time. Here a rank
// leaves within a cache-line transfer of the last rank's arrival. torch.distributed still sets the group up and reduces the results.
//
// Central counter + generation (sense reversal by generation number): the last arriver resets the counter and bumps the generation,
// everybody else spins on the generation with `pause`. Every wait has a deadline, so a rank that died cannot hang the others.
#include "../../include/crt_host.h"
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
struct Shared {
    std::atomic<uint32_t> magic, nRanks, count, generation;
};
struct Handle { Shared* sh; char name[128]; bool owner; };
constexpr uint32_t kMagic = 0x43525442u;   // "CRTB"
inline void relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}
}

extern "C" {

void* crth_shm_barrier_open(const char* name, int nRanks, int create)
{
    if (!name || name[0] != '/' || std::strlen(name) >= sizeof(Handle::name) || nRanks < 1) return nullptr;
    static_assert(std::atomic<uint32_t>::is_always_lock_free, "the barrier words must be plain shared-memory atomics");
    const int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); shm_unlink(name); return nullptr; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (create) shm_unlink(name); return nullptr; }
    Shared* sh = static_cast<Shared*>(p);
    if (create) {
        sh->count.store(0); sh->generation.store(0); sh->nRanks.store((uint32_t)nRanks);
        sh->magic.store(kMagic, std::memory_order_release);
    } else if (sh->magic.load(std::memory_order_acquire) != kMagic || sh->nRanks.load() != (uint32_t)nRanks) {
        munmap(p, sizeof(Shared));
        return nullptr;
    }
    Handle* h = new (std::nothrow) Handle;
    if (!h) { munmap(p, sizeof(Shared)); if (create) shm_unlink(name); return nullptr; }
    h->sh = sh; h->owner = create != 0; std::strcpy(h->name, name);
    return h;
}

int crth_shm_barrier_wait(void* handle, int timeoutMs)
{
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !h->sh) return -2;
    Shared& s = *h->sh;
    const uint32_t n = s.nRanks.load(std::memory_order_relaxed);
    const uint32_t gen = s.generation.load(std::memory_order_acquire);
    if (s.count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
        s.count.store(0, std::memory_order_relaxed);
        s.generation.fetch_add(1, std::memory_order_release);
        return 0;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeoutMs > 0 ? timeoutMs : 1);
    for (unsigned spins = 0;; ++spins) {
        if (s.generation.load(std::memory_order_acquire) != gen) return 0;
        relax();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() > deadline) return -1;   // a rank is missing: the caller gives up loudly
    }
}

void crth_shm_barrier_close(void* handle)
{
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return;
    if (h->sh) munmap(h->sh, sizeof(Shared));
    if (h->owner) shm_unlink(h->name);
    delete h;
}

} // extern "C"
