// Host side of the epilogue's hand-over (reference overiva.py:192-204: Y goes back to the caller as a NumPy array): a process-wide
// ring of pinned staging buffers, a small pool of copy threads, and page pre-faulting of a destination the caller just
// allocated.  Host code only.
//
// Why: the reference signature returns a fresh (T, F, K) array -- 131 MB at the headline shape.  One synchronous hipMemcpy2D
// into such an array (pageable, its pages not yet faulted in) ran at 10 GB/s, 12.7 ms of a 29 ms call whose iterations take 4.
// Here the final demix is cut into slabs of frames; slab k's device-to-host copy (into the pinned ring, at link speed) runs
// while slab k + 1 is computed and while the copy threads move slab k - 1 from the ring into the caller's array, whose pages
// overiva() had faulted in by the same threads while the iterations ran (oiva_host_prefault).
#include "host_io.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace oiva {
namespace {

// N persistent threads; run(f) calls f(j, N) on every thread j and returns when all are done
class CopyPool {
public:
    explicit CopyPool(int n) : n_(n) {
        for (int j = 1; j < n_; ++j) th_.emplace_back([this, j] { loop(j); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return n_; }
    void run(const std::function<void(int, int)>& f) {
        std::lock_guard<std::mutex> one_at_a_time(run_m_);
        {
            std::lock_guard<std::mutex> g(m_);
            task_ = &f;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        f(0, n_);                               // the caller is thread 0
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        task_ = nullptr;
    }

private:
    void loop(int j) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(int, int)>* f = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                f = task_;
            }
            if (f) (*f)(j, n_);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* task_ = nullptr;
    unsigned gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

CopyPool& pool() {
    static CopyPool p([] {
        if (const char* v = std::getenv("OIVA_IO_THREADS")) return std::max(1, std::min(64, std::atoi(v)));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(8u, hw / 2));
    }());
    return p;
}

struct Ring {
    std::mutex m;
    void* slot[kHostRingSlots] = {};
    size_t bytes = 0;
    // (never freed: at process exit the HIP runtime may already be gone when static destructors run)
};
Ring& ring() {
    static Ring r;
    return r;
}

}  // namespace

hipError_t host_ring_slots(size_t bytes, void** slots) {
    Ring& r = ring();
    std::lock_guard<std::mutex> g(r.m);
    if (bytes > r.bytes) {
        for (void*& s : r.slot) {
            if (s) (void)hipHostFree(s);
            s = nullptr;
        }
        r.bytes = 0;
        for (void*& s : r.slot) {
            hipError_t e = hipHostMalloc(&s, bytes, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        r.bytes = bytes;
    }
    for (int i = 0; i < kHostRingSlots; ++i) slots[i] = r.slot[i];
    return hipSuccess;
}

void host_copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, long long nrows) {
    CopyPool& p = pool();
    const bool dense = dst_pitch == row_bytes && src_pitch == row_bytes;
    p.run([&](int j, int n) {
        const long long r0 = nrows * j / n, r1 = nrows * (j + 1) / n;
        if (r1 <= r0) return;
        char* d = static_cast<char*>(dst) + (size_t)r0 * dst_pitch;
        const char* s = static_cast<const char*>(src) + (size_t)r0 * src_pitch;
        if (dense) {
            std::memcpy(d, s, (size_t)(r1 - r0) * row_bytes);
        } else {
            for (long long r = r0; r < r1; ++r) std::memcpy(d + (size_t)(r - r0) * dst_pitch, s + (size_t)(r - r0) * src_pitch, row_bytes);
        }
    });
}

int host_io_threads() { return pool().size(); }

void host_prefault(void* ptr, size_t bytes, bool may_touch) {
    if (!ptr || bytes == 0) return;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    CopyPool& p = pool();
#ifdef MADV_HUGEPAGE
    {
        // $OIVA_IO_THP=1: ask for 2 MB pages where the system gives them on request (transparent huge pages in `madvise`
        // mode).  Off by default: measured on the MI355X box (EPYC 9575F) the pre-fault of 131 MB took 1.9-2.5 ms with the
        // advice against 1.1 ms without, and the hand-over that follows was no faster.
        static const bool thp = [] { const char* v = std::getenv("OIVA_IO_THP"); return v && v[0] == '1'; }();
        const uintptr_t a2 = ((uintptr_t)ptr + page - 1) & ~(uintptr_t)(page - 1), b2 = ((uintptr_t)ptr + bytes) & ~(uintptr_t)(page - 1);
        if (thp && b2 > a2 && b2 - a2 >= ((size_t)4 << 20)) (void)madvise((void*)a2, b2 - a2, MADV_HUGEPAGE);
    }
#endif
    p.run([&](int j, int n) {
        // whole pages of this thread's share of the range (the first and last partial pages are touched by their owners too)
        const uintptr_t a = (uintptr_t)ptr, b = a + bytes;
        const uintptr_t lo = (a + (b - a) / n * j) & ~(uintptr_t)(page - 1);
        const uintptr_t hi = j + 1 == n ? b : ((a + (b - a) / n * (j + 1)) & ~(uintptr_t)(page - 1));
        if (hi <= lo) return;
        const uintptr_t lo_in = std::max(lo, a);
#ifdef MADV_POPULATE_WRITE
        // (Linux >= 5.14: faults the pages in for writing without changing their contents)
        // (the page that holds the first byte belongs to the same mapping: populating all of it is harmless)
        const uintptr_t lo_pg = lo_in & ~(uintptr_t)(page - 1);
        if (madvise((void*)lo_pg, hi - lo_pg, MADV_POPULATE_WRITE) == 0) return;
#endif
        if (!may_touch) return;      // (a range that holds other people's bytes: populate or nothing)
        for (uintptr_t q = lo_in; q < hi; q = (q & ~(uintptr_t)(page - 1)) + page) {
            volatile char* c = (volatile char*)q;
            *c = *c;                             // a write fault that keeps the byte
        }
    });
}

}  // namespace oiva

extern "C" {

int oiva_host_prefault(void* ptr, long long bytes) {
    if (!ptr || bytes < 0) return oiva::fail_with(OIVA_ERR_ARG, "bad arguments");
    oiva::host_prefault(ptr, (size_t)bytes, true);
    return OIVA_OK;
}

}  // extern "C"
