// crt_multidev.h -- several devices behind the one C-ABI (crt_init_devices): per-device worker threads, device selection, the dispatch macros
// Part of the one translation unit crt_shim.hip (included there, in this order: crt_state.h, crt_instances.h, crt_upload.h,
// crt_bvh_driver.h, crt_frame.h, crt_multidev.h); everything here has internal linkage.
#pragma once
namespace {

// ------------------------------------------------------------------------------------------------
// Dispatch: one device (crt_init) or several in one process (crt_init_devices / crt_init_gpus)
//
// The reference drives ONE OpenCL device from one thread (Renderer.cpp:134 asks clGetDeviceIDs for a single GPU). With
// several devices the same C-ABI is kept: the scene is replicated by every upload, the frame is cut into 16-row bands
// dealt round-robin to the devices (device d = rank d of n), every device traces its bands with its own streams and frame
// slots, copies them into the primary device's frame (hipMemcpy2DAsync peer copies, one strided copy per device and frame,
// over xGMI) and the primary's end-of-frame event waits for those copies -- so crt_render keeps upstream's
// Render() + clFinish meaning, crt_read_output / crt_map_host_frame / crt_output_device_ptr return the WHOLE frame, and
// frames in flight work as before. No collective and no host staging in the data path. Each secondary device has a
// host worker thread that submits its share of a frame, so the per-frame submission cost does not grow with the number of
// devices; the calling thread submits the primary's share last (its stream must wait on events the others have recorded).
// ------------------------------------------------------------------------------------------------
struct Worker {
    // Job hand-over by generation counters: the owner bumps `posted`, the worker bumps `finished`. Both sides spin briefly
    // (a frame's share is submitted in ~30 us, a condition-variable wake-up alone costs 5-10 us each way) and fall back
    // to the condition variable, so an idle session does not burn a core.
    std::thread th; std::mutex m; std::condition_variable cv;
    std::function<int()> job; std::atomic<unsigned> posted{0}, finished{0}; std::atomic<bool> quit{false}; int result = 0;
    State* st = nullptr; int device = 0;
    static constexpr int kSpins = 4000;
    void start(State* s, int dev)
    {
        st = s; device = dev;
        th = std::thread([this] {
            (void)hipSetDevice(device);
            G = st;
            unsigned seen = 0;
            for (;;) {
                int spins = 0;
                while (posted.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_acquire)) {
                    if (++spins < kSpins) { crt_cpu_relax(); continue; }
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_acquire); });
                }
                if (quit.load(std::memory_order_acquire)) return;
                seen = posted.load(std::memory_order_acquire);
                result = job();
                finished.store(seen, std::memory_order_release);
                { std::lock_guard<std::mutex> lk(m); }
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> f)
    {
        job = std::move(f);
        { std::lock_guard<std::mutex> lk(m); posted.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
    }
    int wait()
    {
        const unsigned want = posted.load(std::memory_order_acquire);
        int spins = 0;
        while (finished.load(std::memory_order_acquire) != want) {
            if (++spins < kSpins) { crt_cpu_relax(); continue; }
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished.load(std::memory_order_acquire) == want; });
        }
        return result;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit.store(true, std::memory_order_release); }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct Group {
    int n = 0;                                  // 0: no session; 1: crt_init; >1: crt_init_devices
    State* dev[CRT_MAX_DEVICES] = { nullptr };
    int hipDevice[CRT_MAX_DEVICES] = { 0 };
    Worker* worker[CRT_MAX_DEVICES] = { nullptr };
    // how device d's bands reach the primary's frame: 2 = same physical GPU as the primary (rehearsal), 1 = peer mapping
    // (hipDeviceEnablePeerAccess: xGMI), 0 = no peer access, the runtime stages the copy through host memory
    int peer[CRT_MAX_DEVICES] = { 0 };
    unsigned seq = 0;                           // frame-slot rotation of the session (crt_render)
    bool broken = false;                        // a resize failed on some device and could not be rolled back
    int injectFailure = -1;                     // crt_debug_inject_failure
    int pack8W = 0, pack8H = 0;                 // frame size every device's byte frames (RGBA8 gather) were last allocated for
    unsigned long long lastGatherBytes = 0; int lastGatherBpp = 0;   // crt_debug_last_gather: what the secondaries copied into the primary for the last frame
} M;

// selects device d of the session for the calling thread; the primary is re-selected when the scope ends
struct Use {
    explicit Use(int d) { select(d); }
    ~Use() { if (M.n > 1) select(0); }
    static void select(int d) { if (M.n > d && M.dev[d]) { if (M.n > 1) (void)hipSetDevice(M.hipDevice[d]); G = M.dev[d]; } else G = nullptr; }
};
#define NEED_SESSION() do { if (M.n == 0) return CRT_E_NOT_INITIALIZED; } while (0)
// run `expr` on every device of the session (scene uploads, resize, ...); first error wins
#define ON_ALL(expr) do { NEED_SESSION(); int rc_ = CRT_OK; for (int d_ = 0; d_ < M.n; ++d_) { Use u_(d_); const int r_ = (expr); if (r_ != CRT_OK && rc_ == CRT_OK) rc_ = r_; } return rc_; } while (0)
#define ON_PRIMARY(expr) do { NEED_SESSION(); Use u_(0); return (expr); } while (0)

static void destroy_group()
{
    for (int d = 1; d < M.n; ++d) if (M.worker[d]) { M.worker[d]->stop(); delete M.worker[d]; M.worker[d] = nullptr; }
    for (int d = 0; d < M.n; ++d) {
        if (!M.dev[d]) continue;
        if (M.n > 1) (void)hipSetDevice(M.hipDevice[d]);
        G = M.dev[d];
        release_all();
        delete M.dev[d]; M.dev[d] = nullptr;
    }
    G = nullptr; M.n = 0; M.seq = 0; M.broken = false; M.injectFailure = -1; M.lastGatherBytes = 0; M.lastGatherBpp = 0; M.pack8W = M.pack8H = 0;
    for (int& p : M.peer) p = 0;
}

} // namespace
