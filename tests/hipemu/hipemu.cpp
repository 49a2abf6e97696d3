// hipemu.cpp -- fiber scheduler for the SIMT emulator (see hipemu.h).  Test infrastructure only.
#include "hipemu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace hipemu {

thread_local ThreadCtx* cur = nullptr;

static const size_t kStack = 256 * 1024;

struct BlockRunner {
    int nthreads = 0, nwaves = 0;
    std::vector<ucontext_t> fibers;
    std::vector<ThreadCtx> ctx;
    std::vector<char> done;
    std::vector<char*> stacks;
    ucontext_t sched;
    int alive = 0;
    // block barrier
    int barr = 0;
    unsigned bgen = 0;
    // wave barriers + exchange buffers
    std::vector<int> warr;
    std::vector<unsigned> wgen;
    std::vector<uint64_t> xbuf;  // [wave][2][64]
    const std::function<void()>* body = nullptr;

    void ensure(int n) {
        while ((int)stacks.size() < n) {
            void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
            stacks.push_back((char*)p);
        }
        fibers.resize(n);
        ctx.resize(n);
        done.assign(n, 0);
        nthreads = n;
        nwaves = (n + 63) / 64;
        warr.assign(nwaves, 0);
        wgen.assign(nwaves, 0);
        xbuf.assign((size_t)nwaves * 2 * 64, 0);
    }
};

static thread_local BlockRunner* tl_runner = nullptr;

static void yield_() {
    ThreadCtx* me = cur;
    BlockRunner* R = me->runner;
    swapcontext(&R->fibers[me->flat], &R->sched);
}

void sync_block() {
    BlockRunner* R = cur->runner;
    unsigned g = R->bgen;
    if (++R->barr >= R->alive) {
        R->barr = 0;
        R->bgen++;
    } else {
        while (R->bgen == g) yield_();
    }
}

void sync_wave() {
    BlockRunner* R = cur->runner;
    int w = cur->wave;
    int wsize = std::min(64, R->nthreads - w * 64);
    unsigned g = R->wgen[w];
    if (++R->warr[w] == wsize) {
        R->warr[w] = 0;
        R->wgen[w]++;
    } else {
        while (R->wgen[w] == g) yield_();
    }
}

uint64_t* wave_buf(int slot) {
    BlockRunner* R = cur->runner;
    return &R->xbuf[((size_t)cur->wave * 2 + slot) * 64];
}

static void trampoline() {
    ThreadCtx* me = cur;
    BlockRunner* R = me->runner;
    (*R->body)();
    R->done[me->flat] = 1;
    R->alive--;
    // a finished thread no longer participates in barriers (matches s_barrier semantics)
    if (R->barr > 0 && R->barr >= R->alive) {
        R->barr = 0;
        R->bgen++;
    }
    swapcontext(&R->fibers[me->flat], &R->sched);
}

static void run_block(BlockRunner* R, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz,
                      const std::function<void()>& body) {
    int n = block.x * block.y * block.z;
    R->ensure(n);
    R->body = &body;
    R->alive = n;
    R->barr = 0;
    for (int t = 0; t < n; ++t) {
        ThreadCtx& c = R->ctx[t];
        c.tid.x = t % block.x;
        c.tid.y = (t / block.x) % block.y;
        c.tid.z = t / (block.x * block.y);
        c.bid.x = bx; c.bid.y = by; c.bid.z = bz;
        c.bdim = block;
        c.gdim = grid;
        c.flat = t;
        c.lane = t & 63;
        c.wave = t >> 6;
        c.runner = R;
        getcontext(&R->fibers[t]);
        R->fibers[t].uc_stack.ss_sp = R->stacks[t];
        R->fibers[t].uc_stack.ss_size = kStack;
        R->fibers[t].uc_link = nullptr;
        makecontext(&R->fibers[t], (void (*)())trampoline, 0);
    }
    while (R->alive > 0) {
        for (int t = 0; t < n; ++t) {
            if (R->done[t]) continue;
            cur = &R->ctx[t];
            swapcontext(&R->sched, &R->fibers[t]);
        }
    }
    cur = nullptr;
}

// ---------------------------------------------------------------- worker pool
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    bool stop = false;
    uint64_t epoch = 0;
    int working = 0;
    dim3 grid, block;
    const std::function<void()>* body = nullptr;
    std::atomic<uint64_t> next{0};
    uint64_t total = 0;

    Pool() {
        int n = (int)std::thread::hardware_concurrency();
        const char* e = getenv("HIPEMU_THREADS");
        if (e) n = atoi(e);
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void loop() {
        BlockRunner R;
        tl_runner = &R;
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
            }
            for (;;) {
                uint64_t b = next.fetch_add(1);
                if (b >= total) break;
                unsigned bx = b % grid.x, by = (b / grid.x) % grid.y, bz = b / ((uint64_t)grid.x * grid.y);
                run_block(&R, grid, block, bx, by, bz, *body);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--working == 0) cv_done.notify_all();
            }
        }
    }
    void run(dim3 g, dim3 b, const std::function<void()>& f) {
        std::unique_lock<std::mutex> lk(mu);
        grid = g; block = b; body = &f;
        total = (uint64_t)g.x * g.y * g.z;
        next = 0;
        working = (int)th.size();
        epoch++;
        cv.notify_all();
        cv_done.wait(lk, [&] { return working == 0; });
    }
};

static Pool& pool() {
    static Pool* p = new Pool();  // leaked on purpose: avoids static-destruction order issues at exit
    return *p;
}

static std::mutex launch_mu;

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    std::lock_guard<std::mutex> lk(launch_mu);
    if ((uint64_t)grid.x * grid.y * grid.z == 0) return;
    pool().run(grid, block, body);
}

}  // namespace hipemu
