// Development aid, NOT part of the product (see include/hip/hip_runtime.h in this directory).
//
// Lock-step wavefront emulator: a workgroup runs as one fiber per work-item on one OS thread.  A fiber runs until it
// reaches a cross-lane operation and parks there with its operands and its program position: the chain of call sites,
// outermost first, each as (source line, column) looked up from the debug info by an llvm-symbolizer child process (the
// build uses -g -fno-inline -fno-omit-frame-pointer; machine-code layout does not follow source order, line numbers do).  When
// every live lane of a wavefront is parked, the lanes at the LOWEST program position form the group the hardware would
// execute that instruction with (its EXEC mask): lanes inside a divergent branch or still inside a loop are served before
// the lanes that wait behind it, which is how structured control flow reconverges.  The operation is evaluated for the
// group (lanes outside it count as inactive: a shuffle from one of them reads 0, a DPP read keeps `old`) and the group
// runs on.  __syncthreads releases when every live work-item of the workgroup is parked at one.
// Anything else -- lanes parked beyond a barrier others have not passed, a workgroup that cannot make progress -- aborts
// with a dump, which is the point: those are the bugs that reset a GPU.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

namespace emu {

Idx thread_idx, block_idx, block_dim, grid_dim;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");

enum State { RUNNABLE = 0, WAITING = 1, DONE = 2 };
constexpr int MAXPOS = 40;
struct Lane {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    int kind = 0;
    uint64_t a = 0, b = 0, c = 0, d = 0, result = 0;
    uintptr_t pos[MAXPOS];
    int npos = 0;
    unsigned tid = 0;
};

static constexpr size_t STACK_BYTES = 512 * 1024;
static std::vector<Lane> lanes;
static char* stack_pool = nullptr;
static size_t stack_pool_lanes = 0;
static void* sched_sp = nullptr;
static Lane* cur = nullptr;
static const std::function<void()>* body = nullptr;
static unsigned long long n_collectives = 0;

unsigned lane_id() { return cur ? (cur->tid & 63u) : 0u; }

static void lane_entry() {
    (*body)();
    cur->state = DONE;
    emu_switch(&cur->sp, sched_sp);
    abort();
}

static int cmp_pos(const Lane& x, const Lane& y);
static void die(const char* what) {
    std::fprintf(stderr, "[emu] %s (block %u,%u,%u)\n", what, block_idx.x, block_idx.y, block_idx.z);
    for (size_t i = 0; i < lanes.size(); ++i) {
        const Lane& l = lanes[i];
        if (l.state == DONE) continue;
        std::fprintf(stderr, "  lane %3zu state %d kind %d pos", i, l.state, l.kind);
        if (i > 0 && lanes[i - 1].state != DONE && cmp_pos(lanes[i - 1], l) == 0) { std::fprintf(stderr, " (same)\n"); continue; }
        for (int k = 0; k < l.npos; ++k) if (l.pos[k]) std::fprintf(stderr, " %lu:%lu", (unsigned long)(l.pos[k] >> 16), (unsigned long)(l.pos[k] & 0xffff));
        std::fprintf(stderr, "\n");
    }
    abort();
}

// (line << 16 | column) of the call whose return address is `ret`; 0 for code outside this library
static uintptr_t site_key(uintptr_t ret) {
    static std::unordered_map<uintptr_t, uintptr_t> cache;
    static FILE *to_child = nullptr, *from_child = nullptr;
    auto it = cache.find(ret);
    if (it != cache.end()) return it->second;
    uintptr_t key = 0;
    Dl_info info, self;
    if (dladdr(reinterpret_cast<void*>(ret), &info) && dladdr(reinterpret_cast<void*>(&site_key), &self) && info.dli_fbase == self.dli_fbase) {
        if (!to_child) {
            int in_pipe[2], out_pipe[2];
            if (pipe(in_pipe) || pipe(out_pipe)) { std::perror("pipe"); abort(); }
            const pid_t pid = fork();
            if (pid == 0) {
                dup2(in_pipe[0], 0); dup2(out_pipe[1], 1);
                close(in_pipe[1]); close(out_pipe[0]);
                const std::string obj = std::string("--obj=") + self.dli_fname;
                const char* sym = getenv("EMU_SYMBOLIZER") ? getenv("EMU_SYMBOLIZER") : "/opt/rocm/lib/llvm/bin/llvm-symbolizer";
                execl(sym, sym, obj.c_str(), "--no-inlines", "-f=none", (char*)nullptr);
                _exit(127);
            }
            close(in_pipe[0]); close(out_pipe[1]);
            to_child = fdopen(in_pipe[1], "w"); from_child = fdopen(out_pipe[0], "r");
        }
        std::fprintf(to_child, "0x%lx\n", (unsigned long)(ret - 1 - reinterpret_cast<uintptr_t>(info.dli_fbase)));
        std::fflush(to_child);
        char line[1024];
        unsigned long ln = 0, col = 0;
        while (std::fgets(line, sizeof line, from_child)) {
            if (line[0] == '\n') break;
            std::string t(line);
            while (!t.empty() && (t.back() == '\n' || t.back() == '\r')) t.pop_back();
            const size_t c2 = t.rfind(':'), c1 = c2 == std::string::npos ? c2 : t.rfind(':', c2 - 1);
            if (c1 != std::string::npos) { ln = std::strtoul(t.c_str() + c1 + 1, nullptr, 10); col = std::strtoul(t.c_str() + c2 + 1, nullptr, 10); }
        }
        key = (uintptr_t(ln) << 16) | uintptr_t(col & 0xffff);
    }
    cache.emplace(ret, key);
    return key;
}

uint64_t collective(int kind, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    if (!cur) {              // host code calling a device helper outside a launch: single lane semantics
        switch (kind) {
            case K_BALLOT: return a ? 1 : 0;
            case K_SYNC: return 0;
            default: return a;
        }
    }
    Lane* me = cur;
    me->kind = kind; me->a = a; me->b = b; me->c = c; me->d = d;
    // program position: return addresses, outermost frame first
    uintptr_t tmp[MAXPOS];
    int n = 0;
    uintptr_t* fp = reinterpret_cast<uintptr_t*>(__builtin_frame_address(0));
    const uintptr_t lo = reinterpret_cast<uintptr_t>(me->stack), hi = lo + STACK_BYTES;
    while (fp && n < MAXPOS) {
        const uintptr_t f = reinterpret_cast<uintptr_t>(fp);
        if (f < lo || f + 16 > hi) break;
        tmp[n++] = site_key(fp[1]);
        fp = reinterpret_cast<uintptr_t*>(fp[0]);
    }
    me->npos = n;
    for (int i = 0; i < n; ++i) me->pos[i] = tmp[n - 1 - i];
    me->state = WAITING;
    ++n_collectives;
    emu_switch(&me->sp, sched_sp);
    return me->result;
}

static int cmp_pos(const Lane& x, const Lane& y) {
    const int n = x.npos < y.npos ? x.npos : y.npos;
    for (int i = 0; i < n; ++i)
        if (x.pos[i] != y.pos[i]) return x.pos[i] < y.pos[i] ? -1 : 1;
    return x.npos == y.npos ? 0 : (x.npos < y.npos ? -1 : 1);
}

static void resolve(Lane* w, const std::vector<int>& g) {
    // g: lane numbers (0..63) of the group inside wave w[0..63]
    bool in[64] = {};
    for (int l : g) in[l] = true;
    const int kind = w[g[0]].kind;
    switch (kind) {
        case K_BALLOT: {
            uint64_t m = 0;
            for (int l : g) if (w[l].a) m |= 1ull << l;
            for (int l : g) w[l].result = m;
            break;
        }
        case K_SHFL:
            for (int l : g) { const int s = (int)(w[l].b & 63u); w[l].result = in[s] ? w[s].a : 0; }
            break;
        case K_SHFL_XOR:
            for (int l : g) { const int s = (l ^ (int)w[l].b) & 63; w[l].result = in[s] ? w[s].a : 0; }
            break;
        case K_READLANE:
            for (int l : g) { const int s = (int)(w[l].b & 63u); w[l].result = w[s].a; }     // reads the register whatever EXEC says
            break;
        case K_READFIRST: {
            const uint64_t v = w[g[0]].a;
            for (int l : g) w[l].result = v;
            break;
        }
        case K_DPP:
            for (int l : g) {
                const uint32_t old = (uint32_t)w[l].a, ctrl = (uint32_t)w[l].c, rm = (uint32_t)w[l].d & 15u, bm = ((uint32_t)w[l].d >> 8) & 15u;
                const bool bc = ((uint32_t)w[l].d >> 16) & 1u;
                const int row = l >> 4, r = l & 15;
                if (!((rm >> row) & 1u) || !((bm >> (r >> 2)) & 1u)) { w[l].result = old; continue; }
                int s = -1;
                if (ctrl <= 0xff) s = (l & ~3) | (int)((ctrl >> (2 * (l & 3))) & 3u);                        // quad_perm
                else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = (int)ctrl - 0x100; if (r + n < 16) s = l + n; }      // row_shl
                else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = (int)ctrl - 0x110; if (r - n >= 0) s = l - n; }      // row_shr
                else if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = (int)ctrl - 0x120; s = (row << 4) | ((r - n) & 15); } // row_ror
                else if (ctrl == 0x140) s = (row << 4) | (15 - r);                                           // row_mirror
                else if (ctrl == 0x141) s = (l & ~7) | (7 - (l & 7));                                        // row_half_mirror
                else if (ctrl == 0x142) { if (row >= 1) s = ((row - 1) << 4) | 15; }                          // row_bcast:15
                else if (ctrl == 0x143) { if (l >= 32) s = 31; }                                             // row_bcast:31
                else die("unsupported DPP control");
                if (s < 0 || !in[s]) w[l].result = bc ? 0u : old;
                else w[l].result = (uint32_t)w[s].b;
            }
            break;
        default: die("unknown collective");
    }
    for (int l : g) w[l].state = RUNNABLE;
}

static void run_block(unsigned n_threads) {
    if (lanes.size() < n_threads) lanes.resize(n_threads);
    if (stack_pool_lanes < n_threads) {
        if (stack_pool) munmap(stack_pool, stack_pool_lanes * STACK_BYTES);
        stack_pool = static_cast<char*>(mmap(nullptr, size_t(n_threads) * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (stack_pool == MAP_FAILED) { std::perror("mmap"); abort(); }
        stack_pool_lanes = n_threads;
    }
    for (unsigned i = 0; i < n_threads; ++i) {
        Lane& l = lanes[i];
        l.stack = stack_pool + size_t(i) * STACK_BYTES;
        l.tid = i;
        l.state = RUNNABLE;
        uintptr_t top = (reinterpret_cast<uintptr_t>(l.stack) + STACK_BYTES) & ~uintptr_t(15);
        uintptr_t* sp = reinterpret_cast<uintptr_t*>(top) - 8;        // r15 r14 r13 r12 rbx rbp ret pad  (sp % 16 == 0)
        for (int k = 0; k < 6; ++k) sp[k] = 0;
        sp[6] = reinterpret_cast<uintptr_t>(&lane_entry);
        sp[7] = 0;
        l.sp = sp;
    }
    unsigned done = 0;
    std::vector<int> group;
    while (done < n_threads) {
        for (unsigned i = 0; i < n_threads; ++i) {
            Lane& l = lanes[i];
            if (l.state != RUNNABLE) continue;
            cur = &l;
            thread_idx.x = i; thread_idx.y = 0; thread_idx.z = 0;
            emu_switch(&sched_sp, l.sp);
            if (l.state == DONE) ++done;
        }
        cur = nullptr;
        if (done == n_threads) break;
        bool any = false, all_sync = true;
        for (unsigned w0 = 0; w0 < n_threads; w0 += 64) {
            Lane* w = &lanes[w0];
            const unsigned wn = n_threads - w0 < 64 ? n_threads - w0 : 64;
            int best = -1;
            for (unsigned l = 0; l < wn; ++l) {
                if (w[l].state != WAITING) continue;
                if (best < 0 || cmp_pos(w[l], w[best]) < 0) best = (int)l;
            }
            if (best < 0) continue;
            group.clear();
            bool others = false;
            for (unsigned l = 0; l < wn; ++l) {
                if (w[l].state != WAITING) continue;
                if (cmp_pos(w[l], w[best]) == 0) { if (w[l].kind != w[best].kind) die("two operations at one program position"); group.push_back((int)l); }
                else others = true;
            }
            if (w[best].kind == K_SYNC) {
                if (others) die("lanes of a wavefront are parked beyond a barrier that other lanes have not passed (divergent __syncthreads)");
                continue;
            }
            all_sync = false;
            resolve(w, group);
            any = true;
        }
        if (!any) {
            if (!all_sync) die("no progress");
            for (unsigned i = 0; i < n_threads; ++i) if (lanes[i].state == WAITING) { if (lanes[i].kind != K_SYNC) die("no progress at a barrier"); lanes[i].state = RUNNABLE; }
        }
    }
    cur = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& f) {
    if (block.y != 1 || block.z != 1) { std::fprintf(stderr, "[emu] only 1-D workgroups\n"); abort(); }
    body = &f;
    grid_dim = Idx{grid.x, grid.y, grid.z};
    block_dim = Idx{block.x, 1, 1};
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                block_idx = Idx{x, y, z};
                run_block(block.x);
            }
    body = nullptr;
}

}  // namespace emu

unsigned long long emu_clock() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- host runtime ---------------------------------------------------------------------------------------------------------
struct emu_stream { int id; };
struct emu_event { double t_ms; };
static double now_ms() { return (double)emu_clock() * 1e-6; }
hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int, hipStream_t) {
    for (size_t y = 0; y < h; ++y) memcpy(static_cast<char*>(d) + y * dp, static_cast<const char*>(s) + y * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { *st = new emu_stream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t_ms = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "emulated"; }
