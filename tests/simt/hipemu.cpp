// TEST INFRASTRUCTURE ONLY -- see hipemu.h.
#include "hipemu.h"
#include <sys/mman.h>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

extern "C" void simt_swap(void **save_sp, void *load_sp);
asm(R"(
.text
.globl simt_swap
.type simt_swap,@function
simt_swap:
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
.size simt_swap,.-simt_swap
)");

namespace simt {

enum { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
static const size_t STACK_BYTES = 256 * 1024;
static const int MAX_THREADS = 1024;

struct Fiber {
    void *sp;
    int state;
    unsigned tx, ty, tz;
    int wave, lane;
};

struct Wave {
    uint64_t slots[2][64];
    uint64_t deposited[2];
    unsigned seq;
};

static Fiber fibers[MAX_THREADS];
static Wave waves[MAX_THREADS / 64];
static char *stack_pool;
static void *sched_sp;
static int cur = -1, nthreads, nwaves;
static std::function<void()> *cur_body;
static std::vector<char> dyn_smem;

static void yield_to_sched() { simt_swap(&fibers[cur].sp, sched_sp); }

static void fiber_main()
{
    (*cur_body)();
    fibers[cur].state = DONE;
    yield_to_sched();
    abort();
}

int lane_id() { return fibers[cur].lane; }
void *dynamic_shared() { return dyn_smem.data(); }

void sync_block()
{
    fibers[cur].state = WAIT_BLOCK;
    yield_to_sched();
}

const uint64_t *wave_exchange(uint64_t v, uint64_t *active_mask)
{
    Fiber &f = fibers[cur];
    Wave &w = waves[f.wave];
    unsigned par = w.seq & 1;
    w.slots[par][f.lane] = v;
    w.deposited[par] |= 1ull << f.lane;
    f.state = WAIT_WAVE;
    yield_to_sched();
    // released: scheduler advanced w.seq; our data is in the previous parity
    *active_mask = w.deposited[par];
    return w.slots[par];
}

static void run_block(std::function<void()> &body)
{
    cur_body = &body;
    nwaves = (nthreads + 63) / 64;
    for (int w = 0; w < nwaves; w++) { waves[w].deposited[0] = waves[w].deposited[1] = 0; waves[w].seq = 0; }
    for (int t = 0; t < nthreads; t++) {
        Fiber &f = fibers[t];
        f.state = RUNNABLE;
        f.tx = t % blockDim.x; f.ty = (t / blockDim.x) % blockDim.y; f.tz = t / (blockDim.x * blockDim.y);
        f.wave = t / 64; f.lane = t % 64;
        char *top = stack_pool + (size_t)(t + 1) * STACK_BYTES;
        void **sp = (void **)top;
        *--sp = nullptr;                     // fake return address of fiber_main's caller
        *--sp = (void *)&fiber_main;         // popped by simt_swap's ret
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        f.sp = sp;
    }
    int live = nthreads;
    while (live > 0) {
        bool progress = false;
        for (int t = 0; t < nthreads; t++) {
            Fiber &f = fibers[t];
            if (f.state != RUNNABLE) continue;
            cur = t;
            threadIdx.x = f.tx; threadIdx.y = f.ty; threadIdx.z = f.tz;
            simt_swap(&sched_sp, f.sp);
            progress = true;
            if (f.state == DONE) live--;
        }
        // wave-level releases
        for (int w = 0; w < nwaves; w++) {
            int lo = w * 64, hi = std::min(nthreads, lo + 64), waiting = 0, alive = 0;
            for (int t = lo; t < hi; t++) { if (fibers[t].state != DONE) alive++; if (fibers[t].state == WAIT_WAVE) waiting++; }
            if (alive && waiting == alive) {
                unsigned par = waves[w].seq & 1;
                waves[w].seq++;
                waves[w].deposited[(par ^ 1)] = 0;      // the next collective starts clean
                for (int t = lo; t < hi; t++) if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUNNABLE;
                progress = true;
            }
        }
        // block-level release
        int waiting = 0, alive = 0;
        for (int t = 0; t < nthreads; t++) { if (fibers[t].state != DONE) alive++; if (fibers[t].state == WAIT_BLOCK) waiting++; }
        if (alive && waiting == alive) {
            for (int t = 0; t < nthreads; t++) if (fibers[t].state == WAIT_BLOCK) fibers[t].state = RUNNABLE;
            progress = true;
        }
        if (!progress && live > 0) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u): divergent barrier/collective\n", blockIdx.x, blockIdx.y);
            for (int t = 0; t < nthreads; t++) if (fibers[t].state != DONE) fprintf(stderr, "  thread %d state %d\n", t, fibers[t].state);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body)
{
    if (dyn_smem.size() < shmem + 64) dyn_smem.resize(shmem + 64);
    if (!stack_pool) {
        stack_pool = (char *)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE,
                                  MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stack_pool == MAP_FAILED) abort();
    }
    nthreads = block.x * block.y * block.z;
    if (nthreads > MAX_THREADS) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    blockDim = block; gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                run_block(body);
            }
}

}  // namespace simt
