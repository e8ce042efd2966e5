// TEST DOUBLE of librccl — test infrastructure only, never a product fallback.
//
// libm17hip.so binds RCCL by dlopen("librccl.so.1") (m17-cxx-demod_amd/csrc/m17_gather.hpp).  The GPU tests' CHILD processes put this
// directory first in LD_LIBRARY_PATH, so that the product's N > 1 gather protocol (m17hip_gather_frames[_device]) runs between several
// processes that share the ONE device of a test box — something the real library refuses ("duplicate GPU").  It implements the
// symbols the product binds, between processes on one node:
//   rendezvous            POSIX shared memory keyed by the unique id
//   ncclAllGather         stream-ordered, host-staged (stream sync -> D2H -> shared memory -> barrier -> H2D -> barrier; every copy ON the caller's stream)
//   ncclSend / ncclRecv   one mailbox per (source, destination); the data of a message travels in a shared-memory segment of its own;
//                         a message whose size differs from what the receiver asked for is an ERROR here (the real library would
//                         corrupt or hang) — so a protocol whose two sides disagree about a count fails loudly
//   ncclGroupStart / End  collect the calls; at the end all sends are posted, then all receives served, then the sends waited for
//   ncclCommAbort         releases what this rank has hanging on its stream
// Every wait is bounded (M17_FAKE_RCCL_TIMEOUT_MS, default 10000).  What happens at a timeout is the test's choice:
//   M17_FAKE_RCCL_ON_TIMEOUT=error (default)  the call returns ncclSystemError
//   M17_FAKE_RCCL_ON_TIMEOUT=hang             the call returns ncclSuccess and leaves a kernel on the caller's stream that spins until
//                                             the communicator is aborted or destroyed (at most 40 s) — what the real library does
//                                             when a peer never shows up: the product's bounded wait has to deal with it
// M17_FAKE_RCCL_LOG=<prefix>: one line per operation into <prefix>.rank<r> (tests assert on the sequence of calls).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int MAXR = 16;
constexpr size_t AG_MAX = 4096;          // bytes per rank of one all-gather
constexpr uint32_t ID_MAGIC = 0x4D313746;   // "M17F"

struct Box { std::atomic<uint64_t> posted, taken, bytes; };
struct Shared {
    std::atomic<uint64_t> arrive[MAXR];  // barrier tickets: a rank's n-th barrier is passed when every rank has arrived n times
    std::atomic<uint32_t> joined, left;
    unsigned char ag[MAXR][AG_MAX];
    Box box[MAXR][MAXR];                 // [source][destination]
};

double now_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
void nap()
{
    timespec t{0, 20000};
    nanosleep(&t, nullptr);
}
double timeout_ms()
{
    const char* e = getenv("M17_FAKE_RCCL_TIMEOUT_MS");
    return e ? atof(e) : 10000.0;
}
bool hang_on_timeout()
{
    const char* e = getenv("M17_FAKE_RCCL_ON_TIMEOUT");
    return e && !strcmp(e, "hang");
}

// spins until the host releases it (abort / destroy) or 40 s of the 100 MHz clock have passed
__global__ void fake_rccl_wait_kernel(volatile uint32_t* flag)
{
    const uint64_t t0 = wall_clock64();
    while (!__atomic_load_n(flag, __ATOMIC_RELAXED) && wall_clock64() - t0 < 4000000000ull) __builtin_amdgcn_s_sleep(127);
}

// Every copy of the double is made ON the caller's stream and waited for there: a plain hipMemcpy works on the default stream, which orders nothing with a
// non-blocking stream — and a host-to-device copy from pageable memory may return before its last bytes have landed.  (Round 6: the product's streams became
// non-blocking ones of its own; one gather in a few thousand then read the PREVIOUS all-gather's words behind the double's hipMemcpy and answered ECOMM.)
hipError_t copy_on(hipStream_t st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; uint64_t seq; bool posted; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local struct ncclComm* g_group_comm = nullptr;

}  // namespace

struct ncclComm {
    Shared* sh = nullptr;
    int rank = 0, nranks = 1;
    char key[64] = {0};
    uint32_t* release = nullptr;   // pinned: the spin kernels of this rank watch it
    bool hung = false;             // a wait timed out in "hang" mode: nothing more is carried out
    FILE* log = nullptr;
    void say(const char* fmt, ...)
    {
        if (!log) return;
        va_list ap;
        va_start(ap, fmt);
        vfprintf(log, fmt, ap);
        va_end(ap);
        fputc('\n', log);
        fflush(log);
    }
    std::string seg_name(int src, int dst, uint64_t seq) const
    {
        char b[96];
        snprintf(b, sizeof(b), "/%s_%d_%d_%llu", key, src, dst, (unsigned long long)seq);
        return b;
    }
    // a wait that ran out: either an error, or (hang mode) success with a spinning kernel left on the stream
    ncclResult_t timed_out(const char* what, hipStream_t st)
    {
        fprintf(stderr, "FAKE_RCCL_TIMEOUT rank %d: %s\n", rank, what);
        say("timeout %s", what);
        if (!hang_on_timeout()) return ncclSystemError;
        hung = true;
        hipLaunchKernelGGL(fake_rccl_wait_kernel, dim3(1), dim3(1), 0, st, (volatile uint32_t*)release);
        return ncclSuccess;
    }
    template <typename F> bool wait_for(F&& cond)
    {
        const double t0 = now_ms(), lim = timeout_ms();
        while (!cond()) {
            if (now_ms() - t0 > lim) return false;
            nap();
        }
        return true;
    }
    bool barrier()
    {
        const uint64_t n = sh->arrive[rank].fetch_add(1, std::memory_order_acq_rel) + 1;
        return wait_for([&] {
            for (int k = 0; k < nranks; ++k)
                if (sh->arrive[k].load(std::memory_order_acquire) < n) return false;
            return true;
        });
    }
};

namespace {

ncclResult_t run_ops(ncclComm* c, std::vector<Op>& ops)
{
    if (c->hung) return ncclSuccess;
    // 1. post every send
    for (Op& o : ops) {
        if (!o.send) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        Box& b = c->sh->box[c->rank][o.peer];
        if (!c->wait_for([&] { return b.taken.load(std::memory_order_acquire) == b.posted.load(std::memory_order_acquire); }))
            return c->timed_out("send: the previous message to this peer was never taken", o.stream);
        o.seq = b.posted.load() + 1;
        const std::string name = c->seg_name(c->rank, o.peer, o.seq);
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)o.bytes) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
        void* p = mmap(nullptr, o.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return ncclSystemError;
        const hipError_t e = copy_on(o.stream, p, o.buf, o.bytes, hipMemcpyDeviceToHost);
        munmap(p, o.bytes);
        if (e != hipSuccess) return ncclUnhandledCudaError;
        b.bytes.store(o.bytes, std::memory_order_release);
        b.posted.store(o.seq, std::memory_order_release);
        o.posted = true;
        c->say("send peer=%d bytes=%zu seq=%llu", o.peer, o.bytes, (unsigned long long)o.seq);
    }
    // 2. serve every receive
    for (Op& o : ops) {
        if (o.send) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        Box& b = c->sh->box[o.peer][c->rank];
        if (!c->wait_for([&] { return b.posted.load(std::memory_order_acquire) > b.taken.load(std::memory_order_acquire); }))
            return c->timed_out("recv: nothing was sent", o.stream);
        const uint64_t seq = b.posted.load(std::memory_order_acquire);
        const size_t got = (size_t)b.bytes.load(std::memory_order_acquire);
        const std::string name = c->seg_name(o.peer, c->rank, seq);
        if (got != o.bytes) {
            fprintf(stderr, "FAKE_RCCL_SIZE_MISMATCH rank %d: recv of %zu bytes from %d met a send of %zu\n", c->rank, o.bytes, o.peer, got);
            c->say("mismatch peer=%d want=%zu got=%zu", o.peer, o.bytes, got);
            shm_unlink(name.c_str());
            b.taken.store(seq, std::memory_order_release);
            return ncclInvalidArgument;
        }
        const int fd = shm_open(name.c_str(), O_RDWR, 0600);
        if (fd < 0) return ncclSystemError;
        void* p = mmap(nullptr, got, PROT_READ, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return ncclSystemError;
        const hipError_t e = copy_on(o.stream, o.buf, p, got, hipMemcpyHostToDevice);
        munmap(p, got);
        shm_unlink(name.c_str());
        b.taken.store(seq, std::memory_order_release);
        if (e != hipSuccess) return ncclUnhandledCudaError;
        c->say("recv peer=%d bytes=%zu seq=%llu", o.peer, got, (unsigned long long)seq);
    }
    // 3. a send is complete when its receiver has taken it
    for (Op& o : ops) {
        if (!o.send || !o.posted) continue;
        Box& b = c->sh->box[c->rank][o.peer];
        if (!c->wait_for([&] { return b.taken.load(std::memory_order_acquire) >= o.seq; }))
            return c->timed_out("send: nobody received", o.stream);
    }
    return ncclSuccess;
}

ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm* c, hipStream_t st)
{
    const size_t sz = dtype_bytes(t);
    if (!c || !sz || peer < 0 || peer >= c->nranks || peer == c->rank || (count && !buf)) return ncclInvalidArgument;
    if (count == 0) return ncclSuccess;
    Op o{send, buf, count * sz, peer, st, 0, false};
    if (g_depth > 0) {
        if (g_group_comm && g_group_comm != c) return ncclInvalidUsage;   // (one communicator per group is all the product needs)
        g_group_comm = c;
        g_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_ops(c, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    uint32_t w[6] = {ID_MAGIC, (uint32_t)getpid(), 0, 0, 0, 0};
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    w[2] = (uint32_t)t.tv_nsec; w[3] = (uint32_t)t.tv_sec;
    static std::atomic<uint32_t> n{0};
    w[4] = n.fetch_add(1);
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&w[5], 4, 1, f) != 1) w[5] = 0; fclose(f); }
    memcpy(id->internal, w, sizeof(w));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    uint32_t w[6];
    memcpy(w, id.internal, sizeof(w));
    if (w[0] != ID_MAGIC) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->nranks = nranks;
    snprintf(c->key, sizeof(c->key), "m17fakerccl_%08x%08x%08x%08x", w[1], w[2] ^ w[3], w[4], w[5]);
    const std::string name = std::string("/") + c->key;
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shared)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // (a fresh segment is all zeroes = the initial state)
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared*)p;
    if (hipHostMalloc((void**)&c->release, 64, hipHostMallocDefault) != hipSuccess) { munmap(p, sizeof(Shared)); delete c; return ncclUnhandledCudaError; }
    *c->release = 0;
    if (const char* pre = getenv("M17_FAKE_RCCL_LOG")) {
        const std::string path = std::string(pre) + ".rank" + std::to_string(rank);
        c->log = fopen(path.c_str(), "a");
    }
    c->sh->joined.fetch_add(1);
    c->say("init nranks=%d", nranks);
    if (!c->barrier()) {   // creation is collective
        fprintf(stderr, "FAKE_RCCL_TIMEOUT rank %d: communicator creation (%u of %d ranks came)\n", rank, c->sh->joined.load(), nranks);
        (void)hipHostFree(c->release); munmap(p, sizeof(Shared));
        if (c->log) fclose(c->log);
        delete c;
        return ncclSystemError;
    }
    *out = c;
    return ncclSuccess;
}

static void leave(ncclComm* c, bool complain)
{
    __atomic_store_n(c->release, 1u, __ATOMIC_RELEASE);   // whatever spins on this rank's streams ends
    (void)hipDeviceSynchronize();
    for (int k = 0; k < c->nranks; ++k) {
        if (k == c->rank) continue;
        Box& in = c->sh->box[k][c->rank];
        if (in.posted.load() > in.taken.load()) {   // a message nobody asked for
            if (complain) fprintf(stderr, "FAKE_RCCL_LEFTOVER rank %d: a message of %llu bytes from rank %d was never received\n", c->rank,
                                  (unsigned long long)in.bytes.load(), k);
            c->say("leftover from=%d bytes=%llu", k, (unsigned long long)in.bytes.load());
            shm_unlink(c->seg_name(k, c->rank, in.posted.load()).c_str());   // (not marked as taken: its sender goes on waiting, as with the real library)
        }
        Box& outb = c->sh->box[c->rank][k];
        if (outb.posted.load() > outb.taken.load()) {   // our own message that was never taken
            shm_unlink(c->seg_name(c->rank, k, outb.posted.load()).c_str());
            outb.taken.store(outb.posted.load());
        }
    }
    c->say(complain ? "destroy" : "abort");
    const uint32_t gone = c->sh->left.fetch_add(1) + 1;
    if (gone == (uint32_t)c->nranks) shm_unlink((std::string("/") + c->key).c_str());
    munmap(c->sh, sizeof(Shared));
    (void)hipHostFree(c->release);
    if (c->log) fclose(c->log);
    delete c;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    leave(c, true);
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    leave(c, false);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st)
{
    const size_t bytes = count * dtype_bytes(t);
    if (!c || !send || !recv || !bytes || bytes > AG_MAX) return ncclInvalidArgument;
    if (c->hung) return ncclSuccess;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (copy_on(st, c->sh->ag[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!c->barrier()) return c->timed_out("all-gather: not every rank came", st);
    for (int k = 0; k < c->nranks; ++k)
        if (copy_on(st, (char*)recv + (size_t)k * bytes, c->sh->ag[k], bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    c->say("allgather bytes=%zu", bytes);
    if (!c->barrier()) return c->timed_out("all-gather: not every rank read", st);
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    return p2p(true, const_cast<void*>(buf), count, t, peer, c, st);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) { return p2p(false, buf, count, t, peer, c, st); }

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclComm* c = g_group_comm;
    g_group_comm = nullptr;
    if (ops.empty() || !c) return ncclSuccess;
    return run_ops(c, ops);
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (fake rccl)" : "error (fake rccl)"; }

// lets a test make sure that it is this library the product bound
int m17_fake_rccl_marker() { return 0x4D313746; }

}  // extern "C"
