// fake_rccl.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A stand-in for the eight librccl entry points the engine's layer-pipeline transport resolves with dlopen
// (csrc/engine.hip pipe_open: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString), so that rwkv_pipe_decode / rwkv_pipe_prefill EXECUTE with world > 1 on a box with ONE
// GPU: real RCCL refuses two ranks on one device, and a gpurun box has one.  Point RWKV_RCCL_LIB at the built library
// (tests/_build/libfake_rccl.so, rwkv-cpp-accelerated_amd/build.py build_test_helpers); every rank is its own process.
//
// Transport: one POSIX shared-memory segment per communicator (its name travels in the 128-byte unique id), one channel per
// (source, destination) pair: a ring of SLOTS staging slots with a produced / consumed sequence pair.  All of it STREAM
// ORDERED like the real calls, and -- like the real calls -- every wait is made ON THE DEVICE, in the stream:
//   send  = a one-lane kernel that waits for room in the ring, hipMemcpyAsync device -> slot (the segment is hipHostRegister'ed),
//           then a host function that publishes the slot
//   recv  = a one-lane kernel that waits for the slot's sequence number, hipMemcpyAsync slot -> device, then a host function that
//           frees it
// No host function ever blocks: HIP runs the host functions of all of a process's streams on one thread, so a blocking wait in one
// stream's callback held back the publishing callbacks of the process's OTHER streams -- with two communicators on two streams per
// rank (rwkv_pipe_decode_dual) two ranks could wait for each other that way until the time-out (seen once in a full GPU-suite run).
// Inside ncclGroupStart / ncclGroupEnd the operations are queued and issued at GroupEnd, sends first -- a send never waits for
// its peer unless SLOTS messages of that channel are unconsumed -- so a group that sends to one peer and receives from another
// cannot deadlock, which is the property of an RCCL group the engine's tick relies on.
// Bounded waits (60 s of the device's wall clock; 120 s for the ranks to join): a lost peer ends the test with an error instead of hanging the box.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int SLOTS = 4;
constexpr size_t SLOT_BYTES = 3u << 20;       // largest message: a 64-row pass of the residual stream at D = 5120 (2.5 MiB)
constexpr int MAX_RANKS = 8;
constexpr int WAIT_MS = 60000;       // a peer's data or room in the ring (device wall clock)
constexpr int JOIN_MS = 120000;      // all ranks inside ncclCommInitRank (a rank may still be loading its share of the model)

struct Channel {
    std::atomic<uint64_t> produced, consumed;
    uint64_t bytes[SLOTS];
    char pad[64];
};
struct Segment {
    std::atomic<uint32_t> joined, left;
    uint32_t nranks;
    char pad[52];
    Channel ch[MAX_RANKS][MAX_RANKS];          // [src][dst]
    // followed by the slots: [src][dst][SLOTS][SLOT_BYTES]
};
size_t seg_bytes(int n) { return sizeof(Segment) + (size_t)n * n * SLOTS * SLOT_BYTES; }

struct Comm {
    Segment *seg = nullptr;
    size_t bytes = 0;
    int rank = 0, nranks = 1;
    std::string name;
    uint64_t sent[MAX_RANKS] = {}, recvd[MAX_RANKS] = {};     // messages enqueued so far per peer
    Segment *seg_dev = nullptr;                                // the segment as the device addresses it
    int *failed = nullptr;                                     // pinned host word, written by a wait kernel that gave up (or a size mismatch)
    uint64_t wait_ticks = 0;                                   // WAIT_MS in ticks of the device's wall clock
    bool has_failed() const { return __atomic_load_n(failed, __ATOMIC_ACQUIRE) != 0; }
    char *slot(int src, int dst, uint64_t seq) const
    {
        return reinterpret_cast<char *>(seg) + sizeof(Segment) + ((((size_t)src * nranks + dst) * SLOTS) + seq % SLOTS) * SLOT_BYTES;
    }
};

struct Op { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

struct Note { Comm *c; int src, dst; uint64_t seq; size_t bytes; int what; };   // what: 0 publish, 1 check the size of what arrived, 2 free

// stream-ordered wait for *flag >= least (a sequence number another PROCESS advances in the shared segment): one lane polling host memory
__global__ void k_wait(const uint64_t *flag, uint64_t least, int *failed, uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < least) {
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
        if (wall_clock64() - t0 > ticks) { __hip_atomic_store(failed, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(127);
    }
}
void host_note(void *p)
{
    Note *n = static_cast<Note *>(p);
    Channel &ch = n->c->seg->ch[n->src][n->dst];
    switch (n->what) {
    case 0: ch.bytes[n->seq % SLOTS] = n->bytes; ch.produced.store(n->seq + 1, std::memory_order_release); break;
    case 1:
        if (ch.produced.load(std::memory_order_acquire) < n->seq + 1) {
            if (__atomic_load_n(n->c->failed, __ATOMIC_ACQUIRE) == 2)
                fprintf(stderr, "[fake_rccl] rank %d: peer %d did not show up within %d ms\n", n->c->rank, n->src, WAIT_MS);
        } else if (ch.bytes[n->seq % SLOTS] != n->bytes) {
            fprintf(stderr, "[fake_rccl] rank %d: message %llu from %d has %llu bytes, receiver asked for %zu\n", n->c->rank,
                    (unsigned long long)n->seq, n->src, (unsigned long long)ch.bytes[n->seq % SLOTS], n->bytes);
            __atomic_store_n(n->c->failed, 1, __ATOMIC_RELEASE);
        }
        break;
    case 2: ch.consumed.store(n->seq + 1, std::memory_order_release); break;
    }
    delete n;
}
int enqueue(const Op &o)
{
    Comm *c = o.c;
    if (o.bytes > SLOT_BYTES || o.peer < 0 || o.peer >= c->nranks) return 4;   // ncclInvalidArgument (a hop to the rank itself is allowed, as in NCCL >= 2.7 inside a group: the send is enqueued first)
    if (o.send) {
        const uint64_t seq = c->sent[o.peer]++;
        if (seq >= (uint64_t)SLOTS)
            k_wait<<<dim3(1), dim3(1), 0, o.st>>>(reinterpret_cast<const uint64_t *>(&c->seg_dev->ch[c->rank][o.peer].consumed), seq - SLOTS + 1, c->failed, c->wait_ticks);
        if (hipMemcpyAsync(c->slot(c->rank, o.peer, seq), o.buf, o.bytes, hipMemcpyDeviceToHost, o.st) != hipSuccess) return 1;
        if (hipLaunchHostFunc(o.st, host_note, new Note{c, c->rank, o.peer, seq, o.bytes, 0}) != hipSuccess) return 1;
    } else {
        const uint64_t seq = c->recvd[o.peer]++;
        k_wait<<<dim3(1), dim3(1), 0, o.st>>>(reinterpret_cast<const uint64_t *>(&c->seg_dev->ch[o.peer][c->rank].produced), seq + 1, c->failed, c->wait_ticks);
        if (hipLaunchHostFunc(o.st, host_note, new Note{c, o.peer, c->rank, seq, o.bytes, 1}) != hipSuccess) return 1;
        if (hipMemcpyAsync(o.buf, c->slot(o.peer, c->rank, seq), o.bytes, hipMemcpyHostToDevice, o.st) != hipSuccess) return 1;
        if (hipLaunchHostFunc(o.st, host_note, new Note{c, o.peer, c->rank, seq, o.bytes, 2}) != hipSuccess) return 1;
    }
    if (hipGetLastError() != hipSuccess) return 1;
    return c->has_failed() ? 2 : 0;
}
size_t type_bytes(int t)
{
    switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}
int submit(bool send, void *buf, size_t count, int type, int peer, void *comm, hipStream_t st)
{
    const size_t tb = type_bytes(type);
    if (!comm || !buf || !tb) return 4;
    Op o{send, buf, count * tb, peer, static_cast<Comm *>(comm), st};
    if (g_depth > 0) { g_queue.push_back(o); return 0; }
    return enqueue(o);
}

} // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<unsigned> counter{0};
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%u_%llx", (int)getpid(), counter++,
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks; c->name.assign(id.internal, strnlen(id.internal, sizeof(id.internal)));
    c->bytes = seg_bytes(nranks);
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return 2; }      // zero-filled by the kernel: all counters start at 0
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return 2; }
    c->seg = static_cast<Segment *>(p);
    if (hipHostRegister(p, c->bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { munmap(p, c->bytes); delete c; return 1; }
    void *dp = nullptr;
    int dev = 0, khz = 0;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || hipHostMalloc(reinterpret_cast<void **>(&c->failed), 64, hipHostMallocMapped) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
        (void)hipHostUnregister(p); munmap(p, c->bytes); delete c; return 1;
    }
    c->seg_dev = static_cast<Segment *>(dp);
    *c->failed = 0;
    c->wait_ticks = (uint64_t)khz * (uint64_t)WAIT_MS;
    c->seg->nranks = (uint32_t)nranks;
    c->seg->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->seg->joined.load() < (uint32_t)nranks) {       // like the real call: returns once every rank has joined
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(JOIN_MS)) {
            (void)hipHostUnregister(p); (void)hipHostFree(c->failed); munmap(p, c->bytes); shm_unlink(c->name.c_str()); delete c;
            return 2;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 4;
    (void)hipDeviceSynchronize();
    const bool last = c->seg->left.fetch_add(1) + 1 == (uint32_t)c->nranks;
    (void)hipHostUnregister(c->seg);
    (void)hipHostFree(c->failed);
    munmap(c->seg, c->bytes);
    if (last) shm_unlink(c->name.c_str());
    delete c;
    return 0;
}

int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t st) { return submit(true, const_cast<void *>(buf), count, type, peer, comm, st); }
int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t st) { return submit(false, buf, count, type, peer, comm, st); }
int ncclGetVersion(int *v) { if (v) *v = 1; return 0; }      // (1 = this stand-in; real RCCL reports e.g. 22204)
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd()
{
    if (g_depth <= 0) return 4;
    if (--g_depth > 0) return 0;
    int rc = 0;
    for (int pass = 0; pass < 2; pass++)                      // sends first: they do not wait for their peer
        for (const Op &o : g_queue)
            if (o.send == (pass == 0) && !rc) rc = enqueue(o);
    g_queue.clear();
    return rc;
}
const char *ncclGetErrorString(int rc)
{
    switch (rc) { case 0: return "no error"; case 1: return "fake_rccl: HIP call failed"; case 2: return "fake_rccl: peer missing / shared memory failure";
                  case 4: return "fake_rccl: invalid argument"; default: return "fake_rccl: error"; }
}

} // extern "C"
