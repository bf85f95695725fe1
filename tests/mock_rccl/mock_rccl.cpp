// mock_rccl.cpp — TEST DOUBLE for librccl.so: the seven entry points the library's RCCL gather calls (vk_api.hip RcclApi), implemented
// with HIP events and device-to-device copies, so that the ORCHESTRATION of VK_SCENE_RCCL_GATHER — which stream sends, which receives,
// what the unpack kernels wait for, where part 0's slab is read — runs on a ONE-GPU box, where real RCCL refuses a device listed twice.
// Semantics kept: operations take effect in the order of the streams they are enqueued on; sends and receives issued inside one
// ncclGroupStart / ncclGroupEnd are matched pairwise (sender rank -> receiver rank, in order) and progress together; a send's buffer may
// be reused once the stream it was enqueued on has passed it.  Nothing of RCCL's transport is imitated.  Loaded through VK_RCCL_LIB by the
// DEBUG build of the library only (tests/test_gpu_abi2.py).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
struct MockComm { int rank, nranks, device; MockComm **all; };
typedef MockComm *ncclComm_t;

namespace {
struct Op { bool send; void *buf; size_t bytes; int peer; MockComm *comm; hipStream_t stream; };
std::mutex g_mu;
std::vector<Op> g_ops;
int g_depth = 0;
unsigned long long g_pairs = 0;

ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    std::vector<char> used(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); i++) {
        if (used[i] || !ops[i].send) continue;
        const Op &s = ops[i];
        size_t j = 0;
        for (; j < ops.size(); j++)
            if (!used[j] && !ops[j].send && ops[j].comm->rank == s.peer && ops[j].peer == s.comm->rank) break;
        if (j == ops.size() || ops[j].bytes != s.bytes) return ncclInvalidUsage;      // an unmatched send hangs in the real thing
        used[i] = used[j] = 1;
        const Op &r = ops[j];
        hipEvent_t sent = nullptr, received = nullptr;
        if (hipSetDevice(s.comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(sent, s.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipSetDevice(r.comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&received, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamWaitEvent(r.stream, sent, 0) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyPeerAsync(r.buf, r.comm->device, s.buf, s.comm->device, s.bytes, r.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(received, r.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipSetDevice(s.comm->device) != hipSuccess) return ncclUnhandledCudaError;
        if (hipStreamWaitEvent(s.stream, received, 0) != hipSuccess) return ncclUnhandledCudaError;     // the send "completes" with the transfer
        (void)hipEventDestroy(sent); (void)hipEventDestroy(received);      // (destroyed when the work that uses them has run)
        g_pairs++;
    }
    for (size_t i = 0; i < ops.size(); i++) if (!used[i]) return ncclInvalidUsage;          // a receive nobody sends to
    return ncclSuccess;
}
}  // namespace

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    MockComm **all = new MockComm *[ndev];
    for (int i = 0; i < ndev; i++) { comms[i] = new MockComm{i, ndev, devlist ? devlist[i] : i, all}; all[i] = comms[i]; }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    if (c->rank == c->nranks - 1) { /* (the table is leaked on purpose: ranks are destroyed in any order) */ }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> l(g_mu); g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    std::lock_guard<std::mutex> l(g_mu);
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush();
}
static ncclResult_t enqueue(bool send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    if (!buf || !c || peer < 0 || peer >= c->nranks || (t != ncclUint8 && t != ncclInt8)) return ncclInvalidArgument;
    std::lock_guard<std::mutex> l(g_mu);
    g_ops.push_back(Op{send, buf, count, peer, c, st});
    if (g_depth == 0) return ncclInvalidUsage;      // (this double only knows grouped point-to-point, which is what the library issues)
    return ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    return enqueue(true, const_cast<void *>(buf), count, t, peer, c, st);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    return enqueue(false, buf, count, t, peer, c, st);
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidUsage ? "mock: invalid usage (unmatched send / receive, or point-to-point outside a group)" : "mock: error"); }
// test hook: send / receive pairs executed so far
unsigned long long mock_rccl_pairs(void) { std::lock_guard<std::mutex> l(g_mu); return g_pairs; }

}  // extern "C"
