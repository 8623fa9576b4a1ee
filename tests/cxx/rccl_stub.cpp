// rccl_stub.cpp -- a TEST-ONLY stand-in for librccl.so.1 (tests/test_gpu_group_rccl_stub.py builds it into a directory that a
// SUBPROCESS puts first on its loader path).  It lets ptmi_group_gather_color (csrc/ptmi_group.cpp) run its real n > 1 branch --
// ncclCommInitAll over the group's devices, grouped ncclSend / ncclRecv on per-member streams, the stitch on the root -- on a box
// that has ONE physical GPU: several members share the device, which real RCCL refuses.
//
// What it validates: the HOST LOGIC of the gather -- offsets, counts, roots other than 0, members that hold no rows, a second gather
// reusing the communicators, an error between ncclGroupStart and ncclGroupEnd.  What it does NOT validate: RCCL itself, xGMI, or
// any timing.  tests/test_group.py::test_group_gathers_between_two_physical_devices stays armed for the first multi-GPU box.
//
// Semantics implemented (those the gather relies on): inside one group, the k-th ncclSend from rank a to rank b pairs with the k-th
// ncclRecv on rank b from rank a; counts must agree; at ncclGroupEnd every pair becomes one hipMemcpyAsync on the RECEIVER's stream
// that waits for the sender's stream, and the sender's stream then waits for the copy (a send is complete, in stream order, once its
// buffer may be reused).  Sends and receives outside a group are refused (real RCCL would block).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <mutex>
#include <vector>

namespace {

struct Comm { int rank, nranks, device; };
struct Op { bool send; int self, peer; void *buf; size_t bytes; hipStream_t stream; };

std::mutex g_mu;
int g_depth = 0;
std::vector<Op> g_ops;
int g_fail_send_in = 0;        // > 0: the k-th ncclSend from now fails once
long g_counters[8] = {0};      // 0 CommInitAll calls, 1 communicators made, 2 sends, 3 recvs, 4 groups ended, 5 copies issued, 6 CommDestroy calls, 7 bytes copied

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_counters[0];
    for (int i = 0; i < ndev; ++i) {
        comms[i] = reinterpret_cast<ncclComm_t>(new Comm{i, ndev, devlist ? devlist[i] : i});     // duplicate devices are fine HERE
        ++g_counters[1];
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_counters[6];
    delete reinterpret_cast<Comm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const Comm *c = reinterpret_cast<const Comm *>(comm);
    if (!c || !buf || peer < 0 || peer >= c->nranks || type_bytes(type) == 0) return ncclInvalidArgument;
    if (g_depth == 0) return ncclInvalidUsage;
    if (g_fail_send_in > 0 && --g_fail_send_in == 0) return ncclInternalError;
    ++g_counters[2];
    g_ops.push_back(Op{true, c->rank, peer, const_cast<void *>(buf), count * type_bytes(type), stream});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_mu);
    const Comm *c = reinterpret_cast<const Comm *>(comm);
    if (!c || !buf || peer < 0 || peer >= c->nranks || type_bytes(type) == 0) return ncclInvalidArgument;
    if (g_depth == 0) return ncclInvalidUsage;
    ++g_counters[3];
    g_ops.push_back(Op{false, c->rank, peer, buf, count * type_bytes(type), stream});
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth == 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ++g_counters[4];
    std::vector<Op> ops;
    ops.swap(g_ops);
    std::vector<bool> used(ops.size(), false);
    ncclResult_t result = ncclSuccess;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (!ops[i].send) continue;
        size_t j = 0;
        for (; j < ops.size(); ++j)                          // the first unused receive on the peer that names this sender
            if (!used[j] && !ops[j].send && ops[j].self == ops[i].peer && ops[j].peer == ops[i].self) break;
        if (j == ops.size() || ops[j].bytes != ops[i].bytes) { result = ncclInvalidUsage; continue; }     // a send nobody receives: real RCCL hangs
        used[i] = used[j] = true;
        hipEvent_t sent = nullptr, copied = nullptr;
        if (hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventCreateWithFlags(&copied, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(sent); return ncclUnhandledCudaError; }
        bool ok = hipEventRecord(sent, ops[i].stream) == hipSuccess && hipStreamWaitEvent(ops[j].stream, sent, 0) == hipSuccess;
        ok = ok && hipMemcpyAsync(ops[j].buf, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToDevice, ops[j].stream) == hipSuccess;
        ok = ok && hipEventRecord(copied, ops[j].stream) == hipSuccess && hipStreamWaitEvent(ops[i].stream, copied, 0) == hipSuccess;
        (void)hipEventDestroy(sent); (void)hipEventDestroy(copied);      // (destruction is deferred until the events have completed)
        if (!ok) return ncclUnhandledCudaError;
        ++g_counters[5];
        g_counters[7] += (long)ops[i].bytes;
    }
    for (size_t j = 0; j < ops.size(); ++j)
        if (!ops[j].send && !used[j]) result = ncclInvalidUsage;          // a receive nobody sends to
    return result;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (rccl stub)";
    case ncclUnhandledCudaError: return "unhandled HIP error (rccl stub)";
    case ncclInternalError: return "internal error (rccl stub: injected)";
    case ncclInvalidArgument: return "invalid argument (rccl stub)";
    case ncclInvalidUsage: return "invalid usage (rccl stub: unpaired send / receive, or outside a group)";
    default: return "error (rccl stub)";
    }
}

// ---- the test's handles on the stub ----
void rccl_stub_counters(long out[8]) { std::lock_guard<std::mutex> lock(g_mu); for (int i = 0; i < 8; ++i) out[i] = g_counters[i]; }
void rccl_stub_fail_send(int kth) { std::lock_guard<std::mutex> lock(g_mu); g_fail_send_in = kth; }
int rccl_stub_is_the_stub(void) { return 1; }

}  // extern "C"
