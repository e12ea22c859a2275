// A test double for the five RCCL entry points sbwtgpu_index_bcast() uses (ncclCommInitAll, ncclGroupStart, ncclBroadcast,
// ncclGroupEnd, ncclCommDestroy), loaded through SBWTGPU_RCCL_LIB: it lets the single-process call sequence of
// sbwt_amd/csrc/sbwtgpu_capi.cpp (communicators for all ranks, one grouped broadcast on per-rank streams, clean-up on
// every way out) run on a box with ONE GPU, several ranks on the same device (SBWTGPU_BCAST_NO_DEDUP=1), which RCCL itself
// refuses.  A broadcast is one hipMemcpyAsync per non-root rank, issued at ncclGroupEnd like RCCL issues its kernels.
// RCCL_STANDIN_FAIL=init|bcast|groupend makes that call fail (the second ncclBroadcast of a group for "bcast": the first
// rank's call has been recorded by then); rccl_standin_live_comms() counts communicators not yet destroyed.
// Test infrastructure only: what it does NOT exercise is RCCL's transport (xGMI rings, IPC handles, its own streams).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
struct Comm { int rank, n, dev; };
struct Op { const void *send; void *recv; size_t bytes; int root; Comm *comm; hipStream_t stream; };
std::vector<Op> g_ops;
int g_depth = 0, g_live = 0, g_bcasts = 0;
bool fail_at(const char *what) {
    const char *f = getenv("RCCL_STANDIN_FAIL");
    return f && !strcmp(f, what);
}
int run_ops() {
    int rc = 0;
    for (const Op &o : g_ops) {
        if (o.comm->rank == o.root) continue;
        const void *src = nullptr;
        for (const Op &r : g_ops)
            if (r.comm->rank == o.root && r.root == o.root) src = r.send;
        if (!src) { rc = 1; continue; }                       // the root never joined the group
        if (hipSetDevice(o.comm->dev) != hipSuccess ||
            hipMemcpyAsync(o.recv, src, o.bytes, hipMemcpyDefault, o.stream) != hipSuccess) rc = 1;
    }
    g_ops.clear();
    return rc;
}
}  // namespace

extern "C" {
int ncclCommInitAll(void **comms, int n, const int *devs) {
    if (fail_at("init")) return 1;
    for (int i = 0; i < n; i++) { comms[i] = new Comm{i, n, devs[i]}; g_live++; }
    return 0;
}
int ncclGroupStart(void) { g_depth++; return 0; }
int ncclBroadcast(const void *send, void *recv, size_t count, int /*datatype: ncclChar*/, int root, void *comm, hipStream_t stream) {
    if (fail_at("bcast") && ++g_bcasts >= 2) { g_bcasts = 0; return 1; }
    g_ops.push_back(Op{send, recv, count, root, static_cast<Comm *>(comm), stream});
    return g_depth > 0 ? 0 : run_ops();
}
int ncclGroupEnd(void) {
    if (g_depth > 0) g_depth--;
    if (g_depth > 0) return 0;
    if (fail_at("groupend")) { g_ops.clear(); return 1; }
    return run_ops();
}
int ncclCommDestroy(void *comm) { delete static_cast<Comm *>(comm); g_live--; return 0; }
int rccl_standin_live_comms(void) { return g_live; }
}
