// tmat_gather_rows: the one collective of the path for callers that own an RCCL communicator: an
// all-gather of the 32-byte result rows over RCCL / xGMI.  The Python host (bench.py, scripts/compute_branches.py)
// issues the same collective through torch.distributed (backend "nccl" is RCCL on ROCm), which owns its communicator;
// this entry point is the C-ABI form of it.  RCCL is resolved at first use, so libtmat_hip.so has no link-time
// dependency on it (a process that already loaded an RCCL -- PyTorch bundles one -- reuses that copy).
#include "tmat_internal.h"
#include "../../include/tmat.h"

#include <dlfcn.h>

namespace {
typedef int (*all_gather_fn)(const void *, void *, size_t, int, void *, void *);   // ncclAllGather(send, recv, count, datatype, comm, stream)

all_gather_fn resolve()
{
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
        if (void *h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))
            if (void *f = dlsym(h, "ncclAllGather")) return (all_gather_fn)f;
    for (const char *n : names)
        if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))
            if (void *f = dlsym(h, "ncclAllGather")) return (all_gather_fn)f;
    return nullptr;
}
}  // namespace

extern "C" int tmat_gather_rows(void *rccl_comm, const tmat_row *rows_dev, int n_local, tmat_row *out_dev, void *hip_stream)
{
    if (!rccl_comm || !rows_dev || !out_dev || n_local < 0) { tmat::set_error("tmat_gather_rows: bad argument"); return TMAT_E_ARG; }
    if (n_local == 0) return TMAT_OK;
    static all_gather_fn fn = resolve();
    if (!fn) { tmat::set_error("tmat_gather_rows: RCCL (librccl.so) is not available"); return TMAT_E_HIP; }
    const int rc = fn(rows_dev, out_dev, (size_t)n_local * sizeof(tmat_row), /* ncclChar */ 0, rccl_comm, hip_stream);
    if (rc) { tmat::set_error("tmat_gather_rows: ncclAllGather failed with code " + std::to_string(rc)); return TMAT_E_HIP; }
    return TMAT_OK;
}
