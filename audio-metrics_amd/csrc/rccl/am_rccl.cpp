// am_collectives over an RCCL communicator: the hooks am_evaluate_sharded_f32 (include/audio_metrics_hip.h) asks its caller
// for, for one-process-per-GPU hosts that are not Python.  Built as its own small library (libaudio_metrics_rccl.so, links
// librccl) so that libaudio_metrics_hip.so itself depends on no collective library.
//
//   ncclComm_t comm = ...;                         // the host's communicator, one rank per GPU
//   am_collectives coll;
//   am_rccl_collectives(comm, &coll);              // rank / world from the communicator
//   am_evaluate_sharded_f32(ref_shard, ld, cand_shard, ld, D, ref_counts, cand_counts, &coll, AM_EVAL_FAD | AM_EVAL_KD | AM_EVAL_PRDC,
//                           k, idx_cand, idx_ref, S, m, 1.0 / D, 1.0, 3, out, ws, ws_bytes, stream, side_stream, comm_stream);
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include "../../../include/audio_metrics_hip.h"

namespace {

int reduce_hook(void* ctx, void* buf, int64_t count, int dtype, am_stream_t stream) {
    const ncclDataType_t t = dtype == AM_COLL_F64 ? ncclDouble : ncclInt32;
    return ncclAllReduce(buf, buf, (size_t)count, t, ncclSum, static_cast<ncclComm_t>(ctx), static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

// in place; equal shares = one ncclAllGather, unequal ones = one broadcast per rank inside a group (xGMI is point-to-point:
// each broadcast is the root's share over its own links)
int gather_hook(void* ctx, const void* send, void* recv, const int64_t* bytes_per_rank, am_stream_t stream) {
    ncclComm_t comm = static_cast<ncclComm_t>(ctx);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int world = 0;
    if (ncclCommCount(comm, &world) != ncclSuccess) return 1;
    bool equal = true;
    for (int r = 1; r < world; ++r) equal = equal && bytes_per_rank[r] == bytes_per_rank[0];
    if (equal) {
        if (bytes_per_rank[0] == 0) return 0;
        return ncclAllGather(send, recv, (size_t)bytes_per_rank[0], ncclChar, comm, st) == ncclSuccess ? 0 : 1;
    }
    if (ncclGroupStart() != ncclSuccess) return 1;
    int64_t off = 0;
    bool ok = true;
    for (int r = 0; r < world; ++r) {
        char* at = static_cast<char*>(recv) + off;
        if (bytes_per_rank[r] > 0) ok = ok && ncclBroadcast(at, at, (size_t)bytes_per_rank[r], ncclChar, r, comm, st) == ncclSuccess;
        off += bytes_per_rank[r];
    }
    return (ncclGroupEnd() == ncclSuccess && ok) ? 0 : 1;
}

}  // namespace

extern "C" int am_rccl_collectives(void* nccl_comm, am_collectives* out) {
    if (nccl_comm == nullptr || out == nullptr) return 1;
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int rank = 0, world = 0;
    if (ncclCommUserRank(comm, &rank) != ncclSuccess || ncclCommCount(comm, &world) != ncclSuccess) return 1;
    out->ctx = nccl_comm;
    out->rank = rank;
    out->world = world;
    out->all_reduce_sum = reduce_hook;
    out->all_gather_v = gather_hook;
    return 0;
}

// a communicator of ONE rank on the current device (tests on a 1-GPU box; a real host brings its own communicator)
extern "C" int am_rccl_comm_init_single(void** out_comm) {
    if (out_comm == nullptr) return 1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    ncclComm_t comm = nullptr;
    if (ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) return 1;
    *out_comm = comm;
    return 0;
}

extern "C" int am_rccl_comm_destroy(void* comm) { return ncclCommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? 0 : 1; }
