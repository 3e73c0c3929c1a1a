"""The two collectives am_evaluate_sharded_f32 asks its caller for (include/audio_metrics_hip.h: am_collectives), over
torch.distributed - the Python counterpart of csrc/rccl/am_rccl.cpp, which builds the same hooks over an ncclComm_t for hosts
that are not Python.  The library hands the hooks raw device pointers; every buffer it names lies in a tensor this process
allocated (the call's workspace), which `expose` registers so that a pointer can be turned back into a tensor view."""
import contextlib
import ctypes

import torch
import torch.distributed as dist

from . import _lib

COLL_F64, COLL_I32 = 0, 1


class TorchCollectives:
    def __init__(self, group=None):
        self.group = group
        active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if active else 0
        self.world = dist.get_world_size(group) if active else 1
        # a group of ONE rank still runs its collectives for real when asked to (tests: everything of the RCCL path but the wire)
        self.force = False
        self._active = active
        self._tensors = []
        self.error = None
        self.calls = []                                     # (name, bytes) of every hook call, in issue order
        self.gather_form = None                             # which torch.distributed call the last all_gather_v became
        self._reduce = _lib.ALL_REDUCE_SUM_FN(self._all_reduce_sum)
        self._gather = _lib.ALL_GATHER_V_FN(self._all_gather_v)
        self.struct = _lib.CollectivesStruct(None, self.rank, self.world, self._reduce, self._gather)

    def expose(self, tensor):
        self._tensors.append(tensor)
        return tensor

    def _view(self, ptr, nbytes):
        for t in self._tensors:
            base = t.data_ptr()
            if base <= ptr and ptr + nbytes <= base + t.numel() * t.element_size():
                flat = t.view(-1).view(torch.uint8)
                return flat[ptr - base:ptr - base + nbytes]
        raise RuntimeError(f"collective hook: pointer {ptr:#x} (+{nbytes}) lies in no exposed tensor")

    def _in_place_ok(self, tensor):
        """input = a slice of the output: RCCL's in-place all-gather (ncclAllGather with sendbuff = recvbuff + rank * count)"""
        return tensor.is_cuda and dist.get_backend(self.group) == "nccl"

    def _on(self, stream_ptr, device):
        # torch orders a collective after the work of its CURRENT stream: make that the stream the library named
        if device.type != "cuda":
            return contextlib.nullcontext()                  # (host tensors: the hooks' own unit test)
        if stream_ptr:
            return torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr, device=device))
        return torch.cuda.stream(torch.cuda.current_stream(device))

    def _all_reduce_sum(self, ctx, buf, count, dtype, stream):
        try:
            dt, size = (torch.float64, 8) if dtype == COLL_F64 else (torch.int32, 4)
            self.calls.append(("all_reduce_sum", count * size))
            if self.world == 1 and not (self.force and self._active):
                return 0
            view = self._view(buf, count * size).view(dt)
            with self._on(stream, view.device):
                dist.all_reduce(view, group=self.group)
            return 0
        except Exception as exc:                            # an exception must not cross the C frames above
            self.error = exc
            return 1

    def _all_gather_v(self, ctx, send, recv, bytes_per_rank, stream):
        try:
            sizes = [int(bytes_per_rank[r]) for r in range(self.world)]
            self.calls.append(("all_gather_v", sum(sizes)))
            if (self.world == 1 and not (self.force and self._active)) or sum(sizes) == 0:
                return 0
            whole = self._view(recv, sum(sizes))
            offs = [sum(sizes[:r]) for r in range(self.world)]
            if send != recv + offs[self.rank]:
                raise RuntimeError("all_gather_v: the library promises an in-place call")
            mine = whole[offs[self.rank]:offs[self.rank] + sizes[self.rank]]
            with self._on(stream, whole.device):
                if len(set(sizes)) == 1 and self._in_place_ok(whole):
                    # equal shares on RCCL: ONE all-gather straight into the library's buffer, the rank's own share already at
                    # its offset - what csrc/rccl/am_rccl.cpp does with ncclAllGather (no staging copy, no copy back)
                    self.gather_form = "all_gather_into_tensor, in place"
                    dist.all_gather_into_tensor(whole, mine, group=self.group)
                elif len(set(sizes)) == 1:
                    # gloo (the CPU tests, ranks sharing one GPU): the same call from a copy of the own share
                    self.gather_form = "all_gather_into_tensor, own share copied"
                    dist.all_gather_into_tensor(whole, mine.clone(), group=self.group)
                else:
                    # unequal shares: one broadcast per rank into its slice, in place (am_rccl.cpp: the same, inside a group call)
                    self.gather_form = "broadcast per rank, in place"
                    for r in range(self.world):
                        if sizes[r]:
                            src = dist.get_global_rank(self.group, r) if self.group is not None else r
                            dist.broadcast(whole[offs[r]:offs[r] + sizes[r]], src=src, group=self.group)
            return 0
        except Exception as exc:
            self.error = exc
            return 1

    def byref(self):
        return ctypes.byref(self.struct)
