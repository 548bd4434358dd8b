"""Sharded (multi-GPU) evaluation: quadrature directions split contiguously over ranks and ONE sum all-reduce per
evaluation (RCCL over xGMI when the process group is "nccl") -- of the real Q by default (every rank transforms its
own partial sum), or of the complex partial Q_gain_hat with the tail replicated on every rank.

`op` is anything with gainPartial(f, stream) / finish(Q, f, stream) / finishPartial(Q, f, with_loss, stream) and,
optionally, collidePartial(Q, f, with_loss, stream) -- the HIP operator on a GPU box, or the host-emulated operator
in the world-size-2 gloo tests.  `qhat` is a tensor view of the operator's partial-sum buffer.
"""


def sharded_step(op, qhat, Q, f, dist=None, stream=0, reduce_spectral=False, async_op=False):
    """One collision evaluation on this rank's shard.  With dist=None it degenerates to the single-device path.

    Default route (reduce_spectral=False): every rank inverse-transforms its own partial Q_gain_hat (the transform
    is linear), rank 0 also subtracts the loss term, and the ONE collective sums the real Q (G doubles: half the
    bytes of Q_hat, and nothing runs after the collective).  reduce_spectral=True is the textbook route: sum the
    complex Q_gain_hat buffers (2G reals), then run the tail on every rank.

    async_op=True (default route only) returns the collective's work handle instead of waiting for it: nothing of
    this rank's next evaluation depends on the summed Q, so a caller that alternates between two Q buffers can let
    the all-reduce of evaluation i run (on RCCL's own stream) under the gain kernels of evaluation i+1, and calls
    handle.wait() before it reads or reuses that Q.  Returns None when there is nothing to wait for.
    """
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    if multi and not reduce_spectral:
        if hasattr(op, "collidePartial"):      # one call: the library fuses the slab reduce into the tail
            op.collidePartial(Q, f, dist.get_rank() == 0, stream)
        else:
            op.gainPartial(f, stream)
            op.finishPartial(Q, f, dist.get_rank() == 0, stream)
        work = dist.all_reduce(Q, async_op=async_op)      # the single collective of an evaluation (sum)
        return work if async_op else None
    op.gainPartial(f, stream)
    if multi:
        dist.all_reduce(qhat)
    op.finish(Q, f, stream)
    return None


def device_view(torch, ptr, n, precision):
    """Zero-copy torch tensor over a raw device pointer (the handle-owned Q_gain_hat buffer)."""
    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8" if precision == 64 else "<f4",
                                  "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(v, device="cuda")
