"""Multi-GPU plumbing for independent proof batches: one process per GPU, no data-path collective;
the only exchange is the per-proof aggregate (the transcript digest, an Fr element).

Fr is not an RCCL datatype.  Each rank leaves its digest as 8 zero-extended 64-bit lanes (one per
32-bit limb, zk_transcript_digest_dev); ranks all-reduce the lanes with SUM on int64 -- exact for
fewer than 2^31 ranks -- and every rank then carry-propagates and reduces mod r locally.
"""
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def lanes_to_fr(lanes):
    """8 (summed) 64-bit lanes -> the field element sum(lane_j * 2^(32 j)) mod r."""
    v = 0
    for j, x in enumerate(lanes):
        v += int(x) << (32 * j)
    return v % R_MOD


def fr_to_lanes(value):
    value %= R_MOD
    return [(value >> (32 * j)) & 0xFFFFFFFF for j in range(8)]


def allreduce_digest(lanes_tensor, group=None):
    """In-place SUM all-reduce of the 8-lane int64 tensor (nccl = RCCL over xGMI on GPUs, gloo on CPU).
    Returns the aggregate field element as a Python int."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(lanes_tensor, op=dist.ReduceOp.SUM, group=group)
    return lanes_to_fr(lanes_tensor.cpu().tolist())
