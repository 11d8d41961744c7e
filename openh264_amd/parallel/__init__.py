"""Multi-GPU execution of the hot path: one process per GPU, independent sessions sharded over ranks.

The path has no exchange step (sessions / simulcast layers are independent, slices stay
on one GPU), so there is NO data-path collective: `torch.distributed` is used only for the start/stop
barrier and for gathering per-rank results (timings, bitstream digests).  On MI355X nodes the backend
is "nccl" (= RCCL); the CPU test tier runs the same code over "gloo".
"""
import hashlib


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of n_items over world ranks -> (first, count) for `rank`."""
    base, extra = divmod(n_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def encode_sessions_sharded(make_group, session_inputs, frames, rank=0, world=1, dist=None):
    """Encode `len(session_inputs)` independent sessions, sharded over ranks.

    make_group(n) -> EncoderGroup with n sessions (ring_slots >= frames).
    session_inputs[s] -> list of `frames` I420 frame byte strings for session s.
    Returns (on every rank) the list of SHA1 hex digests of all sessions' bitstreams, in session order.
    """
    first, count = shard_range(len(session_inputs), rank, world)
    digests = []
    if count:
        g = make_group(count)
        for s in range(count):
            for f in range(frames):
                g.upload(s, f, session_inputs[first + s][f])
        streams = [bytearray() for _ in range(count)]
        for f in range(frames):
            for s, bs in enumerate(g.step(f)):
                streams[s] += bs
        digests = [hashlib.sha1(bytes(b)).hexdigest() for b in streams]
        g.close()
    if dist is None or world == 1:
        return digests
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    return [d for part in gathered for d in part]
