"""Multi-GPU sharding of the render loop: by samples per pixel.

Every camera sample is independent (the reference parallelises rows with no communication,
HomemadeRayTracer.cpp:107-110).  Rank g of G renders the global sample indices
[g*spp/G, (g+1)*spp/G) of every pixel into its own float32 radiance-sum buffer; the Philox counter
holds the global sample index, so the union of samples is the same for every G.  One sum-reduce of
the W*H*3 float buffers onto rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests) is the only
exchange; rank 0 resolves (mean, gamma, RGBA8).
"""


def spp_range(rank, world, spp):
    """[begin, begin+count) of the global sample indices owned by `rank`; ranges tile [0, spp)."""
    if not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank/world/spp")
    b = (spp * rank) // world
    e = (spp * (rank + 1)) // world
    return b, e - b


def render_sharded(render_range, reduce_to_root, rank, world, spp):
    """Host logic of a sharded render.  render_range(begin, count) -> this rank's partial radiance
    sum (any array type); reduce_to_root(buf) -> summed in place on rank 0.  Returns the buffer
    (complete on rank 0)."""
    b, n = spp_range(rank, world, spp)
    buf = render_range(b, n)
    if world > 1:
        reduce_to_root(buf)
    return buf
