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


def launch_local_ranks(n, argv, env=None, timeout=None):
    """Starts `n` ranks of `argv` on this node, one fresh process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, as torch.distributed.run sets them), and waits for them.  The caller must not have
    touched a GPU: the ranks are children started with subprocess, never an exec of this process.  stdout/stderr are
    inherited, so whatever rank 0 prints is the launcher's output.  If a rank fails the others are stopped (by pid).
    Returns the first non-zero exit status, or 0."""
    import os
    import socket
    import subprocess
    import time
    if n < 1:
        raise ValueError("need at least one rank")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # dmabuf IPC: what RCCL needs on these hosts
    procs = [subprocess.Popen(list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc, t0 = 0, time.monotonic()
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                                                # only on failure / timeout is anything still running
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc
