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


DEFAULT_LAUNCH_TIMEOUT = 900.0      # seconds; bench.py --launch-timeout


def _tail(path, n=25):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode("utf-8", "replace")
    except OSError:
        return ""


def _run_ranks(n, argv, base, timeout, log_dir, attempt):
    """One set of fresh rank processes.  Returns (exit status, ranks that failed, True if every rank had passed its first
    barrier).  Rank r's stderr goes to <log_dir>/bench_rank<r>.err (rank 0's stdout, the JSON line, stays the launcher's)."""
    import os
    import socket
    import subprocess
    import sys
    import time
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(base)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MIRT_LAUNCH_ATTEMPT=str(attempt))
    ready = [os.path.join(log_dir, f"bench_rank{r}.ready") for r in range(n)]
    errs = [os.path.join(log_dir, f"bench_rank{r}.err") for r in range(n)]
    for f in ready:
        if os.path.exists(f):
            os.remove(f)
    logs = [open(f, "wb") for f in errs]
    procs = [subprocess.Popen(list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r), MIRT_READY_FILE=ready[r]), stderr=logs[r]) for r in range(n)]
    rc, t0, failed = 0, time.monotonic(), []
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    failed.append(r)
                    if rc == 0:
                        rc = code
            if rc != 0:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc, failed = 124, sorted(live)
                print(f"[launcher] attempt {attempt}: no result after {timeout:.0f} s, stopping ranks {failed}", file=sys.stderr)
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
        for f in logs:
            f.close()
    all_ready = all(os.path.exists(f) for f in ready)
    for f in ready:
        if os.path.exists(f):
            os.remove(f)
    if rc != 0:
        for r in failed[:2]:
            print(f"[launcher] attempt {attempt}: rank {r} ended with status {procs[r].returncode}; last lines of {errs[r]}:\n{_tail(errs[r])}", file=sys.stderr)
    else:
        for r in range(n):                                             # a clean run: what the ranks wrote to stderr follows the JSON line
            t = _tail(errs[r], 200)
            if t:
                print(t, file=sys.stderr)
    return rc, failed, all_ready


def passed_first_barrier():
    """Called by a rank once its process group works (after the first collective): tells the launcher that a later failure is
    not a failure to start."""
    import os
    f = os.environ.get("MIRT_READY_FILE")
    if f:
        open(f, "w").close()


def launch_local_ranks(n, argv, env=None, timeout=DEFAULT_LAUNCH_TIMEOUT, log_dir=None):
    """Starts `n` ranks of `argv` on this node, one fresh process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, as torch.distributed.run sets them), and waits for them, at most `timeout` seconds
    (None: for ever).  The caller must not have touched a GPU: the ranks are children started with subprocess, never an exec
    of this process.  Rank 0's stdout is the launcher's; every rank's stderr goes to <log_dir>/bench_rank<r>.err (default:
    the current directory) and the failing rank's last lines are printed.  If a rank fails the others are stopped (by pid).

    If the ranks fail BEFORE their first barrier (passed_first_barrier), i.e. the process group never came up, the launcher
    -- which has not touched a GPU -- starts ONE more set of fresh children with HSA_ENABLE_IPC_MODE_LEGACY unset and
    NCCL_DEBUG=WARN: the first environment was chosen on a one-GPU box and may not suit every host.  The ranks see the
    attempt number in MIRT_LAUNCH_ATTEMPT (bench.py reports it as config.launch_attempts).
    Returns the exit status of the last attempt: the first non-zero status of a rank, 124 on timeout, or 0."""
    import os
    import sys
    if n < 1:
        raise ValueError("need at least one rank")
    log_dir = os.path.abspath(log_dir or os.getcwd())
    os.makedirs(log_dir, exist_ok=True)
    base = dict(os.environ if env is None else env)
    first = dict(base)
    first.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # dmabuf IPC: what RCCL needs on these hosts
    rc, failed, all_ready = _run_ranks(n, argv, first, timeout, log_dir, 1)
    if rc == 0 or all_ready or rc == 124:
        return rc
    second = dict(base)
    second.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    second["NCCL_DEBUG"] = "WARN"
    print("[launcher] the ranks failed before their first barrier; one more set of fresh processes with "
          "HSA_ENABLE_IPC_MODE_LEGACY unset and NCCL_DEBUG=WARN", file=sys.stderr)
    rc, _, _ = _run_ranks(n, argv, second, timeout, log_dir, 2)
    return rc
