"""Self-launch of the data-parallel ranks (one process per GPU) for a script run as plain `python3 script.py --gpus N`.

The reference is single-GPU (tools/options.py:295); data parallelism is this build's own capability, so its entry points
have to start their own ranks.  `launch_ranks` is called by a parent that has NOT touched the GPU (importing torch is
fine, any HIP call is not): it starts N CHILD processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
set (the environment `parallel.init_from_env` reads, the same one `torch.distributed.run` would give), relays rank 0's
stdout line by line (the JSON line of bench.py), and returns the worst child exit code.  Nothing is exec'ed and no
process that has initialised the GPU is ever replaced.  If a rank dies, the others are given a grace period (they may be
blocked in a rendezvous or a collective that can no longer complete) and are then terminated by PID.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rc(code):
    """Exit status of a child as a shell would report it (a signal -> 128 + signal number)."""
    return 128 - code if code < 0 else code


def launch_ranks(script, argv, nproc, env=None, out=None, grace_s=30.0, timeout_s=None):
    """Start `nproc` ranks of `python script *argv`; relay rank 0's stdout to `out` (default: this process's stdout);
    other ranks' stdout goes to stderr.  Returns the worst exit code (0 only if every rank exited 0)."""
    if nproc < 1:
        raise ValueError("launch_ranks: nproc must be >= 1")
    out = out if out is not None else sys.stdout
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e, stdout=subprocess.PIPE,
                                      stderr=None, text=True, bufsize=1))

    def relay(p, dst, prefix):
        for line in p.stdout:
            dst.write(prefix + line)
            dst.flush()

    threads = [threading.Thread(target=relay, args=(p, out if r == 0 else sys.stderr, "" if r == 0 else f"[rank {r}] "),
                                daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    t0, first_fail = time.monotonic(), None
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if first_fail is None and any(p.poll() not in (None, 0) for p in procs):
            first_fail = now
        expired = (first_fail is not None and now - first_fail > grace_s) or (timeout_s is not None and now - t0 > timeout_s)
        if expired:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5)
    return max(_rc(p.returncode) for p in procs)
