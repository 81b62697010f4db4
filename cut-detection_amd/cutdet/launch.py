"""One process per GPU, started from a parent that never touches a GPU.

The reference is a single process on a single device (segment_video.py:25-29).  Sharding a clip
(or a set of clips) over the GPUs of a node needs one rank process per GPU; this module starts
them as fresh children - before anything in the parent has initialised the GPU runtime, and never by
replacing a running process - hands each its rank through the environment, relays rank 0's
standard output and turns the ranks' exit codes into one.

Only the standard library is imported here, so a parent can use it without loading torch or the
native library.

Rendezvous without a process group: `FileBarrier` lets ranks that share no collective (one clip per
GPU) line up their timed regions through files in a directory the parent created.
"""
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def in_rank_process():
    """True when a launcher (this module or torch.distributed.run) started this process as a rank."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_info():
    """(rank, local_rank, world) from the environment; (0, 0, 1) outside a launch."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def spawn_ranks(n, argv, extra_env=None, stdout_filter=None):
    """Starts n rank processes `python -u argv...`, waits for them, returns the job's exit code: 0
    only if every rank exits 0; the first failing rank's code otherwise (the others are stopped).

    Every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT and
    CUTDET_RUN_DIR (a directory shared by the ranks of this job, removed afterwards).  Rank 0's
    standard output is relayed line by line: lines for which `stdout_filter(line)` is false go to
    standard error instead (default: everything is relayed to standard output); the other ranks'
    standard output is dropped, their standard error is inherited.

    HSA_ENABLE_IPC_MODE_LEGACY=0 is set for the children unless the caller's environment already
    sets it: the pool's host driver supports only dmabuf IPC, and without it RCCL's buffer exchange
    between rank processes fails in hipIpcGetMemHandle."""
    env = dict(os.environ)
    run_dir = tempfile.mkdtemp(prefix="cutdet_run_")
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               CUTDET_RUN_DIR=run_dir)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-u"] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))

    def relay():
        for line in procs[0].stdout:
            text = line.decode("utf-8", "replace")
            out = sys.stdout if stdout_filter is None or stdout_filter(text) else sys.stderr
            out.write(text)
            out.flush()

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.05)
    for p in procs:                      # the exact processes started above, nothing else
        if p.poll() is None:
            p.kill()
        p.wait()
    reader.join(timeout=10)
    try:
        for name in os.listdir(run_dir):
            os.unlink(os.path.join(run_dir, name))
        os.rmdir(run_dir)
    except OSError:
        pass
    if failed is not None:
        sys.stderr.write(f"{os.path.basename(argv[0])}: rank {failed[0]} exited with code {failed[1]}\n")
        return failed[1] if failed[1] > 0 else 1
    return 0


class FileBarrier:
    """Barrier and small result exchange between the ranks of one job through files in
    CUTDET_RUN_DIR - no process group, no collective, no network.  For jobs whose ranks share no
    data (one clip per GPU) but whose timed regions must start together."""

    def __init__(self, rank, world, run_dir=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.dir = run_dir or os.environ.get("CUTDET_RUN_DIR")
        if self.world > 1 and not self.dir:
            raise RuntimeError("FileBarrier needs CUTDET_RUN_DIR (ranks started by cutdet.launch.spawn_ranks) or "
                               "an explicit directory shared by the ranks")
        self._gen = 0

    def _path(self, tag, rank):
        return os.path.join(self.dir, f"{tag}.{rank}")

    def wait(self):
        """Returns once every rank has called wait() the same number of times."""
        if self.world == 1:
            return
        tag = f"barrier{self._gen}"
        self._gen += 1
        with open(self._path(tag, self.rank), "w"):
            pass
        deadline = time.monotonic() + self.timeout
        for r in range(self.world):
            p = self._path(tag, r)
            while not os.path.exists(p):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not reach {tag} within {self.timeout:.0f} s")
                time.sleep(0.0005)

    def exchange(self, tag, text):
        """Every rank contributes one string; returns the list of all ranks' strings, by rank."""
        if self.world == 1:
            return [text]
        tmp = self._path(tag + ".tmp", self.rank)
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, self._path(tag, self.rank))          # readers never see a half-written file
        out = []
        deadline = time.monotonic() + self.timeout
        for r in range(self.world):
            p = self._path(tag, r)
            while not os.path.exists(p):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not publish {tag} within {self.timeout:.0f} s")
                time.sleep(0.0005)
            with open(p) as f:
                out.append(f.read())
        return out
