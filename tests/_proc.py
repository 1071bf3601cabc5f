"""Child processes of the test suite: never outlive the test, whatever happens to it.

Round 4's driver run lost 247 of 249 GPU tests to one infrastructure test: subprocess.run's
timeout killed only the direct child, its grandchildren (torchrun workers in sessions of their
own) stayed on the GPU, and the next GPU user shared the device with them.  Here every child
  * leads its own process group (start_new_session) -- a timeout SIGKILLs the whole group,
    and the group is swept once more after a normal exit;
  * dies with its parent (PR_SET_PDEATHSIG = SIGKILL), so killing pytest itself -- the
    driver's step limit -- takes the children along (bench.py's rank launcher gives its ranks
    the same property, so the chain reaches every process that holds the GPU).
"""
import ctypes
import os
import signal
import subprocess
import time

PR_SET_PDEATHSIG = 1
SUBPROCESS_TIMEOUT = 240   # default limit of one child; the driver's step limit is 1200 s


def die_with_parent():
    """preexec_fn: SIGKILL this process when the thread that forked it exits."""
    try:
        ctypes.CDLL(None, use_errno=True).prctl(PR_SET_PDEATHSIG, signal.SIGKILL, 0, 0, 0)
    except Exception:
        pass


def _sweep(pgid):
    try:
        os.killpg(pgid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        return
    # wait until the group is gone (zombies of our own child are reaped by communicate/wait)
    for _ in range(100):
        try:
            os.killpg(pgid, 0)
        except (ProcessLookupError, PermissionError):
            return
        time.sleep(0.05)


class Result:
    def __init__(self, returncode, stdout, stderr, timed_out):
        self.returncode, self.stdout, self.stderr, self.timed_out = returncode, stdout, stderr, timed_out


def run(cmd, env=None, timeout=SUBPROCESS_TIMEOUT, cwd=None, merge_stderr=False):
    """subprocess.run(capture_output, text) for a child that may start processes of its own.
    On timeout the whole group is killed and `timed_out` is set (returncode -9); the output so
    far is returned either way.

    The group is swept BEFORE the child is reaped: while the leader is still a zombie its pid --
    and with it the process-group id -- cannot be handed to anybody else, so the SIGKILL can only
    reach our own stragglers (swept after the reap, a recycled id could belong to a stranger)."""
    import threading
    p = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT if merge_stderr else subprocess.PIPE, text=True,
                         start_new_session=True, preexec_fn=die_with_parent)
    bufs = {}

    def reader(name, fh):
        try:
            bufs[name] = fh.read()
        except (OSError, ValueError):
            bufs.setdefault(name, "")

    threads = [threading.Thread(target=reader, args=("out", p.stdout), daemon=True)]
    if not merge_stderr:
        threads.append(threading.Thread(target=reader, args=("err", p.stderr), daemon=True))
    for t in threads:
        t.start()
    deadline = time.time() + timeout
    timed_out = True
    while time.time() < deadline:
        try:   # has it exited?  (WNOWAIT: look, do not reap)
            info = os.waitid(os.P_PID, p.pid, os.WEXITED | os.WNOWAIT | os.WNOHANG)
        except ChildProcessError:
            info = True
        if info is not None:
            timed_out = False
            break
        time.sleep(0.02)
    _sweep(p.pid)        # stragglers of a child that exited by itself, or everything on a timeout
    p.wait()             # only now is the id free for reuse
    for t in threads:    # every writer of the pipes is dead: EOF
        t.join(timeout=10)
    out, err = bufs.get("out", ""), bufs.get("err", "")
    err = err or ""
    if timed_out:
        err += f"\n[tests/_proc.py] killed the process group after {timeout} s\n"
    return Result(p.returncode if not timed_out else -9, out or "", err, timed_out)
