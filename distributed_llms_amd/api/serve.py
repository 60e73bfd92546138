"""Node-level launcher: several HTTP front-end processes on ONE port, each driving its own share of the GPUs.

    python -m distributed_llms_amd.api.serve --port 8000 --frontends 4        # 8 GPUs -> 2 GPUs per front end

Why: one Python front end ingests a 124 MB scanned PDF in ~0.16 s (socket, multipart, SHA-256), i.e. about
800 MB/s or ~650 pages/s of 300-DPI JPEG pages, while ONE MI355X already reads 580 pages/s - a single process
cannot feed eight GPUs.  The reference scales the same way one level up (N stateless replicas behind an ALB,
terraform/main.tf:176-294); this is that picture inside a node:

* the parent binds the listening socket once and hands the descriptor to every child (the kernel spreads
  ``accept()`` over them), supervises the children and restarts one that dies;
* child ``i`` gets ``OCR_GPU_IDS`` = its slice of the visible GPUs and a unique ``OCR_NODE_ID`` (orphan sweeps
  only touch its own documents);
* the front ends are stateless: any of them answers ``/check-status`` and ``/result`` for any document, so the
  stores must be shared - Redis + MongoDB as deployed, or, with nothing configured, an embedded RESP server in
  the parent (``store/resp.py``) plus the file result store under ``OCR_OUTPUT_DIR`` (``OCR_STORE=redis+file``).
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional


def split_gpus(n_gpus: int, frontends: int) -> List[List[int]]:
    """Contiguous slices (neighbouring devices share an xGMI hop / NUMA node); every front end gets >= 1 GPU when
    there are enough, otherwise the extra front ends run without one (text PDFs only need the CPU)."""
    out: List[List[int]] = [[] for _ in range(frontends)]
    if n_gpus <= 0:
        return out
    if n_gpus >= frontends:
        base, extra = divmod(n_gpus, frontends)
        g = 0
        for i in range(frontends):
            k = base + (1 if i < extra else 0)
            out[i] = list(range(g, g + k))
            g += k
    else:
        for g in range(n_gpus):
            out[g] = [g]
    return out


def count_gpus() -> int:
    """Device count WITHOUT touching the GPU runtime: this process forks and execs every front end and every
    supervised restart, and a process that has initialised HIP must not exec (``torch.cuda.device_count()`` falls
    back to ``hipGetDeviceCount`` when amdsmi is missing, which brings HIP up).  Order: ``OCR_NUM_GPUS`` /
    ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` from the environment, else the KFD topology in sysfs (GPU nodes
    are the ones with SIMDs)."""
    env = os.environ.get("OCR_NUM_GPUS", "").strip()
    if env.lstrip("-").isdigit() and int(env) >= 0:
        return int(env)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    return count_kfd_gpus()


def count_kfd_gpus(root: str = "/sys/class/kfd/kfd/topology/nodes") -> int:
    n = 0
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(line.split(None, 1) for line in f if " " in line)
                if int(props.get("simd_count", "0").strip()) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    return n


def child_main(fd: int) -> int:
    """One front end on the inherited listening socket.  ``OCR_HTTP_SERVER=uvicorn`` selects the stock server; the
    default is the in-repo one (api/httpd.py: ~2x cheaper polls, body bytes received straight into the upload buffer)."""
    from ..config import Settings
    from .app import create_app

    app = create_app(Settings.from_env(dotenv=os.environ.get("OCR_DOTENV", ".env")))
    if os.environ.get("OCR_HTTP_SERVER", "dlamd").lower() == "uvicorn":
        import uvicorn

        uvicorn.run(app, fd=fd, log_level=os.environ.get("OCR_LOG_LEVEL", "warning"))
    else:
        from . import httpd

        httpd.run(app, fd=fd)
    return 0


class Supervisor:
    def __init__(self, port: int, host: str, frontends: int, gpus: Optional[int] = None, embedded_redis: Optional[bool] = None):
        self.port, self.host, self.frontends = port, host, frontends
        self.n_gpus = count_gpus() if gpus is None else gpus
        self.slices = split_gpus(self.n_gpus, frontends)
        self.sock: Optional[socket.socket] = None
        self.children: Dict[int, subprocess.Popen] = {}
        self.redis = None
        self.stop = False
        want_embedded = embedded_redis if embedded_redis is not None else (
            not os.environ.get("REDIS_HOST") and not os.environ.get("MONGO_CONNECTION_STRING"))
        self.embedded = want_embedded
        self.restarts = 0

    def child_env(self, i: int) -> Dict[str, str]:
        env = dict(os.environ)
        ids = self.slices[i]
        env["OCR_GPU_IDS"] = ",".join(map(str, ids))
        env["OCR_NUM_GPUS"] = env.get("OCR_NUM_GPUS", "-1") if ids else "0"
        env["OCR_NODE_ID"] = f"{os.uname().nodename}-fe{i}"
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if self.redis is not None:
            env["REDIS_HOST"], env["REDIS_PORT"] = "127.0.0.1", str(self.redis.port)
            env.pop("REDIS_PASSWORD", None)
            env["OCR_STORE"] = "redis+file"
        return env

    def spawn(self, i: int) -> None:
        assert self.sock is not None
        fd = self.sock.fileno()
        self.children[i] = subprocess.Popen([sys.executable, "-m", "distributed_llms_amd.api.serve", "--child-fd", str(fd)],
                                            env=self.child_env(i), pass_fds=(fd,))

    def start(self) -> "Supervisor":
        if self.embedded:
            from ..store.resp import FakeRedisServer

            self.redis = FakeRedisServer(port=0).start()
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((self.host, self.port))
        self.sock.listen(2048)
        self.sock.set_inheritable(True)
        self.port = self.sock.getsockname()[1]
        for i in range(self.frontends):
            self.spawn(i)
        return self

    def poll_once(self) -> None:
        for i, p in list(self.children.items()):
            if p.poll() is not None and not self.stop:
                self.restarts += 1
                print(f"[serve] front end {i} exited with {p.returncode}; restarting", flush=True)
                time.sleep(min(5.0, 0.5 * self.restarts))
                self.spawn(i)

    def run_forever(self) -> None:
        def on_signal(signum, frame):
            self.stop = True

        signal.signal(signal.SIGTERM, on_signal)
        signal.signal(signal.SIGINT, on_signal)
        while not self.stop:
            self.poll_once()
            time.sleep(0.5)
        self.shutdown()

    def shutdown(self) -> None:
        self.stop = True
        for p in self.children.values():
            if p.poll() is None:
                p.terminate()
        for p in self.children.values():
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
        if self.sock is not None:
            self.sock.close()
        if self.redis is not None:
            self.redis.stop()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--port", type=int, default=int(os.environ.get("PORT", "8000")))
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--frontends", type=int, default=0, help="0: one per 2 GPUs (at least 1)")
    ap.add_argument("--gpus", type=int, default=None, help="override the detected GPU count")
    ap.add_argument("--embedded-redis", action="store_true", help="force the embedded status store + file results")
    ap.add_argument("--server", choices=["dlamd", "uvicorn"], default=os.environ.get("OCR_HTTP_SERVER", "dlamd"),
                    help="HTTP server of every front end: the in-repo one (api/httpd.py) or uvicorn")
    ap.add_argument("--child-fd", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    os.environ["OCR_HTTP_SERVER"] = args.server
    if args.child_fd >= 0:
        return child_main(args.child_fd)
    n = count_gpus() if args.gpus is None else args.gpus
    fe = args.frontends if args.frontends > 0 else max(1, n // 2)
    sup = Supervisor(args.port, args.host, fe, gpus=n, embedded_redis=True if args.embedded_redis else None).start()
    print(f"[serve] {fe} front end(s) on {args.host}:{sup.port}; GPU slices {sup.slices}; "
          f"stores: {'embedded RESP + files' if sup.redis is not None else 'from environment'}", flush=True)
    sup.run_forever()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
