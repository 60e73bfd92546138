"""Node-level launcher: N HTTP front-end processes on ONE port in front of one engine process per GPU.

    python -m distributed_llms_amd.api.serve --port 8000            # 8 GPUs -> 8 worker servers + 8 front ends

Why: one Python front end ingests a 124 MB scanned PDF in ~0.1 s of CPU (socket, multipart scan, SHA-256), i.e.
~1000 pages/s of 300-DPI JPEG pages at best, and ONE MI355X reads 1000 pages/s - a single front end cannot feed eight
GPUs.  The reference scales one level up (N stateless replicas behind an ALB, terraform/main.tf:176-294, each with
its own extraction processes, services/ocr_service/ocr_service.py:42); inside a node the same picture would strand
GPUs behind front ends whose clients happen to be idle.  Here instead:

* the parent starts ONE worker server per GPU (``runtime/worker_server.py``, pinned to the cores next to its GPU,
  restarted when it exits) and, by default, one front end per GPU;
* EVERY front end connects to EVERY worker server and sends each shard of a document to the least-loaded one (the
  load a worker reports counts the pages it owes to all front ends); its "home" worker only breaks ties, so an idle
  node keeps a document on the GPU next to the cores that received it;
* the parent binds the listening socket once and hands the descriptor to every front end (the kernel gives a new
  connection to a front end that is waiting in ``accept``, i.e. to one that is not busy); ``--reuseport`` makes
  every front end bind the port itself instead (``SO_REUSEPORT``: connections are hashed over the processes);
* uploads are received into shared memory (``api/app.py``) and mapped, never copied, by the workers;
* the front ends are stateless: any of them answers ``/check-status`` and ``/result`` for any document, so the
  stores must be shared - Redis + MongoDB as deployed, or, with nothing configured, an embedded RESP server in
  the parent (``store/resp.py``) plus the file result store under ``OCR_OUTPUT_DIR`` (``OCR_STORE=redis+file``).

``--owned-workers`` keeps the earlier layout (each front end owns a slice of the GPUs and nobody else can use them);
``--stub-gpus N`` replaces the engines by N sleep-and-answer stubs (``runtime/stub.py``) for front-end capacity
measurements on a machine without GPUs (``bench/frontend_ceiling.py``).
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
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


def child_main(fd: int, port: int = 0, host: str = "0.0.0.0") -> int:
    """One front end on the inherited listening socket (or, with ``port``, on a socket of its own bound with
    ``SO_REUSEPORT``).  ``OCR_HTTP_SERVER=uvicorn`` selects the stock server; the default is the in-repo one
    (api/httpd.py: ~2x cheaper polls, body bytes received straight into the upload buffer)."""
    from ..config import Settings
    from .app import create_app

    cpus = os.environ.get("OCR_PIN_CPUS", "")
    if cpus:
        from ..utils.affinity import parse_cpulist, pin_process

        pin_process(parse_cpulist(cpus))
    if fd < 0:
        sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)
        sock.bind((host, port))
        sock.listen(2048)
        fd = sock.detach()
    app = create_app(Settings.from_env(dotenv=os.environ.get("OCR_DOTENV", ".env")))
    if os.environ.get("OCR_HTTP_SERVER", "dlamd").lower() == "uvicorn":
        import uvicorn

        uvicorn.run(app, fd=fd, log_level=os.environ.get("OCR_LOG_LEVEL", "warning"))
    else:
        from . import httpd

        httpd.run(app, fd=fd)
    return 0


class Supervisor:
    def __init__(self, port: int, host: str, frontends: int, gpus: Optional[int] = None, embedded_redis: Optional[bool] = None,
                 shared_workers: bool = True, stub_gpus: int = 0, stub_pages_per_s: float = 1000.0, reuseport: bool = False,
                 pin: bool = True):
        self.port, self.host, self.frontends = port, host, frontends
        self.stub = stub_gpus > 0
        self.n_gpus = stub_gpus if self.stub else (count_gpus() if gpus is None else gpus)
        self.shared = bool(shared_workers and self.n_gpus > 0)
        self.stub_pages_per_s = stub_pages_per_s
        self.reuseport = reuseport
        self.pin = pin
        self.slices = split_gpus(self.n_gpus, frontends)
        self.sock: Optional[socket.socket] = None
        self.children: Dict[int, subprocess.Popen] = {}
        self.workers: Dict[int, subprocess.Popen] = {}
        self.sock_dir: Optional[str] = None
        self.redis = None
        self.stop = False
        want_embedded = embedded_redis if embedded_redis is not None else (
            not os.environ.get("REDIS_HOST") and not os.environ.get("MONGO_CONNECTION_STRING"))
        self.embedded = want_embedded
        self.restarts = 0
        self.worker_restarts = 0

    # ------------------------------------------------------------------ placement
    def cpus_of(self, gpu: int) -> str:
        """Core list (``"0-15"`` style, comma separated) for GPU ``gpu``'s worker server and its home front end; empty
        when there are too few cores to make slices worth having."""
        if not self.pin:
            return ""
        from ..utils.affinity import allowed_cpus, cpus_for_worker

        if len(allowed_cpus()) < 2 * max(1, self.n_gpus):
            return ""
        return ",".join(map(str, cpus_for_worker(gpu, self.n_gpus, gpu_lists=[] if self.stub else None)))

    def worker_socket(self, i: int) -> str:
        return os.path.join(self.sock_dir, f"w{i}.sock")

    def worker_spec(self) -> Dict[str, object]:
        from ..config import Settings

        st = Settings.from_env(dotenv=os.environ.get("OCR_DOTENV", ".env"))
        cfg = dict(det_limit_side=st.det_limit_side, rec_height=st.rec_height, max_batch_pages=st.max_batch_pages,
                   use_hip_graphs=st.use_hip_graphs, inline_formulas=st.inline_formulas)
        if self.stub:
            return {"model": "stub", "pages_per_s": self.stub_pages_per_s, "cfg": dict(max_batch_pages=st.max_batch_pages)}
        return {"model": "full", "cfg": cfg, "weights_dir": st.weights_dir, "recognizer": st.recognizer}

    def spawn_worker(self, i: int) -> None:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "distributed_llms_amd.runtime.worker_server", "--id", str(i),
               "--device", "cpu" if self.stub else f"cuda:{i}", "--socket", self.worker_socket(i),
               "--spec", json.dumps(self.worker_spec()), "--max-wait-ms", os.environ.get("OCR_MAX_WAIT_MS", "3"),
               "--fault", os.environ.get("FAULT_INJECT", "") if not self.worker_restarts else ""]
        cpus = self.cpus_of(i)
        if cpus:
            cmd += ["--cpus", cpus]
        self.workers[i] = subprocess.Popen(cmd, env=env)

    def child_env(self, i: int) -> Dict[str, str]:
        env = dict(os.environ)
        env["OCR_NODE_ID"] = f"{os.uname().nodename}-fe{i}"
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if self.shared:
            env["OCR_WORKER_SOCKETS"] = ",".join(self.worker_socket(w) for w in range(self.n_gpus))
            env["OCR_HOME_WORKER"] = str(i % self.n_gpus)
            env["OCR_NUM_GPUS"] = "0"                 # the front end itself never touches a GPU
            env["OCR_PIN_CPUS"] = self.cpus_of(i % self.n_gpus)
        else:
            ids = self.slices[i]
            env["OCR_GPU_IDS"] = ",".join(map(str, ids))
            env["OCR_NUM_GPUS"] = env.get("OCR_NUM_GPUS", "-1") if ids else "0"
        if self.redis is not None:
            env["REDIS_HOST"], env["REDIS_PORT"] = "127.0.0.1", str(self.redis.port)
            env.pop("REDIS_PASSWORD", None)
            env["OCR_STORE"] = "redis+file"
        return env

    def spawn(self, i: int) -> None:
        assert self.sock is not None
        if self.reuseport:
            self.children[i] = subprocess.Popen(
                [sys.executable, "-m", "distributed_llms_amd.api.serve", "--child-fd", "-1", "--child-port", str(self.port),
                 "--host", self.host], env=self.child_env(i))
            return
        fd = self.sock.fileno()
        self.children[i] = subprocess.Popen([sys.executable, "-m", "distributed_llms_amd.api.serve", "--child-fd", str(fd)],
                                            env=self.child_env(i), pass_fds=(fd,))

    def start(self) -> "Supervisor":
        if self.embedded:
            from ..store.resp import FakeRedisServer

            self.redis = FakeRedisServer(port=0).start()
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        if self.reuseport:
            self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)    # holds the port number; never listens
        self.sock.bind((self.host, self.port))
        if not self.reuseport:
            self.sock.listen(2048)
            self.sock.set_inheritable(True)
        self.port = self.sock.getsockname()[1]
        if self.shared:
            from ..runtime.shm import sweep_stale

            sweep_stale()             # segments of front ends that died mid-document in an earlier run
            self.sock_dir = tempfile.mkdtemp(prefix="dlamd-workers-")
            for w in range(self.n_gpus):
                self.spawn_worker(w)
        for i in range(self.frontends):
            self.spawn(i)
        return self

    def poll_once(self) -> None:
        for w, p in list(self.workers.items()):
            if p.poll() is not None and not self.stop:
                self.worker_restarts += 1
                print(f"[serve] worker server {w} exited with {p.returncode}; restarting", flush=True)
                self.spawn_worker(w)
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
        procs = list(self.children.values()) + list(self.workers.values())
        for p in self.children.values():
            if p.poll() is None:
                p.terminate()
        for p in self.children.values():
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
        for p in self.workers.values():
            if p.poll() is None:
                p.terminate()
        for p in self.workers.values():
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
        del procs
        if self.sock is not None:
            self.sock.close()
        if self.redis is not None:
            self.redis.stop()
        if self.sock_dir:
            import shutil

            shutil.rmtree(self.sock_dir, ignore_errors=True)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--port", type=int, default=int(os.environ.get("PORT", "8000")))
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--frontends", type=int, default=0, help="0: one per GPU (at least 1)")
    ap.add_argument("--gpus", type=int, default=None, help="override the detected GPU count")
    ap.add_argument("--embedded-redis", action="store_true", help="force the embedded status store + file results")
    ap.add_argument("--owned-workers", action="store_true",
                    help="every front end owns a slice of the GPUs (no sharing across front ends)")
    ap.add_argument("--stub-gpus", type=int, default=0, help="N stub engines instead of GPUs (capacity tests)")
    ap.add_argument("--stub-pages-per-s", type=float, default=1000.0)
    ap.add_argument("--reuseport", action="store_true", help="every front end binds the port itself (SO_REUSEPORT)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin workers / front ends to GPU-local cores")
    ap.add_argument("--server", choices=["dlamd", "uvicorn"], default=os.environ.get("OCR_HTTP_SERVER", "dlamd"),
                    help="HTTP server of every front end: the in-repo one (api/httpd.py) or uvicorn")
    ap.add_argument("--child-fd", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-port", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    os.environ["OCR_HTTP_SERVER"] = args.server
    if args.child_fd is not None:
        return child_main(args.child_fd, args.child_port, args.host)
    n = args.stub_gpus if args.stub_gpus > 0 else (count_gpus() if args.gpus is None else args.gpus)
    fe = args.frontends if args.frontends > 0 else max(1, n)
    sup = Supervisor(args.port, args.host, fe, gpus=n, embedded_redis=True if args.embedded_redis else None,
                     shared_workers=not args.owned_workers, stub_gpus=args.stub_gpus, stub_pages_per_s=args.stub_pages_per_s,
                     reuseport=args.reuseport, pin=not args.no_pin).start()
    layout = (f"{sup.n_gpus} shared worker server(s)" if sup.shared else f"GPU slices {sup.slices}")
    print(f"[serve] {fe} front end(s) on {args.host}:{sup.port}; {layout}; "
          f"stores: {'embedded RESP + files' if sup.redis is not None else 'from environment'}", flush=True)
    sup.run_forever()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
