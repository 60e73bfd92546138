"""One GPU worker as a server that every front end of the node can reach.

    python -m distributed_llms_amd.runtime.worker_server --id 3 --device cuda:3 --socket /run/dlamd/w3.sock

The reference scales out by adding whole service replicas behind a load balancer (terraform/main.tf:176-294); each
replica owns its extraction processes (services/ocr_service/ocr_service.py:42).  Inside a node that shape wastes GPUs:
a front end whose clients are busy cannot lend its GPU to a neighbour whose clients are not.  Here the engines are a
shared pool: ``api/serve.py`` starts one of these servers per GPU, every front end connects to all of them
(``WorkerPool(["unix:..."])``) and sends each shard to the least-loaded one.

* transport: ``multiprocessing.connection`` over a Unix socket - small pickled task tuples in, one bytes object per
  finished shard out (runtime/wire.py); the PDF itself never travels, workers map its shared-memory segment;
* every connection gets a heartbeat (progress counter + pages still owed to ALL front ends, the load figure the front
  ends balance on);
* a front end that disconnects has its shards cancelled;
* watchdog: pages owed but no engine progress for ``--progress-timeout`` seconds means a wedged GPU call; the process
  exits with code 19 and the node supervisor starts a fresh one (front ends re-dispatch what was in flight).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import signal
import sys
import threading
import time
import traceback
from multiprocessing.connection import Listener
from typing import Dict, List, Optional

logger = logging.getLogger("distributed_llms_amd")


class WorkerServer:
    def __init__(self, worker_id: int, device: str, engine_spec: Dict, socket_path: str, fault_spec: str = "",
                 max_wait_ms: float = 3.0, hbm_budget_gb: float = 200.0, cpus: Optional[List[int]] = None,
                 progress_timeout_s: float = 60.0):
        from .workers import WorkerCore

        self.worker_id, self.socket_path = worker_id, socket_path
        self.progress_timeout_s = progress_timeout_s
        self.core = WorkerCore(worker_id, device, engine_spec, fault_spec, max_wait_ms, hbm_budget_gb, cpus)
        try:
            os.unlink(socket_path)
        except OSError:
            pass
        os.makedirs(os.path.dirname(socket_path) or ".", exist_ok=True)
        self.listener = Listener(socket_path, family="AF_UNIX", backlog=64)
        self.conns: Dict[int, tuple] = {}        # id -> (connection, write lock, job ids it has in flight)
        self.mu = threading.Lock()
        self._seq = 0
        self.stopping = False

    # ------------------------------------------------------------------ connections
    def _serve_conn(self, cid: int, conn, wlock: threading.Lock, jobs: set) -> None:
        def reply(msg):
            with wlock:
                conn.send(msg)
            if msg[0] in ("done", "error"):
                with self.mu:
                    jobs.discard(msg[2])

        try:
            reply(("ready", self.worker_id, os.getpid()))
            while True:
                task = conn.recv()
                if task is None:
                    break
                if task[0] == "shard":
                    with self.mu:
                        jobs.add(task[1])
                self.core.handle(task, reply)
        except (EOFError, OSError):
            pass
        except Exception:  # noqa: BLE001
            logger.error("worker connection failed:\n" + traceback.format_exc())
        finally:
            with self.mu:
                self.conns.pop(cid, None)
                left = list(jobs)
            for j in left:                      # nobody is waiting for these any more
                self.core.handle(("cancel", j), lambda m: None)
            try:
                conn.close()
            except OSError:
                pass

    def _accept(self) -> None:
        while not self.stopping:
            try:
                conn = self.listener.accept()
            except OSError:
                return
            with self.mu:
                self._seq += 1
                cid = self._seq
                wlock, jobs = threading.Lock(), set()
                self.conns[cid] = (conn, wlock, jobs)
            threading.Thread(target=self._serve_conn, args=(cid, conn, wlock, jobs), daemon=True,
                             name=f"dlamd-conn-{cid}").start()

    def _beat(self) -> None:
        last_progress, last_moved = -1, time.time()
        while not self.stopping:
            time.sleep(0.2)
            hb = self.core.heartbeat()
            now = time.time()
            if hb[3] != last_progress or hb[4] == 0:
                last_progress, last_moved = hb[3], now
            elif now - last_moved > self.progress_timeout_s:
                logger.error(f"worker {self.worker_id}: {hb[4]} pages owed and no progress for "
                             f"{self.progress_timeout_s:.0f} s; exiting for a restart")
                os._exit(19)
            with self.mu:
                conns = list(self.conns.values())
            for conn, wlock, _ in conns:
                try:
                    with wlock:
                        conn.send(hb)
                except (OSError, ValueError):
                    pass

    def serve_forever(self) -> None:
        threading.Thread(target=self._beat, daemon=True, name="dlamd-heartbeat").start()
        threading.Thread(target=self._accept, daemon=True, name="dlamd-accept").start()
        while not self.stopping:
            time.sleep(0.2)
        self.close()

    def close(self) -> None:
        self.stopping = True
        try:
            self.listener.close()
        except OSError:
            pass
        with self.mu:
            conns = list(self.conns.values())
        for conn, _, _ in conns:
            try:
                conn.close()
            except OSError:
                pass
        self.core.close()
        try:
            os.unlink(self.socket_path)
        except OSError:
            pass


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--id", type=int, required=True)
    ap.add_argument("--device", required=True, help="cuda:N, or cpu")
    ap.add_argument("--socket", required=True)
    ap.add_argument("--spec", default="{}", help="engine spec as JSON (model, recognizer, cfg, weights_dir)")
    ap.add_argument("--fault", default="")
    ap.add_argument("--max-wait-ms", type=float, default=3.0)
    ap.add_argument("--hbm-budget-gb", type=float, default=200.0)
    ap.add_argument("--cpus", default="", help="cores to pin this worker to, e.g. 0-15 (empty: not pinned)")
    ap.add_argument("--progress-timeout", type=float, default=60.0)
    args = ap.parse_args(argv)
    from ..config import configure_logging
    from ..utils.affinity import parse_cpulist

    configure_logging()
    try:
        srv = WorkerServer(args.id, args.device, json.loads(args.spec), args.socket, args.fault, args.max_wait_ms,
                           args.hbm_budget_gb, parse_cpulist(args.cpus) if args.cpus else None, args.progress_timeout)
    except Exception:  # noqa: BLE001
        logger.error(f"worker {args.id} failed to start:\n{traceback.format_exc()}")
        return 1

    def on_signal(signum, frame):
        srv.stopping = True

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    print(f"[worker {args.id}] ready on {args.device} at {args.socket} (pid {os.getpid()})", flush=True)
    srv.serve_forever()
    return 0


if __name__ == "__main__":
    sys.exit(main())
