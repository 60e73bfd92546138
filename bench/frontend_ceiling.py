#!/usr/bin/env python3
"""How many pages per second can the node's front end feed to its GPUs?  Measured without a GPU.

The engines are replaced by stubs (``runtime/stub.py``): a stub worker maps the uploaded PDF, parses it with the native
reader, scans every page's JPEG header on its decode pool, batches the pages, *sleeps* ``pages / --stub-pages-per-s``
per batch and answers ~40 canned lines per page, packed like real results.  Everything around the engine is the real
code: HTTP server, multipart scan, SHA-256, shared-memory hand-off, dispatch over the workers' sockets, result
transport, assembly, language identification, stores.

    python bench/frontend_ceiling.py --mode http --gpus 8 --frontends 8 --clients 8 --docs 48
    python bench/frontend_ceiling.py --mode pool --gpus 8 --frontends 8 --docs 64

``http``: ``api.serve --stub-gpus N`` + client processes that upload 100-page 300-DPI scans (~123 MB each) and poll.
``pool``: no HTTP; front-end processes submit documents that already sit in shared memory to the shared workers.

Reported: pages/s, and CPU seconds per 100 pages for each kind of process (front ends, worker servers, clients) from
``/proc/<pid>/stat``.  On a build host with few cores the wall-clock figure is bound by the cores the clients, the
kernel's loopback copies and the stubs share; the per-100-pages CPU cost is what carries over to a GPU node (cores
needed per GPU = cost x 10 at 1000 pages/s per GPU).

The reference's counterpart is a load balancer in front of N VMs (terraform/main.tf:176-294); it has no measurement
of its own.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
TICK = os.sysconf("SC_CLK_TCK")


def cpu_split(pid: int) -> tuple:
    """(user, system) CPU seconds of a process, all threads."""
    try:
        with open(f"/proc/{pid}/stat") as f:
            parts = f.read().rsplit(")", 1)[1].split()
        return int(parts[11]) / TICK, int(parts[12]) / TICK
    except (OSError, IndexError, ValueError):
        return 0.0, 0.0


def cpu_seconds(pid: int) -> float:
    return sum(cpu_split(pid))


def children_of(pid: int):
    out = []
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            with open(f"/proc/{d}/stat") as f:
                ppid = int(f.read().rsplit(")", 1)[1].split()[1])
            if ppid == pid:
                with open(f"/proc/{d}/cmdline", "rb") as f:
                    out.append((int(d), f.read().replace(b"\0", b" ").decode("utf-8", "replace")))
        except (OSError, ValueError, IndexError):
            continue
    return out


def make_pdf(pages: int, distinct: int = 10) -> bytes:
    from distributed_llms_amd.pdf import synth

    n = min(distinct, pages)
    rendered = [synth.render_page(i) for i in range(n)]
    enc = [synth.encode_page_jpeg(rendered[i][0], restart_blocks=0) for i in range(n)]
    return synth.make_scanned_pdf([rendered[i % n][0] for i in range(pages)], codec="jpeg", encoded=[enc[i % n] for i in range(pages)])


# ------------------------------------------------------------------------------------------------- http mode
def _http(port: int, method: str, path: str, head: bytes = b"", body=()) -> tuple:
    """``body``: pieces sent one after another (a per-document prefix, then the shared file part)."""
    s = socket.create_connection(("127.0.0.1", port))
    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    try:
        n = sum(len(b) for b in body)
        s.sendall(b"%s %s HTTP/1.1\r\nhost: x\r\nconnection: close\r\ncontent-length: %d\r\n%s\r\n"
                  % (method.encode(), path.encode(), n, head))
        for b in body:
            s.sendall(b)
        data = bytearray()
        while True:
            b = s.recv(65536)
            if not b:
                break
            data += b
    finally:
        s.close()
    headb, _, payload = bytes(data).partition(b"\r\n\r\n")
    return int(headb.split(b" ", 2)[1]), payload


ID_SLOT = b"@@DOC-ID-SLOT@@"           # placeholder in the prepared body, replaced by an id of the same length


def client_proc(port: int, body_path: str, ctype: str, ids, threads: int, out_q) -> None:
    import mmap

    with open(body_path, "rb") as f:
        body = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    split = body.find(ID_SLOT) + len(ID_SLOT) + 64
    assert all(len(i) == len(ID_SLOT) for i in ids)
    lat = []
    todo = list(ids)
    mu = threading.Lock()
    cpu0 = time.process_time()

    def run():
        while True:
            with mu:
                if not todo:
                    return
                doc_id = todo.pop()
            t0 = time.perf_counter()
            # the multipart body is the same for every document but the id field, which sits in its first bytes:
            # a patched copy of those, then the shared mapping of the rest
            prefix = bytes(body[:split]).replace(ID_SLOT, doc_id.encode())
            code, _ = _http(port, "POST", "/extract", b"content-type: %s\r\n" % ctype.encode(),
                            (prefix, memoryview(body)[split:]))
            assert code == 200, code
            while True:
                code, payload = _http(port, "GET", "/check-status/" + doc_id)
                if code == 200 and b"completed" in payload:
                    break
                if code == 200 and b'"error"' in payload:
                    raise RuntimeError(f"{doc_id} failed")
                time.sleep(0.02)
            lat.append(time.perf_counter() - t0)

    ts = [threading.Thread(target=run) for _ in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    out_q.put((lat, time.process_time() - cpu0))


def run_http(args, pdf: bytes) -> dict:
    from distributed_llms_amd.api.multipart import encode_multipart

    body, ctype = encode_multipart({"document_id": ID_SLOT.decode()}, {"file": ("scan.pdf", pdf, "application/pdf")})
    tmp = tempfile.mkdtemp(prefix="dlamd-ceiling-")
    body_path = os.path.join(tmp, "body.bin")
    with open(body_path, "wb") as f:
        f.write(body)
    env = dict(os.environ, OCR_OUTPUT_DIR=os.path.join(tmp, "out"), OCR_MAX_WAIT_MS="3",
               OCR_LOG_LEVEL="warning", PYTHONPATH=str(ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "distributed_llms_amd.api.serve", "--port", str(port), "--host", "127.0.0.1",
           "--stub-gpus", str(args.gpus), "--stub-pages-per-s", str(args.stub_pages_per_s), "--frontends", str(args.frontends),
           "--embedded-redis"] + (["--reuseport"] if args.reuseport else []) + (["--no-pin"] if args.no_pin else [])
    sup = subprocess.Popen(cmd, env=env, cwd=str(ROOT))
    try:
        t_end = time.time() + 180
        while time.time() < t_end:
            try:
                if _http(port, "GET", "/health")[0] in (200, 503):
                    break
            except OSError:
                time.sleep(0.3)
        time.sleep(1.0)
        ctx = mp.get_context("spawn")

        def drive(tag, n_docs):
            q = ctx.Queue()
            ids = [f"{tag}-{k}".ljust(len(ID_SLOT), "x") for k in range(n_docs)]
            procs = [ctx.Process(target=client_proc, args=(port, body_path, ctype, ids[c::args.clients], args.client_threads, q))
                     for c in range(args.clients)]
            t0 = time.perf_counter()
            for p in procs:
                p.start()
            got = [q.get(timeout=1200) for _ in procs]
            dt = time.perf_counter() - t0
            for p in procs:
                p.join()
            return dt, [x for lat, _ in got for x in lat], sum(c for _, c in got)

        drive("warm", max(args.clients, args.gpus))
        kids = children_of(sup.pid)
        fes = [p for p, c in kids if "api.serve" in c]
        wks = [p for p, c in kids if "worker_server" in c]
        c0 = {p: cpu_seconds(p) for p in fes + wks + [sup.pid]}
        s0 = sum(cpu_split(p)[1] for p in fes)
        dt, lat, client_cpu = drive("run", args.docs)
        c1 = {p: cpu_seconds(p) for p in c0}
        fe_sys = sum(cpu_split(p)[1] for p in fes) - s0
        pages = args.docs * args.pages
        per100 = lambda pids: round(100.0 * sum(c1[p] - c0[p] for p in pids) / pages, 4)
        lat.sort()
        return {"mode": "http", "pages_per_s": round(pages / dt, 1), "wall_s": round(dt, 3), "documents": args.docs,
                "p50_document_s": round(lat[len(lat) // 2], 3), "p95_document_s": round(lat[int(len(lat) * 0.95)], 3),
                "cpu_s_per_100_pages": {"front_ends": per100(fes), "front_ends_kernel_share": round(100.0 * fe_sys / pages, 4),
                                        "worker_servers": per100(wks),
                                        "supervisor_and_status_store": per100([sup.pid]),
                                        "clients (not part of the node)": round(100.0 * client_cpu / pages, 4)},
                "front_end_processes": len(fes), "worker_processes": len(wks)}
    finally:
        sup.terminate()
        try:
            sup.wait(timeout=20)
        except subprocess.TimeoutExpired:
            sup.kill()
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------- pool mode
def frontend_proc(socks, home: int, seg_name: str, nbytes: int, n_pages: int, n_docs: int, in_flight: int, out_q, go) -> None:
    from concurrent.futures import ThreadPoolExecutor

    from distributed_llms_amd.runtime.extractor import PoolExtractor
    from distributed_llms_amd.runtime.shm import Segment
    from distributed_llms_amd.runtime.workers import ShmDocument, WorkerPool

    pool = WorkerPool(["unix:" + s for s in socks], home=home).start(300.0)
    ext = PoolExtractor(pool)
    seg = Segment.attach(seg_name)
    doc = ShmDocument(seg, 0, nbytes)
    try:
        with ThreadPoolExecutor(max_workers=in_flight) as tp:
            list(tp.map(lambda k: ext.extract_document(doc, f"warm-{home}-{k}"), range(in_flight)))
            out_q.put(("ready", home))
            go.wait()
            c0, t0 = time.process_time(), time.perf_counter()
            docs = list(tp.map(lambda k: ext.extract_document(doc, f"d-{home}-{k}"), range(n_docs)))
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        assert all(len(d.pages) == n_pages and d.text for d in docs)
        out_q.put(("done", home, dt, cpu, pool.describe()["queue_depth"]))
    finally:
        pool.stop()


def run_pool(args, pdf: bytes) -> dict:
    from distributed_llms_amd.runtime.shm import Segment

    tmp = tempfile.mkdtemp(prefix="dlamd-ceiling-")
    socks = [os.path.join(tmp, f"w{i}.sock") for i in range(args.gpus)]
    spec = json.dumps({"model": "stub", "pages_per_s": args.stub_pages_per_s, "cfg": {"max_batch_pages": 36}})
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    wks = [subprocess.Popen([sys.executable, "-m", "distributed_llms_amd.runtime.worker_server", "--id", str(i), "--device", "cpu",
                             "--socket", socks[i], "--spec", spec], env=env, cwd=str(ROOT)) for i in range(args.gpus)]
    seg = Segment.create(len(pdf))
    seg.buf[: len(pdf)] = pdf
    ctx = mp.get_context("spawn")
    q, go = ctx.Queue(), ctx.Event()
    per_fe = max(1, args.docs // args.frontends)
    try:
        t_end = time.time() + 120
        while not all(os.path.exists(s) for s in socks) and time.time() < t_end:
            time.sleep(0.2)
        fes = [ctx.Process(target=frontend_proc, args=(socks, i % args.gpus, seg.name, len(pdf), args.pages, per_fe,
                                                       args.in_flight, q, go)) for i in range(args.frontends)]
        for p in fes:
            p.start()
        for _ in fes:
            assert q.get(timeout=600)[0] == "ready"
        c0 = {p.pid: cpu_seconds(p.pid) for p in wks}
        t0 = time.perf_counter()
        go.set()
        got = [q.get(timeout=1200) for _ in fes]
        dt = time.perf_counter() - t0
        c1 = {p.pid: cpu_seconds(p.pid) for p in wks}
        for p in fes:
            p.join()
        pages = per_fe * args.frontends * args.pages
        return {"mode": "pool", "pages_per_s": round(pages / dt, 1), "wall_s": round(dt, 3), "documents": per_fe * args.frontends,
                "cpu_s_per_100_pages": {"front_ends": round(100.0 * sum(g[3] for g in got) / pages, 4),
                                        "worker_servers": round(100.0 * sum(c1[k] - c0[k] for k in c0) / pages, 4)},
                "front_end_processes": args.frontends, "worker_processes": args.gpus}
    finally:
        for p in wks:
            p.terminate()
        for p in wks:
            p.wait()
        seg.release()
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--mode", choices=["http", "pool"], default="http")
    ap.add_argument("--gpus", type=int, default=8, help="stub workers")
    ap.add_argument("--frontends", type=int, default=0, help="0: one per stub GPU")
    ap.add_argument("--stub-pages-per-s", type=float, default=1000.0)
    ap.add_argument("--pages", type=int, default=100)
    ap.add_argument("--docs", type=int, default=48)
    ap.add_argument("--clients", type=int, default=4, help="http: client processes")
    ap.add_argument("--client-threads", type=int, default=3, help="http: uploads in flight per client process")
    ap.add_argument("--in-flight", type=int, default=3, help="pool: documents in flight per front end")
    ap.add_argument("--reuseport", action="store_true")
    ap.add_argument("--no-pin", action="store_true")
    args = ap.parse_args()
    args.frontends = args.frontends or args.gpus
    pdf = make_pdf(args.pages)
    res = run_http(args, pdf) if args.mode == "http" else run_pool(args, pdf)
    res.update({"stub_gpus": args.gpus, "stub_pages_per_s_each": args.stub_pages_per_s, "pages_per_document": args.pages,
                "document_mb": round(len(pdf) / 1e6, 1), "host_cores": len(os.sched_getaffinity(0)),
                "engine_capacity_pages_per_s": args.gpus * args.stub_pages_per_s})
    c = res["cpu_s_per_100_pages"]
    node = c["front_ends"] + c["worker_servers"] + c.get("supervisor_and_status_store", 0.0)      # kernel share is inside front_ends
    res["node_cpu_s_per_100_pages"] = round(node, 4)
    res["cores_per_gpu_at_1000_pages_per_s"] = round(node * 10.0, 2)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
