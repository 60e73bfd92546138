"""HTTP API.

Same public contract as the reference service
(services/ocr_service/ocr_service.py:127-195, 412-606; API.md):

    POST /extract                multipart ``file`` + ``document_id``  -> {"status": "ok"} | raw status of a duplicate
    GET  /check-status/{id}      -> {"status": processing|completed|error} | 404 {'status': 'unknown', 'error': ...}
    GET  /result/{id}            -> {"document_id", "text", "status": "completed" [, "warnings"]} | in-progress status | 404 | 500
    GET  /health                 -> healthy 200 | degraded 200 | unhealthy 503, version "1.0.0"
    GET  /metrics                -> Prometheus text (new)

Built as a FastAPI app (CORS wide open like the reference, lifespan-managed stores).  The
multipart body is parsed by ``api.multipart`` because python-multipart - which FastAPI's
``File``/``Form`` need - is not a dependency here.  Deliberate fixes:
an empty ``document_id`` really answers 400 (the reference *returns* the exception object);
the duplicate check and the ``queued`` write are one atomic claim; no Redis call blocks the
event loop; there is no default Mongo credential.
"""
from __future__ import annotations

import asyncio
import datetime
import hashlib
import json
import logging
import os
import sys
import traceback
from contextlib import asynccontextmanager
from typing import Optional

from fastapi import FastAPI, Request
from fastapi.middleware.cors import CORSMiddleware
from fastapi.responses import JSONResponse, Response

from ..config import SERVICE_VERSION, Settings, configure_logging
from ..jobs.manager import JobManager
from ..store.resp import RespClient
from ..store.results import ErrorSink, LocalFileWriter, MemoryResultStore, MongoResultStore, ResultStore
from ..store.status import MemoryStatusStore, RedisStatusStore, StatusStore, public_status
from ..utils import metrics
from .multipart import MultipartError, StreamingForm, UploadTooLarge, parse_multipart

logger = logging.getLogger("distributed_llms_amd")

MAX_UPLOAD_BYTES = int(os.environ.get("OCR_MAX_UPLOAD_BYTES", str(2 << 30)))
# upload bytes that may be in flight in one front-end process at a time (announced lengths of the requests being
# received): beyond it new uploads are answered 503 instead of each pinning a buffer of its announced size
MAX_INFLIGHT_UPLOAD_BYTES = int(os.environ.get("OCR_MAX_INFLIGHT_UPLOAD_BYTES", str(16 << 30)))


def make_stores(settings: Settings):
    """(status store, result store) for the configured mode; never invents credentials."""
    if settings.resolved_store() == "memory":
        return MemoryStatusStore(ttl=settings.status_ttl), MemoryResultStore()
    client = RespClient(settings.redis_host, settings.redis_port, settings.redis_password)
    status = RedisStatusStore(client, ttl=settings.status_ttl)
    if settings.resolved_store() == "redis+file":
        from ..store.results import FileResultStore

        return status, FileResultStore(os.path.join(settings.output_dir, "results"))
    results = MongoResultStore(settings.mongo_connection_string, settings.mongo_db_name,
                               settings.mongo_collection_name, tls=settings.mongo_tls)
    return status, results


class PollingFastPath:
    """Pure ASGI layer in front of the framework for ``GET /check-status/{id}``.

    A client polls this endpoint every few milliseconds per document in flight (the reference's API has no other way
    to learn that a document is done, ocr_service.py:473-501), so under load it is hit a hundred times more often
    than every other route together.  Through the framework (routing, dependency resolution, response model
    handling, a thread hop for the store) one poll costs ~1.2 ms of the interpreter; answered here it is a dictionary
    lookup and three ASGI messages.  Same status codes and bodies as the route below; requests that carry an
    ``Origin`` header (browsers: CORS response headers needed) go through the framework."""

    def __init__(self, app, ctx: dict, count):
        self.app, self.ctx, self.count = app, ctx, count

    async def __call__(self, scope, receive, send):
        if scope["type"] == "http" and scope["method"] == "GET" and scope["path"].startswith("/check-status/"):
            doc_id = scope["path"][len("/check-status/"):]
            jobs = self.ctx.get("jobs")
            if doc_id and "/" not in doc_id and jobs is not None and not any(k == b"origin" for k, _ in scope["headers"]):
                code, body = await self._status(jobs, doc_id)
                await send({"type": "http.response.start", "status": code,
                            "headers": [(b"content-type", b"application/json"),
                                        (b"content-length", str(len(body)).encode())]})
                await send({"type": "http.response.body", "body": body})
                return
        await self.app(scope, receive, send)

    async def _status(self, jobs, doc_id: str):
        try:
            store = jobs.status
            data = store.get(doc_id) if store.in_process else await jobs._io(store.get, doc_id)
            if data:
                self.count("check-status", 200)
                return 200, _json_bytes({"status": public_status(data["status"])})
            self.count("check-status", 404)
            return 404, _json_bytes({"status": "unknown", "error": f"No record found for document ID: {doc_id}"})
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error checking status for {doc_id}: {str(e)}")
            self.count("check-status", 200)
            return 200, _json_bytes({"status": "error", "message": str(e)})


def _json_bytes(obj) -> bytes:
    # the framework's JSONResponse rendering
    return json.dumps(obj, ensure_ascii=False, allow_nan=False, indent=None, separators=(",", ":")).encode("utf-8")


def create_app(settings: Optional[Settings] = None, status: Optional[StatusStore] = None,
               results: Optional[ResultStore] = None, extractor=None) -> FastAPI:
    """Application factory.  Tests inject in-memory stores and a stub / CPU extractor."""
    settings = settings or Settings.from_env()
    configure_logging()
    state = {"status": status, "results": results, "extractor": extractor, "jobs": None, "own_extractor": False}

    @asynccontextmanager
    async def lifespan(app: FastAPI):
        # ---- startup (ocr_service.py:132-163): connect, ping, indexes; errors are logged, not fatal
        if getattr(settings, "gil_switch_ms", 0) > 0:
            sys.setswitchinterval(settings.gil_switch_ms / 1000.0)
        try:
            if state["status"] is None or state["results"] is None:
                st, rs = make_stores(settings)
                state["status"] = state["status"] or st
                state["results"] = state["results"] or rs
            try:
                state["results"].connect()
                state["results"].ensure_indexes()
            except Exception as e:  # noqa: BLE001
                logger.error(f"Error initializing MongoDB connection: {str(e)}")
            os.makedirs(settings.image_dir, exist_ok=True)
            state["image_writer"] = LocalFileWriter(settings.image_dir)
            if state["extractor"] is None:
                from ..runtime.extractor import make_extractor

                loop = asyncio.get_running_loop()
                # figure crops of every document go through the image writer (ocr_service.py:111-124, 198, 216)
                state["extractor"] = await loop.run_in_executor(None, make_extractor, settings, state["image_writer"])
                state["own_extractor"] = True
            jobs = JobManager(state["status"], state["results"], ErrorSink(settings.output_dir), state["extractor"],
                              node_id=settings.node_id, deadline_s=settings.doc_deadline_s)
            state["jobs"] = jobs
            n = jobs.sweep_orphans()
            if n:
                logger.warning(f"marked {n} orphaned document(s) of node {settings.node_id} as failed")
            logger.info(f"Services initialized: {json.dumps(state['extractor'].describe(), default=str)}")
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error initializing services: {str(e)}\n{traceback.format_exc()}")
        app.state.ctx = state
        yield
        # ---- shutdown
        jobs = state.get("jobs")
        if jobs is not None:
            await jobs.drain(timeout=5.0)
            jobs.shutdown()
        try:
            state["results"].close()
        except Exception:  # noqa: BLE001
            pass
        try:
            state["status"].close()
        except Exception:  # noqa: BLE001
            pass
        if state.get("own_extractor") and state.get("extractor") is not None:
            state["extractor"].close()
        seg_pool.clear()
        logger.info("Services shut down")

    app = FastAPI(title="PDF OCR Service", description="MI355X-native PDF text extraction / OCR service",
                  version=SERVICE_VERSION, lifespan=lifespan)
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"],
                       allow_headers=["*"])

    _counters: dict = {}

    def _count(route: str, code: int) -> None:
        c = _counters.get((route, code))
        if c is None:
            c = _counters[(route, code)] = metrics.REQUESTS.labels(route, str(code))
        c.inc()

    app.add_middleware(PollingFastPath, ctx=state, count=_count)

    # ------------------------------------------------------------------ POST /extract
    inflight = {"bytes": 0}
    from ..runtime.shm import SegmentPool

    # finished uploads' segments are kept for the next upload of that size (their pages are already faulted in)
    seg_pool = SegmentPool(int(os.environ.get("OCR_SHM_POOL_BYTES", str(2 << 30))))

    @app.post("/extract")
    async def upload_pdf(request: Request):
        seg = None                 # shared-memory segment the upload is received into (GPU worker pool back ends)
        handed_over = False
        announced = 0
        try:
            clen = request.headers.get("content-length")
            jobs: JobManager = request.app.state.ctx["jobs"]
            try:
                # the body is consumed as it arrives: the cap holds for chunked uploads too, and the file's SHA-256
                # is computed on a helper thread while the rest of the upload is still on the wire
                factory = None
                if getattr(jobs.extractor, "wants_shm", False):
                    # the workers map the upload instead of receiving a copy: it is received INTO the segment
                    def factory(n):
                        nonlocal seg
                        seg = seg_pool.acquire(n)
                        return seg.buf

                announced = int(clen) if clen and clen.isdigit() else 0
                if inflight["bytes"] + announced > MAX_INFLIGHT_UPLOAD_BYTES and inflight["bytes"] > 0:
                    announced = 0
                    _count("extract", 503)
                    return JSONResponse({"detail": "too many uploads in flight, retry shortly"}, status_code=503,
                                        headers={"retry-after": "1"})
                inflight["bytes"] += announced
                form = StreamingForm(request.headers.get("content-type"), MAX_UPLOAD_BYTES,
                                     int(clen) if clen and clen.isdigit() else None,
                                     hash_pool=jobs.hash_pool, buffer_factory=factory)
                sink = (request.scope.get("extensions") or {}).get("dlamd.body_sink")
                if sink is not None and form.fixed and form.content_length:
                    # api/httpd.py: the socket's bytes are written to the form's buffer by the kernel itself
                    sink["set"](form.buffer())
                async for chunk in request.stream():
                    form.feed(chunk)
                parts = form.finish(defer_hash=True)
            except UploadTooLarge:
                _count("extract", 413)
                return JSONResponse({"detail": "upload too large"}, status_code=413)
            except MultipartError as e:
                _count("extract", 422)
                return JSONResponse({"detail": f"invalid multipart body: {e}"}, status_code=422)
            file_part = next((p for p in parts if p.name == "file" and p.is_file), None)
            id_part = next((p for p in parts if p.name == "document_id" and not p.is_file), None)
            if file_part is None or id_part is None:
                # FastAPI's own answer for a missing File(...)/Form(...) field
                missing = [n for n, p in (("file", file_part), ("document_id", id_part)) if p is None]
                _count("extract", 422)
                return JSONResponse({"detail": [{"type": "missing", "loc": ["body", m], "msg": "Field required"}
                                                for m in missing]}, status_code=422)
            document_id = id_part.text().strip()
            logger.info(f"Received file: {document_id}")
            if not (file_part.filename or "").lower().endswith(".pdf"):
                _count("extract", 400)
                return JSONResponse({"detail": "Only PDF files are accepted"}, status_code=400)
            if not document_id:
                _count("extract", 400)
                return JSONResponse({"detail": "document_id is required"}, status_code=400)
            file_bytes = file_part.data                 # a view of the receive buffer: the PDF is never copied again
            # the hash thread may still be a few pieces behind the wire: the job starts without waiting for it, and
            # the digest joins the status record with the job's first write (the 'queued' record carries it only
            # when it is ready by now)
            hash_fut = form.sha256_pending()
            if hash_fut is not None and not hash_fut.done():
                document_hash = None
            else:
                hash_fut = None
                document_hash = form.sha256_hex() or hashlib.sha256(file_bytes).hexdigest()
            existing = await jobs.claim(document_id, document_hash, file_part.filename or "")
            if existing is not None:
                _count("extract", 200)
                return {"status": existing.get("status", "unknown")}
            payload = file_bytes
            if seg is not None and file_part.offset >= 0:
                from ..runtime.workers import ShmDocument

                del file_bytes, parts                   # views of the mapping: the job's owner unmaps it when done
                payload = ShmDocument(seg, file_part.offset, len(file_part.data), seg_pool)
                file_part = None
                handed_over = True
            jobs.start(payload, document_id, document_hash, hash_future=hash_fut)
            _count("extract", 200)
            return {"status": "ok"}
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error handling PDF upload: {str(e)}")
            _count("extract", 500)
            return JSONResponse({"detail": f"Error processing PDF: {str(e)}"}, status_code=500)
        finally:
            inflight["bytes"] -= announced
            if seg is not None and not handed_over:
                seg.release()

    # ------------------------------------------------------------------ GET /check-status/{id}
    @app.get("/check-status/{document_id}")
    async def check_processing_status(document_id: str, request: Request):
        try:
            jobs: JobManager = request.app.state.ctx["jobs"]
            status_data = await jobs._io(jobs.status.get, document_id)
            if status_data:
                _count("check-status", 200)
                return {"status": public_status(status_data["status"])}
            _count("check-status", 404)
            return JSONResponse({"status": "unknown", "error": f"No record found for document ID: {document_id}"},
                                status_code=404)
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error checking status for {document_id}: {str(e)}")
            _count("check-status", 200)
            return {"status": "error", "message": str(e)}

    # ------------------------------------------------------------------ GET /result/{id}
    @app.get("/result/{document_id}")
    async def get_processing_result(document_id: str, request: Request):
        try:
            jobs: JobManager = request.app.state.ctx["jobs"]
            status_json = await jobs._io(jobs.status.get_raw, document_id)
            status_data = None
            if status_json:
                try:
                    status_data = json.loads(status_json)
                except ValueError:
                    logger.warning(f"Could not parse Redis status for {document_id}")
            if status_data and status_data.get("status") != "completed":
                st = status_data.get("status", "unknown")
                _count("result", 200)
                return JSONResponse({"document_id": document_id, "status": st,
                                     "message": f"Document processing status: {st}"})
            doc = await jobs._io(jobs.results.find, document_id)
            if doc:
                _count("result", 200)
                body = {"document_id": document_id, "text": doc.get("combined_text", ""), "status": "completed"}
                if status_data and status_data.get("warnings"):
                    # beyond the reference's three keys, and only when there is something to say: pages whose image was
                    # damaged or whose text was read with low confidence (a script the recogniser does not know)
                    body["warnings"] = status_data["warnings"]
                return JSONResponse(body)
            _count("result", 404)
            return JSONResponse({"error": f"No record found for document ID: {document_id}"}, status_code=404)
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error retrieving result for {document_id}: {str(e)}")
            logger.error(traceback.format_exc())
            _count("result", 500)
            return JSONResponse({"error": f"Error retrieving result: {str(e)}"}, status_code=500)

    # ------------------------------------------------------------------ GET /health
    @app.get("/health")
    async def health_check(request: Request):
        ctx = request.app.state.ctx
        status = {"status": "healthy", "version": SERVICE_VERSION, "timestamp": datetime.datetime.now().isoformat(),
                  "dependencies": {"redis": "unknown", "mongodb": "unknown"}}
        loop = asyncio.get_running_loop()
        jobs: Optional[JobManager] = ctx.get("jobs")
        pool = jobs.io_pool if jobs is not None else None
        try:
            ok = await loop.run_in_executor(pool, ctx["status"].ping)
            status["dependencies"]["redis"] = "healthy" if ok else "degraded"
        except Exception as e:  # noqa: BLE001
            logger.error(f"Redis health check failed: {str(e)}")
            status["dependencies"]["redis"] = "unhealthy"
        try:
            res = await loop.run_in_executor(pool, ctx["results"].ping)
            status["dependencies"]["mongodb"] = "healthy" if res.get("ok") == 1 else "degraded"
        except Exception as e:  # noqa: BLE001
            logger.error(f"MongoDB health check failed: {str(e)}")
            status["dependencies"]["mongodb"] = "unhealthy"
        deps = status["dependencies"].values()
        if all(v == "healthy" for v in deps):
            status["status"] = "healthy"
        elif any(v == "unhealthy" for v in deps):
            status["status"] = "unhealthy"
            _count("health", 503)
            return JSONResponse(status, status_code=503, headers={"x-ocr-frontend": settings.node_id})
        else:
            status["status"] = "degraded"
        _count("health", 200)
        # the body is the reference's (ocr_service.py:565-606); which front-end process answered goes in a header
        return JSONResponse(status, headers={"x-ocr-frontend": settings.node_id})

    # ------------------------------------------------------------------ GET /metrics
    @app.get("/metrics")
    async def prometheus_metrics(request: Request):
        ext = request.app.state.ctx.get("extractor")
        if ext is not None:
            metrics.refresh_from_extractor(ext)
        return Response(metrics.render(), media_type="text/plain; version=0.0.4")

    return app


def main() -> None:
    """``python -m distributed_llms_amd.api.app`` - the equivalent of the reference's ``uvicorn ocr_service:app``
    (services/ocr_service/Dockerfile:28, ocr_service.py:608-610)."""
    import uvicorn

    settings = Settings.from_env(dotenv=os.environ.get("OCR_DOTENV", ".env"))
    uvicorn.run(create_app(settings), host="0.0.0.0", port=settings.port, log_level="info")


if __name__ == "__main__":
    main()
