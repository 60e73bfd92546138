"""Status / result stores, RESP client against the in-process fake server, config, multipart."""
import json
import os

import pytest

from distributed_llms_amd.api.multipart import MultipartError, encode_multipart, parse_multipart
from distributed_llms_amd.config import Settings, load_dotenv
from distributed_llms_amd.runtime.document import combine_text
from distributed_llms_amd.store.resp import FakeRedisServer, RespClient, RespConnectionError, RespError, encode_command
from distributed_llms_amd.store.results import ErrorSink, LocalFileWriter, MemoryResultStore, make_result_document
from distributed_llms_amd.store.status import MemoryStatusStore, RedisStatusStore, public_status


class Clock:
    def __init__(self):
        self.t = 1000.0

    def __call__(self):
        return self.t


def test_resp_wire_format():
    assert encode_command("SET", "k", "v") == b"*3\r\n$3\r\nSET\r\n$1\r\nk\r\n$1\r\nv\r\n"
    assert encode_command("SETEX", "k", 5, "é") == b"*4\r\n$5\r\nSETEX\r\n$1\r\nk\r\n$1\r\n5\r\n$2\r\n\xc3\xa9\r\n"


def test_resp_client_against_fake_server_with_auth_and_ttl():
    clk = Clock()
    with FakeRedisServer(password="s3cret", clock=clk) as srv:
        c = RespClient("127.0.0.1", srv.port, password="s3cret")
        assert c.ping()
        assert c.get("missing") is None
        assert c.setex("a", 10, "1") and c.get("a") == "1" and c.exists("a") == 1
        assert 0 < c.ttl("a") <= 10
        assert c.set("a", "2", ex=10, nx=True) is False and c.get("a") == "1"     # NX honours existing keys
        assert c.set("b", "x", ex=10, nx=True) is True
        clk.t += 11
        assert c.get("a") is None and c.exists("a", "b") == 0                      # expired
        for i in range(7):
            c.setex(f"doc:{i}", 100, "v")
        c.setex("hash:zz", 100, "v")
        assert sorted(c.scan_iter("doc:*", count=3)) == [f"doc:{i}" for i in range(7)]
        assert c.delete("doc:0", "doc:1") == 2
        with pytest.raises(RespError):
            c.execute("NOSUCHCOMMAND")
        c.close()
        bad = RespClient("127.0.0.1", srv.port, password="wrong")
        with pytest.raises((RespError, RespConnectionError)):
            bad.ping()
    down = RespClient("127.0.0.1", srv.port, timeout=0.5)
    with pytest.raises(RespConnectionError):
        down.ping()


@pytest.mark.parametrize("kind", ["memory", "redis"])
def test_status_store_lifecycle_ttl_claim_and_sweep(kind):
    clk = Clock()
    srv = None
    if kind == "redis":
        srv = FakeRedisServer(clock=clk).start()
        store = RedisStatusStore(RespClient("127.0.0.1", srv.port), ttl=86400)
    else:
        store = MemoryStatusStore(ttl=86400, clock=clk)
    try:
        assert store.ping()
        assert store.get("d") is None and not store.exists("d")
        assert store.claim("d", {"status": "queued", "document_hash": "abc", "owner": "n1"})
        assert not store.claim("d", {"status": "queued", "document_hash": "abc"})   # second replica loses the race
        assert store.document_for_hash("abc") == "d"
        assert store.get("d")["status"] == "queued" and public_status("queued") == "processing"
        clk.t += 80000
        assert store.update("d", {"status": "processing", "owner": "n1"})           # every write re-arms the 24 h TTL
        clk.t += 80000
        assert store.get("d")["status"] == "processing"
        assert store.update("e", {"status": "processing", "owner": "n2"})
        assert store.sweep_orphans(owner="n1") == 1
        assert store.get("d")["status"] == "failed" and "orphaned" in store.get("d")["error"]
        assert store.get("e")["status"] == "processing"
        assert public_status("failed") == "error" and public_status("completed") == "completed"
        clk.t += 86401
        assert store.get("d") is None                                                # TTL lapsed
        assert store.update(None, {"status": "queued"}) is False
    finally:
        store.close()
        if srv:
            srv.stop()


def test_status_store_swallows_connection_errors():
    store = RedisStatusStore(RespClient("127.0.0.1", 1, timeout=0.2))
    assert store.update("x", {"status": "queued"}) is False
    assert store.get("x") is None and store.exists("x") is False
    with pytest.raises(Exception):
        store.get_raw("x")          # /result maps this to a 500 like the reference


def test_result_document_schema_and_error_sink(tmp_path):
    doc = make_result_document("id1", "two words\n\nmore")
    assert list(doc) == ["document_id", "combined_text", "status", "processed_at", "word_count", "character_count"]
    assert doc["word_count"] == 3 and doc["character_count"] == 15 and doc["status"] == "completed"
    rs = MemoryResultStore()
    rs.ensure_indexes()
    assert rs.indexes == ["document_id", "document_hash", "status", "processed_at"]
    rs.upsert(doc)
    rs.upsert(make_result_document("id1", "new"))
    assert rs.count() == 1 and rs.find("id1")["combined_text"] == "new" and rs.find("zz") is None
    sink = ErrorSink(str(tmp_path / "out"))
    assert sink.save("a/b", "bad thing")
    payload = json.loads(open(sink.path_for("a/b")).read())
    assert payload["document_id"] == "a/b" and payload["status"] == "failed" and payload["error"] == "bad thing"
    w = LocalFileWriter(str(tmp_path / "img"))
    p = w.write("fig/1.png", b"\x89PNG")
    assert open(p, "rb").read() == b"\x89PNG"
    with pytest.raises(ValueError):
        w.write("../escape.bin", b"x")


def test_combine_text_rule_and_markdown_fallback():
    items = [{"type": "text", "text": "First"}, {"type": "image", "img_path": "x.png"}, "bare string",
             {"type": "table", "html": "<table/>"}, {"type": "text", "text": "Last"}]
    assert combine_text(items) == "First\n\nbare string\n\nLast"
    md = "# Title\n\n**bold** and *it* __u__ | cell | --- ___ text"
    out = combine_text([{"type": "image"}], md)
    assert "#" not in out and "*" not in out and "|" not in out and "Title" in out and "bold" in out
    assert combine_text([], "") == ""


def test_settings_env_names_defaults_and_no_default_credential(tmp_path):
    s = Settings.from_env({})
    assert s.mongo_connection_string is None and s.resolved_store() == "memory"
    assert (s.mongo_db_name, s.mongo_collection_name, s.redis_host, s.redis_port, s.port) == \
        ("ocr_service", "documents", "localhost", 6379, 8000)
    with pytest.raises(ValueError):
        Settings.from_env({"OCR_STORE": "redis+mongo"})
    s = Settings.from_env({"MONGO_CONNECTION_STRING": "mongodb://u:p@h/db", "REDIS_PASSWORD": "pw", "PORT": "9000",
                           "OCR_NUM_GPUS": "8", "OCR_RECOGNIZER": "trocr"})
    assert s.resolved_store() == "redis+mongo" and s.port == 9000 and s.num_gpus == 8 and s.recognizer == "trocr"
    assert s.describe()["mongo_connection_string"] == "***" and s.describe()["redis_password"] == "***"
    with pytest.raises(ValueError):
        Settings.from_env({"REDIS_PORT": "abc"})
    env = {}
    p = tmp_path / ".env"
    p.write_text("# comment\nexport REDIS_HOST=10.0.0.5\nMONGO_DB_NAME='quoted name'\nPORT=8123 # trailing\n")
    load_dotenv(p, environ=env)
    assert env == {"REDIS_HOST": "10.0.0.5", "MONGO_DB_NAME": "quoted name", "PORT": "8123"}


def test_multipart_roundtrip_and_errors():
    body, ctype = encode_multipart({"document_id": "id-7", "note": "ü"}, {"file": ("a b.pdf", b"%PDF\r\n--x\r\n\x00\xff", "application/pdf")})
    parts = {p.name: p for p in parse_multipart(body, ctype)}
    assert parts["document_id"].text() == "id-7" and parts["note"].text() == "ü"
    assert parts["file"].is_file and parts["file"].filename == "a b.pdf" and parts["file"].data == b"%PDF\r\n--x\r\n\x00\xff"
    with pytest.raises(MultipartError):
        parse_multipart(body, "application/json")
    with pytest.raises(MultipartError):
        parse_multipart(b"no boundary here", ctype)
