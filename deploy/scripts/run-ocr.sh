#!/usr/bin/env bash
# Start (or restart) the service container with every GPU of the node.
set -euo pipefail
REPO_DIR="$(cd "$(dirname "$0")/../.." && pwd)"
ENV_FILE="${ENV_FILE:-$REPO_DIR/.env}"
[ -f "$ENV_FILE" ] || { echo "no $ENV_FILE (copy deploy/.env.template)" >&2; exit 1; }
docker rm -f ocr-service >/dev/null 2>&1 || true
docker run -d --name ocr-service --restart always \
  --device /dev/kfd --device /dev/dri --group-add video --group-add render \
  --security-opt seccomp=unconfined --ipc host --shm-size 16g \
  --env-file "$ENV_FILE" -p "${PORT:-8000}:8000" \
  -v "$REPO_DIR/output:/app/output" -v "$REPO_DIR/weights:/app/weights:ro" \
  dlamd/ocr-service:latest
sleep 5
docker ps --filter name=ocr-service
curl -fsS "http://127.0.0.1:${PORT:-8000}/health" || echo "(service still warming up: models load and kernels warm at start-up)"
