#!/usr/bin/env bash
# Redis with append-only persistence and a data volume (status store, 24 h TTL keys).
set -euo pipefail
command -v docker >/dev/null 2>&1 || { curl -fsSL https://get.docker.com | sh; }
sudo systemctl enable --now docker
mkdir -p "${REDIS_DATA_DIR:-$HOME/redis-data}"
docker rm -f ocr-redis >/dev/null 2>&1 || true
ARGS=(redis-server --appendonly yes)
[ -n "${REDIS_PASSWORD:-}" ] && ARGS+=(--requirepass "$REDIS_PASSWORD")
docker run -d --name ocr-redis --restart always -p 6379:6379 \
  -v "${REDIS_DATA_DIR:-$HOME/redis-data}:/data" redis:7-alpine "${ARGS[@]}"
docker ps --filter name=ocr-redis
