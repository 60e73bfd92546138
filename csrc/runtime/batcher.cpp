// Page-level dynamic batcher - native host component of the GPU worker.
//
// Pages (detector inputs) and crops arrive one by one from the decode pool, possibly from many
// documents at once.  A batch must be shape-uniform (one detector bucket Hd x Wd), fit the HBM
// budget of the device (288 GB on MI355X minus weights/workspaces), and must not make the
// oldest item wait longer than ``max_wait``.  ``pop`` returns the next batch to launch:
//   * a bucket that already holds ``max_items`` (or whose byte cost reaches the budget) is ready;
//   * otherwise the bucket holding the oldest item becomes ready when that item has waited
//     ``max_wait_us`` (or immediately when ``flush`` is requested);
//   * among ready buckets the one with the oldest head goes first (FIFO fairness across documents).
// Thread-safe: producers call push() from decode threads, the engine thread blocks in pop().
// Exposed through a C ABI for ctypes; there is no Python in the hot path besides the call itself.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {

using Clock = std::chrono::steady_clock;

struct Item {
  uint64_t id;
  uint64_t cost;       // bytes of device memory this item needs while in flight
  int64_t arrival_us;
};

struct Bucket {
  std::deque<Item> q;
  uint64_t bytes = 0;
};

struct Batcher {
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, Bucket> buckets;  // key = (Hd << 32) | Wd, or any caller-defined shape key
  uint64_t budget_bytes;
  int max_items;
  int64_t max_wait_us;
  bool closed = false;
  uint64_t pushed = 0, popped = 0, batches = 0, full_batches = 0, deadline_batches = 0;
  Clock::time_point t0 = Clock::now();

  int64_t now_us() const { return std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count(); }
};

// Must be called with the lock held.  Returns the key of the bucket to launch, or false.
bool pick(Batcher* b, bool flush, int64_t now, uint64_t* key_out, int64_t* next_deadline, bool* by_deadline) {
  bool found = false;
  int64_t best_age = -1;
  *next_deadline = -1;
  for (auto& kv : b->buckets) {
    Bucket& bk = kv.second;
    if (bk.q.empty()) continue;
    const int64_t head = bk.q.front().arrival_us;
    const bool full = (int)bk.q.size() >= b->max_items || bk.bytes >= b->budget_bytes;
    const bool due = flush || b->closed || (now - head) >= b->max_wait_us;
    if (full || due) {
      const int64_t age = now - head;
      if (!found || age > best_age) {
        found = true;
        best_age = age;
        *key_out = kv.first;
        *by_deadline = !full;
      }
    } else {
      const int64_t dl = head + b->max_wait_us;
      if (*next_deadline < 0 || dl < *next_deadline) *next_deadline = dl;
    }
  }
  return found;
}

}  // namespace

extern "C" {

void* dlrt_batcher_create(uint64_t budget_bytes, int max_items, int64_t max_wait_us) {
  if (max_items <= 0) return nullptr;
  auto* b = new Batcher();
  b->budget_bytes = budget_bytes ? budget_bytes : ~0ull;
  b->max_items = max_items;
  b->max_wait_us = max_wait_us < 0 ? 0 : max_wait_us;
  return b;
}

void dlrt_batcher_destroy(void* h) { delete (Batcher*)h; }

int dlrt_batcher_push(void* h, uint64_t key, uint64_t id, uint64_t cost) {
  auto* b = (Batcher*)h;
  {
    std::lock_guard<std::mutex> g(b->mu);
    if (b->closed) return -1;
    Bucket& bk = b->buckets[key];
    bk.q.push_back(Item{id, cost, b->now_us()});
    bk.bytes += cost;
    ++b->pushed;
  }
  b->cv.notify_all();
  return 0;
}

// No more items will arrive: pending buckets drain immediately, then pop returns 0 forever.
void dlrt_batcher_close(void* h) {
  auto* b = (Batcher*)h;
  {
    std::lock_guard<std::mutex> g(b->mu);
    b->closed = true;
  }
  b->cv.notify_all();
}

// Blocks up to timeout_us (0: poll, < 0: forever) for the next batch.  Writes up to ``cap`` ids
// (cap >= max_items) and the bucket key.  Returns the batch size, 0 on timeout / closed-and-empty.
int dlrt_batcher_pop(void* h, int flush, int64_t timeout_us, uint64_t* ids, int cap, uint64_t* key_out) {
  auto* b = (Batcher*)h;
  std::unique_lock<std::mutex> lk(b->mu);
  const int64_t start = b->now_us();
  while (true) {
    const int64_t now = b->now_us();
    uint64_t key = 0;
    int64_t next_deadline = -1;
    bool by_deadline = false;
    if (pick(b, flush != 0, now, &key, &next_deadline, &by_deadline)) {
      Bucket& bk = b->buckets[key];
      int n = 0;
      uint64_t bytes = 0;
      while (!bk.q.empty() && n < cap && n < b->max_items) {
        const Item& it = bk.q.front();
        if (n > 0 && bytes + it.cost > b->budget_bytes) break;  // always admit at least one item
        bytes += it.cost;
        ids[n++] = it.id;
        bk.q.pop_front();
      }
      bk.bytes -= bytes;
      if (bk.q.empty()) b->buckets.erase(key);
      b->popped += n;
      ++b->batches;
      if (by_deadline) ++b->deadline_batches; else ++b->full_batches;
      if (key_out) *key_out = key;
      return n;
    }
    bool empty = true;
    for (auto& kv : b->buckets) if (!kv.second.q.empty()) { empty = false; break; }
    if (b->closed && empty) return 0;
    int64_t wait_us = -1;  // forever
    if (timeout_us >= 0) {
      wait_us = timeout_us - (now - start);
      if (wait_us <= 0) return 0;
    }
    if (next_deadline >= 0) {
      const int64_t d = next_deadline - now;
      if (wait_us < 0 || d < wait_us) wait_us = d > 0 ? d : 0;
    }
    if (wait_us < 0) b->cv.wait(lk);
    else b->cv.wait_for(lk, std::chrono::microseconds(wait_us + 50));
  }
}

int dlrt_batcher_pending(void* h) {
  auto* b = (Batcher*)h;
  std::lock_guard<std::mutex> g(b->mu);
  int n = 0;
  for (auto& kv : b->buckets) n += (int)kv.second.q.size();
  return n;
}

// stats[0..4] = pushed, popped, batches, full_batches, deadline_batches
void dlrt_batcher_stats(void* h, uint64_t* stats) {
  auto* b = (Batcher*)h;
  std::lock_guard<std::mutex> g(b->mu);
  stats[0] = b->pushed; stats[1] = b->popped; stats[2] = b->batches; stats[3] = b->full_batches;
  stats[4] = b->deadline_batches;
}

}  // extern "C"
