// zkdl_amd/csrc/recorder.hip -- deferred execution of recorded launch chains, zipped across layers
// (see launch.hpp for the why).  Everything here is host code.
#include <algorithm>
#include <cstring>

#include "ctx.hpp"

namespace zk {

int rec_begin_chain(zk_ctx* ctx, Recorder* r) {
  if (!r->blob_dev) {
    r->blob_dev = static_cast<uint8_t*>(ctx->pool_alloc(REC_BLOB_CAP));
    if (!r->blob_dev) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: recorder blob");
    r->blob_cap = REC_BLOB_CAP;
    r->blob.clear();
  }
  if (r->chains.empty()) ctx->pending_recs.push_back(r);
  r->chains.emplace_back();
  ctx->rec = r;
  return ZK_OK;
}

void rec_end_chain(zk_ctx* ctx) { ctx->rec = nullptr; }

// small host data that recorded kernels read: lands in the blob, which is uploaded once before the chains run
void* rec_blob_put(zk_ctx* ctx, const void* src, size_t bytes) {
  Recorder* r = ctx->rec;
  size_t off = (r->blob.size() + 63) & ~(size_t)63;
  if (off + bytes > r->blob_cap / 2) return nullptr;  // the other half is kept for the job tables
  r->blob.resize(off + bytes);
  memcpy(r->blob.data() + off, src, bytes);
  return r->blob_dev + off;
}

int rec_opaque(zk_ctx* ctx, const char* name, uint64_t bytes, std::function<int(hipStream_t)> fn) {
  RecOp op;
  op.name = name;
  op.bytes = bytes;
  op.opaque = std::move(fn);
  ctx->rec->chains.back().push_back(std::move(op));
  return ZK_OK;
}

namespace {
struct Batch {
  const RecOp* first;
  dim3 grid;
  size_t table_off;  // job table inside the blob
  uint64_t bytes;
};
}  // namespace

int rec_execute(zk_ctx* ctx, Recorder* r) {
  if (r->chains.empty()) return ZK_OK;
  Recorder* outer = ctx->rec;
  ctx->rec = nullptr;  // launches below are real
  auto it = std::find(ctx->pending_recs.begin(), ctx->pending_recs.end(), r);
  if (it != ctx->pending_recs.end()) ctx->pending_recs.erase(it);

  // ---- zip: repeatedly take the largest group of chain heads that launch the same kernel
  const size_t nc = r->chains.size();
  std::vector<size_t> head(nc, 0);
  std::vector<Batch> batches;
  size_t remaining = 0;
  for (auto& c : r->chains) remaining += c.size();
  int rc = ZK_OK;
  std::vector<size_t> members;
  while (remaining) {
    // pick the key of the earliest chain's head unless another key has strictly more members
    size_t best = nc, best_count = 0;
    for (size_t c = 0; c < nc; c++) {
      if (head[c] >= r->chains[c].size()) continue;
      const RecOp& a = r->chains[c][head[c]];
      size_t count = 1;
      if (a.k)
        for (size_t d = c + 1; d < nc; d++) {
          if (head[d] >= r->chains[d].size()) continue;
          const RecOp& b = r->chains[d][head[d]];
          if (b.k == a.k && b.gy == a.gy && b.shmem == a.shmem) count++;
        }
      if (count > best_count) {
        best = c;
        best_count = count;
      }
    }
    const RecOp& a = r->chains[best][head[best]];
    Batch b;
    b.first = &a;
    b.bytes = 0;
    b.table_off = 0;
    if (!a.k) {
      b.grid = dim3(1);
      head[best]++;
      remaining--;
      batches.push_back(b);
      continue;
    }
    members.clear();
    uint32_t gx = 0;
    for (size_t d = best; d < nc; d++) {
      if (head[d] >= r->chains[d].size()) continue;
      const RecOp& o = r->chains[d][head[d]];
      if (o.k == a.k && o.gy == a.gy && o.shmem == a.shmem) {
        members.push_back(d);
        gx = std::max(gx, o.gx);
        b.bytes += o.bytes;
      }
    }
    size_t off = (r->blob.size() + 63) & ~(size_t)63;
    size_t need = (size_t)a.k->job_size * members.size();
    if (off + need > r->blob_cap) {
      rc = ctx->fail(ZK_ERR_ALLOC, "recorder blob exhausted");
      break;
    }
    r->blob.resize(off + need);
    for (size_t m = 0; m < members.size(); m++) {
      const RecOp& o = r->chains[members[m]][head[members[m]]];
      memcpy(r->blob.data() + off + m * a.k->job_size, r->args.data() + o.args_off, a.k->job_size);
      head[members[m]]++;
    }
    remaining -= members.size();
    b.table_off = off;
    b.grid = dim3(gx, a.gy, (uint32_t)members.size());
    batches.push_back(b);
  }

  // ---- upload the blob (uploads + job tables) through the pinned staging ring, then launch
  if (rc == ZK_OK) {
    const size_t total = r->blob.size(), step = 1u << 20;
    for (size_t o = 0; o < total && rc == ZK_OK; o += step) {
      size_t nb = std::min(step, total - o);
      void* st = ctx->stage_put(r->blob.data() + o, nb);
      rc = st ? dev_copy(ctx, r->blob_dev + o, st, (nb + 15) & ~(size_t)15) : ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
    }
  }
  for (size_t i = 0; i < batches.size() && rc == ZK_OK; i++) {
    const Batch& b = batches[i];
    int lr;
    {
      ProfScope ps(ctx, b.first->name, b.first->k ? b.bytes : b.first->bytes);
      lr = b.first->k ? b.first->k->launch_many(ctx->stream, b.grid, b.first->shmem, r->blob_dev + b.table_off) : b.first->opaque(ctx->stream);
    }
    if (lr != 0) rc = ctx->fail(ZK_ERR_DEVICE, std::string("launch of ") + b.first->name + " failed: " + hipGetErrorString(hipGetLastError()));
  }

  // ---- temporaries of the recorded calls: stream-ordered release behind the launches above
  ctx->pool_release(r->blob_dev);
  r->blob_dev = nullptr;
  for (void* p : r->deferred_free) ctx->pool_release(p);
  r->deferred_free.clear();
  r->chains.clear();
  r->args.clear();
  r->blob.clear();
  ctx->rec = outer;
  return rc;
}

}  // namespace zk

bool zk_ctx::rec_defer_free(void* p) {
  if (!rec) return false;
  rec->deferred_free.push_back(p);
  return true;
}

int zk_ctx::settle() {
  int rc = ZK_OK;
  while (!pending_recs.empty()) {
    int r2 = zk::rec_execute(this, pending_recs.front());
    if (rc == ZK_OK) rc = r2;
  }
  join_lanes();
  return rc;
}
