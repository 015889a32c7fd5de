// zkdl_amd/csrc/recorder.hip -- deferred execution of recorded launch chains, zipped across layers
// (see launch.hpp for the why).  Everything here is host code.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "ctx.hpp"

namespace zk {

int rec_begin_chain(zk_ctx* ctx, Recorder* r, uint32_t group) {
  if (!r->blob_dev) {
    r->blob_dev = static_cast<uint8_t*>(ctx->pool_alloc(REC_BLOB_CAP));
    if (!r->blob_dev) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: recorder blob");
    r->blob_cap = REC_BLOB_CAP;
    r->blob.clear();
  }
  if (r->chains.empty()) ctx->pending_recs.push_back(r);
  r->chains.emplace_back();
  r->chains.back().group = group;
  ctx->rec = r;
  return ZK_OK;
}

void rec_split(zk_ctx* ctx, uint32_t group, bool early) {
  Recorder* r = ctx->rec;
  if (!r) return;
  if (!r->chains.back().ops.empty()) r->chains.emplace_back();
  r->chains.back().group = group;
  r->chains.back().early = early;
}

void rec_end_chain(zk_ctx* ctx) { ctx->rec = nullptr; }

// A prove call failed while recording: forget the chains it added (everything after the first `keep_chains`).
// Nothing of those chains was launched, so their temporaries could go back at once; they simply stay on the
// deferred list until the surviving chains run -- unless no chain survives, in which case the recorder is
// emptied here (rec_execute treats a recorder without chains as "nothing to do").
void rec_rollback(zk_ctx* ctx, Recorder* r, size_t keep_chains) {
  if (ctx->rec == r) ctx->rec = nullptr;
  if (r->chains.size() > keep_chains) r->chains.resize(keep_chains);
  if (!r->chains.empty()) return;
  auto it = std::find(ctx->pending_recs.begin(), ctx->pending_recs.end(), r);
  if (it != ctx->pending_recs.end()) ctx->pending_recs.erase(it);
  if (r->blob_dev) ctx->pool_release(r->blob_dev);
  r->blob_dev = nullptr;
  for (void* p : r->deferred_free) ctx->pool_release(p);
  r->deferred_free.clear();
  r->args.clear();
  r->blob.clear();
}

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
  ctx->rec->chains.back().ops.push_back(std::move(op));
  return ZK_OK;
}

namespace {
struct Batch {
  const RecOp* first;
  dim3 grid;
  size_t table_off;  // job table inside the blob
  uint64_t bytes;
  uint32_t group;
  bool early = false;      // belongs to the group's early chains
  bool early_end = false;  // last launch of the group's early chains
};
inline bool same_key(const RecOp& a, const RecOp& b) { return a.k && b.k == a.k && b.gy == a.gy && b.shmem == a.shmem; }
}  // namespace

int rec_execute(zk_ctx* ctx, Recorder* r, bool join, std::vector<hipEvent_t>* early_out) {
  if (r->chains.empty()) return ZK_OK;
  Recorder* outer = ctx->rec;
  ctx->rec = nullptr;  // launches below are real
  auto it = std::find(ctx->pending_recs.begin(), ctx->pending_recs.end(), r);
  if (it != ctx->pending_recs.end()) ctx->pending_recs.erase(it);

  // ---- zip, group by group: the chain with the most launches left picks the kernel (so a chain that fell
  // behind -- one round fewer, a different variant -- catches up instead of trailing to the end), and every
  // other chain of the group whose head is the same kernel joins that launch
  const size_t nc = r->chains.size();
  std::vector<size_t> head(nc, 0);
  std::vector<Batch> batches;
  std::vector<uint32_t> groups;
  for (auto& c : r->chains)
    if (std::find(groups.begin(), groups.end(), c.group) == groups.end()) groups.push_back(c.group);
  std::sort(groups.begin(), groups.end());
  int rc = ZK_OK;
  std::vector<size_t> members;
  for (uint32_t g : groups) {
    bool early_phase = false;  // the group's early chains are zipped (among themselves) and launched before its other chains
    for (auto& c : r->chains)
      if (c.group == g && c.early && !c.ops.empty()) early_phase = true;
    for (;;) {
      size_t best = nc, best_left = 0;
      for (size_t c = 0; c < nc; c++) {
        if (r->chains[c].group != g || (early_phase && !r->chains[c].early)) continue;
        size_t left = r->chains[c].ops.size() - head[c];
        if (left > best_left) {
          best = c;
          best_left = left;
        }
      }
      if (best == nc) {
        if (!early_phase) break;
        early_phase = false;
        if (!batches.empty() && batches.back().group == g) batches.back().early_end = true;
        continue;
      }
      const RecOp& a = r->chains[best].ops[head[best]];
      Batch b;
      b.first = &a;
      b.bytes = 0;
      b.table_off = 0;
      b.group = g;
      b.early = early_phase;
      if (!a.k) {
        b.grid = dim3(1);
        head[best]++;
        batches.push_back(b);
        continue;
      }
      members.clear();
      uint32_t gx = 0;
      for (size_t d = 0; d < nc; d++) {
        if (r->chains[d].group != g || head[d] >= r->chains[d].ops.size() || (early_phase && !r->chains[d].early)) continue;
        const RecOp& o = r->chains[d].ops[head[d]];
        if (same_key(a, o)) {
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
        const RecOp& o = r->chains[members[m]].ops[head[members[m]]];
        memcpy(r->blob.data() + off + m * a.k->job_size, r->args.data() + o.args_off, a.k->job_size);
        head[members[m]]++;
      }
      b.table_off = off;
      b.grid = dim3(gx, a.gy, (uint32_t)members.size());
      batches.push_back(b);
    }
    if (rc != ZK_OK) break;
  }

  // ---- upload the blob (uploads + job tables) through the pinned staging ring
  if (rc == ZK_OK) {
    const size_t total = r->blob.size(), step = 1u << 20;
    for (size_t o = 0; o < total && rc == ZK_OK; o += step) {
      size_t nb = std::min(step, total - o);
      void* st = ctx->stage_put(r->blob.data() + o, nb);
      rc = st ? dev_copy(ctx, r->blob_dev + o, st, (nb + 15) & ~(size_t)15) : ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
    }
  }
  // ---- launch: group 0 on the current stream, group g > 0 on worker lane g-1 (forked behind the upload)
  hipStream_t home = ctx->stream;
  const bool fan_out = ctx->lanes_enabled && !ctx->lanes.empty() && home == ctx->main_stream;
  std::vector<zk_ctx::Lane*> used;
  if (fan_out) {  // fork first, so that no group waits for another group's launches
    (void)hipEventRecord(ctx->fork_ev, home);
    for (uint32_t g : groups) {
      if (g == 0) continue;
      zk_ctx::Lane* l = &ctx->lanes[(g - 1) % ctx->lanes.size()];
      if (std::find(used.begin(), used.end(), l) != used.end()) continue;
      (void)hipStreamWaitEvent(l->stream, ctx->fork_ev, 0);
      used.push_back(l);
    }
  }
  auto stream_of = [&](uint32_t g) -> hipStream_t { return (g == 0 || !fan_out) ? home : ctx->lanes[(g - 1) % ctx->lanes.size()].stream; };
  static const bool skip_fr = experiment_on("ZKDL_EXPERIMENT_SKIP_FR");  // timing experiments only: nothing recorded is launched
  // early batches of every group are enqueued first (the host needs ~0.5 ms to enqueue a proof's ~90 launches: whoever waits for the
  // early events should not wait for the host to get through the other groups); per stream the order is unchanged
  std::vector<size_t> order;
  order.reserve(batches.size());
  for (size_t i = 0; i < batches.size(); i++)
    if (batches[i].early) order.push_back(i);
  for (size_t i = 0; i < batches.size(); i++)
    if (!batches[i].early) order.push_back(i);
  for (size_t oi = 0; oi < order.size() && rc == ZK_OK && !skip_fr; oi++) {
    const Batch& b = batches[order[oi]];
    ctx->stream = stream_of(b.group);
    int lr;
    {
      ProfScope ps(ctx, b.first->name, b.first->k ? b.bytes : b.first->bytes);
      lr = b.first->k ? b.first->k->launch_many(ctx->stream, b.grid, b.first->shmem, r->blob_dev + b.table_off) : b.first->opaque(ctx->stream);
    }
    if (lr != 0) rc = ctx->fail(ZK_ERR_DEVICE, std::string("launch of ") + b.first->name + " failed: " + hipGetErrorString(hipGetLastError()));
    if (b.early_end && early_out && rc == ZK_OK) {
      hipEvent_t e = ctx->get_event();
      if (e) {
        (void)hipEventRecord(e, ctx->stream);
        early_out->push_back(e);
      }
    }
  }
  ctx->stream = home;
  for (zk_ctx::Lane* l : used) {
    (void)hipEventRecord(l->done, l->stream);
    if (join) (void)hipStreamWaitEvent(home, l->done, 0);  // everything after this on the home stream sees the groups' results
    else l->dirty = true;                                   // unjoined: whoever needs the results waits on the lanes' events
  }

  // ---- temporaries of the recorded calls.  Joined: stream-ordered release behind the launches above.  Unjoined: the next proof's
  // groups run right behind this one's, stream by stream, so a block freed here could be handed to ANOTHER group's stream while this
  // proof still reads it: the blocks (and the blob) stay with the recorder until rec_release_held().
  if (join) {
    ctx->pool_release(r->blob_dev);
    for (void* p : r->deferred_free) ctx->pool_release(p);
  } else {
    r->held_free.push_back(r->blob_dev);
    for (void* p : r->deferred_free) r->held_free.push_back(p);
  }
  r->blob_dev = nullptr;
  r->deferred_free.clear();
  r->chains.clear();
  r->args.clear();
  r->blob.clear();
  ctx->rec = outer;
  return rc;
}

// Move every recorded chain of `src` into `dst`, so that ONE rec_execute(dst) zips the launches of both (two proofs recorded on two
// transcripts: every batched launch then carries the jobs of both, and the latency-bound small launches of a proof are shared).
// src's uploads already have their device addresses inside src's blob: it is uploaded here and travels with dst's temporaries.
int rec_merge(zk_ctx* ctx, Recorder* dst, Recorder* src) {
  if (src->chains.empty()) return ZK_OK;
  if (dst->chains.empty()) return ZK_OK;  // nothing to zip with: the caller executes src on its own
  // upload src's blob FIRST: if staging fails nothing has been spliced and both recorders are as they were
  {
    const size_t total = src->blob.size(), step = 1u << 20;
    for (size_t o = 0; o < total; o += step) {
      size_t nb = std::min(step, total - o);
      void* st = ctx->stage_put(src->blob.data() + o, nb);
      if (!st) return ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
      ZK_TRY(dev_copy(ctx, src->blob_dev + o, st, (nb + 15) & ~(size_t)15));
    }
  }
  const size_t base = dst->args.size();
  dst->args.insert(dst->args.end(), src->args.begin(), src->args.end());
  for (auto& c : src->chains) {
    for (auto& op : c.ops)
      if (op.k) op.args_off += base;
    dst->chains.push_back(std::move(c));
  }
  if (src->blob_dev) dst->deferred_free.push_back(src->blob_dev);
  for (void* p : src->deferred_free) dst->deferred_free.push_back(p);
  src->blob_dev = nullptr;
  src->deferred_free.clear();
  src->chains.clear();
  src->args.clear();
  src->blob.clear();
  auto it = std::find(ctx->pending_recs.begin(), ctx->pending_recs.end(), src);
  if (it != ctx->pending_recs.end()) ctx->pending_recs.erase(it);
  return ZK_OK;
}

void rec_release_held(zk_ctx* ctx, Recorder* r) {
  for (void* p : r->held_free) ctx->pool_release(p);
  r->held_free.clear();
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
