// zkdl_amd/csrc/stage_ring.hpp -- index arithmetic of the pinned staging ring (zk_ctx::stage_put), kept free of
// HIP so that tests/test_stage_ring.py can compile it with g++ and walk it around the ring on the CPU.
//
// The ring is cut into SEGS segments.  A put never straddles a segment boundary: when it does not fit behind
// the previous one, the segment being filled is LEFT (the caller records events on every consuming stream) and
// the next one is ENTERED (the caller waits on the events recorded when it was last left).  The segment is
// tracked explicitly: after an exact fit the offset sits on the boundary and would otherwise name the next one.
#pragma once
#include <stddef.h>

namespace zk {

struct StageRing {
  size_t cap = 0, segs = 8, off = 0, seg = 0;
  size_t seg_size() const { return cap / segs; }
  // Places `need` bytes (already rounded up by the caller).  Returns false if it can never fit.  On success *at
  // is the offset to write at; *rolled says a boundary was crossed: *left is the segment to mark, seg the one
  // to wait on before writing.
  bool place(size_t need, size_t* at, bool* rolled, size_t* left) {
    const size_t ss = seg_size();
    if (need > ss) return false;
    *rolled = false;
    if (off + need > (seg + 1) * ss) {
      *rolled = true;
      *left = seg;
      seg = (seg + 1) % segs;
      off = seg * ss;
    }
    *at = off;
    off += need;
    return true;
  }
};

}  // namespace zk
