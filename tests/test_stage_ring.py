"""Host-only check of the pinned staging ring's index arithmetic (zkdl_amd/csrc/stage_ring.hpp, used by
zk_ctx::stage_put): exact-fit puts (2 x 1 MiB, 1 x 2 MiB -- the chunk size rec_execute uploads its blob in)
walked around the whole ring must never write past the buffer, never straddle a segment, and must mark the
segment they leave / wait on the one they enter exactly once per lap."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "stage_ring.hpp"
int main() {
  const size_t CAP = 16u << 20, SEGS = 8, SS = CAP / SEGS;
  const size_t patterns[][4] = {{1u << 20, 1u << 20, 0, 0}, {2u << 20, 0, 0, 0}, {64, 1u << 20, (1u << 20) - 64, 0},
                                {1u << 20, 64, 0, 0}, {(2u << 20) - 64, 64, 64, 0}, {192, 0, 0, 0}};
  for (auto& pat : patterns) {
    zk::StageRing r;
    r.cap = CAP;
    r.segs = SEGS;
    std::vector<int> marked(SEGS, 0), waited(SEGS, 0);
    size_t puts = 0, rolls = 0;
    for (int lap = 0; lap < 40000; lap++) {
      for (int i = 0; i < 4 && pat[i]; i++) {
        size_t at = 0, left = 0;
        bool rolled = false;
        if (!r.place(pat[i], &at, &rolled, &left)) { printf("FAIL place\n"); return 1; }
        if (at + pat[i] > CAP) { printf("FAIL overflow at=%zu need=%zu\n", at, pat[i]); return 1; }
        if (at / SS != (at + pat[i] - 1) / SS) { printf("FAIL straddle at=%zu need=%zu\n", at, pat[i]); return 1; }
        if (at / SS != r.seg) { printf("FAIL seg mismatch at=%zu seg=%zu\n", at, r.seg); return 1; }
        if (rolled) {
          if (left >= SEGS || r.seg != (left + 1) % SEGS || at != r.seg * SS) { printf("FAIL roll left=%zu seg=%zu at=%zu\n", left, r.seg, at); return 1; }
          marked[left]++;
          waited[r.seg]++;
          rolls++;
        }
        puts++;
      }
      if (puts > 200000) break;
    }
    // every segment is left and entered equally often (+-1 for the lap in flight)
    for (size_t s = 0; s < SEGS; s++)
      if (abs(marked[s] - marked[0]) > 1 || abs(waited[s] - waited[0]) > 1) { printf("FAIL uneven marks\n"); return 1; }
    if (rolls == 0) { printf("FAIL never rolled\n"); return 1; }
  }
  zk::StageRing r;
  r.cap = CAP;
  r.segs = SEGS;
  size_t at, left;
  bool rolled;
  if (r.place(SS + 64, &at, &rolled, &left)) { printf("FAIL oversize accepted\n"); return 1; }
  printf("OK\n");
  return 0;
}
"""


def test_stage_ring_exact_fit_sizes_never_leave_the_ring(tmp_path):
    src = tmp_path / "ring.cpp"
    src.write_text(SRC)
    exe = tmp_path / "ring"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "zkdl_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr
