"""The in-kernel scheduler hands out macroblocks in the order built by common/mb_order.h.  Deadlock freedom rests on
that order being topological for the (left, top, top-left, top-right) neighbour graph of a slice / picture, and on
wh_mb_deps naming MBs whose completion implies all of those.  Checked here on the host for row-aligned, ragged and
single-MB-wide layouts."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <vector>
#include "mb_order.h"
static int check (int mb_w, int first, int last) {
  const int n = last - first;
  std::vector<uint16_t> ord (n);
  wh_build_mb_order (mb_w, first, last, ord.data());
  std::vector<int> pos (last, -1);
  for (int t = 0; t < n; ++t) { const int xy = ord[t]; if (xy < first || xy >= last || pos[xy] >= 0) return 1; pos[xy] = t; }
  for (int t = 0; t < n; ++t) {
    const int xy = ord[t], x = xy % mb_w;
    const int nb[4] = {x > 0 ? xy - 1 : -1, xy - mb_w, x > 0 ? xy - mb_w - 1 : -1, x < mb_w - 1 ? xy - mb_w + 1 : -1};
    for (int k = 0; k < 4; ++k) if (nb[k] >= first && pos[nb[k]] > t) return 2;          // neighbours come first
    int a, b;
    wh_mb_deps (mb_w, xy, first, &a, &b);
    // completion of a and b must imply completion of every in-range neighbour: each neighbour is a or b, or precedes
    // one of them in its row (rows complete left to right because of the left dependency)
    for (int k = 0; k < 4; ++k) {
      if (nb[k] < first) continue;
      bool ok = false;
      const int d[2] = {a, b};
      for (int j = 0; j < 2; ++j) if (d[j] >= 0 && d[j] / mb_w == nb[k] / mb_w && d[j] >= nb[k]) ok = true;
      if (!ok) return 3;
    }
  }
  return 0;
}
int main() {
  const int cases[][3] = {{120, 0, 8160}, {120, 2040, 4080}, {11, 0, 99}, {11, 25, 58}, {40, 13, 600}, {1, 0, 7}, {2, 1, 9}, {5, 0, 5}, {80, 3599, 3600}};
  for (auto& c : cases) { const int rc = check (c[0], c[1], c[2]); if (rc) { printf ("FAIL mb_w %d [%d,%d): %d\n", c[0], c[1], c[2], rc); return 1; } }
  printf ("OK\n");
  return 0;
}
'''


def test_mb_order_is_topological_and_deps_are_sufficient(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "openh264_amd", "csrc", "common"), "-o", str(exe), str(src)])
    assert subprocess.check_output([str(exe)]).decode().strip() == "OK"
