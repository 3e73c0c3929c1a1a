"""The XCD-grouped work order of the symmetric k-NN sweep (audio-metrics_amd/csrc/sym_groups.h, used by knn_wide_kernel):
host arithmetic, compiled here with g++ from the very header the kernel includes.  For a range of set sizes, chunk counts
and rank partitions the items of the grid(s) must cover every (row block, column tile) pair of the cyclic half-range
EXACTLY ONCE - the pairs sym_item() of pairwise_common.h assigns to a row block: offsets 0 .. T/2, the antipodal offset of
an even T to the lower-numbered block only - all items of a grid must have nearly the same length, and the 32 workgroups
an XCD holds at a time must share 8 row blocks and 4 column streams."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))                  # tools/experiments: sym_groups.h lives beside this file
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "audio-metrics_amd", "csrc")

PROGRAM = r"""
#include "sym_groups.h"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
using namespace am;
int main(int argc, char** argv) {
    const int64_t T = atoll(argv[1]);
    const int chunks = atoi(argv[2]), nparts = atoi(argv[3]), order = atoi(argv[4]);
    std::map<std::pair<int64_t, int64_t>, int> seen;
    long long items = 0, min_len = 1 << 30, max_len = 0, bad_groups = 0;
    for (int part = 0; part < nparts; ++part) {
        const int64_t grid = grp_grid(T, chunks, part, nparts);
        if (grid % 256 != 0) { printf("FAIL grid %lld not whole rounds\n", (long long)grid); return 1; }
        std::map<int64_t, std::set<int64_t>> rows_of_group, streams_of_group;
        for (int64_t b = 0; b < grid; ++b) {
            const GrpWork w = grp_item(T, chunks, part, nparts, b, order);
            if (w.ntiles == 0) continue;
            if ((int)(w.pb * nparts / T) != part) { printf("FAIL ownership pb %lld part %d\n", (long long)w.pb, part); return 1; }
            ++items;
            const GrpTiles tm(w.c0, w.h0, w.t0, T, w.ncommon, w.nhead);
            for (int t = 0; t < w.ntiles; ++t) {
                const int64_t q = tm(t);
                if (q < 0 || q >= T) { printf("FAIL tile %lld out of range\n", (long long)q); return 1; }
                ++seen[{w.pb, q}];
            }
            if (w.chunk < chunks - 1 || chunks == 1) {            // the last chunk carries the member's own tiles too
                min_len = w.ntiles < min_len ? w.ntiles : min_len;
            }
            max_len = w.ntiles > max_len ? w.ntiles : max_len;
            const int64_t group = ((b >> 3) >> 5) * 8 + (b & 7);   // the 32 workgroups an XCD holds at a time
            rows_of_group[group].insert(w.pb);
            if (w.ncommon > 0) streams_of_group[group].insert(w.c0);
        }
        for (auto& kv : rows_of_group)
            if (kv.second.size() > 8 || streams_of_group[kv.first].size() > 4) ++bad_groups;
    }
    // the reference assignment (sym_item): offsets 0 .. noff - 1 of every row block
    long long want = 0;
    for (int64_t pb = 0; pb < T; ++pb) {
        const int64_t noff = T / 2 + 1 - (((T % 2) == 0 && pb >= T / 2) ? 1 : 0);
        for (int64_t o = 0; o < noff; ++o) {
            ++want;
            auto it = seen.find({pb, (pb + o) % T});
            if (it == seen.end() || it->second != 1) { printf("FAIL pair (%lld, +%lld) covered %d times\n", (long long)pb, (long long)o, it == seen.end() ? 0 : it->second); return 1; }
        }
    }
    if ((long long)seen.size() != want) { printf("FAIL %zu pairs covered, %lld wanted\n", seen.size(), want); return 1; }
    printf("OK items %lld len %lld..%lld bad_groups %lld\n", items, min_len, max_len, bad_groups);
    return 0;
}
"""


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("sym_groups")
    src = d / "check.cpp"
    src.write_text(PROGRAM)
    exe = d / "check"
    r = subprocess.run([gxx, "-O1", "-std=c++17", "-I", HERE, "-I", CSRC, str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(exe)


@pytest.mark.parametrize("tiles,chunks,nparts", [
    (391, 12, 1), (391, 16, 1), (392, 12, 1),             # the BASELINE size (100 000 rows), odd and even tile counts
    (391, 12, 2), (391, 12, 8), (392, 16, 3),             # rank partitions: contiguous ranges of row blocks
    (128, 4, 1), (129, 4, 1), (130, 4, 4), (200, 8, 1),   # the smallest sets that take this order (plan_knn: >= 8 common tiles per chunk)
    (3907, 44, 1), (3907, 44, 8),                          # 1 000 000 rows
    (24, 4, 1), (17, 4, 2), (9, 4, 1), (40, 12, 16),      # degenerate: half-ranges shorter than a group, ranks without blocks
])
@pytest.mark.parametrize("order", [0, 1])
def test_grouped_items_cover_the_half_range_exactly_once(checker, tiles, chunks, nparts, order):
    r = subprocess.run([checker, str(tiles), str(chunks), str(nparts), str(order)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    fields = r.stdout.split()
    lo, hi = (int(v) for v in fields[4].split(".."))
    assert int(fields[6]) == 0, r.stdout                                       # every XCD group: <= 8 P blocks, <= 4 Q streams
    if tiles >= 128:
        # items of equal length to within two tiles (the last chunk's own part is 7 tiles; rounding of the chunk bounds)
        assert hi - lo <= 2 + (0 if tiles % 2 else 1), r.stdout
