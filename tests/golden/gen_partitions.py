"""Generates tests/golden/partitions.json — partition-plan fixtures the reference's own tests pin:

 * equal plan: rows_per_rank = ceil(N / W), clipped (cpp/src/wholememory/memory_handle.cpp:1618-1635);
 * host_random_partition (cpp/tests/wholememory_ops/embedding_test_utils.cu:531-546): libstdc++
   std::default_random_engine(0) + std::uniform_int_distribution<size_t>(90, 100), scaled to the total and the
   remainder added to rank 0 — regenerated here by compiling a 20-line C++ program with this image's g++
   (same libstdc++ algorithms as the reference build would use);
 * python random_partition (python/.../test_utils/test_comm.py:188-195): np.random.seed(42),
   np.random.uniform(90, 100, world) scaled to the total, remainder to rank 0 — restated below.
The prefix-sum / same_chunk expectations are derived in plain Python from the rule of
memory_handle.cpp:69-79 and :1607-1616. Run: python tests/golden/gen_partitions.py
"""
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CPP = r"""
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char** argv) {
  size_t total = strtoull(argv[1], nullptr, 10); int count = atoi(argv[2]);
  std::default_random_engine random_engine(0);
  std::uniform_int_distribution<size_t> uniform(90, 100);
  std::vector<size_t> p(count); size_t acc = 0, sum = 0;
  for (int i = 0; i < count; i++) { p[i] = (size_t)uniform(random_engine); sum += p[i]; }
  for (int i = 0; i < count; i++) { p[i] = (size_t)((p[i] / (double)sum) * total); acc += p[i]; }
  p[0] += total - acc;
  for (int i = 0; i < count; i++) printf("%zu ", p[i]);
  return 0;
}
"""


def prefix_and_same(sizes):
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    same = all(sizes[i] == sizes[i + 1] for i in range(len(sizes) - 2))
    return offs, same


def python_random_partition(total, world):
    # test_comm.py:188-195
    np.random.seed(42)
    random_array = np.random.uniform(90, 100, size=world)
    random_sum = np.sum(random_array)
    partition = ((random_array / random_sum) * total).astype(np.uintp)
    diff = total - np.sum(partition)
    partition[0] += diff
    return [int(x) for x in partition]


def main():
    out = {"equal": [], "host_random_partition": [], "python_random_partition": []}
    for n, w in [(1003, 1), (1003, 2), (1003, 3), (1003, 8), (5, 8), (1000000, 8), (7, 3), (16, 4)]:
        per = -(-n // w)
        sizes = [max(0, min((i + 1) * per, n) - min(i * per, n)) for i in range(w)]
        offs = [min(i * per, n) for i in range(w)] + [n]
        out["equal"].append({"n": n, "world": w, "sizes": sizes, "offsets": offs})
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "p.cpp"), os.path.join(td, "p")
        open(src, "w").write(CPP)
        subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe])
        for n, w in [(1003, 2), (1003, 3), (1000000, 8), (400001, 4), (262147, 8)]:
            sizes = [int(x) for x in subprocess.check_output([exe, str(n), str(w)]).split()]
            offs, same = prefix_and_same(sizes)
            out["host_random_partition"].append({"n": n, "world": w, "sizes": sizes, "offsets": offs, "same_chunk": same})
    for n, w in [(1003, 2), (1003, 3), (262147 * 8 + 3, 8), (1048579, 4)]:
        sizes = python_random_partition(n, w)
        offs, same = prefix_and_same(sizes)
        out["python_random_partition"].append({"n": n, "world": w, "sizes": sizes, "offsets": offs, "same_chunk": same})
    with open(os.path.join(HERE, "partitions.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote partitions.json")


if __name__ == "__main__":
    main()
