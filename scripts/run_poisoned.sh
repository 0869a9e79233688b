#!/bin/bash
# Every GPU test file on its own under TGP_POISON=1 (fresh device allocations filled with 0xFF bytes: NaN as a double, a set flag as a mask),
# so that a kernel reading memory nobody wrote fails deterministically. Per file: a poisoned NaN must not leak from one file's handles to the next.
cd $GRAFT_REPO_ROOT
export TGP_POISON=1
total=0
for f in tests/test_gpu_*.py tests/test_space_time.py tests/test_pseudo_point.py tests/test_second_tier_lgc.py; do
  out=$(timeout 1500 python -m pytest $f -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$f: $out"
done
