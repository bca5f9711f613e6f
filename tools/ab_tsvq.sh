#!/bin/bash
# TSVQ build A/B on one box: build times + per-kernel sums for several library builds ("new" = the tree's own)
#   bash tools/ab_tsvq.sh ab/libvqhip_base.so new
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for L in "$@"; do
  if [ "$L" != "new" ]; then export VQHIP_LIB_PATH=$REPO/$L; else unset VQHIP_LIB_PATH; fi
  echo "== $L (round $round)"
  python3 $REPO/tools/tsvq_time.py c4 2>&1 | tail -1
  python3 $REPO/tools/tsvq_time.py normal 2>&1 | tail -1
done
done
for L in "$@"; do
  if [ "$L" != "new" ]; then export VQHIP_LIB_PATH=$REPO/$L; else unset VQHIP_LIB_PATH; fi
  echo "== $L kernels (C4)"
  bash $REPO/tools/tsvq_prof.sh tsvq_time.py c4 --sum 2>&1 | head -12
done
