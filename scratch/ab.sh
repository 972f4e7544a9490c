#!/bin/bash
# usage: scratch/ab.sh <entries> lib1.so lib2.so ... : per-kernel times of the core micro-benchmark for each library build
n=$1; shift
for l in "$@"; do
  echo "== $l"
  HOTK_LIB=$GRAFT_REPO_ROOT/$l scratch/kfused.sh ab_$(basename $l .so) $n | grep -E "entries|k_graph" || exit 1
done
