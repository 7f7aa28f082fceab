#!/bin/bash
# A/B of library builds on the four heavy bench scenes: tools/ab_all.sh lib1.so lib2.so ...
for s in "" "--scene mustang_red_reduced" "--scene staircase_reduced" "--scene synthetic_4k"; do
  bash tools/ab2.sh "$s" "$@" || exit 1
done
