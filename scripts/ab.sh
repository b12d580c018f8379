#!/bin/bash
# same-box A/B of library variants: scripts/ab.sh ab/libB.so ab/libC.so ...   (A = the in-tree build)
for r in 1 2 3; do
  echo -n "A "; python scripts/ablate.py 2>/dev/null | tail -1
  for L in "$@"; do echo -n "$L "; TDE_HIP_LIB=$PWD/$L python scripts/ablate.py 2>/dev/null | tail -1; done
done
